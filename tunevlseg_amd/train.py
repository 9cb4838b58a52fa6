"""Entry point with the reference's command line: ``python -m tunevlseg_amd.train experiment=coop/clipseg model=vpt_clipseg ...``
(reference ``src/train.py:55-158``).  Composes the Hydra-shaped config tree, seeds, instantiates the task module and
runs fit / test.  The on-disk image pipeline (``src/data``) is outside this round's scope: data comes
from ``SyntheticImageTextMaskLoader`` unless the caller passes loaders to :func:`train`.
"""
from __future__ import annotations

import os
import sys
from typing import Any

import torch

from . import config_loader as CL
from . import dist as tdist
from .trainer import SyntheticImageTextMaskLoader, Trainer


def train(cfg: dict[str, Any], train_loader=None, val_loader=None, test_loader=None) -> dict[str, float]:
    rank, local_rank, world = tdist.init_distributed("cuda")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if cfg.get("seed") is not None:
        torch.manual_seed(int(cfg["seed"]))  # L.seed_everything (src/train.py:67-68)
    module = CL.instantiate(CL.select(cfg, "model")).to(device)
    tcfg = CL.resolve(cfg, cfg.get("trainer", {})) or {}
    trainer = Trainer(**{k: v for k, v in tcfg.items() if k != "_target_"},
                      default_root_dir=CL.resolve(cfg, cfg.get("paths", {}).get("output_dir")) if cfg.get("paths") else None)
    if train_loader is None:
        d = CL.resolve(cfg, cfg.get("data", {})) or {}
        bs, size = int(d.get("batch_size", 32)), int(CL.resolve(cfg, cfg.get("img_size", 352)))
        train_loader = SyntheticImageTextMaskLoader(int(d.get("synthetic_train_batches", 8)), bs, size, device, seed=1)
        val_loader = SyntheticImageTextMaskLoader(int(d.get("synthetic_val_batches", 2)), bs, size, device, seed=2)
        test_loader = SyntheticImageTextMaskLoader(int(d.get("synthetic_val_batches", 2)), bs, size, device, seed=3)
    metrics: dict[str, float] = {}
    if cfg.get("train", True):
        metrics.update(trainer.fit(module, train_loader, val_loader, ckpt_path=cfg.get("ckpt_path")))
    if cfg.get("test", True) and test_loader is not None:
        metrics.update(trainer.test(module, test_loader, ckpt_path="best" if trainer.best_path else None))
    if cfg.get("predict") and test_loader is not None:  # reference src/train.py: save_predictions after test
        from .predict import save_predictions

        metrics["saved_masks"] = float(save_predictions(module, test_loader, CL.resolve(cfg, cfg.get("output_masks_dir")),
                                                        bool(cfg.get("overwrite_outputs"))))
    return metrics


def main(argv: list[str] | None = None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    config_dir = os.environ.get("TVL_CONFIG_DIR", "configs")
    for a in list(argv):
        if a.startswith("--config-dir="):
            config_dir = a.split("=", 1)[1]
            argv.remove(a)
    cfg = CL.Composer(config_dir).compose("train", argv)
    metrics = train(cfg)
    if tdist.env_world()[0] == 0:
        print({k: round(v, 6) for k, v in metrics.items()})


if __name__ == "__main__":
    main()
