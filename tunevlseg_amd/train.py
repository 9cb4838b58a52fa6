"""Entry point with the reference's command line: ``python -m tunevlseg_amd.train experiment=coop/clipseg model=vpt_clipseg ...``
(reference ``src/train.py:55-158``).  Composes the Hydra-shaped config tree, seeds, instantiates ``cfg.data`` (the reference's
``ImageTextDatamodule`` surface: datasets in the reference's wire format, per-rank shards, transforms on the device) and the task module
and runs fit / test / predict.  Synthetic device-resident batches are used only when the config ASKS for them (``data.synthetic: true``
or a ``data`` node with ``synthetic_*`` keys and no ``_target_``); a ``data`` node that names a datamodule which cannot be built raises.
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
    tcfg = CL.resolve(cfg, cfg.get("trainer", {})) or {}
    if tcfg.get("graph_step"):   # +trainer.graph_step=true: the capture stream must be current before the parameters exist (tunevlseg_amd/graph.py)
        from .graph import use_private_stream

        use_private_stream(device)
    module = CL.instantiate(CL.select(cfg, "model")).to(device)
    trainer = Trainer(**{k: v for k, v in tcfg.items() if k != "_target_"},
                      default_root_dir=CL.resolve(cfg, cfg.get("paths", {}).get("output_dir")) if cfg.get("paths") else None)
    if train_loader is None:
        dnode = cfg.get("data") or {}
        if "_target_" in dnode:
            # the reference's datamodule (src/train.py:71 instantiate(cfg.data)): built or the run fails -- never a silent fall-back
            dm = CL.instantiate(CL.select(cfg, "data"))
            dm.seed = int(cfg.get("seed") or 0)
            dm.setup("fit", world_size=world, rank=rank, device=device)
            train_loader = dm.train_dataloader() if dm.train_ds is not None else None
            val_loader = dm.val_dataloader() if dm.val_ds is not None else None
            test_loader = dm.test_dataloader() if dm.test_ds is not None else None
        elif dnode.get("synthetic") or any(str(k).startswith("synthetic_") for k in dnode):
            d = CL.resolve(cfg, dnode) or {}
            bs, size = int(d.get("batch_size", 32)), int(CL.resolve(cfg, cfg.get("img_size", 352)))
            train_loader = SyntheticImageTextMaskLoader(int(d.get("synthetic_train_batches", 8)), bs, size, device, seed=1)
            val_loader = SyntheticImageTextMaskLoader(int(d.get("synthetic_val_batches", 2)), bs, size, device, seed=2)
            test_loader = SyntheticImageTextMaskLoader(int(d.get("synthetic_val_batches", 2)), bs, size, device, seed=3)
        else:
            raise ValueError("the config has no buildable `data` node: give a datamodule (`_target_: src.data.image_text_mask_datamodule."
                             "ImageTextDatamodule`, as every experiment file of the reference does) or ask for synthetic batches explicitly "
                             "(`data.synthetic=true`)")
    metrics: dict[str, float] = {}
    if cfg.get("train", True):
        if train_loader is None:
            raise ValueError("train=true but the datamodule has no train_ds")
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
