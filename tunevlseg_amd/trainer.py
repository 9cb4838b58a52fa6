"""Minimal fit / validate / test loop standing in for ``pytorch_lightning.Trainer`` on the hot path.

Mirrors what the reference's entry point does around the task module (``src/train.py:55-136``): seed, fit with
per-epoch validation, ReduceLROnPlateau on ``val_loss``, early stopping on ``val_loss`` (patience 12), checkpoint
best-by-``val_dice`` (mode max) + ``last`` (``configs/callbacks/default.yaml:9-21``), gradient accumulation
(``+trainer.accumulate_grad_batches``), data-parallel gradient averaging, then test from the best checkpoint.
Batches are dicts with ``image``, ``mask``, ``input_ids``, ``attention_mask`` already on the device.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Any, Iterable

import torch

from . import dist as tdist
from .task import FusedAdamW, ImageTextMaskModule, ReduceLROnPlateau


class Trainer:
    def __init__(self, max_epochs: int = 10, min_epochs: int = 1, accumulate_grad_batches: int = 1, check_val_every_n_epoch: int = 1,
                 default_root_dir: str | None = None, early_stopping_patience: int | None = 12, monitor: str = "val_dice",
                 monitor_mode: str = "max", log_fn=print, **_ignored: Any) -> None:
        self.max_epochs, self.min_epochs = max_epochs, min_epochs
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.root = Path(default_root_dir) if default_root_dir else None
        self.patience = early_stopping_patience
        self.monitor, self.monitor_mode = monitor, monitor_mode
        self.log = log_fn if tdist.env_world()[0] == 0 else (lambda *a, **k: None)
        self.best_score: float | None = None
        self.best_path: Path | None = None
        self.callback_metrics: dict[str, float] = {}

    # ------------------------------------------------------------------ checkpoints
    @staticmethod
    def trainable_state(module: ImageTextMaskModule) -> dict[str, torch.Tensor]:
        """Lightning saves the full state_dict; the frozen 150 M backbone never changes, so only trainable tensors are
        written here (same key names: ``net.context_learner.context_vectors`` ...)."""
        return {k: p.detach().cpu().clone() for k, p in module.named_parameters() if p.requires_grad}

    def save(self, module, opt: FusedAdamW, epoch: int, name: str) -> Path | None:
        if self.root is None or tdist.env_world()[0] != 0:
            return None
        self.root.mkdir(parents=True, exist_ok=True)
        path = self.root / f"{name}.ckpt"
        torch.save({"epoch": epoch, "state_dict": self.trainable_state(module), "optimizer_step": opt.step_count,
                    "adam": [{"m": g["m"].cpu(), "v": g["v"].cpu(), "lr": g["lr"]} for g in opt.param_groups],
                    "callback_metrics": dict(self.callback_metrics)}, path)
        return path

    @staticmethod
    def load(module, path: str | Path, opt: FusedAdamW | None = None) -> dict:
        ck = torch.load(path, map_location="cpu", weights_only=False)
        own = dict(module.named_parameters())
        with torch.no_grad():
            for k, v in ck["state_dict"].items():
                own[k].copy_(v.to(own[k].device))
        if opt is not None:
            opt.step_count = ck.get("optimizer_step", 0)
            for g, s in zip(opt.param_groups, ck.get("adam", [])):
                g["m"].copy_(s["m"].to(g["m"].device))
                g["v"].copy_(s["v"].to(g["v"].device))
                g["lr"] = s["lr"]
        return ck

    # ------------------------------------------------------------------ loops
    def _run_eval(self, module: ImageTextMaskModule, loader: Iterable, stage: str) -> dict[str, float]:
        step = module.validation_step if stage == "val" else module.test_step
        total, n = 0.0, 0
        for i, batch in enumerate(loader):
            loss = step(batch, i)
            total += float(torch.nan_to_num(loss.detach(), nan=float("inf")).item()) * len(batch["image"])
            n += len(batch["image"])
        out = module.epoch_metrics(stage)
        out[f"{stage}_loss"] = total / max(n, 1)  # rank-local, as the reference logs it (sync_dist unset)
        return out

    def fit(self, module: ImageTextMaskModule, train_loader: Iterable, val_loader: Iterable | None = None, ckpt_path: str | None = None):
        module.setup("fit")
        conf = module.configure_optimizers()
        opt: FusedAdamW = conf["optimizer"]
        sched = conf.get("lr_scheduler", {}).get("scheduler")
        start_epoch = 0
        if ckpt_path:
            start_epoch = self.load(module, ckpt_path, opt)["epoch"] + 1
        bad_epochs, best_val_loss = 0, math.inf
        for epoch in range(start_epoch, self.max_epochs):
            opt.zero_grad()
            running, n_batches = 0.0, 0
            for i, batch in enumerate(train_loader):
                loss = module.training_step(batch, i)
                (loss / self.accumulate).backward()
                if (i + 1) % self.accumulate == 0:
                    opt.step()
                    opt.zero_grad()
                running += float(loss.detach().item())
                n_batches += 1
            metrics = module.epoch_metrics("train")
            metrics["train_loss"] = running / max(n_batches, 1)
            if val_loader is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                metrics.update(self._run_eval(module, val_loader, "val"))
                if isinstance(sched, ReduceLROnPlateau):
                    sched.step(metrics["val_loss"])
                score = metrics.get(self.monitor)
                improved = score is not None and (self.best_score is None or (
                    score > self.best_score if self.monitor_mode == "max" else score < self.best_score))
                if improved:
                    self.best_score = score
                    self.best_path = self.save(module, opt, epoch, "best") or self.best_path
                if metrics["val_loss"] < best_val_loss:
                    best_val_loss, bad_epochs = metrics["val_loss"], 0
                else:
                    bad_epochs += 1
            self.callback_metrics = metrics
            self.save(module, opt, epoch, "last")
            self.log(f"epoch {epoch}: " + " ".join(f"{k}={v:.5f}" for k, v in sorted(metrics.items())))
            if self.patience is not None and bad_epochs > self.patience and epoch + 1 >= self.min_epochs:
                self.log(f"early stopping at epoch {epoch} (val_loss did not improve for {bad_epochs} epochs)")
                break
        return self.callback_metrics

    def validate(self, module, loader) -> dict[str, float]:
        module.setup("validate")
        return self._run_eval(module, loader, "val")

    def test(self, module, loader, ckpt_path: str | Path | None = None) -> dict[str, float]:
        module.setup("test")
        if ckpt_path == "best":
            ckpt_path = self.best_path
        if ckpt_path:
            self.load(module, ckpt_path)
        return self._run_eval(module, loader, "test")


class SyntheticImageTextMaskLoader:
    """Device-resident synthetic batches with the datamodule's batch layout (SURVEY.md §8d distributions); the global
    ``batch_size`` is split over ranks like ``ImageTextDatamodule`` (image_text_mask_datamodule.py:40-47)."""

    def __init__(self, n_batches: int, batch_size: int, image_size: int, device, seed: int = 0, vocab: int = 49408,
                 bos: int = 49406, eos: int = 49407, pad: int = 1, max_len: int = 8):
        rank, _, world = tdist.env_world()
        self.per_device = tdist.per_device_batch_size(batch_size, world)
        self.batches = []
        g = torch.Generator().manual_seed(seed * 1000 + rank)
        for _ in range(n_batches):
            B = self.per_device
            img = torch.randn(B, 3, image_size, image_size, generator=g)
            ids = torch.full((B, max_len), pad, dtype=torch.long)
            am = torch.zeros(B, max_len, dtype=torch.long)
            for b in range(B):
                nw = 1 + int(torch.randint(0, max_len - 2, (1,), generator=g))
                row = [bos, *torch.randint(2, min(vocab - 2, 40000), (nw,), generator=g).tolist(), eos]
                ids[b, : len(row)] = torch.tensor(row)
                am[b, : len(row)] = 1
            # a learnable target: a disc whose position follows the first image channel's mean sign
            yy, xx = torch.meshgrid(torch.arange(image_size), torch.arange(image_size), indexing="ij")
            cx = image_size * (0.3 + 0.4 * torch.rand(B, generator=g))
            mask = (((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - image_size / 2) ** 2) < (image_size / 4) ** 2).float()[:, None]
            k = len(self.batches)
            self.batches.append({"image": img.to(device), "input_ids": ids.to(device), "attention_mask": am.to(device),
                                 "mask": mask.to(device).contiguous(),
                                 # what the predict tail needs (reference image_text_mask_dataset.py:74-96)
                                 "mask_name": [f"synthetic/r{rank}_b{k}_{b}.png" for b in range(B)],
                                 "mask_shape": [torch.tensor([image_size, image_size])] * B})

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)
