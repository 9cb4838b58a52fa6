"""Minimal fit / validate / test loop standing in for ``pytorch_lightning.Trainer`` on the hot path.

Mirrors what the reference's entry point does around the task module (``src/train.py:55-136``): seed, fit with
per-epoch validation, ReduceLROnPlateau on ``val_loss``, early stopping on ``val_loss`` (patience 12), checkpoint
best-by-``val_dice`` (mode max) + ``last`` (``configs/callbacks/default.yaml:9-21``), gradient accumulation
(``+trainer.accumulate_grad_batches``), data-parallel gradient averaging, then test from the best checkpoint.
Batches are dicts with ``image``, ``mask``, ``input_ids``, ``attention_mask`` already on the device.
"""
from __future__ import annotations

import contextlib
import math
from pathlib import Path
from typing import Any, Iterable

import torch

from . import hip

from . import dist as tdist
from .task import ReduceLROnPlateau


@contextlib.contextmanager
def evaluation_mode(module: torch.nn.Module):
    """``module.eval()`` + ``no_grad`` for validation / test / predict (Lightning runs those stages in eval mode), restoring every
    submodule's own flag afterwards: the frozen towers stay in eval during training while the learner trains (dropout of the
    SharedAttn learner's encoder layer: on in fit, off here)."""
    modes = [(m, m.training) for m in module.modules()]
    module.eval()
    try:
        with torch.no_grad():
            yield
    finally:
        for m, was in modes:
            m.training = was


class Trainer:
    def __init__(self, max_epochs: int = 10, min_epochs: int = 1, accumulate_grad_batches: int = 1, check_val_every_n_epoch: int = 1,
                 default_root_dir: str | None = None, early_stopping_patience: int | None = 12, early_stopping_min_delta: float = 1e-4,
                 monitor: str = "val_dice", monitor_mode: str = "max", log_fn=print, graph_step: bool = False, **_ignored: Any) -> None:
        self.max_epochs, self.min_epochs = max_epochs, min_epochs
        self.accumulate = max(1, int(accumulate_grad_batches))
        # forward + backward of a train step replayed as a hipGraph (tunevlseg_amd/graph.py; opt-in, +trainer.graph_step=true)
        self.graph_step = bool(graph_step)
        if self.graph_step and self.accumulate != 1:
            raise ValueError("graph_step captures zero_grad + forward + backward as one unit: accumulate_grad_batches must be 1")
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.root = Path(default_root_dir) if default_root_dir else None
        # EarlyStopping(monitor=val_loss, patience=12, min_delta=1e-4, mode=min) -- configs/callbacks/default.yaml:17-21
        self.patience, self.min_delta = early_stopping_patience, float(early_stopping_min_delta)
        self.monitor, self.monitor_mode = monitor, monitor_mode
        self.log = log_fn if tdist.env_world()[0] == 0 else (lambda *a, **k: None)
        self.best_score: float | None = None
        self.best_path: Path | None = None
        self.callback_metrics: dict[str, float] = {}
        self.wait_count, self.best_val_loss = 0, math.inf

    # ------------------------------------------------------------------ checkpoints
    @staticmethod
    def trainable_state(module) -> dict[str, torch.Tensor]:
        """Lightning saves the full state_dict; the frozen 150 M backbone never changes, so only trainable tensors are
        written here (same key names: ``net.context_learner.context_vectors`` ...)."""
        return {k: p.detach().cpu().clone() for k, p in module.named_parameters() if p.requires_grad}

    def save(self, module, opt, epoch: int, name: str, sched=None) -> Path | None:
        """Rank 0 writes ``<root>/<name>.ckpt``; EVERY rank returns the path once the file is complete (barrier), so that
        ``test(ckpt_path="best")`` loads the same weights everywhere."""
        if self.root is None:
            return None
        path = self.root / f"{name}.ckpt"
        if tdist.env_world()[0] == 0:
            self.root.mkdir(parents=True, exist_ok=True)
            ck = {"epoch": epoch, "state_dict": self.trainable_state(module), "callback_metrics": dict(self.callback_metrics),
                  "early_stopping": {"wait_count": self.wait_count, "best_val_loss": self.best_val_loss},
                  "checkpoint_callback": {"best_score": self.best_score}}
            if hasattr(opt, "state_dict"):
                ck["optimizer"] = opt.state_dict()
            if sched is not None and hasattr(sched, "state_dict"):
                ck["lr_scheduler"] = sched.state_dict()
            tmp = path.with_suffix(".tmp")
            torch.save(ck, tmp)
            tmp.replace(path)
        tdist.barrier()
        return path

    def load(self, module, path: str | Path, opt=None, sched=None) -> dict:
        """Restores trainable tensors (matched by name against parameters AND buffers; a reference Lightning checkpoint carries
        the frozen backbone and buffers too: equal-shaped entries are copied, unknown ones reported), optimiser moments,
        scheduler and early-stopping state."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        own = {**dict(module.named_buffers()), **dict(module.named_parameters())}
        unknown = []
        with torch.no_grad():
            for k, v in ck["state_dict"].items():
                if k in own and tuple(own[k].shape) == tuple(v.shape):
                    own[k].copy_(v.to(own[k].device))
                else:
                    unknown.append(k)
        if unknown:
            self.log(f"checkpoint {path}: {len(unknown)} entries without a matching tensor here were skipped, e.g. {unknown[:3]}")
        if opt is not None and "optimizer" in ck and hasattr(opt, "load_state_dict"):
            opt.load_state_dict(ck["optimizer"])
        if sched is not None and "lr_scheduler" in ck and hasattr(sched, "load_state_dict"):
            sched.load_state_dict(ck["lr_scheduler"])
        es = ck.get("early_stopping")
        if es:
            self.wait_count, self.best_val_loss = int(es["wait_count"]), float(es["best_val_loss"])
        self.best_score = ck.get("checkpoint_callback", {}).get("best_score", self.best_score)
        return ck

    # ------------------------------------------------------------------ loops
    def _run_eval(self, module, loader: Iterable, stage: str) -> dict[str, float]:
        step = module.validation_step if stage == "val" else module.test_step
        total_dev, n = None, 0
        with evaluation_mode(module):
            for i, batch in enumerate(loader):
                loss = step(batch, i)
                # accumulated on the device: a .item() per batch would make the host wait for the GPU before it may enqueue the next batch
                term = torch.nan_to_num(loss.detach().double(), nan=float("inf")) * len(batch["image"])
                total_dev = term if total_dev is None else total_dev + term
                n += len(batch["image"])
            out = module.epoch_metrics(stage)
        total = float(total_dev.item()) if total_dev is not None else 0.0
        # every rank sees its own shard of the data: the scheduler, the early-stopping test and the logs must all read ONE number,
        # the sample-weighted mean over ranks (Lightning syncs the stopping decision; a rank-local loss desynchronises the LR)
        total, n = tdist.reduce_sums([total, float(n)])
        out[f"{stage}_loss"] = total / max(n, 1.0)
        return out

    def fit(self, module, train_loader: Iterable, val_loader: Iterable | None = None, ckpt_path: str | None = None):
        module.setup("fit")
        conf = module.configure_optimizers()
        opt = conf["optimizer"]
        sched = conf.get("lr_scheduler", {}).get("scheduler")
        start_epoch = 0
        if ckpt_path:
            start_epoch = self.load(module, ckpt_path, opt, sched)["epoch"] + 1
        arm = getattr(opt, "set_exchange_armed", lambda armed: None)
        stepper = None
        if self.graph_step:
            from .graph import GraphedStep

            stepper = GraphedStep(module, opt)
        for epoch in range(start_epoch, self.max_epochs):
            if hasattr(train_loader, "set_epoch"):   # per-rank shard order and augmentation draws of this epoch (DistributedSampler.set_epoch)
                train_loader.set_epoch(epoch)
            opt.zero_grad()
            running, n_batches, pending = None, 0, 0   # running: device-side sum of the step losses (no host synchronisation per step)
            n_train = len(train_loader) if hasattr(train_loader, "__len__") else None
            for i, batch in enumerate(train_loader):
                boundary = (i + 1) % self.accumulate == 0 or (n_train is not None and i + 1 == n_train)
                arm(boundary)  # the gradient all-reduce rides on the backward of the micro-step that ends in step()
                if stepper is not None:
                    loss = stepper(batch)   # zero_grad + forward + backward: eager the first time a shape is seen, a graph replay from its third batch on
                else:
                    loss = module.training_step(batch, i)
                    (loss / self.accumulate).backward()
                pending += 1
                if boundary:
                    opt.step()
                    opt.zero_grad()
                    pending = 0
                running = loss.detach().double() if running is None else running + loss.detach().double()
                n_batches += 1
            if pending:  # loader without a length: the leftover micro-batches of the epoch still make a step
                arm(True)
                opt.step()
                opt.zero_grad()
            metrics = module.epoch_metrics("train")
            metrics["train_loss"] = (float(running.item()) if running is not None else 0.0) / max(n_batches, 1)
            # the loss and optimiser kernels keep sticky device-side NaN / Inf flags (hip.nonfinite_flags): read here, where the host has just
            # waited for the epoch anyway.  A non-finite step ends the fit with an exception instead of a NaN in the log
            hip.check_finite(what=f"fit, epoch {epoch}")
            stop = False
            if val_loader is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                metrics.update(self._run_eval(module, val_loader, "val"))
                if isinstance(sched, ReduceLROnPlateau):   # Lightning hands the monitored value to ReduceLROnPlateau only
                    sched.step(metrics["val_loss"])
                score = metrics.get(self.monitor)
                improved = score is not None and (self.best_score is None or (
                    score > self.best_score if self.monitor_mode == "max" else score < self.best_score))
                self.callback_metrics = metrics
                if improved:  # the metrics are all-reduced, so every rank takes the same branch (save() holds a barrier)
                    self.best_score = score
                    self.best_path = self.save(module, opt, epoch, "best", sched) or self.best_path
                # Lightning EarlyStopping: improvement = monitor < best - min_delta; stop when wait_count >= patience
                if not math.isfinite(metrics["val_loss"]):
                    stop = True
                elif metrics["val_loss"] < self.best_val_loss - self.min_delta:
                    self.best_val_loss, self.wait_count = metrics["val_loss"], 0
                else:
                    self.wait_count += 1
                    stop = self.patience is not None and self.wait_count >= self.patience
            if sched is not None and not isinstance(sched, ReduceLROnPlateau) and hasattr(sched, "step"):
                sched.step()   # every other scheduler: once per epoch, no argument (Lightning's interval="epoch" default)
            self.callback_metrics = metrics
            self.save(module, opt, epoch, "last", sched)
            self.log(f"epoch {epoch}: " + " ".join(f"{k}={v:.5f}" for k, v in sorted(metrics.items())))
            # one decision for all ranks (they computed it from reduced numbers; the reduction makes that an invariant, not a hope)
            stop = tdist.reduce_sums([1.0 if stop else 0.0])[0] > 0
            if stop and epoch + 1 >= self.min_epochs:
                self.log(f"early stopping at epoch {epoch} (val_loss did not improve by {self.min_delta} for {self.wait_count} checks)")
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
