"""Task module with the reference's ``ImageTextMaskModule`` surface, minus Lightning
(reference ``src/models/image_text_mask_module.py:23-395``).

``model_step`` / ``get_logits`` / ``get_optim_groups`` / ``configure_optimizers`` / the ``*_step`` methods and the
metric names (``train_loss``, ``train_dice_step``, ``val_dice``, ``val_iou``, ``val_loss`` ...) are kept.  The loss and
the metric statistics come from ONE fused HIP pass over logits and mask (``tvl_dicece_stats``).
"""
from __future__ import annotations

from typing import Any, Callable, Mapping

import torch
from torch import nn

from . import dist as tdist
from . import hip, ops


class DiceCELoss(nn.Module):
    """``monai.losses.DiceCELoss`` for the configuration the reference uses (``configs/model/*.yaml:21-25``:
    ``sigmoid=True, lambda_dice=1, lambda_ce=0.2``; one output channel -> BCE-with-logits as the CE term)."""

    def __init__(self, sigmoid: bool = True, lambda_dice: float = 1.0, lambda_ce: float = 1.0, threshold: float = 0.5, **unsupported):
        super().__init__()
        if not sigmoid or any(v for v in unsupported.values()):
            raise NotImplementedError(f"only DiceCELoss(sigmoid=True, lambda_dice, lambda_ce) is on the hot path, got {unsupported}")
        self.lambda_dice, self.lambda_ce, self.threshold = float(lambda_dice), float(lambda_ce), float(threshold)
        self.last_counts: torch.Tensor | None = None  # int64 [B,4] TP/FP/FN/TN of the last call

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        loss, counts = ops.DiceCELossFn.apply(logits, target, self.lambda_dice, self.lambda_ce, self.threshold)
        self.last_counts = counts
        return loss


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return hip.bias_act(ops._c(x).view(-1, x.shape[-1]), None, hip.ACT_SIGMOID).view(x.shape)

    @staticmethod
    def backward(ctx, dy):  # pragma: no cover - predictions are never differentiated in the reference
        raise NotImplementedError


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    return SigmoidFn.apply(x.detach())


class DiceSamples:
    """``torchmetrics.Dice(threshold, zero_division=1, average="samples")``: mean over samples of 2TP/(2TP+FP+FN)."""

    def __init__(self, zero_division: float = 1.0):
        self.zero_division = zero_division
        self.reset()

    def reset(self):
        self.values: list[torch.Tensor] = []

    def update(self, counts: torch.Tensor):
        self.values.append(counts)   # int64 [B, 4]; the arithmetic waits for compute(): a train step launches nothing for its metrics

    def compute(self) -> float:
        c = torch.cat(self.values).to(torch.float64)
        den = 2 * c[:, 0] + c[:, 1] + c[:, 2]
        v = torch.where(den > 0, 2 * c[:, 0] / den.clamp(min=1), torch.full_like(den, self.zero_division))
        return float(tdist.allgather_cat(v).mean().item())


class JaccardBinary:
    """``torchmetrics.JaccardIndex(task="binary", threshold, zero_division=1)``: global TP/(TP+FP+FN), int64 state."""

    def __init__(self, zero_division: float = 1.0):
        self.zero_division = zero_division
        self.reset()

    def reset(self):
        self.batches: list[torch.Tensor] = []

    def update(self, counts: torch.Tensor):
        self.batches.append(counts)   # summed in compute()

    def compute(self) -> float:
        c = tdist.allreduce_counts(torch.cat(self.batches).sum(0))
        tp, fp, fn = (float(c[i].item()) for i in range(3))
        den = tp + fp + fn
        return tp / den if den > 0 else self.zero_division


class ImageTextMaskModule(nn.Module):
    def __init__(self, net: nn.Module, loss_fn: nn.Module, optimizer: Callable | None = None, scheduler: Callable | None = None,
                 compile: bool = False, task: str = "binary", threshold: float = 0.5, weight_decay: float = 0.0,
                 log_image_num: int = 8, lr_scheduler_config: Mapping[str, Any] | None = None,
                 activation_fn: Callable | None = sigmoid, cache_outputs: bool = False, *args, **kwargs) -> None:
        super().__init__()
        if task != "binary":
            raise NotImplementedError("only task='binary' is used by the reference configs")
        self.hparams = dict(compile=compile, task=task, threshold=threshold, weight_decay=weight_decay, log_image_num=log_image_num,
                            lr_scheduler_config=lr_scheduler_config, cache_outputs=cache_outputs)
        self.net = net
        self.loss_fn = loss_fn
        if hasattr(loss_fn, "threshold"):
            loss_fn.threshold = float(threshold)
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.activation_fn = (lambda x: x) if activation_fn is None else activation_fn
        self.metrics: dict[str, Any] = {}
        self.logged: dict[str, float] = {}

    def forward(self, *args, **kwargs) -> torch.Tensor:
        return self.net(*args, **kwargs)

    def setup(self, stage: str) -> None:
        """reference ``setup`` (``image_text_mask_module.py:272-302``)."""
        stages = {"fit": ("train", "val"), "validate": ("val",), "test": ("test",)}.get(stage, ())
        for s in stages:
            self.metrics[f"{s}_dice"] = DiceSamples()
            self.metrics[f"{s}_iou"] = JaccardBinary()

    def get_logits(self, batch: Mapping[str, Any]) -> torch.Tensor:
        text_input = {k: batch[k] for k in ("input_ids", "attention_mask")}
        if self.hparams.get("cache_outputs"):
            text_input["cache_name"] = batch["cache_name"]
        return self(image_input=batch["image"], text_input=text_input)

    def model_step(self, batch: Mapping[str, Any]):
        """reference ``model_step`` (``image_text_mask_module.py:87-107``): (loss, predictions, integer targets)."""
        logits = self.get_logits(batch)
        mask = batch["mask"]
        loss = self.loss_fn(logits, mask)
        preds = self.activation_fn(logits)
        return loss, preds, mask.long()

    def _step(self, stage: str, batch) -> torch.Tensor:
        """``model_step`` + metric update as the reference's ``*_step`` methods do it -- but the confusion counts the metrics need were
        already taken by the loss's own pass over (logits, mask), so neither the probabilities nor ``mask.long()`` are materialised."""
        logits = self.get_logits(batch)
        mask = batch["mask"]
        loss = self.loss_fn(logits, mask)
        counts = getattr(self.loss_fn, "last_counts", None)
        if counts is None:  # foreign loss: statistics from a dedicated pass over the logits (sigmoid(x) > thr and (int64)mask)
            _, counts, _ = hip.dicece_stats(ops._c(logits.detach()), ops._c(mask.to(torch.float32)), self.hparams["threshold"])
        if f"{stage}_dice" in self.metrics:
            self.metrics[f"{stage}_dice"].update(counts)
            self.metrics[f"{stage}_iou"].update(counts)
        return loss

    def training_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        return self._step("train", batch)

    def validation_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        with torch.no_grad():
            return self._step("val", batch)

    def test_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        with torch.no_grad():
            return self._step("test", batch)

    def predict_step(self, batch, batch_idx: int = 0):
        """reference image_text_mask_module.py:244-255: probabilities + what is needed to save them at the original size."""
        with torch.no_grad():
            preds = self.activation_fn(self.get_logits(batch))
        return {"preds": preds, "mask_name": batch.get("mask_name"), "mask_shape": batch.get("mask_shape")}

    def epoch_metrics(self, stage: str, reset: bool = True) -> dict[str, float]:
        out = {f"{stage}_dice": self.metrics[f"{stage}_dice"].compute(), f"{stage}_iou": self.metrics[f"{stage}_iou"].compute()}
        if reset:
            self.metrics[f"{stage}_dice"].reset()
            self.metrics[f"{stage}_iou"].reset()
        return out

    # ---- optimiser groups (reference ``get_optim_groups``, image_text_mask_module.py:304-361) ----
    def get_optim_groups(self):
        """Without weight decay: all parameters in one group.  With it: weights of ``nn.Linear`` / conv modules and fused
        attention ``*proj_weight`` tensors decay; everything else (prompt vectors, ``residual_ratio``, norm weights, biases,
        embeddings and the frozen backbone tensors, which live in plain containers) does not."""
        wd = self.hparams["weight_decay"]
        if wd <= 0:
            return self.parameters()
        decaying_modules = (nn.Linear, nn.modules.conv._ConvNd)

        def decays(module: nn.Module, leaf: str) -> bool:
            return leaf.endswith("proj_weight") or (leaf.endswith("weight") and isinstance(module, decaying_modules))

        owner = {}  # full parameter name -> does it decay?  (a tensor reachable under two names must agree with itself)
        for prefix, module in self.named_modules():
            for leaf, _ in module.named_parameters(recurse=False):
                full = f"{prefix}.{leaf}" if prefix else leaf
                verdict = decays(module, leaf)
                if owner.setdefault(full, verdict) != verdict:
                    raise ValueError(f"parameter {full} made it into both decay/no_decay sets!")
        named = dict(self.named_parameters())
        yes = [named[k] for k in sorted(named) if owner.get(k, False)]
        no = [named[k] for k in sorted(named) if not owner.get(k, False)]
        return [{"params": yes, "weight_decay": wd}, {"params": no, "weight_decay": 0.0}]

    def configure_optimizers(self) -> dict[str, Any]:
        optimizer = self.optimizer(self.get_optim_groups())
        if self.scheduler is not None:
            scheduler = self.scheduler(optimizer=optimizer)
            return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "monitor": "val_loss", "interval": "epoch",
                                                             "frequency": 1, **(self.hparams["lr_scheduler_config"] or {})}}
        return {"optimizer": optimizer}


class FusedAdamW:
    """``torch.optim.AdamW`` semantics over ONE :class:`tunevlseg_amd.dist.FlatParams` buffer.

    All trainable parameters of all groups are re-homed into one flat fp32 buffer (group after group), so the data-parallel
    exchange is one bucketed, backward-overlapped all-reduce (:class:`tunevlseg_amd.dist.GradExchange`) whatever the number of
    groups, and each group -- a contiguous segment with its own ``lr`` / ``weight_decay`` (decay / no-decay split of
    ``get_optim_groups``) -- is one ``tvl_adamw`` launch.  Frozen parameters passed in by the reference-style
    ``self.parameters()`` call are skipped, as AdamW skips grad-less ones.
    """

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        kept = [(g, [p for p in g["params"] if p.requires_grad]) for g in groups]
        kept = [(g, ps) for g, ps in kept if ps]
        if not kept:
            raise ValueError("no trainable parameters")
        self.flat = tdist.FlatParams([p for _, ps in kept for p in ps])
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self.exchange = tdist.GradExchange(self.flat)
        self.param_groups = []
        off = 0
        for g, ps in kept:
            n = sum(p.numel() for p in ps)
            self.param_groups.append({"params": ps, "span": (off, off + n), "lr": g.get("lr", lr), "betas": g.get("betas", betas),
                                      "eps": g.get("eps", eps), "weight_decay": g.get("weight_decay", weight_decay),
                                      "m": self.m[off:off + n], "v": self.v[off:off + n]})
            off += n
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat.zero_grad()

    def set_exchange_armed(self, armed: bool) -> None:
        """Gradient accumulation: disarm on the micro-steps that do not end in ``step()`` (Lightning's ``no_sync``)."""
        self.exchange.armed = armed

    def step(self) -> None:
        self.step_count += 1
        scale = self.exchange.finish()  # DDP: SUM over ranks (enqueued during backward), averaged inside the update kernel
        for g in self.param_groups:
            a, b = g["span"]
            hip.adamw(self.flat.data[a:b], self.flat.grad[a:b], g["m"], g["v"], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                      g["weight_decay"], self.step_count, scale)

    def state_dict(self) -> dict[str, Any]:
        return {"step": self.step_count, "m": self.m.cpu(), "v": self.v.cpu(), "lr": [g["lr"] for g in self.param_groups]}

    def load_state_dict(self, sd: Mapping[str, Any]) -> None:
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        for g, lr in zip(self.param_groups, sd["lr"]):
            g["lr"] = lr


class ReduceLROnPlateau:
    """``torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", factor, patience)`` for :class:`FusedAdamW`
    (reference ``configs/model/vpt_clipseg.yaml:43-48``; threshold 1e-4 rel, as torch's default)."""

    def __init__(self, optimizer, mode: str = "min", factor: float = 0.1, patience: int = 10, threshold: float = 1e-4, min_lr: float = 0.0):
        if mode != "min":
            raise NotImplementedError("mode='min' only")
        self.optimizer, self.factor, self.patience, self.threshold, self.min_lr = optimizer, factor, patience, threshold, min_lr
        self.best, self.num_bad = float("inf"), 0

    def state_dict(self) -> dict[str, float]:
        return {"best": self.best, "num_bad": self.num_bad}

    def load_state_dict(self, sd: Mapping[str, float]) -> None:
        self.best, self.num_bad = float(sd["best"]), int(sd["num_bad"])

    def step(self, metric: float) -> None:
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            for g in self.optimizer.param_groups:
                g["lr"] = max(g["lr"] * self.factor, self.min_lr)
            self.num_bad = 0
