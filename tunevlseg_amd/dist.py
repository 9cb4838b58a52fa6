"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` over RCCL (xGMI) or gloo (CPU tests).

The reference gets DDP from Lightning (``configs/trainer/ddp.yaml:4-9``): every rank runs the same
net on ``batch_size // world_size`` samples and the gradients of the *trainable* parameters are
averaged.  Here that is ONE all-reduce per optimiser step over ONE flat fp32 buffer that aliases
every trainable gradient (9 K floats for VPT-10 ... 772 K for MaPLe depth 9 -- latency-bound, so a
single small collective instead of per-parameter buckets).
"""
from __future__ import annotations

import os
from typing import Iterable

import torch
import torch.distributed as dist


def env_world() -> tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(device_type: str = "cuda") -> tuple[int, int, int]:
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    # TVL_DIST_SINGLE_RANK_GROUP=1: a process group of ONE rank (RCCL on one GPU): the collectives of the N > 1 path -- async all-reduce enqueued from
    # the backward's hooks, waits, barrier -- run for real, as identities (tests/test_train_gpu.py; the only RCCL a one-GPU box can exercise)
    single = world == 1 and os.environ.get("TVL_DIST_SINGLE_RANK_GROUP") == "1"
    if (world > 1 or single) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL ("nccl" on ROCm) for GPUs; TVL_DIST_BACKEND=gloo lets the N>1 code path be rehearsed with several ranks on ONE
        # device (RCCL refuses duplicate devices), gloo staging the tiny gradient buffer through the host
        backend = os.environ.get("TVL_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda" and backend == "gloo" and os.environ.get("TVL_ALLOW_SHARED_DEVICE") != "1":
            # One process per device is a requirement: with several ranks on one device, steps went non-finite in round 3 (first
            # step only, about half of the 4-rank runs) for a reason that was never pinned to this code or to the platform.  The rehearsal
            # mode stays available for debugging, behind a flag that says what it is.
            raise RuntimeError("TVL_DIST_BACKEND=gloo puts every rank on cuda:0: unsupported (one process per GPU; DESIGN.md §6).  For debugging only, "
                               "set TVL_ALLOW_SHARED_DEVICE=1.")
        if device_type == "cuda":
            torch.cuda.set_device(0 if backend == "gloo" else local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def group_active() -> bool:
    """A process group exists (also one of a single rank, whose collectives are identities): the exchange code runs."""
    return dist.is_available() and dist.is_initialized()


def per_device_batch_size(global_batch_size: int, world: int) -> int:
    """``data.batch_size`` is the GLOBAL batch (reference image_text_mask_datamodule.py:40-47)."""
    if global_batch_size % world != 0:
        raise ValueError(f"Batch size ({global_batch_size}) is not divisible by the number of devices ({world}).")
    return global_batch_size // world


class FlatParams:
    """Re-homes the trainable parameters (and their ``.grad``) into two contiguous fp32 buffers.

    ``p.data`` / ``p.grad`` become views, so autograd accumulates straight into the flat gradient,
    one all-reduce covers everything, and the fused AdamW kernel updates all parameters in one launch.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.data = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets = []
        off = 0
        for p in self.params:
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            self.offsets.append((off, k))
            off += k
        self.numel = n

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, (off, k) in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view(p.shape)

    def allreduce_grads(self) -> float:
        """SUM all-reduce of the flat gradient; returns the scale (1/world) the optimiser must apply."""
        w = world_size()
        if group_active():
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        return 1.0 / w


class GradExchange:
    """Gradient exchange overlapped with the backward pass (north_star: "RCCL all-reduce of prompt-parameter grads ...
    overlapped with the backward"; reference: Lightning DDP's bucketed reducer, configs/trainer/ddp.yaml:4-9).

    The flat gradient is cut into contiguous buckets (whole parameters, >= ``bucket_bytes`` each).  A post-accumulate-grad hook
    per parameter counts arrivals; the moment a bucket's last gradient has been accumulated its SUM all-reduce is enqueued
    asynchronously (RCCL runs it on its own stream, behind an event on the compute stream), so for deep prompts the slices of
    the upper layers travel while the lower layers are still in backward.  ``finish()`` enqueues what never fired (parameters
    without a gradient this step), waits for every collective and returns the 1/world scale the optimiser applies.  With
    gradient accumulation the hooks stay disarmed until the boundary micro-step, so only the accumulated total is exchanged.
    """

    def __init__(self, flat: FlatParams, bucket_bytes: int | None = None):
        if bucket_bytes is None:  # 1 MiB: VPT-10 is one bucket, MaPLe depth 9 (3 MB of gradients) three
            bucket_bytes = int(os.environ.get("TVL_DDP_BUCKET_BYTES", 1 << 20))
        self.flat = flat
        self.armed = True
        self.buckets: list[tuple[int, int, int]] = []  # (start, end, number of parameters)
        self.bucket_of: list[int] = []
        start, count = 0, 0
        for i, (off, k) in enumerate(flat.offsets):
            self.bucket_of.append(len(self.buckets))
            count += 1
            if (off + k - start) * 4 >= bucket_bytes or i == len(flat.offsets) - 1:
                self.buckets.append((start, off + k, count))
                start, count = off + k, 0
        self._arrived = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works: list = []
        self.launched_in_backward = 0  # statistic: buckets whose all-reduce was enqueued from inside backward (tests / DESIGN.md)
        for i, prm in enumerate(flat.params):
            prm.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))

    def _launch(self, b: int) -> None:
        s, e, _ = self.buckets[b]
        self._works.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, async_op=True))
        self._launched[b] = True

    def _on_grad(self, i: int) -> None:
        if not self.armed or not group_active():
            return
        b = self.bucket_of[i]
        self._arrived[b] += 1
        if self._arrived[b] == self.buckets[b][2] and not self._launched[b]:
            self._launch(b)
            self.launched_in_backward += 1

    def finish(self) -> float:
        w = world_size()
        if group_active():
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
            for wk in self._works:
                wk.wait()
        self._works.clear()
        self._arrived = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        return 1.0 / w


def reduce_sums(values: list[float]) -> list[float]:
    """SUM of a few python floats over the ranks (validation loss totals, stop flags); identity for world size 1."""
    if not group_active():
        return list(values)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def barrier() -> None:
    if group_active():
        dist.barrier()


def allreduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """Integer confusion counts: exact SUM across ranks (torchmetrics JaccardIndex state sync)."""
    if group_active():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def allgather_cat(values: torch.Tensor) -> torch.Tensor:
    """Per-sample values gathered over ranks (torchmetrics Dice(average='samples') 'cat' state sync)."""
    w = world_size()
    if not group_active():
        return values
    out = [torch.empty_like(values) for _ in range(w)]
    dist.all_gather(out, values.contiguous())
    return torch.cat(out)
