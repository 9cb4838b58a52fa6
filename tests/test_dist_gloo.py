"""N>1 path on CPU: world_size-2 gloo.  The flat-gradient all-reduce of ``tunevlseg_amd.dist`` must reproduce the
single-process gradient of the concatenated global batch (DDP semantics of the reference's trainer=ddp / ddp_sim,
configs/trainer/ddp_sim.yaml:5-7); metric state sync must reproduce the global Dice/IoU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import clipseg_oracle as O
from tunevlseg_amd import dist as tdist
from tunevlseg_amd.config import CLIPSegConfig
from tunevlseg_amd.task import DiceSamples, JaccardBinary
from tunevlseg_amd.weights import init_clipseg_state_dict


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(B, 3, 64, 64, generator=g)
    ids = torch.tensor([[62, 5, 9, 63, 1, 1]] * B)
    am = (ids != 1).long()
    mask = (torch.rand(B, 1, 64, 64, generator=g) > 0.7).float()
    return pix, ids, am, mask


def _local_grads(rank, world, global_b):
    cfg = CLIPSegConfig.tiny()
    sd = init_clipseg_state_dict(cfg, 11)
    pix, ids, am, mask = _inputs(global_b, 0)
    per = tdist.per_device_batch_size(global_b, world)
    sl = slice(rank * per, (rank + 1) * per)
    ctx = torch.nn.Parameter(torch.randn(2, 4, 32, generator=torch.Generator().manual_seed(1)) * 0.02)
    conv_w = torch.nn.Parameter(torch.randn(1, 16, 5, 5, generator=torch.Generator().manual_seed(2)) * 0.1)
    conv_b = torch.nn.Parameter(torch.zeros(1))
    logits = O.vpt_forward(sd, cfg, {"kind": "vpt", "ctx": ctx}, pix[sl], ids[sl], am[sl], (conv_w, conv_b, torch.tensor(0.5)))
    O.dice_ce_loss(logits, mask[sl]).backward()
    tp, fp, fn, tn = O.confusion_counts(torch.sigmoid(logits.detach()), mask[sl].long())
    return [ctx, conv_w, conv_b], torch.stack((tp, fp, fn, tn), 1)


def _worker(rank, world, port, global_b, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    tdist.init_distributed("cpu")
    params, counts = _local_grads(rank, world, global_b)
    grads = [p.grad.clone() for p in params]
    flat = tdist.FlatParams(params)
    assert all(p.grad.abs().sum() == 0 for p in params)  # grads re-homed into the (zeroed) flat buffer
    for p, g in zip(params, grads):
        p.grad.add_(g)  # autograd accumulates in place into the views
    scale = flat.allreduce_grads()
    d, j = DiceSamples(), JaccardBinary()
    d.update(counts)
    j.update(counts)
    # plain numpy / floats through the queue: a tensor travels as a shared-memory handle that dies with this process
    res = (flat.grad * scale).numpy().copy(), float(d.compute()), float(j.compute())
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_allreduce_equals_single_process_global_batch():
    world, global_b = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_b, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat_grad, dice, iou = q.get(timeout=240)
    flat_grad = torch.from_numpy(flat_grad)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process on the whole global batch.  The Dice term is a per-sample mean and BCE a per-pixel mean, so the
    # average of the two half-batch gradients equals the full-batch gradient exactly (up to fp32 summation order).
    params, counts = _local_grads(0, 1, global_b)
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    assert (flat_grad - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 1e-10
    tp, fp, fn = counts[:, 0], counts[:, 1], counts[:, 2]
    assert abs(dice - O.dice_samples(tp, fp, fn).item()) < 1e-12
    assert abs(iou - O.jaccard_binary(tp, fp, fn).item()) < 1e-12


# ---------------------------------------------------------------------------------------------------------------------------------
# overlapped gradient exchange + the fit loop with two ranks (CPU stand-in module: the Trainer / GradExchange logic is host code)
# ---------------------------------------------------------------------------------------------------------------------------------
class _ToyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.a = torch.nn.Parameter(torch.randn(300, generator=g))
        self.b = torch.nn.Parameter(torch.randn(40, 10, generator=g))
        self.unused = torch.nn.Parameter(torch.zeros(7))  # never receives a gradient: its bucket is exchanged by finish()
        self.c = torch.nn.Parameter(torch.randn(5, generator=g))

    def forward(self, x):
        return (x @ self.b).sum(1) * self.c.sum() + (self.a ** 2).sum()


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    tdist.init_distributed("cpu")
    net = _ToyNet()
    flat = tdist.FlatParams(net.parameters())
    ex = tdist.GradExchange(flat, bucket_bytes=256)
    x = torch.randn(8, 40, generator=torch.Generator().manual_seed(0))[rank * 4:(rank + 1) * 4]
    # micro-step 1 of 2 (gradient accumulation): disarmed, nothing is exchanged
    ex.armed = False
    net(x).sum().backward()
    assert ex.launched_in_backward == 0 and not ex._works
    ex.armed = True
    net(x).sum().backward()
    in_backward = ex.launched_in_backward
    scale = ex.finish()
    res = (flat.grad * scale).numpy().copy(), in_backward, len(ex.buckets)   # numpy: see _worker
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_exchange_buckets_fire_during_backward_and_sum_like_one_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat_grad, in_backward, n_buckets = q.get(timeout=240)
    flat_grad = torch.from_numpy(flat_grad)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n_buckets >= 3 and 1 <= in_backward < n_buckets  # the bucket holding `unused` can only go out in finish()
    net = _ToyNet()
    x = torch.randn(8, 40, generator=torch.Generator().manual_seed(0))
    (2 * net(x).sum() / 2).backward()  # two accumulated micro-steps per rank, averaged over two ranks = 1x the global sum
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    assert torch.allclose(flat_grad, ref, rtol=1e-5, atol=1e-5)


class _ToyTask(torch.nn.Module):
    """The ImageTextMaskModule surface the Trainer drives, over a one-parameter model whose optimum differs per rank's data."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.drop = torch.nn.Dropout(0.5)
        self.eval_modes = []

    def setup(self, stage):
        self._n = {}

    def _loss(self, batch):
        return ((self.w - batch["target"]) ** 2).mean()

    def training_step(self, batch, i=0):
        return self._loss(batch)

    def validation_step(self, batch, i=0):
        self.eval_modes.append(self.drop.training)
        return self._loss(batch)

    test_step = validation_step

    def epoch_metrics(self, stage):
        return {f"{stage}_dice": float(-abs(self.w.item() - 1.0)), f"{stage}_iou": 0.0}

    def configure_optimizers(self):
        from tunevlseg_amd.task import ReduceLROnPlateau

        opt = torch.optim.SGD(self.parameters(), lr=0.2)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=1)}}


def _trainer_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    tdist.init_distributed("cpu")
    from tunevlseg_amd.trainer import Trainer

    task = _ToyTask()
    train = [{"image": torch.zeros(2), "target": torch.full((2,), 1.0)}] * 3
    # validation shards that disagree: rank 0's loss keeps falling towards target 1, rank 1's (target 5, twice the samples) rises
    val = [{"image": torch.zeros(2 if rank == 0 else 4), "target": torch.full((2 if rank == 0 else 4,), 1.0 if rank == 0 else 5.0)}]
    logs = []
    tr = Trainer(max_epochs=12, min_epochs=1, default_root_dir=root, early_stopping_patience=2, log_fn=logs.append)
    task.train()

    def sync_grads():  # the toy optimiser is plain SGD: average the one gradient by hand (FusedAdamW does this on the HIP path)
        dist.all_reduce(task.w.grad)
        task.w.grad /= world

    task.w.register_post_accumulate_grad_hook(lambda p: sync_grads())
    final = tr.fit(task, train, val)
    lr = task.configure_optimizers  # noqa: F841 (keep the bound method alive for the log below)
    epochs_run = len([line for line in logs if line.startswith("epoch")]) if rank == 0 else None
    test = tr.test(task, val, ckpt_path="best")
    q.put((rank, final["val_loss"], str(tr.best_path), tr.wait_count, task.w.item(), test["test_loss"], epochs_run, task.training,
           any(task.eval_modes)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_trainer_two_ranks_take_identical_decisions(tmp_path):
    """ranks with different validation shards must agree on val_loss (sample-weighted mean), LR schedule,
    early stopping epoch and the checkpoint that test() loads; validation runs in eval mode and training mode is restored."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, vl0, best0, wait0, w0, tl0, epochs, training0, evalmode0), (_, vl1, best1, wait1, w1, tl1, _, training1, evalmode1) = got
    assert vl0 == vl1 and tl0 == tl1 and best0 == best1 and wait0 == wait1 and w0 == w1  # one job, one set of decisions
    assert best0.endswith("best.ckpt")
    assert epochs is not None and epochs < 12  # the pooled val_loss stops improving -> both ranks stopped early, together
    assert training0 and training1 and not evalmode0 and not evalmode1
    # pooled loss = (2 * (w-1)^2 + 4 * (w-5)^2) / 6 at the restored best weights
    assert abs(tl0 - (2 * (w0 - 1) ** 2 + 4 * (w0 - 5) ** 2) / 6) < 1e-5  # the losses are fp32


def _one_rank_worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TVL_DIST_SINGLE_RANK_GROUP="1")
    torch.set_num_threads(2)
    assert not tdist.group_active()
    tdist.init_distributed("cpu")
    assert tdist.group_active() and tdist.world_size() == 1 and dist.get_backend() == "gloo"
    g = torch.Generator().manual_seed(5)
    a = torch.nn.Parameter(torch.randn(300, generator=g))
    b = torch.nn.Parameter(torch.randn(40, 20, generator=g))
    flat = tdist.FlatParams([a, b])
    ex = tdist.GradExchange(flat, bucket_bytes=1024)   # two buckets
    x = torch.randn(20, generator=g)
    ((a * 2.0).sum() + (b @ x).square().sum()).backward()
    scale = ex.finish()
    want_a, want_b = torch.full((300,), 2.0), 2.0 * (b.detach() @ x)[:, None] * x[None, :]
    tdist.barrier()
    q.put((scale, ex.launched_in_backward, torch.equal(a.grad, want_a), torch.allclose(b.grad, want_b, rtol=1e-6, atol=1e-6),
           tdist.reduce_sums([1.5, 2.5]), tdist.allgather_cat(torch.arange(3.0)).tolist()))
    dist.destroy_process_group()


def test_group_of_one_rank_runs_the_exchange_as_an_identity():
    """TVL_DIST_SINGLE_RANK_GROUP=1: a process group of a single rank (here gloo; on a GPU box RCCL, tests/test_train_gpu.py) -- the bucketed
    all-reduces are launched from the backward's hooks and waited for, barrier / metric syncs run, and nothing changes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(0, _free_port(), q))
    p.start()
    scale, launched, a_ok, b_ok, sums, gathered = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert scale == 1.0 and launched == 2 and a_ok and b_ok
    assert sums == [1.5, 2.5] and gathered == [0.0, 1.0, 2.0]
