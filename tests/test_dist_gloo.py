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
    res = (flat.grad * scale).clone(), d.compute(), j.compute()
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
