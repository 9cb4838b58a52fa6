"""N ranks (torch.distributed.run, TVL_DIST_BACKEND=gloo to share one device), the bench's VPT step: per step and rank, is the local gradient
finite before the exchange, after it, and is the loss finite?  Prints the first bad (step, rank, stage).

    TVL_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 tools/dist_nan_hunt.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from tunevlseg_amd import dist as tdist
    from tunevlseg_amd import hip

    rank, local_rank, world = tdist.init_distributed("cuda")
    if os.environ.get("TVL_DIST_BACKEND") == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    hip.load()
    module, opt = bench.build_module(device)
    batch = bench.make_batch(32, 352, 100 + rank, device)
    ex = opt.exchange
    log = []
    orig = ex._launch

    def launch(b):
        s, e, _ = ex.buckets[b]
        log.append(("pre", torch.isfinite(ex.flat.grad[s:e]).all(), ex.flat.grad[s:e].abs().max()))
        orig(b)

    ex._launch = launch
    steps = int(os.environ.get("STEPS", "10"))
    bad = None
    if os.environ.get("ASYNC"):   # as bench.py runs: no host synchronisation inside the loop, the flags stay on the device until the end
        flags = torch.zeros(steps, 4, device=device)
        for i in range(steps):
            log.clear()
            opt.zero_grad()
            loss = module.training_step(batch, 0)
            loss.backward()
            pre_ok = torch.stack([a for _, a, _ in log]).all() if log else torch.ones((), device=device, dtype=torch.bool)
            opt.step()
            flags[i, 0] = pre_ok
            flags[i, 1] = torch.isfinite(opt.flat.grad).all()
            flags[i, 2] = torch.isfinite(opt.flat.data).all()
            flags[i, 3] = torch.isfinite(loss.detach())
        torch.cuda.synchronize()
        f = flags.cpu()
        for i in range(steps):
            if not f[i].all():
                print(f"rank {rank} ASYNC step {i}: pre-exchange grad finite {bool(f[i, 0])}, post-exchange {bool(f[i, 1])}, params {bool(f[i, 2])}, loss {bool(f[i, 3])}", flush=True)
                bad = i if bad is None else bad
        print(f"rank {rank}: ASYNC first bad step {bad}; launched in backward {ex.launched_in_backward}", flush=True)
        torch.distributed.barrier()
        return 0
    for i in range(steps):
        log.clear()
        opt.zero_grad()
        loss = module.training_step(batch, 0)
        loss.backward()
        opt.step()
        post = torch.isfinite(opt.flat.grad).all().item()
        pmax = opt.flat.grad.abs().max().item()
        pre = [(bool(a.item()), float(m.item())) for _, a, m in log]
        l = loss.item()
        par = torch.isfinite(opt.flat.data).all().item()
        print(f"rank {rank} step {i}: loss {l:.6f} pre-exchange finite/max {pre} post-exchange finite {post} max {pmax:.3e} params finite {par}", flush=True)
        if bad is None and not (post and par and l == l):
            bad = i
    print(f"rank {rank}: first bad step {bad}", flush=True)
    torch.distributed.barrier()
    return 0


if __name__ == "__main__":
    sys.exit(main())
