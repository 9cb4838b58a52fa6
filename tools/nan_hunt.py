"""Find the first step of the bench workload whose loss or gradient is not finite, and whether the same step reproduces it.

    python tools/nan_hunt.py --workload vpt --steps 80            # no host sync: losses collected on the device, first bad step reported
    python tools/nan_hunt.py --workload vpt --steps 80 --sync     # sync every step; on the first bad value re-run the step's forward/backward
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vpt", choices=("vpt", "maple", "cris"))
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--no-opt", action="store_true", help="never update the parameters: every step computes the same thing")
    args = ap.parse_args()

    from tunevlseg_amd import hip

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip.load()
    if os.environ.get("TVL_DIST_SINGLE_RANK_GROUP") == "1":   # the N > 1 exchange code on a one-rank RCCL group
        from tunevlseg_amd import dist as tdist

        os.environ.setdefault("WORLD_SIZE", "1")
        tdist.init_distributed("cuda")
    cris, maple = args.workload == "cris", args.workload == "maple"
    module, opt = bench.build_cris_module(device) if cris else (bench.build_maple_module(device) if maple else bench.build_module(device))
    batch = bench.make_batch(args.batch, 416 if cris else 352, 100, device, pad_id=0 if cris else 1)
    params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]

    losses = torch.zeros(args.steps, device=device)
    gnorm = torch.zeros(args.steps, device=device)
    for i in range(args.steps):
        opt.zero_grad()
        loss = module.training_step(batch, 0)
        if args.sync and not torch.isfinite(loss).item():
            again = module.training_step(batch, 0)
            print(f"step {i}: forward loss {loss.item()}; the same forward again: {again.item()}")
            return 1
        loss.backward()
        g = torch.stack([p.grad.float().norm() for _, p in params if p.grad is not None]).norm()
        losses[i], gnorm[i] = loss.detach(), g
        if args.sync and not torch.isfinite(g).item():
            bad = [n for n, p in params if p.grad is not None and not torch.isfinite(p.grad).all().item()]
            print(f"step {i}: loss {loss.item():.6f} finite, gradient not finite in {bad}")
            for n, p in params:
                if n in bad:
                    gg = p.grad
                    nb = (~torch.isfinite(gg)).sum().item()
                    print(f"   {n} {tuple(gg.shape)}: {nb} bad of {gg.numel()}; bad rows: {(~torch.isfinite(gg)).reshape(gg.shape[0], -1).any(1).nonzero().flatten().tolist()[:20]}")
            opt.zero_grad()
            l2 = module.training_step(batch, 0)
            l2.backward()
            bad2 = [n for n, p in params if p.grad is not None and not torch.isfinite(p.grad).all().item()]
            print(f"   the same step again: loss {l2.item():.6f}, gradient not finite in {bad2}")
            return 1
        if not args.no_opt:
            opt.step()
    torch.cuda.synchronize()
    l, g = losses.cpu(), gnorm.cpu()
    bad = (~(torch.isfinite(l) & torch.isfinite(g))).nonzero().flatten().tolist()
    print(f"{args.workload}: {args.steps} steps, loss[0] {l[0]:.6f} loss[-1] {l[-1]:.6f} max|g| {g[torch.isfinite(g)].max():.4g}; "
          f"first bad step: {bad[0] if bad else None} (loss {l[bad[0]].item() if bad else ''}, |g| {g[bad[0]].item() if bad else ''})")
    if bad:
        k = bad[0]
        print("   |g| around it:", [round(float(x), 5) for x in g[max(0, k - 5): k + 2]])
        print("   loss around it:", [round(float(x), 6) for x in l[max(0, k - 5): k + 2]])
    return 0


if __name__ == "__main__":
    sys.exit(main())
