"""Experiment: one train step as a hipGraph whose two half-batches run as parallel branches (two streams forked inside the capture), against
the one-branch graph of the whole batch.  Question: do two de-phased kernel streams fill the prologue / epilogue bubbles of the ring GEMMs
 now that no host has to interleave two enqueues?

    python tools/graph_two_halves.py --workload vpt
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vpt", choices=("vpt", "maple", "cris"))
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="", help="capture only this variant (whole | halves)")
    ap.add_argument("--bwd-on-main", action="store_true", help="call the second half's backward with the main stream current")
    args = ap.parse_args()

    from tunevlseg_amd import hip
    from tunevlseg_amd.graph import use_private_stream

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip.load()
    s = use_private_stream(device)
    s2 = torch.cuda.Stream()
    cris, maple = args.workload == "cris", args.workload == "maple"
    module, opt = bench.build_cris_module(device) if cris else (bench.build_maple_module(device) if maple else bench.build_module(device))
    batch = bench.make_batch(args.batch, 416 if cris else 352, 100, device, pad_id=0 if cris else 1)
    h = args.batch // 2
    half1 = {k: v[:h].contiguous() for k, v in batch.items()}
    half2 = {k: v[h:].contiguous() for k, v in batch.items()}

    def whole():
        opt.zero_grad()
        loss = module.training_step(batch, 0)
        loss.backward()
        return loss

    def halves():
        opt.zero_grad()
        s2.wait_stream(s)
        with torch.cuda.stream(s2):
            l2 = module.training_step(half2, 0)
        l1 = module.training_step(half1, 0)
        (l1 * 0.5).backward()
        if args.bwd_on_main:
            (l2 * 0.5).backward()
        else:
            with torch.cuda.stream(s2):
                (l2 * 0.5).backward()
        s.wait_stream(s2)
        return (l1.detach() + l2.detach()) * 0.5

    for fn in (whole, halves, whole, halves):   # eager warm-up of both shapes
        fn()
        opt.step()
    torch.cuda.synchronize()
    snap = (opt.flat.data.clone(), opt.m.clone(), opt.v.clone(), opt.step_count)

    params = [p for p in module.parameters() if p.requires_grad]

    class Pair:
        """Two graphs, one per half-batch, each captured like the whole step (one fork-free capture per stream; a single capture with both halves as
        branches dies in hipStreamEndCapture).  Half 1 accumulates into the flat gradient as usual; half 2 returns its gradients (torch.autograd.grad: no
        AccumulateGrad, no shared buffer written by two streams) and they are added after the join."""

        def __init__(self):
            self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1, stream=s):
                for pool in hip._MAX_SLOTS.values():
                    pool[0].zero_()
                opt.zero_grad()
                self.l1 = module.training_step(half1, 0)
                (self.l1 * 0.5).backward()
            with torch.cuda.graph(self.g2, stream=s2):
                self.l2 = module.training_step(half2, 0)
                self.gs = torch.autograd.grad(self.l2 * 0.5, params)
            self.loss = torch.zeros((), device=device)

        def replay(self):
            s2.wait_stream(s)
            self.g1.replay()
            with torch.cuda.stream(s2):
                self.g2.replay()
            s.wait_stream(s2)
            for p, g in zip(params, self.gs):
                p.grad.add_(g)
            self.loss.copy_((self.l1.detach() + self.l2.detach()) * 0.5)

    graphs = {}
    if args.only != "halves":
        print("capture whole", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for pool in hip._MAX_SLOTS.values():
                pool[0].zero_()
            loss = whole()
        graphs["whole"] = (g, loss)
    if args.only != "whole":
        print("capture halves (two graphs)", flush=True)
        pair = Pair()
        graphs["halves"] = (pair, pair.loss)
    torch.cuda.synchronize()

    grads = {}
    for name, (g, loss) in graphs.items():
        opt.flat.data.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2]); opt.step_count = snap[3]
        g.replay()
        torch.cuda.synchronize()
        grads[name] = (loss.item(), opt.flat.grad.clone())
    if len(grads) == 2:
        dl = abs(grads["whole"][0] - grads["halves"][0])
        dg = (grads["whole"][1] - grads["halves"][1]).abs().max().item() / grads["whole"][1].abs().max().item()
        print(f"loss whole {grads['whole'][0]:.7f} halves {grads['halves'][0]:.7f} (diff {dl:.2e}); gradient rel diff {dg:.2e}", flush=True)

    for rep in range(3):
        for name, (g, _) in graphs.items():
            for _ in range(3):
                g.replay(); opt.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g.replay(); opt.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            print(f"{args.workload} {name}: {ms:.2f} ms/step ({args.batch / ms * 1e3:.1f} img/s)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
