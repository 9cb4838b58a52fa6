"""Capture forward + backward of one train step into a hipGraph (torch.cuda.graph) and replay it: equality with the eager step, timing.

    python -X faulthandler tools/graph_step.py --workload vpt [--steps 30]

The optimiser update stays outside the graph (its bias correction takes the step count as a launch argument).  The slot pool of the tagged
atomicMax hand-over (hip._max_slot) is cleared inside the captured region: a replay re-uses the slots and tags of the capture.
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
    args = ap.parse_args()

    from tunevlseg_amd import hip

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip.load()
    cris, maple = args.workload == "cris", args.workload == "maple"
    # Everything on ONE non-default stream, module construction included: the AccumulateGrad nodes of the flat parameter views are created (and kept
    # alive by the gradient-exchange hooks) when the optimiser is built; autograd runs each on the stream it was created on, and a node of the
    # default stream drags that stream into the capture, where nothing joins it back (hipStreamEndCapture then dies).
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    module, opt = bench.build_cris_module(device) if cris else (bench.build_maple_module(device) if maple else bench.build_module(device))
    batch = bench.make_batch(args.batch, 416 if cris else 352, 100, device, pad_id=0 if cris else 1)

    def fwd_bwd():
        opt.zero_grad()
        loss = module.training_step(batch, 0)
        loss.backward()
        return loss

    def eager_step():
        loss = fwd_bwd()
        opt.step()
        return loss

    print("warm-up (eager, on the capture stream) ...", flush=True)
    for _ in range(4):
        eager_step()
    torch.cuda.synchronize()

    snap = (opt.flat.data.clone(), opt.m.clone(), opt.v.clone(), opt.step_count)

    def restore():
        opt.flat.data.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2]); opt.step_count = snap[3]

    # eager reference: 3 steps from the snapshot
    ref = []
    for _ in range(3):
        l = eager_step()
        ref.append((l.item(), opt.flat.grad.clone()))
    restore()
    torch.cuda.synchronize()

    print("capture ...", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for pool in hip._MAX_SLOTS.values():
            pool[0].zero_()
        static_loss = fwd_bwd()
    torch.cuda.synchronize()
    print("captured; replay ...", flush=True)
    ok = True
    for i in range(3):
        g.replay()
        opt.step()
        torch.cuda.synchronize()
        dl = abs(static_loss.item() - ref[i][0])
        dg = (opt.flat.grad - ref[i][1]).abs().max().item() / ref[i][1].abs().max().item()
        print(f"replay {i}: loss {static_loss.item():.7f} (eager {ref[i][0]:.7f}, diff {dl:.2e}), gradient rel diff {dg:.2e}", flush=True)
        ok = ok and dl <= 1e-6 and dg <= 1e-5

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3

    def replay_step():
        g.replay()
        opt.step()

    for name, fn in (("eager", eager_step), ("graph", replay_step), ("eager", eager_step), ("graph", replay_step)):
        fn(); fn()
        host, step = timed(fn, args.steps)
        print(f"{args.workload} {name}: host enqueue {host:.2f} ms/step, step {step:.2f} ms ({args.batch / step * 1e3:.1f} img/s)", flush=True)
    print("EQUAL" if ok else "DIFFERENT")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
