#!/usr/bin/env python
"""Can the train step's forward + backward be captured in a HIP graph, and does replaying it change the step time?

    python tools/graph_step.py [--workload vpt|maple]

Captures `loss = training_step(batch); loss.backward()` (every launch goes through ctypes onto torch's current stream, which is the
capturing stream inside torch.cuda.graph); the fused AdamW stays outside (its bias corrections are host scalars that change every
step).  Prints eager and replayed ms/step and the two losses."""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("vpt", "maple"), default="vpt")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from tunevlseg_amd import hip

    hip.load()
    dev = torch.device("cuda", 0)
    module, opt = bench.build_maple_module(dev) if args.workload == "maple" else bench.build_module(dev)
    batch = bench.make_batch(32, 352, 100, dev)

    def fwd_bwd():
        loss = module.training_step(batch, 0)
        loss.backward()
        return loss

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        opt.zero_grad()
        fwd_bwd()
        opt.step()

    for _ in range(3):
        eager()
    t_eager = timed(eager, args.steps)
    host0 = time.perf_counter()
    opt.zero_grad()
    l_eager = fwd_bwd()
    host_enqueue = (time.perf_counter() - host0) * 1e3   # host time to enqueue forward + backward (no sync)
    opt.step()
    torch.cuda.synchronize()

    # capture (torch wants the warm-up on a side stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            opt.zero_grad(set_to_none=False)
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=False)
    try:
        with torch.cuda.graph(g):
            l_graph = fwd_bwd()
    except Exception as e:  # noqa: BLE001
        import traceback

        tb = traceback.extract_tb(e.__traceback__)
        print("capture stopped at: " + " <- ".join(f"{Path(f.filename).name}:{f.lineno} {f.name}" for f in reversed(tb[-8:])))
        print(f"capture FAILED: {type(e).__name__}: {str(e)[:300].splitlines()[0] if str(e) else repr(e)}")
        print(f"eager {t_eager:.3f} ms/step, host enqueue of forward + backward {host_enqueue:.2f} ms")
        return

    def replay():
        opt.zero_grad(set_to_none=False)
        g.replay()
        opt.step()

    for _ in range(2):
        replay()
    t_graph = timed(replay, args.steps)
    print(f"{args.workload}: eager {t_eager:.3f} ms/step (host enqueue of forward + backward {host_enqueue:.2f} ms), graph replay {t_graph:.3f} ms/step; "
          f"loss eager {float(l_eager):.6f}, captured {float(l_graph):.6f}")


if __name__ == "__main__":
    main()
