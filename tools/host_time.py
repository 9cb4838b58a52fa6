#!/usr/bin/env python
"""Host enqueue time vs step time: how long Python needs to issue one train step's launches (no synchronisation inside the step),
next to the synchronised step time.  python tools/host_time.py [vpt|maple|cris]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "vpt"
dev = torch.device("cuda:0")
size = 416 if wl == "cris" else 352
module, opt = {"cris": bench.build_cris_module, "maple": bench.build_maple_module}.get(wl, bench.build_module)(dev)
batch = bench.make_batch(32, size, 1, dev)


def step():
    opt.zero_grad()
    module.training_step(batch).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
host.sort(), total.sort()
print(f"{wl}: host enqueue {host[len(host) // 2]:.2f} ms, step (enqueue + drain) {total[len(total) // 2]:.2f} ms")
