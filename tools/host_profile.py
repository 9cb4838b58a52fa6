#!/usr/bin/env python
"""Where the host spends a train step's enqueue time: cProfile over 5 steps (no synchronisation inside), top functions by own time.
python tools/host_profile.py [vpt|maple|cris]"""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "vpt"
dev = torch.device("cuda:0")
module, opt = {"cris": bench.build_cris_module, "maple": bench.build_maple_module}.get(wl, bench.build_module)(dev)
batch = bench.make_batch(32, 416 if wl == "cris" else 352, 1, dev)


def step():
    opt.zero_grad()
    module.training_step(batch).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
