#!/usr/bin/env python
"""Which torch ops (not our HIP library) launch kernels inside one train step, with the Python line that asked for them.

    python tools/glue_ops.py [vpt|maple|cris]
"""
import sys
from collections import Counter
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "vpt"
dev = torch.device("cuda:0")
if wl == "cris":
    module, opt = bench.build_cris_module(dev)
    batch = bench.make_batch(32, 416, 1, dev)
elif wl == "maple":
    module, opt = bench.build_maple_module(dev)
    batch = bench.make_batch(32, 352, 1, dev)
else:
    module, opt = bench.build_module(dev)
    batch = bench.make_batch(32, 352, 1, dev)


def step():
    opt.zero_grad()
    module.training_step(batch).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt, dur, where = Counter(), Counter(), {}
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cuda_time_total > 0 and not any(
            c.name.startswith("aten::") and c.cuda_time_total > 0 for c in ev.cpu_children):
        st = [s for s in (ev.stack or []) if "tunevlseg_amd" in s or "bench.py" in s]
        key = (ev.name, st[0].split("/root/repo/")[-1] if st else "?")
        cnt[key] += 1
        dur[key] += ev.cuda_time_total
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:45]:
    print(f"{n:4d} x {dur[key] / n:7.1f} us = {dur[key] / 1e3:6.3f} ms  {key[0]:28s} {key[1]}")
print("total", sum(cnt.values()), "launching aten ops,", round(sum(dur.values()) / 1e3, 3), "ms")
