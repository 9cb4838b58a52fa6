#!/usr/bin/env python
"""Time tvl_h2_pack (per-row and per-tensor mode) over the activation shapes of the CRIS step: GB/s of fp32 read + image written."""
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

hip.load()
shapes = [(21632, 512), (21632, 1024), (21632, 2048), (86528, 256), (86528, 512), (346112, 64), (346112, 128), (346112, 256), (5408, 2048)]
for per_row in (True, False):
    for M, K in shapes:
        x = torch.randn(M, K, device="cuda")
        for _ in range(3):
            hip.h2_pack(x, per_row=per_row)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            hip.h2_pack(x, per_row=per_row)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"{'row' if per_row else 'tensor'} M={M} K={K}: {us:8.1f} us  {M * K * 8 / us / 1e3:7.1f} GB/s (read once + write once)")
