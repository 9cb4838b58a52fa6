#!/usr/bin/env python
"""3x3 convs of the CRIS step on tvl_conv3x3_h2, one shape at a time, 192- against 256-row tiles (hip.CONV_TILE).

    python tools/bench_conv.py [--reps 5]
"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tunevlseg_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
# (B, H, W, C_in, N): tools/gemm_shapes.py cris
SHAPES = [(32, 104, 104, 512, 256), (32, 104, 104, 256, 512), (32, 52, 52, 512, 512), (32, 104, 104, 128, 128), (32, 52, 52, 512, 256),
          (32, 52, 52, 256, 256), (32, 26, 26, 512, 512), (32, 26, 26, 1024, 512), (32, 26, 26, 256, 256), (32, 13, 13, 512, 512), (32, 13, 13, 512, 1024)]
g = torch.Generator(device="cpu").manual_seed(0)
for B, H, W, Cc, N in SHAPES:
    M = B * H * W
    x = torch.randn(M, Cc, generator=g).to(dev)
    Wm = hip.mark_frozen((torch.randn(N, 9 * Cc, generator=g) / (9 * Cc) ** 0.5).to(dev))
    bias = torch.randn(N, generator=g).to(dev)
    packed = hip.h2_pack(x, per_row=False, zero_tail=True)
    y = torch.empty(M, N, device=dev)
    hip.CONV_TILE = 0
    line = f"M={M:7d} C={Cc:4d} N={N:4d} own rule {hip.conv_h2_tile(M, N)}:"
    ref = None
    for tile in (192, 256):
        hip.CONV_TILE = tile
        hip.conv3x3(None, B, H, W, Wm, bias, hip.ACT_RELU, out=y, packed=packed)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        elif not torch.equal(ref, y):
            line += f" [tiles differ by {(ref - y).abs().max().item():.2e}]"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            hip.conv3x3(None, B, H, W, Wm, bias, hip.ACT_RELU, out=y, packed=packed)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        line += f"  {tile}: {us:8.1f} us {2.0 * M * N * 9 * Cc / us / 1e6:6.1f} TF"
    print(line, flush=True)
