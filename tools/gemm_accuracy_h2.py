#!/usr/bin/env python
"""Accuracy of the two h2 ring-GEMM generations (32x32x16 vs 16x16x32 fp16 MFMA) against float64: signed mean (a truncating adder
shows as a bias) and RMS of err / sum|a||b|, (a) on operands that ARE fp16 numbers (second pieces zero: isolates the instruction's
own accumulation of exact products) and (b) on full fp32 operands; all-positive and zero-mean data."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

hip.load()
torch.manual_seed(0)
M, N = 1024, 768
for K in (768, 3072):
    for kind in ("uniform01", "normal"):
        for exact16 in (True, False):
            A = torch.rand(M, K) if kind == "uniform01" else torch.randn(M, K)
            B = torch.rand(N, K) if kind == "uniform01" else torch.randn(N, K)
            if exact16:   # fp16-representable values whose row / tensor scale is exactly 1 * 2^k: the second piece is zero
                A, B = A.half().float(), B.half().float()
            ref = A.double() @ B.double().T
            scale = A.double().abs() @ B.double().abs().T
            Ah, Bh = hip.h2_pack(A.cuda(), True), hip.h2_pack(B.cuda(), False)
            row = []
            for name, tile in (("32x32x16", 1920), ("16x16x32", 1926)):
                C, _ = hip.gemm_h2(Ah, Bh, tile_m=tile)
                e = (C.cpu().double() - ref) / scale
                row.append(f"{name}: mean {e.mean().item():+.2e} rms {e.pow(2).mean().sqrt().item():.2e} max {e.abs().max().item():.2e}")
            e = ((A @ B.T).double() - ref) / scale
            print(f"K={K:4d} {kind:9s} {'fp16-exact operands' if exact16 else 'fp32 operands      '}  " + "  |  ".join(row) + f"  |  torch cpu f32: rms {e.pow(2).mean().sqrt().item():.2e}")
