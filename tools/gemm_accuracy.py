#!/usr/bin/env python
"""Accuracy probe of the GEMM arithmetic families against a float64 product: signed mean and RMS of the relative error, on
all-positive operands (a biased accumulation shows as a signed mean) and on zero-mean operands."""
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
        A = torch.rand(M, K) if kind == "uniform01" else torch.randn(M, K)
        B = torch.rand(N, K) if kind == "uniform01" else torch.randn(N, K)
        ref = A.double() @ B.double().T
        scale = (A.double().abs() @ B.double().abs().T)  # sum |a||b|: the natural error scale of a dot product
        Ad, Bd = A.cuda(), B.cuda()
        outs = {}
        for mode in ("f32", "bf16x6"):
            hip.set_gemm_mode(mode)
            C = torch.empty(M, N, device="cuda")
            hip.gemm(hip.NT, M, N, K, Ad, K, Bd, K, C, N)
            outs[mode] = C.cpu().double()
        hip.set_gemm_mode("bf16x6")
        C, _ = hip.gemm_tp3(hip.tp3_pack(Ad), hip.tp3_pack(Bd))
        outs["tp3"] = C.cpu().double()
        C, _ = hip.gemm_h2(hip.h2_pack(Ad, True), hip.h2_pack(Bd, False))
        outs["h2 (fp16 x 3)"] = C.cpu().double()
        Bs = Bd * 0.02   # weight-like magnitudes: the tensor scale has to lift them
        C, _ = hip.gemm_h2(hip.h2_pack(Ad * 37.0, True), hip.h2_pack(Bs, False))
        outs["h2 scaled in"] = C.cpu().double() / (37.0 * 0.02)
        outs["torch_cpu_f32"] = (A @ B.T).double()
        for k, v in outs.items():
            e = (v - ref) / scale
            print(f"K={K} {kind:9s} {k:14s} mean(err/sum|a||b|) {e.mean().item():+.3e}  rms {e.pow(2).mean().sqrt().item():.3e}  max {e.abs().max().item():.3e}")
