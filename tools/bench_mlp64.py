#!/usr/bin/env python
"""The decoder's feed-forward block at the headline shape (M = 32 * 485 rows, 64 -> 2048 -> 64): one-kernel forward / backward
(csrc/mlp64.hip) against the op-by-op path it replaces (fc1 + relu, fc2 + residual, LayerNorm; LayerNorm', fc2 dgrad with the relu gate,
fc1 dgrad + residual).  HIP events, interleaved rounds, one process."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

M, F = 32 * 485, 2048
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 64, device=dev, generator=g)
W1, b1 = torch.randn(F, 64, device=dev, generator=g) * 0.1, torch.randn(F, device=dev, generator=g) * 0.1
W2, b2 = torch.randn(64, F, device=dev, generator=g) * 0.02, torch.randn(64, device=dev, generator=g) * 0.1
gm, bt = torch.ones(64, device=dev), torch.zeros(64, device=dev)
dout = torch.randn(M, 64, device=dev, generator=g)
W1t, W2t = W1.t().contiguous(), W2.t().contiguous()
w = hip.Mlp64Weights(W1, b1, W2)


def fused_fwd():
    return hip.mlp64_fwd(x, w, b1, b2, gm, bt, 1e-5)


def ops_fwd():
    u, z = hip.linear_fwd(x, W1, b1, act=hip.ACT_RELU, want_pre=True)
    t2 = hip.linear_fwd(u, W2, b2, residual=x)
    out, m, r = hip.layernorm_fwd(t2, gm, bt, 1e-5, want_stats=True)
    return out, t2, m, r, z


out, t2, m, r = fused_fwd()
_, t2o, mo, ro, z = ops_fwd()


def fused_bwd():
    return hip.mlp64_bwd(dout, x, t2, m, r, w, b1, gm)


def ops_bwd():
    dt2 = hip.layernorm_bwd(dout, t2o, gm, mo, ro)
    dz = hip.linear_dgrad(dt2, W2, dact=hip.ACT_RELU, dact_aux=z, Wt=W2t)
    return hip.linear_dgrad(dz, W1, residual=dt2, Wt=W1t)


print("max |fused - ops| fwd", (out - ops_fwd()[0]).abs().max().item(), "bwd", (fused_bwd() - ops_bwd()).abs().max().item())
fns = {"fused fwd": fused_fwd, "ops fwd": ops_fwd, "fused bwd": fused_bwd, "ops bwd": ops_bwd}
tot = {k: 0.0 for k in fns}
R, N = 5, 20
for rnd in range(R + 1):
    for k, f in fns.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            f()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            tot[k] += e0.elapsed_time(e1) / N
fl = {"fused fwd": 4.0 * M * F * 64, "ops fwd": 4.0 * M * F * 64, "fused bwd": 6.0 * M * F * 64, "ops bwd": 4.0 * M * F * 64}
for k in fns:
    us = tot[k] / R * 1e3
    print(f"{k:10s} {us:8.1f} us  {fl[k] / us / 1e6:7.1f} TFLOP/s (fp32-equivalent, recompute counted for the fused backward)")
