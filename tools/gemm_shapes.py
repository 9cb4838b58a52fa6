#!/usr/bin/env python
"""Which GEMM / conv shapes a workload's step launches, per kernel instantiation: wraps hip.gemm / hip.gemm_h2 / hip.conv3x3 for ONE eager step.

    python tools/gemm_shapes.py [vpt|maple|cris|denseclip]
"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from tunevlseg_amd import hip  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cris"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
module, opt = {"cris": bench.build_cris_module, "maple": bench.build_maple_module, "vpt": bench.build_module}[wl](dev)
batch = bench.make_batch(32, 416 if wl == "cris" else 352, 100, dev, pad_id=0 if wl == "cris" else 1)


def step():
    opt.zero_grad()
    loss = module.training_step(batch, 0)
    loss.backward()
    opt.step()


step()
log = collections.Counter()
orig_gemm, orig_h2, orig_conv = hip.gemm, hip.gemm_h2, hip.conv3x3


def gemm(layout, M, N, K, A, lda, B, ldb, Cout, ldc, **kw):
    takes = hip._gemm_takes_h2(layout, M, N, K, A, lda, B, ldb, Cout, ldc, kw.get("residual"), kw.get("ldr", 0), kw.get("pre_out"), kw.get("dact_aux"), kw.get("ld_aux", 0),
                               kw.get("alpha", 1.0), kw.get("a_map"), kw.get("c_map"))
    log[("gemm", "h2-packed" if takes else ("NT", "NN", "TN")[layout], M, N, K, bool(getattr(B, "_tvl_frozen", False)))] += 1
    return orig_gemm(layout, M, N, K, A, lda, B, ldb, Cout, ldc, **kw)


def conv(x2d, B, H, W, Wm, bias=None, act=0, stride=1, out=None, packed=None, x_relu_mask=None):
    Cc = packed.cols if packed is not None else x2d.shape[1]
    log[("conv3x3", "h2" if (packed is not None or hip.conv3x3_takes_h2(B * H * W, Cc, Wm, stride)) else "bf16s/f32", B * ((H - 1) // stride + 1) * ((W - 1) // stride + 1), Wm.shape[0], 9 * Cc, stride)] += 1
    return orig_conv(x2d, B, H, W, Wm, bias, act, stride, out, packed, x_relu_mask)


hip.gemm, hip.conv3x3 = gemm, conv
import tunevlseg_amd.cris_ops as C_  # noqa: E402

step()
torch.cuda.synchronize()
for k, n in sorted(log.items(), key=lambda kv: -kv[0][2] * kv[0][3] * kv[0][4] * kv[1]):
    fl = 2.0 * k[2] * k[3] * k[4] * n
    print(f"{n:3d} x {k}  {fl / 1e9:8.1f} GFLOP")
