#!/usr/bin/env python
"""Diagnostic (needs a `make DIAG=1` library): where a two-piece fp16 ring-GEMM workgroup spends its time.  Variant 32 stamps
s_memrealtime / s_memtime at kernel entry, after the prologue, after the k-loop, after the workgroup barrier, after the epilogue's
stores were issued and after they drained; 36 = the same with no DMA inside the k-loop.  Shares and the in-loop clock only -- never
quote the run time of this build."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

hip.load()
M = 15840
for tile in (192, 256):
    for N, K in ((768, 3072), (768, 768), (3072, 768)):
        A = hip.h2_pack(torch.randn(M, K, device="cuda"), per_row=True)
        B = hip.weight_h2(torch.randn(N, K, device="cuda") * K**-0.5)
        C = torch.empty(M, N, device="cuda")
        nwg = ((M + tile - 1) // tile) * (N // 256)
        for variant in (32, 36):
            dbg = torch.zeros(nwg * 12 * 2, device="cuda", dtype=torch.float32)
            hip.set_tp3_variant(0)
            for _ in range(20):   # reach the steady-state clock with the production kernel
                hip.gemm_h2(A, B, out=C, tile_m=tile)
            args = hip.GemmTp3Args(M, N, K, A.buf.data_ptr(), A.rows, B.buf.data_ptr(), B.rows, C.data_ptr(), N, None, None, None, 0, 0,
                                   dbg.data_ptr(), None, 0, 0, B.alpha(), tile, variant)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip._call("tvl_gemm_h2", ctypes.byref(args), A.inv_scale.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            d = dbg.view(torch.int64).view(nwg, 12).cpu()
            rt, ck = d[:, :6].double(), d[:, 6:].double()
            us = (rt - rt[:, 0].min()) / 100.0
            print(f"tile {tile} N={N} K={K} variant {variant}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us (events), {nwg} workgroups")
            for i, name in ((0, "entry"), (1, "prologue done"), (2, "k-loop done"), (4, "WG barrier"), (5, "stores issued"), (3, "stores drained")):
                print(f"   {name:15s} median {us[:, i].median():7.2f} us   min {us[:, i].min():7.2f}   max {us[:, i].max():7.2f}")
            dclk = (ck[:, 2] - ck[:, 1]) / ((rt[:, 2] - rt[:, 1]) / 100.0) / 1e3
            print(f"   shader clock in the k-loop: median {dclk.median():.3f} GHz  min {dclk.min():.3f} max {dclk.max():.3f};  "
                  f"cycles per 16-deep slab: {((ck[:, 2] - ck[:, 1]) / (K // 16)).median():.0f}  (pure MFMA issue: {3 * (tile // 64) * 2 * 2 * 32})")
