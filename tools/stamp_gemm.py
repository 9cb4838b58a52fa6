#!/usr/bin/env python
"""Diagnostic: where a gemm_tp3 workgroup spends its time (variant bit 5 stamps s_memrealtime / s_memtime at kernel entry,
after the prologue, after the k-loop and after the epilogue's stores have drained).  Timing shares only -- never quote the
run time of this build."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

hip.load()
hip.GEMM_TP3_TILE, hip.GEMM_TP3_VARIANT = 192, 0
STAMP = int(sys.argv[1]) if len(sys.argv) > 1 else 33  # 32: generic epilogue, 33: specialised (fp32 out only)
M = 15840
for N, K in ((768, 768), (768, 3072), (2304, 768)):
    A, B = hip.tp3_pack(torch.randn(M, K, device="cuda")), hip.tp3_pack(torch.randn(N, K, device="cuda"))
    C = torch.empty(M, N, device="cuda")
    nwg = ((M + 191) // 192) * (N // 256)
    dbg = torch.zeros(nwg * 12 * 2, device="cuda", dtype=torch.float32)  # 8 uint64 per workgroup
    hip.GEMM_TP3_VARIANT = 0
    for _ in range(20):  # reach the steady-state clock with the production variant
        hip.gemm_tp3(A, B, out=C)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    args = hip.GemmTp3Args(M, N, K, A.buf.data_ptr(), A.rows, B.buf.data_ptr(), B.rows, C.data_ptr(), N, None, None, None, 0, 0,
                           dbg.data_ptr(), None, 0, 0, 1.0, 192, STAMP)
    import ctypes
    hip._call("tvl_gemm_tp3", ctypes.byref(args))
    e1.record()
    torch.cuda.synchronize()
    d = dbg.view(torch.int64).view(nwg, 12).cpu()
    rt, ck = d[:, :6].double(), d[:, 6:].double()
    t0 = rt[:, 0].min()
    us = (rt - t0) / 100.0  # 100 MHz -> us
    print(f"N={N} K={K}: kernel {e0.elapsed_time(e1)*1e3:.1f} us (events), {nwg} workgroups")
    for i, name in ((0, "entry"), (1, "prologue done"), (2, "k-loop done"), (4, "WG barrier"), (5, "stores issued"), (3, "stores drained")):
        print(f"   {name:15s} median {us[:, i].median():7.2f} us   min {us[:, i].min():7.2f}   max {us[:, i].max():7.2f}")
    dclk = (ck[:, 2] - ck[:, 1]) / ((rt[:, 2] - rt[:, 1]) / 100.0) / 1e3
    print(f"   shader clock in the k-loop: median {dclk.median():.3f} GHz  min {dclk.min():.3f} max {dclk.max():.3f};  "
          f"cycles per slab: {((ck[:, 2] - ck[:, 1]) / (K // 16)).median():.0f}")
