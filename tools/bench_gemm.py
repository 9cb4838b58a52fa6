#!/usr/bin/env python
"""GEMM micro-benchmark on the ViT-B/16 shapes of the hot path (M = 32*495 = 15840). Interleaved rounds, HIP events."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

M = 15840
SHAPES = [("qkv  NT", hip.NT, M, 2304, 768), ("out  NT", hip.NT, M, 768, 768), ("fc1  NT", hip.NT, M, 3072, 768),
          ("fc2  NT", hip.NT, M, 768, 3072), ("dz   NN", hip.NN, M, 3072, 768), ("dx2  NN", hip.NN, M, 768, 3072),
          ("do   NN", hip.NN, M, 768, 768), ("dx1  NN", hip.NN, M, 768, 2304), ("sq4k NT", hip.NT, 4096, 4096, 4096)]
if len(sys.argv) > 1 and sys.argv[1] != "f32":  # split-bf16 runs data gradients as NT over W^T
    SHAPES = [(n.replace("NN", "NT"), hip.NT, m, nn_, k) for n, _, m, nn_, k in SHAPES]


TP3 = len(sys.argv) > 1 and sys.argv[1].startswith("tp3")
H2 = len(sys.argv) > 1 and sys.argv[1].startswith("h2")   # h2 | h2:<tile>: two fp16 pieces, 3 MFMAs per product (csrc/gemm_h2.hip)
if H2:
    SHAPES += [("qkv* NT", hip.NT, M, 2304, 768), ("fc1* NT", hip.NT, M, 3072, 768)]
if TP3:  # the two GEMMs whose epilogue moves the most bytes, with their real epilogues (ops.EncoderLayerTp3Fn)
    SHAPES += [("fc1* NT", hip.NT, M, 3072, 768), ("dz*  NT", hip.NT, M, 3072, 768), ("out* NT", hip.NT, M, 768, 768)]


H2_TILE = int(sys.argv[1].split(":")[1]) if H2 and ":" in sys.argv[1] else 0


def main():
    hip.load()
    if len(sys.argv) > 1:
        if H2:
            pass
        elif sys.argv[1].startswith("tp3"):  # tp3 | tp3:<tile>:<variant>
            parts = sys.argv[1].split(":")
            hip.GEMM_TP3_TILE = int(parts[1]) if len(parts) > 1 else 0
            hip.GEMM_TP3_VARIANT = int(parts[2]) if len(parts) > 2 else hip.GEMM_TP3_VARIANT
        else:
            hip.set_gemm_mode(sys.argv[1])
    print("mode", hip.GEMM_MODE)
    torch.manual_seed(0)
    bufs = {}
    for name, layout, m, n, k in SHAPES:
        A = torch.randn(m, k, device="cuda")
        B = torch.randn(n, k, device="cuda") if layout == hip.NT else torch.randn(k, n, device="cuda")
        if os.environ.get("TVL_BENCH_ZERO"):  # DVFS probe: zero operands let the chip hold a higher clock (guide rule 25)
            A.zero_(), B.zero_()
        hip.mark_frozen(B)
        C = torch.empty(m, n, device="cuda")
        if H2:
            A, B = hip.h2_pack(A, True), hip.h2_pack(B, False)
            B._alpha = None
        if TP3:  # operands handed over pre-tiled (activations by their producer, weights once)
            A, B = hip.tp3_pack(A), hip.tp3_pack(B)
        extra = {}
        if H2 and name.startswith("qkv*"):
            extra = dict(want_f32=False, out_tp3=hip.Tp3(m, n, "cuda"), bias=torch.randn(n, device="cuda"))
        elif name.startswith("fc1*"):
            extra = dict(want_f32=False, out_tp3=hip.Tp3(m, n, "cuda"), bias=torch.randn(n, device="cuda"), act=hip.ACT_QUICK_GELU, pre_out=C)
        elif name.startswith("dz*"):
            extra = dict(want_f32=False, out_tp3=hip.Tp3(m, n, "cuda"), dact=hip.ACT_QUICK_GELU, dact_aux=torch.randn(m, n, device="cuda"))
        elif name.startswith("out*"):
            extra = dict(out=C, bias=torch.randn(n, device="cuda"), residual=torch.randn(m, n, device="cuda"))
        elif TP3 or H2:
            extra = dict(out=C)
        bufs[name] = (A, B, C, extra)
    rounds = 5
    times = {s[0]: [] for s in SHAPES}
    for r in range(rounds + 1):
        for name, layout, m, n, k in SHAPES:
            A, B, C, extra = bufs[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                if H2:
                    hip.gemm_h2(A, B, tile_m=H2_TILE, **extra)
                elif TP3:
                    hip.gemm_tp3(A, B, **extra)
                else:
                    hip.gemm(layout, m, n, k, A, A.shape[1], B, B.shape[1], C, n)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1) / 3)
    tot_f = tot_t = 0
    for name, layout, m, n, k in SHAPES:
        t = sorted(times[name])[len(times[name]) // 2]
        fl = 2.0 * m * n * k
        if not name.startswith("sq"):
            tot_f += fl
            tot_t += t
        key = f"gemm_h2<tile {H2_TILE or 'auto'}>" if H2 else f"gemm_tp3<{hip.tp3_tile(m, n)},256,{hip.GEMM_TP3_VARIANT}>" if TP3 else hip.gemm_kernel_key(layout, m, n, True, hip._NSPLIT.get(hip.GEMM_MODE, 0), k)
        print(f"{name} M={m} N={n} K={k}: {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF/s  {key}")
    print(f"layer GEMMs total: {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
