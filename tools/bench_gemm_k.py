#!/usr/bin/env python
"""GEMM time vs K at fixed M, N (split-bf16 default mode): separates the per-launch fixed cost from the k-loop rate."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402


TP3 = len(sys.argv) > 1 and sys.argv[1].startswith("tp3")  # tp3:<tile>:<variant>


def main():
    hip.load()
    if TP3:
        parts = sys.argv[1].split(":")
        hip.GEMM_TP3_TILE = int(parts[1]) if len(parts) > 1 else 0
        hip.GEMM_TP3_VARIANT = int(parts[2]) if len(parts) > 2 else hip.GEMM_TP3_VARIANT
    M = 15840
    for N in (768, 2304, 3072):
        rows = []
        for K in ((64, 128, 256, 512, 768, 1536, 3072, 6144) if TP3 else (32, 64, 128, 256, 512, 768, 1536, 3072, 6144)):
            A, B, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
            if TP3:
                At, Bt = hip.tp3_pack(A), hip.tp3_pack(B)
                gemm = lambda: hip.gemm_tp3(At, Bt, out=C)  # noqa: E731
            else:
                gemm = lambda: hip.gemm(hip.NT, M, N, K, A, K, B, K, C, N)  # noqa: E731
            for _ in range(3):
                gemm()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    gemm()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4 * 1e3)
            t = sorted(ts)[2]
            rows.append((K, t))
            print(f"N={N} K={K}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s  {'tp3' if TP3 else hip.gemm_kernel_key(hip.NT, M, N, True, 3, K)[17:30]}")
        (k0, t0), (k1, t1) = rows[-4], rows[-2]
        b = (t1 - t0) / (k1 - k0)
        print(f"   N={N}: slope {b*1e3:.1f} ns per k  => asymptotic {2.0*M*N/b/1e6:.1f} TF/s, intercept {t0 - b*k0:.1f} us")


if __name__ == "__main__":
    main()
