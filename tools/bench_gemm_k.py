#!/usr/bin/env python
"""GEMM time vs K at fixed M, N (split-bf16 default mode): separates the per-launch fixed cost from the k-loop rate."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402


def main():
    hip.load()
    M = 15840
    for N in (768, 2304, 3072):
        rows = []
        for K in (32, 64, 128, 256, 512, 768, 1536, 3072, 6144):
            A, B, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
            for _ in range(3):
                hip.gemm(hip.NT, M, N, K, A, K, B, K, C, N)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    hip.gemm(hip.NT, M, N, K, A, K, B, K, C, N)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4 * 1e3)
            t = sorted(ts)[2]
            rows.append((K, t))
            print(f"N={N} K={K}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s  {hip.gemm_kernel_key(hip.NT, M, N, True, 3, K)[17:30]}")
        (k0, t0), (k1, t1) = rows[5], rows[7]
        b = (t1 - t0) / (k1 - k0)
        print(f"   N={N}: slope {b*1e3:.1f} ns per k  => asymptotic {2.0*M*N/b/1e6:.1f} TF/s, intercept {t0 - b*k0:.1f} us")


if __name__ == "__main__":
    main()
