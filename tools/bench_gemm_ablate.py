#!/usr/bin/env python
"""Timing-only ablations of the two-piece fp16 ring GEMM (needs a `make DIAG=1` library; results of variants != 0 are WRONG by design):
4 = no DMA inside the k-loop, 8 = every DMA re-reads slab 0 (operands L2-resident), 16 = no stores, 24 = 8 + 16.
Interleaved rounds in one process, plain fp32-output epilogue, the vision layer's shapes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

M = 15840
SHAPES = [("N=768  K=3072", 768, 3072), ("N=768  K=768 ", 768, 768), ("N=3072 K=768 ", 3072, 768), ("N=2304 K=768 ", 2304, 768)]
VARIANTS = [0, 4, 8, 16, 24]


def main():
    hip.load()
    torch.manual_seed(0)
    ops = {}
    for name, n, k in SHAPES:
        A = hip.h2_pack(torch.randn(M, k, device="cuda"), per_row=True)
        B = hip.weight_h2(torch.randn(n, k, device="cuda") * k**-0.5)
        ops[name] = (A, B, torch.empty(M, n, device="cuda"))
    times = {(s[0], v): [] for s in SHAPES for v in VARIANTS}
    for r in range(6):
        for name, n, k in SHAPES:
            A, B, C = ops[name]
            for v in VARIANTS:
                hip.set_tp3_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    hip.gemm_h2(A, B, out=C)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[(name, v)].append(e0.elapsed_time(e1) / 3)
    hip.set_tp3_variant(0)
    for name, n, k in SHAPES:
        fl = 2.0 * M * n * k
        row = []
        for v in VARIANTS:
            t = sorted(times[(name, v)])[len(times[(name, v)]) // 2]
            row.append(f"v{v:<2d} {t * 1e3:7.1f} us {fl / t / 1e9:6.1f} TF/s")
        print(f"{name}:  " + "  |  ".join(row))


if __name__ == "__main__":
    main()
