#!/usr/bin/env python
"""LayerNorm -> tp3 micro-benchmark at the vision-tower shape (15,840 rows x 768): time and algorithmic HBM rate."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402


def main():
    hip.load()
    rows, cols = 32 * 495, 768
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device="cuda")
    dy = torch.randn(rows, cols, device="cuda")
    dres = torch.randn(rows, cols, device="cuda")
    g, b = torch.randn(cols, device="cuda"), torch.randn(cols, device="cuda")
    _, mean, rstd = hip.layernorm_fwd_tp3(x, g, b, 1e-5)

    def timeit(fn, n=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    n = rows * cols
    tf = timeit(lambda: hip.layernorm_fwd_tp3(x, g, b, 1e-5))
    tb = timeit(lambda: hip.layernorm_bwd_tp3(dy, x, g, mean, rstd, dres))
    bf, bb = n * (4 + 6), n * (4 * 3 + 4 + 6)   # fwd: read x, write tp3; bwd: read dy, x, dres, write dx fp32 + tp3
    print(f"ln_fwd_tp3 {tf*1e6:.1f} us  {bf/tf/1e12:.2f} TB/s algorithmic   ln_bwd_tp3 {tb*1e6:.1f} us  {bb/tb/1e12:.2f} TB/s algorithmic")
    _, mean, rstd = hip.layernorm_fwd_h2(x, g, b, 1e-5)
    tf = timeit(lambda: hip.layernorm_fwd_h2(x, g, b, 1e-5))
    tb = timeit(lambda: hip.layernorm_bwd_h2(dy, x, g, mean, rstd, dres))
    bf, bb = n * (4 + 4), n * (4 * 3 + 4 + 4)
    print(f"ln_fwd_h2  {tf*1e6:.1f} us  {bf/tf/1e12:.2f} TB/s algorithmic   ln_bwd_h2  {tb*1e6:.1f} us  {bb/tb/1e12:.2f} TB/s algorithmic")
    # wide rows (CRIS decoder FFN: LayerNorm(2048) over B * 676 rows)
    rows2, cols2 = 32 * 676, 2048
    x2, dy2 = torch.randn(rows2, cols2, device="cuda"), torch.randn(rows2, cols2, device="cuda")
    g2, b2 = torch.randn(cols2, device="cuda"), torch.randn(cols2, device="cuda")
    _, m2, r2 = hip.layernorm_fwd_h2(x2, g2, b2, 1e-5)
    tf = timeit(lambda: hip.layernorm_fwd_h2(x2, g2, b2, 1e-5))
    tb = timeit(lambda: hip.layernorm_bwd_h2(dy2, x2, g2, m2, r2, None))
    n2 = rows2 * cols2
    print(f"2048 cols: ln_fwd_h2 {tf*1e6:.1f} us  {n2*8/tf/1e12:.2f} TB/s   ln_bwd_h2 {tb*1e6:.1f} us  {n2*16/tb/1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
