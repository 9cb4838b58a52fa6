#!/usr/bin/env python
"""Attention micro-benchmark at the vision-tower shape (B=32, T=495, H=12, d_h=64)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402


def main():
    hip.load()
    B, T, H, dh = 32, 495, 12, 64
    D = H * dh
    torch.manual_seed(0)
    qkv = torch.randn(B * T, 3 * D, device="cuda")
    d_o = torch.randn(B * T, D, device="cuda")
    o, lse = hip.attn_fwd_packed(qkv, B, T, H, dh, dh ** -0.5)

    def timeit(fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    tf = timeit(lambda: hip.attn_fwd_packed(qkv, B, T, H, dh, dh ** -0.5))
    tb = timeit(lambda: hip.attn_bwd_packed(qkv, o, d_o, lse, B, T, H, dh, dh ** -0.5))
    fl = 4.0 * T * T * dh * H * B
    qkv_t = hip.tp3_pack(qkv)
    tf3 = timeit(lambda: hip.attn_tp3_fwd(qkv_t, B, T, H, dh ** -0.5))
    print(f"fwd on the tp3 QKV image (LDS-DMA key tiles): {tf3*1e3:.1f} us  {fl/tf3/1e9:.1f} TF/s")
    print(f"fwd {tf*1e3:.1f} us  {fl/tf/1e9:.1f} TF/s   bwd {tb*1e3:.1f} us  {2.5*fl/tb/1e9:.1f} TF/s (algorithmic 10 T^2 d)")


if __name__ == "__main__":
    main()
