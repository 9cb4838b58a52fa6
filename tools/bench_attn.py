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
    import os
    qkv = torch.randn(B * T, 3 * D, device="cuda")
    d_o = torch.randn(B * T, D, device="cuda")
    if os.environ.get("TVL_BENCH_ZERO") == "1":   # all-zero operands: the same instruction stream at a fraction of the switching power
        qkv.zero_(); d_o.zero_()
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
    o_t, lse_t = hip.attn_tp3_fwd(qkv_t, B, T, H, dh ** -0.5)
    do_t = hip.tp3_pack(d_o)
    tb3 = timeit(lambda: hip.attn_tp3_bwd(qkv_t, o_t, do_t, lse_t, B, T, H, dh ** -0.5))
    print(f"bwd on tp3 QKV / O / dO images (dQ kernel incl. delta + dK/dV kernel): {tb3*1e3:.1f} us  {2.5*fl/tb3/1e9:.1f} TF/s")
    qh, dh_ = hip.h2_pack(qkv, per_row=False), hip.h2_pack(d_o, per_row=False)
    tfh = timeit(lambda: hip.attn_h2_fwd(qh, B, T, H, dh ** -0.5))
    oh, lseh = hip.attn_h2_fwd(qh, B, T, H, dh ** -0.5)
    tbh = timeit(lambda: hip.attn_h2_bwd(qh, oh, dh_, lseh, B, T, H, dh ** -0.5))
    print(f"two fp16 pieces (3 MFMAs per product): fwd {tfh*1e3:.1f} us  {fl/tfh/1e9:.1f} TF/s   bwd {tbh*1e3:.1f} us  {2.5*fl/tbh/1e9:.1f} TF/s")
    if "--diag" in sys.argv:
        o3 = hip.Tp3(B * T, D, qkv.device)
        lse3 = torch.empty(B, H, T, device="cuda")
        nwg = (T + 127) // 128 * H * B
        stamps = torch.zeros(nwg, 4, dtype=torch.int64, device="cuda")

        def diag(v):
            hip._call("tvl_attn_tp3_fwd_diag", qkv_t.buf.data_ptr(), o3.buf.data_ptr(), lse3.data_ptr(), B, H, T, dh ** -0.5, v, stamps.data_ptr())

        for v, what in ((0, "product"),  (2, "no static priority"), (1, "no DMA after the prologue")):
            t = timeit(lambda: diag(v))
            print(f"  variant {v:2d} ({what}): {t*1e3:.1f} us")
        for pv_ in (32, 34):
            ph = torch.zeros(nwg, 5, dtype=torch.int64, device="cuda")
            hip._call("tvl_attn_tp3_fwd_diag", qkv_t.buf.data_ptr(), o3.buf.data_ptr(), lse3.data_ptr(), B, H, T, dh ** -0.5, pv_, ph.data_ptr())
            torch.cuda.synchronize()
            m = ph.cpu().double().mean(0) / ((T + 31) // 32 + 1)
            print(f"  phase cycles per tile, wave 0 of every workgroup (variant {pv_}{' = no static priority' if pv_ & 2 else ''}): barrier wait {m[0]:.0f}, "
                  f"K reads + S MFMAs {m[1]:.0f}, softmax + split {m[2]:.0f}, P.V MFMAs {m[3]:.0f}, DMA issue + vm wait {m[4]:.0f}; sum {m.sum():.0f}")
        diag(16)
        torch.cuda.synchronize()
        raw = stamps.cpu()
        where = raw[:, 3] >> 40
        raw[:, 3] &= (1 << 40) - 1
        raw[:, 2] &= (1 << 40) - 1
        st = raw.double()
        cyc, wall = st[:, 1] - st[:, 0], (st[:, 3] - st[:, 2]) * 10.0   # wall clock ticks at 100 MHz -> ns
        t_start = (st[:, 2] - st[:, 2].min()) * 10.0 / 1e3
        span = (st[:, 3].max() - st[:, 2].min()) * 10.0
        ntile = (T + 31) // 32 + 1
        print(f"  stamps: workgroup duration {cyc.mean():.0f} cycles (min {cyc.min():.0f}, max {cyc.max():.0f}) / {wall.mean()/1e3:.1f} us "
              f"(min {wall.min()/1e3:.1f}, max {wall.max()/1e3:.1f}); clock {(cyc / wall).mean():.2f} GHz (min {(cyc / wall).min():.2f}, "
              f"max {(cyc / wall).max():.2f}); first start -> last end {span/1e3:.1f} us; per tile {cyc.mean() / ntile:.0f} cycles")
        first = t_start < 5.0
        print(f"  first round: {int(first.sum())} workgroups, {cyc[first].mean():.0f} cycles; later: {cyc[~first].mean():.0f} cycles")
        print("  start-time histogram (us):", torch.histc(t_start.float(), bins=8, min=0, max=float(span / 1e3)).int().tolist())
        # per CU: how many workgroups it ran, and how their durations compare
        cus = {}
        for w, c, f in zip(where.tolist(), cyc.tolist(), first.tolist()):
            cus.setdefault(w, []).append((c, f))
        n_first = [sum(1 for _, f in v if f) for v in cus.values()]
        print(f"  {len(cus)} distinct (xcc, se, sh, cu); first-round workgroups per CU: " + str({k: n_first.count(k) for k in sorted(set(n_first))})
              + "; total per CU: " + str({k: [len(v) for v in cus.values()].count(k) for k in sorted({len(v) for v in cus.values()})}))
        by_n = {}
        for v in cus.values():
            nf = sum(1 for _, f in v if f)
            by_n.setdefault(nf, []).extend(c for c, f in v if f)
        print("  first-round duration by co-residents:", {k: f"{sum(v)/len(v):.0f}" for k, v in sorted(by_n.items())})
        xcc = {}
        for w, c in zip(where.tolist(), cyc.tolist()):
            xcc.setdefault(w >> 8, []).append(c)
        print("  mean cycles by XCC:", {k: f"{sum(v)/len(v):.0f} (n={len(v)})" for k, v in sorted(xcc.items())})
    print(f"fwd {tf*1e3:.1f} us  {fl/tf/1e9:.1f} TF/s   bwd {tb*1e3:.1f} us  {2.5*fl/tb/1e9:.1f} TF/s (algorithmic 10 T^2 d)")


if __name__ == "__main__":
    main()
