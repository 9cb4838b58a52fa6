#!/usr/bin/env python
"""The eight GEMMs of one ViT-B/16 vision layer (M = 32 * 495 = 15840) with their REAL epilogues (ops.EncoderLayerTp3Fn), per row tile.

    python tools/bench_layer_gemms.py [tiles=0,128,...] [M=15840] [rounds=7]

Interleaved rounds in ONE process (guide §5.4 rule 24), HIP events around 3 back-to-back launches, median.  Tile 0 = the library's own
choice (256 / 192 rows, 8-wave workgroups); 128 = the 4-wave workgroups, two per CU (csrc/gemm_h2_w4.hip).  Before timing, every
tile's outputs are compared with tile 0's: same k order and piece order per element, so the results must agree bit for bit."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
TILES = [int(t) for t in args.get("tiles", "0,128").split(",")]   # a negative tile code = the same tile WITHOUT the persistent tile walk
M = int(args.get("M", 15840))
ROUNDS = int(args.get("rounds", 7))
D, F = 768, 3072


def main():
    hip.load()
    torch.manual_seed(0)
    dev = "cuda"
    rn = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    W = {k: hip.weight_h2(rn(n, kk) * kk**-0.5) for k, (n, kk) in dict(wqkv=(3 * D, D), wo=(D, D), w1=(F, D), w2=(D, F), w2_t=(F, D), w1_t=(D, F), wo_t=(D, D)).items()}
    x = hip.h2_pack(rn(M, D), per_row=True, want_norm=True)
    o = hip.h2_pack(rn(M, D), per_row=False)
    a = hip.h2_pack(rn(M, F), per_row=True, want_norm=True)
    res, z = rn(M, D), rn(M, F)
    bq, bo, b1 = rn(3 * D), rn(D), rn(F)
    zout = torch.empty(M, F, device=dev)

    calls = {
        "qkv  385 N=2304 K=768 ": lambda t: hip.gemm_h2(x, W["wqkv"], want_f32=False, want_h2=True, out_per_tensor=True, out_add=4.0, bias=bq, tile_m=abs(t), persistent=t >= 0),
        "out  163 N=768  K=768 ": lambda t: hip.gemm_h2(o, W["wo"], bias=bo, residual=res, tile_m=abs(t)),
        "fc1  405 N=3072 K=768 ": lambda t: hip.gemm_h2(x, W["w1"], want_f32=False, want_h2=True, out_add=4.0, bias=b1, act=hip.ACT_QUICK_GELU, pre_out=zout, tile_m=abs(t), persistent=t >= 0),
        "fc2  163 N=768  K=3072": lambda t: hip.gemm_h2(a, W["w2"], bias=bo, residual=res, tile_m=abs(t)),
        "dz   392 N=3072 K=768 ": lambda t: hip.gemm_h2(x, W["w2_t"], want_f32=False, want_h2=True, out_mul=1.125 * W["w2_t"]._bound, dact=hip.ACT_QUICK_GELU, dact_aux=z, tile_m=abs(t), persistent=t >= 0),
        "dx2  160 N=768  K=3072": lambda t: hip.gemm_h2(a, W["w1_t"], tile_m=abs(t)),
        "do   384 N=768  K=768 ": lambda t: hip.gemm_h2(x, W["wo_t"], want_f32=False, want_h2=True, out_per_tensor=True, tile_m=abs(t)),
    }
    flops = {k: 2.0 * M * int(k.split("N=")[1].split()[0]) * int(k.split("K=")[1]) for k in calls}

    def result(out):
        cf, ch = out
        parts = []
        if cf is not None:
            parts.append(cf)
        if ch is not None:
            parts += [ch.float(), ch.inv_scale]
        return parts

    ref = {k: result(f(2560 if "N=3072" in k else 1920)) for k, f in calls.items()}   # the 32x32x16 generation (scratch epilogue): independent code
    for t in TILES:
        for k, f in calls.items():
            got = result(f(t))
            same = all(torch.equal(g_, r_) for g_, r_ in zip(got, ref[k]))
            if not same:   # (the 16x16x32 kernels sum k in another order inside the instruction: equal to rounding, not bit for bit)
                worst = max(((g_ - r_).abs().max() / r_.abs().max().clamp(min=1e-30)).item() for g_, r_ in zip(got, ref[k]))
                print(f"{'MISMATCH' if worst > 2e-6 else 'differs in rounding'} tile {t} {k}: max |diff| / max |ref| = {worst:.3e}")
    times = {(k, t): [] for k in calls for t in TILES}
    for r in range(ROUNDS + 1):
        for k, f in calls.items():
            for t in TILES:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    f(t)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[(k, t)].append(e0.elapsed_time(e1) / 3)
    tot = {t: 0.0 for t in TILES}
    print(f"M = {M}; stagger {os.environ.get('TVL_GEMM_STAGGER_US', '0')} us")
    for k in calls:
        row = []
        for t in TILES:
            v = sorted(times[(k, t)])
            med = v[len(v) // 2]
            tot[t] += med
            row.append(f"tile {t:5d}: {med * 1e3:7.1f} us {flops[k] / med / 1e9:6.1f} TF/s (min {v[0] * 1e3:6.1f})")
        print(f"{k}  " + "  |  ".join(row))
    fl = sum(flops.values())
    print("seven GEMMs (without the QKV data gradient): " + "  |  ".join(f"tile {t}: {tot[t]:.3f} ms {fl / tot[t] / 1e9:.1f} TF/s" for t in TILES))


if __name__ == "__main__":
    main()
