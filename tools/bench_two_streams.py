#!/usr/bin/env python
"""Experiment: one vision layer's GEMM chain (QKV, out-proj, fc1, fc2 forward; dz, dx2, dO backward: the seven of tools/bench_layer_gemms.py)
for the full batch on one stream against TWO half batches on two streams, launched alternately from one host thread with a skew of one
kernel (half A runs kernel k + 1 while half B runs kernel k), so that one half's epilogue store burst meets the other half's k-loop.

    python tools/bench_two_streams.py [layers=10] [rounds=5]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
LAYERS, ROUNDS = int(args.get("layers", 10)), int(args.get("rounds", 5))
D, F = 768, 3072
dev = "cuda"
hip.load()
torch.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
W = {k: hip.weight_h2(rn(n, kk) * kk**-0.5) for k, (n, kk) in dict(wqkv=(3 * D, D), wo=(D, D), w1=(F, D), w2=(D, F), w2_t=(F, D), w1_t=(D, F), wo_t=(D, D)).items()}
bq, bo, b1 = rn(3 * D), rn(D), rn(F)


def operands(M):
    return dict(x=hip.h2_pack(rn(M, D), per_row=True, want_norm=True), o=hip.h2_pack(rn(M, D), per_row=False),
                a=hip.h2_pack(rn(M, F), per_row=True, want_norm=True), res=rn(M, D), z=(hip.gemm_aux(M, F, dev) if hip.gemm_aux(M, F, dev) is not None else rn(M, F)), M=M)


def chain(t):
    """the seven launches as closures over one operand set"""
    x, o, a, res, z = t["x"], t["o"], t["a"], t["res"], t["z"]
    return [
        lambda: hip.gemm_h2(x, W["wqkv"], want_f32=False, want_h2=True, out_per_tensor=True, out_add=4.0, bias=bq),
        lambda: hip.gemm_h2(o, W["wo"], bias=bo, residual=res),
        lambda: hip.gemm_h2(x, W["w1"], want_f32=False, want_h2=True, out_add=4.0, bias=b1, act=hip.ACT_QUICK_GELU, pre_out=z, aux_blocked=z.dim() == 1),
        lambda: hip.gemm_h2(a, W["w2"], bias=bo, residual=res),
        lambda: hip.gemm_h2(x, W["w2_t"], want_f32=False, want_h2=True, out_mul=1.125 * W["w2_t"]._bound, dact=hip.ACT_QUICK_GELU, dact_aux=z, aux_blocked=z.dim() == 1),
        lambda: hip.gemm_h2(a, W["w1_t"]),
        lambda: hip.gemm_h2(x, W["wo_t"], want_f32=False, want_h2=True, out_per_tensor=True),
    ]


full = chain(operands(15840))
halves = [chain(operands(7920)) for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run_full():
    for _ in range(LAYERS):
        for f in full:
            f()


def run_two(skew=1):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    seq = [f for _ in range(LAYERS) for f in halves[0]], [f for _ in range(LAYERS) for f in halves[1]]
    n = len(seq[0])
    for i in range(n + skew):
        if i < n:
            with torch.cuda.stream(s1):
                seq[0][i]()
        if i >= skew:
            with torch.cuda.stream(s2):
                seq[1][i - skew]()
    cur.wait_stream(s1); cur.wait_stream(s2)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for r in range(ROUNDS):
    print(f"{LAYERS} layers x 7 GEMMs: full batch, one stream {timeit(run_full):.3f} ms | two halves, two streams, skew 1 {timeit(lambda: run_two(1)):.3f} ms | skew 0 "
          f"{timeit(lambda: run_two(0)):.3f} ms | skew 3 {timeit(lambda: run_two(3)):.3f} ms")
