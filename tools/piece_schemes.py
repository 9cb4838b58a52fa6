#!/usr/bin/env python
"""CPU simulation: representation error of split-operand GEMM schemes against fp64 (K = 3072), relative to sum|a||b|.

Products of pieces are exact in fp32 for both bf16 and fp16 pieces, so what a scheme loses is (a) what the pieces do not represent and
(b) the piece products it drops; the fp32 accumulation error (~1e-8 here, torch's own fp32 matmul) comes on top for every scheme.

    bf16 x6 : x = p0 + p1 + p2 (bf16), products with i + j <= 2       -> 6 MFMAs per k-step (what tunevlseg_amd runs)
    bf16 x3 : two bf16 pieces, 3 products                             -> 3 MFMAs, 16 bits
    fp16 x3 : x = h0 + h1 (fp16, 11 bits each), products h0h0 + h0h1 + h1h0 -> 3 MFMAs, ~22 bits -- but fp16's exponent range:
              the second piece of anything below 0.125 is subnormal, so operands must be pre-scaled by (exact) powers of two
"""
import torch

torch.manual_seed(0)
M, N, K = 256, 256, 3072
a = torch.randn(M, K)
b = torch.randn(N, K) * 0.03
ref = a.double() @ b.double().T
den = a.abs().double() @ b.abs().double().T


def rms(x):
    return ((x.double() - ref) / den).pow(2).mean().sqrt().item()


def mm(x, y):
    return (x.double() @ y.double().T).float()


def split_bf16(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def split_f16(x):
    h0 = x.half().float()
    return h0, (x - h0).half().float()


print(f"torch fp32 matmul                      {rms(a @ b.T):.2e}")
pa, pb = split_bf16(a, 3), split_bf16(b, 3)
print(f"bf16 x6                                {rms(sum(mm(pa[i], pb[j]) for i in range(3) for j in range(3) if i + j <= 2)):.2e}")
pa, pb = split_bf16(a, 2), split_bf16(b, 2)
print(f"bf16 x3                                {rms(mm(pa[0], pb[0]) + mm(pa[0], pb[1]) + mm(pa[1], pb[0])):.2e}")
for sa, sb in ((0, 0), (6, 11)):
    (a0, a1), (b0, b1) = split_f16(a * 2.0**sa), split_f16(b * 2.0**sb)
    acc = (mm(a0, b0) + mm(a0, b1) + mm(a1, b0)) * 2.0 ** -(sa + sb)
    print(f"fp16 x3, operands pre-scaled 2^{sa} / 2^{sb}   {rms(acc):.2e}")
