// "tp3": an fp32 matrix held as three bf16 pieces per element in MFMA-fragment order (include/tvl_hip.h).  Shared by the GEMM
// (gemm_tp3_kernel.h) and by the producers that write their result directly in this form (LayerNorm, attention).
#pragma once
#include "common.h"

namespace tp3 {

constexpr int PIECE = 1024;       // bytes: one MFMA operand (32 rows x 16 k bf16), lane-linear
constexpr int BLK = 3 * PIECE;    // one (32 x 16) block: three pieces

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
// two fp32 bit patterns -> one dword holding their upper halves (bf16 by truncation): low half = lo, high half = hi
__device__ __forceinline__ unsigned pack_trunc(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    bf16x2_t t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}

// 4 consecutive k-elements -> 3 pieces of 4 bf16 (8 bytes) each.  Every piece is the round-to-nearest bf16 of the running
// residual (v_cvt_pk_bf16_f32): |x - p0| <= 2^-9 |x|, |x - p0 - p1| <= 2^-18 |x|, and the last residual has at most 8
// significant bits, so x == p0 + p1 + p2 exactly.  With rounding (not truncation) the piece products a GEMM drops
// (a2*b3, a3*b2, a3*b3) are <= 2^-27 |a||b|, below the fp32 rounding of the sum itself.
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&out)[3]) {
    float x[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const unsigned lo = pack_rn(x[0], x[1]), hi = pack_rn(x[2], x[3]);
        out[s] = make_uint2(lo, hi);
        if (s < 2) {
            x[0] -= bfloat(lo << 16); x[1] -= bfloat(lo & 0xFFFF0000u);
            x[2] -= bfloat(hi << 16); x[3] -= bfloat(hi & 0xFFFF0000u);
        }
    }
}
// 8 consecutive k-elements -> 3 pieces of 16 bytes
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[3]) {
    const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    uint2 lo[3], hi[3];
    split4(a, lo);
    split4(b, hi);
#pragma unroll
    for (int s = 0; s < 3; ++s) out[s] = make_uint4(lo[s].x, lo[s].y, hi[s].x, hi[s].y);
}
// 16 bytes of each piece -> the 8 fp32 values they encode
__device__ __forceinline__ void join8(const uint4 (&in)[3], float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
    for (int s = 2; s >= 0; --s) {
        const unsigned w[4] = {in[s].x, in[s].y, in[s].z, in[s].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += bfloat(w[e] << 16);
            v[2 * e + 1] += bfloat(w[e] & 0xFFFF0000u);
        }
    }
}

// byte offset of 4 consecutive elements (row, col .. col+3), col % 4 == 0, of piece 0 inside an image with kblocks = cols / 16
__device__ __forceinline__ long off4(long row, int col, int kblocks) {
    return ((row >> 5) * kblocks + (col >> 4)) * (long)BLK + ((((col >> 3) & 1) * 32 + (int)(row & 31)) * 16 + (col & 4) * 2);
}
// split 4 consecutive elements and store them (8 bytes per piece)
__device__ __forceinline__ void store4(unsigned char* __restrict__ img, int kblocks, long row, int col, const float (&v)[4]) {
    uint2 pl[3];
    split4(v, pl);
    unsigned char* o = img + off4(row, col, kblocks);
#pragma unroll
    for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(o + s * PIECE) = pl[s];
}

}  // namespace tp3

// "h2": two fp16 pieces of the value scaled by an exact power of two (csrc/gemm_h2.hip); same block order as tp3, 2 KiB per block.
namespace h2 {

constexpr int PIECE = 1024;
constexpr int BLK = 2 * PIECE;

// 1 / s for the power of two s that puts amax * s in [2^13, 2^14) (1 for a zero / non-finite maximum)
__device__ __forceinline__ float inv_scale_of(float amax) {
    if (!(amax > 1.0e-30f) || !(amax < 3.0e38f)) return 1.0f;   // (a row that small is left unscaled: its pieces are subnormal noise either way)
    int e;
    frexpf(amax, &e);               // amax = m * 2^e, m in [0.5, 1)
    return ldexpf(1.0f, e - 14);    // s = 2^(14 - e)
}
// 8 consecutive k-elements (already scaled) -> 2 pieces of 16 bytes: h0 = fp16(x), h1 = fp16(x - h0)
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[2]) {
    _Float16 h0[8], h1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h0[e] = (_Float16)v[e];
        h1[e] = (_Float16)(v[e] - (float)h0[e]);
    }
    out[0] = *reinterpret_cast<const uint4*>(h0);
    out[1] = *reinterpret_cast<const uint4*>(h1);
}

// byte offset of 4 consecutive elements (row, col .. col+3), col % 4 == 0, of piece 0 inside an h2 image with kblocks = cols / 16
__device__ __forceinline__ long off4(long row, int col, int kblocks) {
    return ((row >> 5) * kblocks + (col >> 4)) * (long)BLK + ((((col >> 3) & 1) * 32 + (int)(row & 31)) * 16 + (col & 4) * 2);
}
// split 4 consecutive (already scaled) elements and store them (8 bytes per piece)
__device__ __forceinline__ void store4(unsigned char* __restrict__ img, int kblocks, long row, int col, const float (&v)[4]) {
    _Float16 h0[4], h1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h0[e] = (_Float16)v[e];
        h1[e] = (_Float16)(v[e] - (float)h0[e]);
    }
    unsigned char* o = img + off4(row, col, kblocks);
    *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(h0);
    *reinterpret_cast<uint2*>(o + PIECE) = *reinterpret_cast<const uint2*>(h1);
}

}  // namespace h2

