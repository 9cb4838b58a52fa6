// fp32-in / fp32-out GEMM on the bf16 matrix cores by operand splitting (gfx950, v_mfma_f32_32x32x16_bf16).
//
// Each fp32 operand element is split, while its tile is staged to LDS, into S bf16 pieces
//     x = x1 + x2 + ... + xS      (every piece = round-to-nearest bf16 of the running residual)
// and the product is accumulated in fp32 over the piece pairs (i, j) with i + j <= S + 1:
//     S = 1  plain bf16 operands                 1 MFMA  per 32x32x16 step   (16x the f32-MFMA rate)
//     S = 2  a1b1 + a1b2 + a2b1                  3 MFMAs                     rel. error ~2^-16 per product
//     S = 3  + a1b3 + a2b2 + a3b1                6 MFMAs                     three bf16 pieces carry all 24 significand
//                                                                            bits of an fp32 value: the dropped terms are
//                                                                            <= 2^-24 relative, i.e. fp32-equivalent
// bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so S = 3 reproduces an fp32 GEMM to fp32
// round-off at 6/16 of the f32-MFMA cost (8 f32 MFMAs of 64 cycles vs 6 bf16 MFMAs of 32 cycles per 16-deep k-step).
//
// Tuning log (profiles/r1_gemm_experiments.md): warp-specialised producer/consumer waves, a two-tile ping-pong workgroup,
// source-level MFMA/VALU interleaving (sched_group_barrier) and a 3-deep register prefetch ring were all measured on
// MI355X and landed within +-4 % of this simple structure (146-151 TFLOP/s fp32-equivalent on the ViT-B/16 shapes); the
// PMC counters show 5.2 VALU instructions per MFMA and a 42 % busy matrix pipe at a ~2.3 GHz reported clock.
//
// Layout: NT only (A [M,K] and B [N,K], both k-contiguous); the frozen weights are kept in both orientations by the host
// so data gradients are NT as well.  256 threads = 4 waves, tiles 128x128 (2x2 waves) / 96x128 (1x4) / 64x64 (2x2),
// BK = 32, ONE LDS stage (S planes per operand) + register prefetch of the next slab, two barriers per slab.
// LDS rows are 32 bf16 + 8 pad = 80 bytes = 20 dwords, so the 16 rows of a ds_read_b128 lane group start on 16
// distinct 16-byte slots: conflict free.  Fragment map (guide §3): lane l -> row l&31, k = 8*(l>>5) .. +7.
#pragma once
#include <stdlib.h>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// 8-wave instantiations (192x256 tile) live in gemm_bf16s_w8.hip; GemmParams is TU-local, hence the opaque pointer
int tvl_gemm_bf16s_w8(const void* gemm_params, bool vec, bool conv, hipStream_t s);

namespace {

// 256 threads = 4 waves per workgroup; gemm_bf16s_w8.hip includes this header with 512 threads (8 waves) for the 192x256 tile
#ifndef TVL_GEMM_NTHREADS
#define TVL_GEMM_NTHREADS 256
#endif
constexpr int NTHREADS = TVL_GEMM_NTHREADS;
constexpr int NWAVES = NTHREADS / 64;

struct GemmParams {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* residual; int ldr;
    int act;
    float* pre_out;
    const float* dact_aux; int ld_aux; int dact;
    float alpha;
    tvlRowMap a_map, c_map;
    int tiles_m, tiles_n;
    // implicit 3x3 / pad 1 convolution (CONV kernels): A is an NHWC pixel matrix [B*H*W, lda], the GEMM row m is the output
    // pixel (b, oy, ox) and the GEMM column (ky*3+kx)*C + c addresses x[b, oy*stride+ky-1, ox*stride+kx-1, c] (0 outside)
    int cH, cW, cC, cStride, cHo, cWo;
};

__device__ __forceinline__ long map_row(int r, const tvlRowMap& m) {
    return m.div > 0 ? (long)(r / m.div) * m.mul + (r % m.div) + m.off : (long)r;
}

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ base, long row, int col, int cols, int ld) {
    const float* p = base + row * (long)ld + col;
    if (VEC) {
        if (col + 3 < cols) return *reinterpret_cast<const float4*>(p);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < cols) v.x = p[0];
    if (col + 1 < cols) v.y = p[1];
    if (col + 2 < cols) v.z = p[2];
    if (col + 3 < cols) v.w = p[3];
    return v;
}

// one operand slab = ROWS x BK fp32 = ROWS*BK/4 float4, dealt round-robin over the 256 threads
template <int ROWS, int BK>
struct StageRegs {
    static constexpr int TOTAL = ROWS * BK / 4;
    static constexpr int N = (TOTAL + NTHREADS - 1) / NTHREADS;
    float4 v[N];
};

template <int ROWS, int BK, bool VEC>
__device__ __forceinline__ void gload(StageRegs<ROWS, BK>& s, const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K,
                                      const tvlRowMap& map) {
    constexpr int F4 = BK / 4;
#pragma unroll
    for (int i = 0; i < StageRegs<ROWS, BK>::N; ++i) {
        const int idx = threadIdx.x + NTHREADS * i;
        if (StageRegs<ROWS, BK>::TOTAL % NTHREADS == 0 || idx < StageRegs<ROWS, BK>::TOTAL) {
            int r = row0 + idx / F4;
            r = r < nrows ? r : nrows - 1;
            s.v[i] = load4<VEC>(base, map_row(r, map), k0 + 4 * (idx % F4), K, ld);
        }
    }
}

// implicit-GEMM operand load for the 3x3 conv: the im2col matrix is never materialised.  The rows a thread stages are the
// same for every k-slab, so their (b, oy, ox) decomposition is done once (ConvRows); per slab only the tap changes.
template <int ROWS, int BK>
struct ConvRows {
    static constexpr int N = StageRegs<ROWS, BK>::N;
    long pix0[N];   // b * H * W
    int oy[N], ox[N];
};
template <int ROWS, int BK>
__device__ __forceinline__ void conv_rows_init(ConvRows<ROWS, BK>& cr, int row0, int nrows, const GemmParams& p) {
    constexpr int F4 = BK / 4;
#pragma unroll
    for (int i = 0; i < ConvRows<ROWS, BK>::N; ++i) {
        const int idx = threadIdx.x + NTHREADS * i;
        int r = row0 + idx / F4;
        r = r < nrows ? r : nrows - 1;
        const int hw = p.cHo * p.cWo;
        const int b = r / hw, rem = r - b * hw;
        const int oy = rem / p.cWo;
        cr.pix0[i] = (long)b * p.cH * p.cW;
        cr.oy[i] = oy * p.cStride - 1;
        cr.ox[i] = (rem - oy * p.cWo) * p.cStride - 1;
    }
}
template <int ROWS, int BK>
__device__ __forceinline__ void gload_conv(StageRegs<ROWS, BK>& s, const ConvRows<ROWS, BK>& cr, const GemmParams& p, int k0) {
    constexpr int F4 = BK / 4;
    const int kcol = k0 + 4 * (threadIdx.x % F4);  // NTHREADS % F4 == 0: the same column for every row this thread stages
    const int tap = kcol / p.cC;
    const int c = kcol - tap * p.cC;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const bool kok = kcol < p.K;
#pragma unroll
    for (int i = 0; i < StageRegs<ROWS, BK>::N; ++i) {
        const int idx = threadIdx.x + NTHREADS * i;
        if (StageRegs<ROWS, BK>::TOTAL % NTHREADS == 0 || idx < StageRegs<ROWS, BK>::TOTAL) {
            const int iy = cr.oy[i] + ky, ix = cr.ox[i] + kx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW)
                v = *reinterpret_cast<const float4*>(p.A + (cr.pix0[i] + (long)iy * p.cW + ix) * p.lda + c);
            s.v[i] = v;
        }
    }
}

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
// two fp32 bit patterns -> one dword holding their upper halves (bf16 by truncation): low half = lo, high half = hi
__device__ __forceinline__ unsigned pack_trunc(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}

// split 4 consecutive k-elements into S planes; plane s gets two dwords (4 bf16).  Every plane is the round-to-nearest bf16 of
// the running residual (see tp3.h): with rounding the dropped piece products are <= 2^-27 |a||b| for S = 3 (2^-24 with truncation)
template <int S>
__device__ __forceinline__ void split4(const float4 v, uint2 (&out)[S]) {
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const unsigned lo = pack_rn(x[0], x[1]), hi = pack_rn(x[2], x[3]);
        out[s] = make_uint2(lo, hi);
        if (s < S - 1) {
            x[0] -= bfloat(lo << 16); x[1] -= bfloat(lo & 0xFFFF0000u);
            x[2] -= bfloat(hi << 16); x[3] -= bfloat(hi & 0xFFFF0000u);
        }
    }
}

template <int ROWS, int BK, int S>
__device__ __forceinline__ void sstore(const StageRegs<ROWS, BK>& sr, __bf16* __restrict__ lds) {
    constexpr int F4 = BK / 4;
    constexpr int LDB = BK + 8;
#pragma unroll
    for (int i = 0; i < StageRegs<ROWS, BK>::N; ++i) {
        const int idx = threadIdx.x + NTHREADS * i;
        if (StageRegs<ROWS, BK>::TOTAL % NTHREADS == 0 || idx < StageRegs<ROWS, BK>::TOTAL) {
            uint2 pl[S];
            split4<S>(sr.v[i], pl);
#pragma unroll
            for (int s = 0; s < S; ++s) *reinterpret_cast<uint2*>(&lds[(s * ROWS + idx / F4) * LDB + 4 * (idx % F4)]) = pl[s];
        }
    }
}

// One float4 of the epilogue (4 consecutive columns of one row), all operands 16-byte aligned and inside the matrix.
__device__ __forceinline__ void emit4(const GemmParams& p, bool post, long crow, int col, float4 a) {
    float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
    if (p.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    if (p.dact) {
        const float4 z4 = *reinterpret_cast<const float4*>(p.dact_aux + crow * p.ld_aux + col);
        v[0] *= dact_f(z4.x, p.dact); v[1] *= dact_f(z4.y, p.dact); v[2] *= dact_f(z4.z, p.dact); v[3] *= dact_f(z4.w, p.dact);
    }
    if (p.pre_out) *reinterpret_cast<float4*>(p.pre_out + crow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.residual) r4 = *reinterpret_cast<const float4*>(p.residual + crow * p.ldr + col);
    if (post) { v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act & 0xff);
    if (!post) { v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
    *reinterpret_cast<float4*>(p.C + crow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
}

// Coalesced epilogue.  Straight from the accumulators a store instruction covers 32 rows x 32 bytes (lane l31 owns a row, the
// two lane halves 8 of its columns): partial-line writes that made the C write-out cost ~0.5 us per MB, un-overlapped (time vs K
// at fixed M, N has an intercept proportional to M*N).  Each 32x32 block therefore takes a round trip through a per-wave LDS
// scratch (free after the k-loop's last barrier) and leaves as 8 rows x 128 contiguous bytes per instruction; the epilogue's own
// loads (residual, activation-derivative operand) get the same mapping.
// The loop over row groups is ROLLED and the option handling is run-time: the fully unrolled version (TM x TN x 4 copies of emit4,
// twice for the two residual orders) is executed once per workgroup and arrived cold from the instruction cache -- on the tp3 kernel
// the same structure cost 20 us per tile against 4 for this one (profiles/r2_gemm_experiments.md).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_lds(const GemmParams& p, bool post, f32x16 (&acc)[TM][TN], int row_base, int col_base, int lane, float* scratch) {
    constexpr int LDS_ROW = 36;  // floats: 32 + 4 pad, ds_write_b128 of 8 consecutive rows hits 32 distinct banks
    const int l31 = lane & 31, h = lane >> 5;
    const int rr = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&scratch[(i * 32 + l31) * LDS_ROW + 8 * g + 4 * h]) =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        const int col = col_base + j * 32 + cc;
        if (col + 3 < p.N) {
#pragma unroll 1
            for (int rl = rr; rl < TM * 32; rl += 8) {
                const float4 v = *reinterpret_cast<const float4*>(&scratch[rl * LDS_ROW + cc]);
                const int row = row_base + rl;
                if (row < p.M) emit4(p, post, map_row(row, p.c_map), col, v);
            }
        }
    }
}

// epilogue of a transposed-product accumulator tile: lane -> row l31, registers 4g..4g+3 -> columns 8g+4h .. 8g+4h+3
template <int TM, int TN, bool VEC, bool POST>
__device__ __forceinline__ void epilogue_t(const GemmParams& p, f32x16 (&acc)[TM][TN], int row_base, int col_base, int l31, int h) {
    const bool vec_c = VEC && (p.ldc % 4 == 0) && tvl_dev_aligned16(p.C) && (!p.pre_out || tvl_dev_aligned16(p.pre_out)) &&
                       (!p.residual || (p.ldr % 4 == 0 && tvl_dev_aligned16(p.residual))) &&
                       (!p.dact || (p.ld_aux % 4 == 0 && tvl_dev_aligned16(p.dact_aux))) && (!p.bias || tvl_dev_aligned16(p.bias));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row_base + i * 32 + l31;
        if (row >= p.M) continue;
        const long crow = map_row(row, p.c_map);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = col_base + j * 32 + 8 * g + 4 * h;
                if (col >= p.N) continue;
                float v[4] = {acc[i][j][4 * g] * p.alpha, acc[i][j][4 * g + 1] * p.alpha, acc[i][j][4 * g + 2] * p.alpha,
                              acc[i][j][4 * g + 3] * p.alpha};
                if (vec_c && col + 3 < p.N) {
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    }
                    if (p.dact) {
                        const float4 z4 = *reinterpret_cast<const float4*>(p.dact_aux + crow * p.ld_aux + col);
                        v[0] *= dact_f(z4.x, p.dact); v[1] *= dact_f(z4.y, p.dact); v[2] *= dact_f(z4.z, p.dact); v[3] *= dact_f(z4.w, p.dact);
                    }
                    if (p.pre_out) *reinterpret_cast<float4*>(p.pre_out + crow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                    if (POST && p.residual) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + crow * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act & 0xff);
                    if (!POST && p.residual) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + crow * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                    *reinterpret_cast<float4*>(p.C + crow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = col + e;
                        if (c >= p.N) continue;
                        float x = v[e] + (p.bias ? p.bias[c] : 0.f);
                        if (p.dact) x *= dact_f(p.dact_aux[crow * p.ld_aux + c], p.dact);
                        if (p.pre_out) p.pre_out[crow * p.ldc + c] = x;
                        if (POST && p.residual) x += p.residual[crow * p.ldr + c];
                        x = act_f(x, p.act & 0xff);
                        if (!POST && p.residual) x += p.residual[crow * p.ldr + c];
                        p.C[crow * p.ldc + c] = x;
                    }
                }
            }
        }
    }
}

// STAGES: 1 = one LDS buffer, the next slab waits in registers; 2 = two LDS buffers; 3 = one LDS buffer and TWO slabs ahead in registers
// (the latency-bound 64x64 tile: a slab's global loads get two k-steps to land instead of one)
template <int BM, int BN, int WGM, int S, bool VEC, int BK, int STAGES, bool CONV>
__device__ __forceinline__ void gemm_bf16s_body(const GemmParams& p) {
    constexpr bool AHEAD2 = STAGES == 3;
    static_assert(!AHEAD2 || !CONV, "two-ahead prefetch: plain operands only");
    constexpr int LDB = BK + 8;  // bf16 elements per LDS row: 80 B (BK 32) / 48 B (BK 16), both conflict free for ds_read_b128
    constexpr int STAGE_ELEMS = S * (BM + BN) * LDB;
    constexpr int WGN = NWAVES / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);  // per stage: A [S][BM][LDB] then B [S][BN][LDB]

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped order: consecutive tile ids walk GROUP_M row-tiles of one column-tile before moving to the next column, so
    // the ~64 tiles resident on an XCD cover an ~8 x 8 block: 8 A panels + 8 W slices instead of 3 A panels + all of W
    // (rocprofv3: FETCH_SIZE of the fc1 GEMM was 19x its algorithmic bytes with the row-major order)
    constexpr int GROUP_M = 8;
    const int gsize_full = GROUP_M * p.tiles_n;
    const int group = bid / gsize_full;
    const int gm0 = group * GROUP_M;
    const int gm = p.tiles_m - gm0 < GROUP_M ? p.tiles_m - gm0 : GROUP_M;
    const int in_group = bid - group * gsize_full;
    const int tile_m = gm0 + in_group % gm, tile_n = in_group / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StageRegs<BM, BK> sa;
    StageRegs<BN, BK> sb;
    StageRegs<BM, BK> sa2;   // AHEAD2 only: the slab after next
    StageRegs<BN, BK> sb2;
    const tvlRowMap ident = {0, 0, 0};
    const int nk = (p.K + BK - 1) / BK;

    ConvRows<CONV ? BM : 1, BK> crows;
    if constexpr (CONV) {
        conv_rows_init<BM, BK>(crows, m0, p.M, p);
        gload_conv<BM, BK>(sa, crows, p, 0);
    } else {
        gload<BM, BK, VEC>(sa, p.A, p.lda, m0, p.M, 0, p.K, p.a_map);
    }
    gload<BN, BK, VEC>(sb, p.B, p.ldb, n0, p.N, 0, p.K, ident);
    sstore<BM, BK, S>(sa, smem);
    sstore<BN, BK, S>(sb, smem + S * BM * LDB);
    if constexpr (AHEAD2) {   // slab 1 -> (sa, sb), slab 2 -> (sa2, sb2); inside the loop the two register sets alternate
        if (nk > 1) {
            gload<BM, BK, VEC>(sa, p.A, p.lda, m0, p.M, BK, p.K, p.a_map);
            gload<BN, BK, VEC>(sb, p.B, p.ldb, n0, p.N, BK, p.K, ident);
        }
        if (nk > 2) {
            gload<BM, BK, VEC>(sa2, p.A, p.lda, m0, p.M, 2 * BK, p.K, p.a_map);
            gload<BN, BK, VEC>(sb2, p.B, p.ldb, n0, p.N, 2 * BK, p.K, ident);
        }
    }
    __syncthreads();

    // one k-slab: MFMAs of slab kt out of LDS, then slab kt+1 goes from registers (ra, rb) to LDS; AHEAD2: (ra, rb) then receive slab kt+3
    auto slab = [&](int kt, StageRegs<BM, BK>& ra, StageRegs<BN, BK>& rb) {
        const int cur = STAGES == 2 ? (kt & 1) : 0;
        const __bf16* As = smem + cur * STAGE_ELEMS;
        const __bf16* Bs = As + S * BM * LDB;
        if constexpr (!AHEAD2) {
            if (kt + 1 < nk) {
                if constexpr (CONV) gload_conv<BM, BK>(ra, crows, p, (kt + 1) * BK);
                else gload<BM, BK, VEC>(ra, p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, p.a_map);
                gload<BN, BK, VEC>(rb, p.B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, ident);
            }
        }
        if constexpr (TM * TN <= 4) {
            // all fragment reads of the slab are issued before its first MFMA; the compiler then waits with counted
            // lgkmcnt(N) so the later reads land underneath the earlier MFMAs
            bf16x8 af[BK / 16][TM][S], bf[BK / 16][TN][S];
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        af[ks][i][s] = *reinterpret_cast<const bf16x8*>(&As[(s * BM + wm * WM + i * 32 + l31) * LDB + ks * 16 + 8 * h]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        bf[ks][j][s] = *reinterpret_cast<const bf16x8*>(&Bs[(s * BN + wn * WN + j * 32 + l31) * LDB + ks * 16 + 8 * h]);
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                // smallest-magnitude piece pairs first, the a1*b1 term last; mfma(b, a) = transposed product, so that a lane's
                // 4 consecutive accumulator registers are 4 consecutive COLUMNS of one row of C (16-byte epilogue accesses)
#pragma unroll
                for (int order = S - 1; order >= 0; --order)
#pragma unroll
                    for (int sa_ = 0; sa_ <= order; ++sa_) {
                        const int sb_ = order - sa_;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][j][sb_], af[ks][i][sa_], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            // large wave tiles (96x64): fragments of one 16-deep k-step at a time, or the slab's fragments alone would take
            // 120 VGPRs next to the 96 accumulator registers.  Piece pairs from a table (rectangular loops unroll reliably).
            constexpr int NPAIR = S * (S + 1) / 2;
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB3[6] = {2, 1, 0, 1, 0, 0};  // S = 3: (0,2) (1,1) (2,0) (0,1) (1,0) (0,0)
            constexpr int PA2[3] = {0, 1, 0}, PB2[3] = {1, 0, 0};                    // S = 2: (0,1) (1,0) (0,0)
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[TM][S], bf[TN][S];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        af[i][s] = *reinterpret_cast<const bf16x8*>(&As[(s * BM + wm * WM + i * 32 + l31) * LDB + ks * 16 + 8 * h]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        bf[j][s] = *reinterpret_cast<const bf16x8*>(&Bs[(s * BN + wn * WN + j * 32 + l31) * LDB + ks * 16 + 8 * h]);
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int sa_ = S == 3 ? PA[q] : (S == 2 ? PA2[q] : 0);
                    const int sb_ = S == 3 ? PB3[q] : (S == 2 ? PB2[q] : 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][sb_], af[i][sa_], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (STAGES != 2) __syncthreads();  // every wave is done reading this slab before it is overwritten
        if (kt + 1 < nk) {
            __bf16* dst = smem + (STAGES == 2 ? (cur ^ 1) * STAGE_ELEMS : 0);
            sstore<BM, BK, S>(ra, dst);
            sstore<BN, BK, S>(rb, dst + S * BM * LDB);
            if constexpr (AHEAD2) {
                if (kt + 3 < nk) {
                    gload<BM, BK, VEC>(ra, p.A, p.lda, m0, p.M, (kt + 3) * BK, p.K, p.a_map);
                    gload<BN, BK, VEC>(rb, p.B, p.ldb, n0, p.N, (kt + 3) * BK, p.K, ident);
                }
            }
        }
        __syncthreads();
    };
    if constexpr (AHEAD2) {
        for (int kt = 0; kt < nk; kt += 2) {   // the two register sets alternate: static register indices, counted vmcnt waits
            slab(kt, sa, sb);
            if (kt + 1 < nk) slab(kt + 1, sa2, sb2);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) slab(kt, sa, sb);
    }
    // two copies of the epilogue so that the common order (activation, then residual) keeps its straight-line code
    // fast path: every epilogue operand 16-byte aligned and the wave's columns inside N -> coalesced stores through LDS
    const bool aligned = VEC && (p.ldc % 4 == 0) && tvl_dev_aligned16(p.C) && (!p.pre_out || tvl_dev_aligned16(p.pre_out)) &&
                         (!p.residual || (p.ldr % 4 == 0 && tvl_dev_aligned16(p.residual))) &&
                         (!p.dact || (p.ld_aux % 4 == 0 && tvl_dev_aligned16(p.dact_aux))) && (!p.bias || tvl_dev_aligned16(p.bias));
    const bool post = (p.act & TVL_ACT_POST_RESIDUAL) != 0;
    if (aligned && n0 + BN <= p.N) {  // workgroup-uniform: LDS scratch is only touched on this path
        float* scratch = reinterpret_cast<float*>(smem) + wave * (TM * 32 * 36);
        epilogue_lds<TM, TN>(p, post, acc, m0 + wm * WM, n0 + wn * WN, lane, scratch);
    } else if (post) {
        epilogue_t<TM, TN, VEC, true>(p, acc, m0 + wm * WM, n0 + wn * WN, l31, h);
    } else {
        epilogue_t<TM, TN, VEC, false>(p, acc, m0 + wm * WM, n0 + wn * WN, l31, h);
    }
}

// Split-K variant of the 64x64 tile for skinny problems (few output tiles, deep K: the text tower's M = B*L-row GEMMs with
// K = 1536 / 2048 walk 48-64 k-slabs serially on 32 workgroups, 55-73 us).  blockIdx.y owns a k-range and writes its raw
// partial tile to a workspace [splits][M][N]; splitk_reduce_kernel sums the partials in a fixed order (deterministic) and
// applies the epilogue.
template <int S, bool VEC>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16s_splitk_kernel(GemmParams p, float* __restrict__ ws, int kchunk) {
    TVL_KERNEL_ENTRY();
    constexpr int BM = 64, BN = 64, BK = 32, WGN = 2, WM = 32, WN = 32;
    constexpr int LDB = BK + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.y * kchunk;
    const int kend = kbeg + kchunk < p.K ? kbeg + kchunk : p.K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    StageRegs<BM, BK> sa;
    StageRegs<BN, BK> sb;
    const tvlRowMap ident = {0, 0, 0};
    const int nk = (kend - kbeg + BK - 1) / BK;
    gload<BM, BK, VEC>(sa, p.A, p.lda, m0, p.M, kbeg, kend, p.a_map);
    gload<BN, BK, VEC>(sb, p.B, p.ldb, n0, p.N, kbeg, kend, ident);
    sstore<BM, BK, S>(sa, smem);
    sstore<BN, BK, S>(sb, smem + S * BM * LDB);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const __bf16* As = smem;
        const __bf16* Bs = As + S * BM * LDB;
        if (kt + 1 < nk) {
            gload<BM, BK, VEC>(sa, p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, p.a_map);
            gload<BN, BK, VEC>(sb, p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, ident);
        }
        bf16x8 af[2][S], bf[2][S];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                af[ks][s] = *reinterpret_cast<const bf16x8*>(&As[(s * BM + wm * WM + l31) * LDB + ks * 16 + 8 * h]);
                bf[ks][s] = *reinterpret_cast<const bf16x8*>(&Bs[(s * BN + wn * WN + l31) * LDB + ks * 16 + 8 * h]);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int order = S - 1; order >= 0; --order)
#pragma unroll
                for (int sa_ = 0; sa_ <= order; ++sa_)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][order - sa_], af[ks][sa_], acc, 0, 0, 0);
        __syncthreads();
        if (kt + 1 < nk) {
            sstore<BM, BK, S>(sa, smem);
            sstore<BN, BK, S>(sb, smem + S * BM * LDB);
        }
        __syncthreads();
    }
    // raw partial: lane -> row l31, registers 4g..4g+3 -> columns 8g+4h..+3 (transposed product)
    float* wp = ws + (long)blockIdx.y * p.M * p.N;
    const int row = m0 + wm * WM + l31;
    if (row < p.M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = n0 + wn * WN + 8 * g + 4 * h;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < p.N) wp[(long)row * p.N + col + e] = acc[4 * g + e];
        }
    }
}

// C[map(m), n] = epilogue(alpha * sum_s ws[s][m][n]): same epilogue order as the GEMM kernels
__global__ void splitk_reduce_kernel(GemmParams p, const float* __restrict__ ws, int splits) {
    TVL_KERNEL_ENTRY();
    const long total = (long)p.M * p.N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % p.N);
        const int row = (int)(i / p.N);
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += ws[(long)s * total + i];
        const long crow = map_row(row, p.c_map);
        v = v * p.alpha + (p.bias ? p.bias[col] : 0.f);
        if (p.dact) v *= dact_f(p.dact_aux[crow * p.ld_aux + col], p.dact);
        if (p.pre_out) p.pre_out[crow * p.ldc + col] = v;
        const bool post = (p.act & TVL_ACT_POST_RESIDUAL) != 0;
        if (post && p.residual) v += p.residual[crow * p.ldr + col];
        v = act_f(v, p.act & 0xff);
        if (!post && p.residual) v += p.residual[crow * p.ldr + col];
        p.C[crow * p.ldc + col] = v;
    }
}

template <int S, bool VEC>
int launch_splitk(const GemmParams& p0, float* ws, int splits, hipStream_t s) {
    GemmParams p = p0;
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.N + 63) / 64;
    const int kchunk = ((p.K + splits - 1) / splits + 31) / 32 * 32;
    const int real_splits = (p.K + kchunk - 1) / kchunk;
    constexpr size_t smem = (size_t)S * 128 * 40 * sizeof(__bf16);
    hipLaunchKernelGGL((gemm_bf16s_splitk_kernel<S, VEC>), dim3((unsigned)(p.tiles_m * p.tiles_n), (unsigned)real_splits), dim3(NTHREADS), smem, s,
                       p, ws, kchunk);
    const long total = (long)p.M * p.N;
    long nb = (total + 255) / 256;
    nb = nb > 65536 ? 65536 : nb;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, (const float*)ws, real_splits);
    return 0;
}

template <int BM, int BN, int WGM, int S, bool VEC, int BK, int STAGES, bool CONV = false>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16s_kernel(GemmParams p) {
    TVL_KERNEL_ENTRY();
    gemm_bf16s_body<BM, BN, WGM, S, VEC, BK, STAGES, CONV>(p);
}
template <int BM, int BN, int WGM, int S, bool VEC, int BK, int STAGES, bool CONV = false>
int launch_v(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    constexpr size_t stage_bytes = (size_t)(STAGES == 2 ? 2 : 1) * S * (BM + BN) * (BK + 8) * sizeof(__bf16);
    constexpr size_t epi_bytes = (size_t)NWAVES * (BM / WGM) * 36 * sizeof(float);  // per-wave scratch of the coalesced epilogue: its WM rows
    constexpr size_t smem = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static int attr_dev_mask = 0;  // the > 64 KiB dynamic-LDS opt-in is a per-device function attribute
    auto kern = gemm_bf16s_kernel<BM, BN, WGM, S, VEC, BK, STAGES, CONV>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    if (!(attr_dev_mask & (1 << dev))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
        attr_dev_mask |= 1 << dev;
    }
    const long nwg = (long)p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NTHREADS), smem, s, p);
    return 0;
}

template <int BM, int BN, int WGM, int S, bool VEC>
int launch(const GemmParams& p, hipStream_t s) {
    return launch_v<BM, BN, WGM, S, VEC, 32, 1>(p, s);
}

#if TVL_GEMM_NTHREADS == 256
template <int S, bool VEC>
int launch_tile(int bm, const GemmParams& p, hipStream_t s) {
    if constexpr (S == 3) {
        if (bm == 256) return tvl_gemm_bf16s_w8(&p, VEC, false, s);
        if (bm == 192) return launch<192, 128, 2, S, VEC>(p, s);
    }
    if (bm == 256) bm = 128;
    if (bm == 192) bm = 96;  // the reduced-precision modes keep the three original tiles
    if (bm == 128) return launch<128, 128, 2, S, VEC>(p, s);
    if (bm == 96) return launch<96, 128, 1, S, VEC>(p, s);
    // 64x64 tiles serve the launch-latency-bound problems (text tower, decoder: M = B*L rows): a k-step there is one exposed memory
    // round trip (~1 us) around ~100 cycles of MFMA, so a 64-deep slab halves the kernel's duration; TVL_GEMM_SMALL_BK=32 restores it
    static const int small_bk = getenv("TVL_GEMM_SMALL_BK") ? atoi(getenv("TVL_GEMM_SMALL_BK")) : 64;
    if constexpr (S == 3) {
        if (small_bk == 64) return launch_v<64, 64, 2, S, VEC, 64, 3>(p, s);   // + two slabs ahead in registers
    }
    return launch<64, 64, 2, S, VEC>(p, s);
}

int launch_conv_tile(int bm, const GemmParams& p, hipStream_t s) {
    if (bm == 256) return tvl_gemm_bf16s_w8(&p, true, true, s);
    if (bm == 192) return launch_v<192, 128, 2, 3, true, 32, 1, true>(p, s);
    if (bm == 128) return launch_v<128, 128, 2, 3, true, 32, 1, true>(p, s);
    if (bm == 96) return launch_v<96, 128, 1, 3, true, 32, 1, true>(p, s);
    return launch_v<64, 64, 2, 3, true, 32, 1, true>(p, s);
}
#endif  // TVL_GEMM_NTHREADS == 256

int choose_bm(long M, long N, long K) {
    const long cus = 256;
    // 192x128 (wave tile 96x64) moves the fewest operand bytes per FLOP from L2 (the measured bound of this kernel: operand
    // loads alone take half its run time) and reads the fewest LDS bytes per MFMA: it replaces 96x128 outright (+19 % on the
    // K = 3072 shapes) and beats 128x128 except on short-K problems, where the bigger prologue / epilogue shows;
    // TVL_GEMM_TILE192=0 disables it.  192x256 (code 256: 8 waves, one workgroup per CU, same wave tile) halves the number of
    // tile rounds of the wide-N shapes; TVL_GEMM_TILE256=0 disables it.
    static const int use192 = getenv("TVL_GEMM_TILE192") ? atoi(getenv("TVL_GEMM_TILE192")) : 1;
    static const int use256 = getenv("TVL_GEMM_TILE256") ? atoi(getenv("TVL_GEMM_TILE256")) : 1;
    struct Cand { int bm, bn, per_cu, code; double w; } cands[5] = {{192, 256, 1, 256, 0.90}, {192, 128, 2, 192, 0.93}, {128, 128, 2, 128, 1.0},
                                                                    {96, 128, 2, 96, 1.0}, {64, 64, 4, 64, 1.12}};
    double best = 1e300, raw[5] = {0, 0, 0, 0, 0};
    int out = 64;
    for (int ci = 0; ci < 5; ++ci) {
        const Cand& c = cands[ci];
        // long K (+4 %) or very wide N (N = 3072 at K = 768: 165 vs 163 TFLOP/s for 128x128); 1-3 % slower on the other K = 768 shapes
        if (c.code == 256 && (!use256 || !use192 || N % 256 != 0 || (K < 1536 && N < 3072))) continue;
        if (c.code == 192 && !use192) continue;
        if (c.code == 96 && use192) continue;
        const long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const long slots = cus * c.per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        raw[ci] = (double)rounds * c.per_cu * c.bm * c.bn;
        const double cost = raw[ci] * c.w;
        if (cost < best) { best = cost; out = c.code; }
    }
    if (out == 192 && K < 1536 && raw[2] <= raw[1]) out = 128;
    if (out == 128 && use256 && use192 && N % 256 == 0 && N >= 3072 && raw[0] > 0 && raw[0] <= raw[2]) out = 256;
    return out;
}

}  // namespace
