// fp32-equivalent GEMM over operands that already live in HBM as three bf16 pieces, in fragment order ("tp3" format).
// gfx950: LDS-DMA ring (global_load_lds_dwordx4, counted vmcnt, raw s_barrier) + v_mfma_f32_32x32x16_bf16.
//
// Why a second GEMM.  gemm_bf16s_kernel.h splits fp32 operands into bf16 pieces while it stages a tile: ~5 VALU per MFMA,
// ds_write fills and two barriers per k-step, with nothing overlapping the fill (profiles/r1_gemm_experiments.md: matrix
// pipe 42 % busy).  The vision tower's eight GEMMs per layer always have a frozen weight on one side and an activation that
// one of our own kernels has just produced on the other, so both sides can be handed over pre-split and PRE-TILED:
//
//   tp3 image of a logical fp32 matrix X[R, K] (K % 16 == 0), R rounded up to 32-row blocks:
//     block (rb, kb) = rows 32 rb .. 32 rb + 31, k = 16 kb .. 16 kb + 15, at byte ((rb * K/16 + kb) * 3 + piece) * 1024
//     inside a piece: element (r, k) at ((k % 16) / 8 * 32 + r % 32) * 16 + (k % 8) * 2        (bf16)
//   x = piece0 + piece1 + piece2 (pieces 0, 1 by truncating the running residual, piece 2 rounded): 24 significand bits.
//
// One 1-KiB piece is exactly the A (or B) operand of one v_mfma_f32_32x32x16_bf16 in lane order (lane l = row l & 31,
// k = 8 (l >> 5) .. + 7, guide §3): the LDS image equals the HBM image, an LDS-DMA instruction copies one piece (fully
// coalesced, no swizzle needed), and a fragment read is ds_read_b128 at lane * 16 -- conflict free by construction.
//
// Kernel: 512 threads = 8 waves (2 x 4), tile BM x 256, wave tile (BM/2) x 64, one workgroup per CU.  k-slab = 16 deep =
// 3 * (BM + 256) / 32 pieces; three LDS stages.  Per slab ONE barrier:
//     wait (counted vmcnt) until slab t+1 has landed -> s_barrier -> DMA slab t+3 into the stage slab t just left ->
//     ds_read slab t+1's fragments into the second register set, interleaved with the 6*TM*TN MFMAs of slab t.
// Fragments are double-buffered in registers, so the MFMAs of a slab never wait for LDS, and the DMA has two full slabs
// (>= 2 x 1150 MFMA cycles) to land.  The ds_reads are inline asm: hipcc would otherwise put a vmcnt(0) in front of every
// LDS read that follows an LDS-DMA (it cannot tell the stages apart) and drain the ring (guide §5, "Pipelining across
// barriers"); their completion is our own `s_waitcnt lgkmcnt(0)` at the top of the next slab.
//
// Epilogue: same contract as tvl_gemm_f32 (bias, act', pre_out, act, residual) + optional tp3 output of the final value, so
// the consumer GEMM needs no split pass (fc1 -> QuickGELU -> fc2, and the dz of the backward).
#pragma once
#include <stdlib.h>
#include <utility>
#include "common.h"
#include "tp3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int NTH = 512;     // the 8-wave workgroup (2 x 4 waves, one workgroup per CU)
constexpr int NWAVE = 8;
// NW = 4: a 4-wave workgroup (1 x 4 waves, tile BM x 256 with BM <= 128, LDS <= 80 KB) of which TWO fit a CU -- one wave of each per SIMD.
// The two are independent (own barriers, own DMA ring): whatever one of them waits for (its slab barrier, a counted vmcnt, the
// HBM burst of its epilogue, its launch prologue) the other one's MFMAs fill, which a single 8-wave workgroup in lockstep cannot do.
using tp3::PIECE;
using tp3::BLK;

struct Tp3Params {
    int M, N, K;
    const unsigned char* A; int a_rb;   // tp3 image of [M, K]; a_rb = 32-row blocks present
    const unsigned char* B; int b_rb;   // tp3 image of [N, K]
    float* C; int ldc;                  // fp32 output (may be null when Cp is given)
    unsigned char* Cp;                  // tp3 image of the output [M, N] (as the next GEMM's A), or null
    const float* bias;
    const float* residual; int ldr;
    int act;
    float* pre_out;
    const float* dact_aux; int ld_aux; int dact;
    float alpha;
    const float* a_scale;   // optional per-row factor of the result (two-piece fp16 operands carry power-of-two row scales), or null
    int a_sstride;          // 1: a_scale[row]; 0: one factor for every row (a tensor-scaled A operand: the conv's pixel matrix)
    // h2 OUTPUT (the next GEMM's A operand as two fp16 pieces): its row scale comes from a Cauchy-Schwarz bound on the row,
    // |out[m, n]| <= out_norm[m] * out_mul + out_add (out_norm = ||A row m||_2 from A's producer, out_mul = max_n ||B row n||_2 times
    // the activation's Lipschitz bound, out_add = max |bias|); the epilogue writes the inverse scale to out_inv[m]
    unsigned char* Ch2; const float* out_norm; float out_mul, out_add; float* out_inv;
    // A with one scale per (row, 64-column chunk of K) -- dQ | dK | dV: every (row, head) block is written by one attention workgroup that
    // knows its exact maximum.  a_kscale[m * k_chunks + c] = inverse scale of A's columns 64 c .. 64 c + 63 in row m (KS kernels only)
    const float* a_kscale; int k_chunks;
    int out_stride;   // 1: out_norm / out_inv are per row; 0: one bound and one scale for the whole output (out_norm[0], out_inv[0])
    // CONV kernels: A is the image of the NHWC pixel matrix [B*cH*cW, 16*cC16] with one all-zero 32-row block appended (block row a_rb);
    // row m of the GEMM gathers its nine taps from it (3x3, pad 1, stride 1; K = 9 * 16 * cC16 ordered (c / 16, ky, kx, c % 16))
    int cH, cW, cC16;
    int tiles_m, tiles_n;
    int row_walk;        // gemm_h2m_kernel: column tiles fastest in the tile walk (few column tiles: all of a row tile's in one XCD's chunk)
    int stagger_ticks;   // NW = 4 kernels: start delay (10 ns ticks of s_memrealtime) of the second workgroup of each CU, 0 = none
    int f32_direct;                 // gemm_h2m_kernel: plain fp32-output epilogues straight from the accumulators (64-byte row segments, no LDS round trip)
    int aux_blocked;                // gemm_h2m_kernel's direct epilogue: pre_out / dact_aux in accumulator order (tvlGemmTp3Args.aux_blocked)
    float* work; long work_bytes;   // gemm_h2m_kernel<PERSIST>: where a workgroup parks the partial sums of its split first tile (or null)
};

__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(size_t)lds_byte, 16, 0, 0);
}

template <int OFF>
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most G of this wave's DMA groups (one group = its pieces of one slab) are still in flight
template <int PT, int NW, int G>
__device__ __forceinline__ void wait_groups_ct(bool extra) {
    constexpr int n0 = PT / NW;
    static_assert(G >= 0 && G * (n0 + 1) <= 63, "vmcnt is a 6-bit count");
    if constexpr (G == 0) wait_vm<0>();
    else if constexpr (PT % NW == 0) wait_vm<G * n0>();
    else { if (extra) wait_vm<G * (n0 + 1)>(); else wait_vm<G * n0>(); }
}
template <int PT, int NW, int GMAX>
__device__ __forceinline__ void wait_groups_rt(int groups, bool extra) {   // prologue only: the count depends on K
    if constexpr (GMAX <= 0) wait_groups_ct<PT, NW, 0>(extra);
    else { if (groups >= GMAX) wait_groups_ct<PT, NW, GMAX>(extra); else wait_groups_rt<PT, NW, GMAX - 1>(groups, extra); }
}

using tp3::split4;
__device__ __forceinline__ long tp3_off(long row, int col, int kblocks) { return tp3::off4(row, col, kblocks); }

// Epilogue modes.  EPI < 0: every option read from the arguments at run time (any combination the C ABI allows).  EPI >= 0: a
// bit set of the options below, resolved at compile time -- the epilogue is cold, straight-line code executed once per tile, and
// with all options live it was ~5 KB per 4-column group, 24 groups per wave: the write-out of a tile then ran at the
// instruction-fetch rate (20 us per 192x256 tile; stamps in profiles/r2_gemm_experiments.md), not at the store rate.
enum { E_BIAS = 1, E_RES = 2, E_QGELU = 4, E_DQGELU = 8, E_PRE = 16, E_F32 = 32, E_TP3 = 64, E_RSCALE = 128, E_H2OUT = 256, E_RELU = 512 };

// h2_sc: the row's power-of-two scale of an h2 output (computed once per row by epilogue(), not once per four columns)
template <int EPI, bool NOSTORE = false>
__device__ __forceinline__ void emit4(const Tp3Params& p, long row, int col, float4 a, float h2_sc = 1.0f) {
    constexpr bool G = EPI < 0;
    float v[4] = {a.x, a.y, a.z, a.w};
    if (G || (EPI & E_RSCALE)) {   // alpha, and the operand row scale of the two-piece fp16 format
        const float f = p.a_kscale ? p.alpha * p.a_kscale[row * p.k_chunks + p.k_chunks - 1]   // the accumulator carries the LAST chunk's scale
                                   : ((G ? (p.a_scale != nullptr) : true) ? p.alpha * p.a_scale[row * p.a_sstride] : p.alpha);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= f;
    }
    if (G ? p.bias != nullptr : (EPI & E_BIAS) != 0) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    if (G ? p.dact != 0 : (EPI & E_DQGELU) != 0) {
        const float4 z4 = *reinterpret_cast<const float4*>(p.dact_aux + row * p.ld_aux + col);
        const int d = G ? p.dact : (int)TVL_ACT_QUICK_GELU;
        v[0] *= dact_f(z4.x, d); v[1] *= dact_f(z4.y, d); v[2] *= dact_f(z4.z, d); v[3] *= dact_f(z4.w, d);
    }
    if (G ? p.pre_out != nullptr : (EPI & E_PRE) != 0) *reinterpret_cast<float4*>(p.pre_out + row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    const bool post = G && (p.act & TVL_ACT_POST_RESIDUAL) != 0;
    const bool has_res = G ? p.residual != nullptr : (EPI & E_RES) != 0;
    if (post && has_res) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + row * p.ldr + col);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
    }
    if (G || (EPI & (E_QGELU | E_RELU))) {
        const int act = G ? (p.act & 0xff) : ((EPI & E_RELU) ? (int)TVL_ACT_RELU : (int)TVL_ACT_QUICK_GELU);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], act);
    }
    if (!post && has_res) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + row * p.ldr + col);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
    }
    if (NOSTORE) {  // timing-only ablation: keep the values alive, store nothing (the condition never holds)
        if (v[0] + v[1] + v[2] + v[3] == 1.2345e-33f) p.C[0] = v[0];
        return;
    }
    if (G ? p.C != nullptr : (EPI & E_F32) != 0) *reinterpret_cast<float4*>(p.C + row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    if (G ? p.Cp != nullptr : (EPI & E_TP3) != 0) {
        uint2 pl[3];
        split4(v, pl);
        unsigned char* o = p.Cp + tp3_off(row, col, p.N >> 4);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(o + s * PIECE) = pl[s];
    }
    if (G ? p.Ch2 != nullptr : (EPI & E_H2OUT) != 0) {
        const float w[4] = {v[0] * h2_sc, v[1] * h2_sc, v[2] * h2_sc, v[3] * h2_sc};
        h2::store4(p.Ch2, p.N >> 4, row, col, w);
    }
}

// accumulators -> per-wave LDS scratch (one 32-column strip of the wave tile at a time, all its rows) -> a ROLLED loop over
// groups of 8 rows: rows leave as 128 contiguous bytes (8 rows per store instruction) and the option handling exists once per
// strip instead of once per 32x32 block.  The tp3 output of the same pass leaves as four 128-byte segments per instruction.
template <int TM, int TN, int EPI, bool NOSTORE = false>
__device__ __forceinline__ void epilogue(const Tp3Params& p, f32x16 (&acc)[TM][TN], int row_base, int col_base, int lane, float* scratch) {
    constexpr int LDS_ROW = 36;
    const int l31 = lane & 31, h = lane >> 5;
    const int rr = lane >> 3, cc = (lane & 7) * 4;
    // h2 output: the rows' scales (a bound -> a power of two, tp3.h) once per row into the tail of the wave's scratch; the workgroup
    // that owns the rows' first columns also publishes the inverse scales for the consuming GEMM
    constexpr bool H2O = EPI < 0 || (EPI & E_H2OUT) != 0;
    float* rowsc = scratch + TM * 32 * LDS_ROW;
    const bool h2o = H2O && (EPI >= 0 || p.Ch2 != nullptr);
    if (h2o) {
        for (int rl = lane; rl < TM * 32; rl += 64) {
            const int row = row_base + rl;
            float inv = 1.0f;
            if (row < p.M) {
                inv = h2::inv_scale_of(p.out_norm[(long)row * p.out_stride] * p.out_mul + p.out_add);
                if (col_base == 0 && (p.out_stride || row == 0)) p.out_inv[(long)row * p.out_stride] = inv;
            }
            rowsc[rl] = 1.0f / inv;   // a power of two: exact
        }
    }
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
                if (row < p.M) emit4<EPI, NOSTORE>(p, row, col, v, h2o ? rowsc[rl] : 1.0f);
            }
        }
    }
}

template <int INFLIGHT, bool ISSUE, bool LAST>
struct StepMode {};

template <int TM, int TN, int NP>
struct Frags {
    bf16x8 a[TM][NP], b[TN][NP];
};

// piece q of the slab's (memory-op, MFMA) interleave: DMA pieces first, then the fragment reads of the NEXT slab
template <int TM, int TN, int NP, int IDX>
__device__ __forceinline__ void read_one(Frags<TM, TN, NP>& f, unsigned a_addr, unsigned b_addr) {
    if constexpr (IDX < NP * TM) f.a[IDX / NP][IDX % NP] = lds_frag<IDX * PIECE>(a_addr);
    else if constexpr (IDX < NP * (TM + TN)) f.b[(IDX - NP * TM) / NP][(IDX - NP * TM) % NP] = lds_frag<(IDX - NP * TM) * PIECE>(b_addr);
}

template <int TM, int TN, int NP, int... I>
__device__ __forceinline__ void read_all(Frags<TM, TN, NP>& f, unsigned a_addr, unsigned b_addr, std::integer_sequence<int, I...>) {
    (read_one<TM, TN, NP, I>(f, a_addr, b_addr), ...);
}

// NP = pieces per operand element: 3 (bf16 pieces, 6 products per k-step: the tp3 format) or 2 (fp16 pieces, 3 products: the "h2" format,
// gemm_h2.hip -- same block order, 2 KiB per 32 x 16 block, operands pre-scaled by exact powers of two)
// KS: A carries one power-of-two scale per (row, 64-column chunk of K) (a_kscale).  The accumulators hold the sum scaled by the CURRENT
// chunk's scale; at a chunk boundary (every four 16-deep slabs) each lane multiplies its rows' accumulators by s_next / s_cur -- a power of
// two, exact -- and the epilogue undoes the last chunk's scale.  The ratios sit in LDS behind the three stages (BM x k_chunks floats).
// NS = LDS stages of the DMA ring (NS - 1 slabs requested ahead of the one being multiplied).  The ring is what hides the operand
// stream's latency: bytes in flight per CU / latency = fill rate, and with two fp16 pieces a slab is multiplied in half the time
// three bf16 pieces took, so the same two slabs in flight cover half the time (profiles/r3_gemm_experiments.md)
template <int BM, int BN, int VARIANT, int EPI, int NP = 3, bool KS = false, bool CONV = false, int NW = NWAVE, int NS = 3>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_tp3_kernel(Tp3Params p_in) {
    TVL_KERNEL_ENTRY();
    Tp3Params p = p_in;
    float* const stamp_buf = p_in.pre_out;
    if constexpr ((VARIANT & 32) != 0) p.pre_out = nullptr;
    constexpr int NWAVE = NW, NTH = NW * 64;   // (shadow the 8-wave constants of the namespace)
    constexpr int WGN = 4, WGM = NW / WGN;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(NS >= 3 && NS <= 6, "ring stages");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int BLKP = NP * PIECE;                                      // one 32 x 16 block of an operand image
    constexpr int PA = NP * BM / 32, PB = NP * BN / 32, PT = PA + PB;   // 1-KiB pieces per slab
    constexpr int PW = (PT + NWAVE - 1) / NWAVE;                       // pieces per wave (the first PT % 8 waves carry one more)
    constexpr int STAGE = PT * PIECE;
    constexpr int NREAD = NP * (TM + TN);
    constexpr int NPROD = NP == 3 ? 6 : 3;   // piece products kept per k-step
    constexpr int NMFMA = NPROD * TM * TN;
    constexpr bool PIN = (VARIANT & 1) != 0, DMA_SPREAD = (VARIANT & 2) != 0;
    // timing-only ablations (wrong results): bit 2 = no DMA inside the loop, bit 3 = every DMA re-reads slab 0 (L2-resident)
    constexpr bool ABL_NODMA = (VARIANT & 4) != 0, ABL_SLAB0 = (VARIANT & 8) != 0, ABL_NOSTORE = (VARIANT & 16) != 0;
    // diagnostic build (bit 5): wave 0 stamps s_memrealtime (100 MHz) / s_memtime (shader clock) at 4 points into p.pre_out
    constexpr bool STAMP = (VARIANT & 32) != 0;
    unsigned long long stamp_rt[6] = {0, 0, 0, 0, 0, 0}, stamp_clk[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            stamp_rt[i] = __builtin_amdgcn_s_memrealtime();
            stamp_clk[i] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int gsize_full = GROUP_M * p.tiles_n;
    const int group = bid / gsize_full;
    const int gm0 = group * GROUP_M;
    const int gm = p.tiles_m - gm0 < GROUP_M ? p.tiles_m - gm0 : GROUP_M;
    const int in_group = bid - group * gsize_full;
    const int tile_m = gm0 + in_group % gm, tile_n = in_group / gm;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const bool extra = wave < PT % NWAVE;
    const int KB = p.K >> 4;
    const int nk = KB;

    // this wave's DMA sources (slab 0): wave-uniform 64-bit bases (SGPRs) + one 32-bit lane offset
    const unsigned char* src[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pc = wave + NWAVE * i;
        if (pc < PA) {
            int rb = tile_m * (BM / 32) + pc / NP;
            rb = rb < p.a_rb ? rb : p.a_rb - 1;
            src[i] = p.A + ((long)rb * KB) * BLKP + (pc % NP) * PIECE;
        } else {
            const int q = pc < PT ? pc - PA : 0;
            int rb = tile_n * (BN / 32) + q / NP;
            rb = rb < p.b_rb ? rb : p.b_rb - 1;
            src[i] = p.B + ((long)rb * KB) * BLKP + (q % NP) * PIECE;
        }
    }
    const unsigned lane16 = lane * 16;
    // CONV: the output pixel behind this lane's row of the wave's A pieces (only pieces 0 and 1 of a wave can be A pieces), and the
    // (tap, channel block) of the NEXT slab to be requested -- slabs are requested in order, conv_advance() after each
    static_assert(!CONV || PA <= 2 * NWAVE, "conv: at most two A pieces per wave");
    // first-round stagger (speed only, never correctness): the workgroups that the dispatcher is observed to place SECOND on each CU
    // (ids 256 .. 511 of the first wave of dispatches) start half a tile late, so that the two workgroups of a CU reach their
    // epilogues at different times from then on
    if constexpr (NW == 4) {
        if (p.stagger_ticks > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(64);
        }
    }
    int cv_row[2] = {-1, -1}, cv_yx[2] = {0, 0};
    int cv_dy = -1, cv_dx = -1, cv_cb = 0;
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + NWAVE * i;
            if (pc < PA) {
                const int r = tile_m * BM + (pc / NP) * 32 + (lane & 31);
                const int hw = p.cH * p.cW;
                const int rem = r % hw, oy = rem / p.cW;
                cv_row[i] = r < p.M ? r : -1;
                cv_yx[i] = (oy << 16) | (rem - oy * p.cW);
            }
        }
    }
    auto conv_advance = [&]() {
        if constexpr (CONV) {
            // taps innermost: the nine requests for one channel block hit the same few rows while they are still in L2
            if (++cv_dx == 2) {
                cv_dx = -1;
                if (++cv_dy == 2) { cv_dy = -1; ++cv_cb; }
            }
        }
    };
    // request piece i of this wave for k-slab `slab` into LDS stage `stage`
    auto issue_piece = [&](auto idx, int slab, int stage) {
        constexpr int i = decltype(idx)::value;
        const int pc = wave + NWAVE * i;
        if constexpr (CONV && i < 2) {
            if (pc < PA) {   // wave-uniform
                const int iy = (cv_yx[i] >> 16) + cv_dy, ix = (cv_yx[i] & 0xffff) + cv_dx;
                const bool ok = cv_row[i] >= 0 && (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
                const int rin = ok ? cv_row[i] + cv_dy * p.cW + cv_dx : p.a_rb * 32;
                const unsigned long off = ((unsigned long)((rin >> 5) * p.cC16 + cv_cb)) * BLKP + (unsigned)((pc % NP) * PIECE + (((lane >> 5) * 32 + (rin & 31)) * 16));
                glds16(p.A + off, lds0 + stage * STAGE + pc * PIECE);
                return;
            }
        }
        if ((i + 1) * NWAVE <= PT || pc < PT) glds16((src[i] + (long)(ABL_SLAB0 ? (slab & 1) : slab) * BLKP) + lane16, lds0 + stage * STAGE + pc * PIECE);
    };
    auto issue = [&](int slab, int stage) {
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue_piece(std::integral_constant<int, I>{}, slab, stage), ...); }(std::make_integer_sequence<int, PW>{});
    };
    auto issue_next = [&](int slab, int stage) { issue(slab, stage); conv_advance(); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned a_frag = lds0 + lane * 16 + wm * (TM * BLKP);
    const unsigned b_frag = lds0 + lane * 16 + PA * PIECE + wn * (TN * BLKP);

    // KS: ratio table R[row][c] = inv[row][c-1] / inv[row][c] (c >= 1) behind the stages, before any LDS-DMA is in flight
    const unsigned ktab = lds0 + NS * STAGE;
    if constexpr (KS) {
        float* tab = reinterpret_cast<float*>(smem + NS * STAGE);
        const int nch = p.k_chunks;
        for (int e = threadIdx.x; e < BM * nch; e += NTH) {
            const int rl = e / nch, c = e - rl * nch;
            long row = (long)tile_m * BM + rl;
            row = row < p.M ? row : p.M - 1;
            tab[e] = c ? p.a_kscale[row * nch + c - 1] / p.a_kscale[row * nch + c] : 1.0f;
        }
        __syncthreads();
    }
    // prologue: NS slabs in flight (the host guarantees nk > NS), wait for the first
#pragma unroll
    for (int i = 0; i < NS; ++i) issue_next(i, i);
    wait_groups_ct<PT, NW, NS - 1>(extra);
    __builtin_amdgcn_s_barrier();
    stamp(1);
    Frags<TM, TN, NP> f0, f1;
    read_all<TM, TN, NP>(f0, a_frag, b_frag, std::make_integer_sequence<int, NREAD>{});

    // smallest piece products first, a0*b0 last
    constexpr int PA_[6] = {NP == 3 ? 0 : 0, 1, NP == 3 ? 2 : 0, 0, 1, 0}, PB_[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0};

    // One slab.  `cur` holds slab kt's fragments (requested one slab ago); unless LAST, `nxt` receives slab kt+1's.
    //   INFLIGHT: DMA groups that may stay in flight while slab kt+1 is awaited (1 in steady state, 0 near the end);
    //   ISSUE:    slab kt+3 exists and is requested into the stage slab kt has just left.
    float kratio[TM];   // KS: this lane's rows' ratios for the next chunk boundary (requested one boundary ahead)
    auto kratio_request = [&](int c) {
        if constexpr (KS) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                asm volatile("ds_read_b32 %0, %1" : "=v"(kratio[i]) : "v"(ktab + (unsigned)(((wm * WM + i * 32 + (lane & 31)) * p.k_chunks + c) * 4)));
        }
    };
    if constexpr (KS) kratio_request(1);
    auto step = [&]<int INFLIGHT, bool ISSUE, bool LAST>(StepMode<INFLIGHT, ISSUE, LAST>, int kt, int st_cur, Frags<TM, TN, NP>& cur,
                                                         Frags<TM, TN, NP>& nxt) {
        if constexpr (!LAST && !ABL_NODMA) wait_groups_ct<PT, NW, INFLIGHT>(extra);   // this wave's pieces of slab kt+1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // `cur` is complete and this wave no longer reads stage st_cur
        if constexpr (KS) {
            if ((kt & 3) == 0 && kt) {   // slab kt opens chunk kt / 4: bring the accumulators to its scale, request the next boundary's ratios
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= kratio[i];
                const int cn = (kt >> 2) + 1;
                kratio_request(cn < p.k_chunks ? cn : p.k_chunks - 1);
            }
        }
        if constexpr (!LAST) __builtin_amdgcn_s_barrier();        // ... nor does any other wave, and their pieces landed too
        __builtin_amdgcn_sched_barrier(0);
        const int st_nxt = st_cur == NS - 1 ? 0 : st_cur + 1;
        const unsigned a_addr = a_frag + st_nxt * STAGE, b_addr = b_frag + st_nxt * STAGE;
        // NMFMA MFMAs of slab kt.  The memory instructions ride between them so that the matrix pipe restarts right after the
        // barrier: the first NREAD MFMAs are each preceded by one fragment read of slab kt+1, and the PW DMA requests of slab
        // kt+3 are spread over the slab (a request costs ~60-180 issue cycles: all of them up front left the pipe idle for
        // ~700 of the slab's 2300 cycles on both waves of a SIMD at once)
        constexpr int DMA_EVERY = NMFMA / PW;
        auto body = [&](auto idx) {
            constexpr int q = decltype(idx)::value;
            if constexpr (!LAST && q < NREAD) read_one<TM, TN, NP, q>(nxt, a_addr, b_addr);
            if constexpr (ISSUE && !ABL_NODMA && DMA_SPREAD && q % DMA_EVERY == 1 && q / DMA_EVERY < PW)
                issue_piece(std::integral_constant<int, q / DMA_EVERY>{}, kt + NS, st_cur);
            constexpr int pair = q % NPROD, ij = q / NPROD, i = ij / TN, j = ij % TN;
            if constexpr (NP == 3) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.b[j][PB_[pair]], cur.a[i][PA_[pair]], acc[i][j], 0, 0, 0);
            } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.b[j][PB_[pair]]),
                                                                   __builtin_bit_cast(f16x8, cur.a[i][PA_[pair]]), acc[i][j], 0, 0, 0);
            }
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (ISSUE && !ABL_NODMA && !DMA_SPREAD) issue(kt + NS, st_cur);
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (body(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, NMFMA>{});
        if constexpr (ISSUE) conv_advance();
    };
    using Steady = StepMode<NS - 2, true, false>;
    auto adv = [](int st) { return st == NS - 1 ? 0 : st + 1; };

    // nk is even and > NS (K % 32 == 0; the host picks NS): two slabs per trip while there is a slab kt + NS to request, then the
    // NS last slabs peeled (straight-line: every wait count and every DMA request is unconditional).  nk - NS steady steps: their
    // parity is NS's, so which register set holds the first peeled slab is known at compile time.
    int kt = 0, st = 0;
    for (; kt + 1 < nk - NS; kt += 2) {
        step(Steady{}, kt, st, f0, f1);
        st = adv(st);
        step(Steady{}, kt + 1, st, f1, f0);
        st = adv(st);
    }
    if constexpr (NS % 2 == 1) {   // one more steady step: requests the last slab
        step(Steady{}, kt, st, f0, f1);
        st = adv(st);
        ++kt;
    }
    // slab nk-NS+j awaits slab nk-NS+j+1 with NS-2-j younger groups still in flight; the last one awaits nothing
    auto tail = [&]<int J>(auto&& self, std::integral_constant<int, J>, Frags<TM, TN, NP>& cur, Frags<TM, TN, NP>& nxt) {
        if constexpr (J == NS - 1) {
            step(StepMode<0, false, true>{}, kt + J, st, cur, nxt);
        } else {
            step(StepMode<NS - 2 - J, false, false>{}, kt + J, st, cur, nxt);
            st = adv(st);
            self(self, std::integral_constant<int, J + 1>{}, nxt, cur);
        }
    };
    if constexpr (NS % 2 == 1) tail(tail, std::integral_constant<int, 0>{}, f1, f0);
    else tail(tail, std::integral_constant<int, 0>{}, f0, f1);

    stamp(2);
    __syncthreads();  // every wave is past its last LDS read: the stages become epilogue scratch
    stamp(4);
    float* scratch = reinterpret_cast<float*>(smem) + wave * (TM * 32 * 37);   // 36 floats per row of the strip + 1 per row for an h2 output's scale
    epilogue<TM, TN, EPI, ABL_NOSTORE>(p, acc, tile_m * BM + wm * WM, tile_n * BN + wn * WN, lane, scratch);
    if constexpr (STAMP) {
        stamp(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
        if (threadIdx.x == 0 && stamp_buf) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(stamp_buf) + (long)blockIdx.x * 12;
            for (int i = 0; i < 6; ++i) { d[i] = stamp_rt[i]; d[6 + i] = stamp_clk[i]; }
        }
    }
}

template <int BM, int BN, int VARIANT, int EPI, int NP = 3, bool KS = false, bool CONV = false, int NW = NWAVE, int NS = 3>
int launch(const Tp3Params& p0, hipStream_t s) {
    if (NS > 3 && (p0.K >> 4) <= NS) return launch<BM, BN, VARIANT, EPI, NP, KS, CONV, NW, 3>(p0, s);   // a ring deeper than the k-loop: the 3-stage one
    Tp3Params p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    constexpr size_t stage_bytes = (size_t)NS * (NP * (BM + BN) / 32) * PIECE;
    constexpr size_t epi_bytes = (size_t)NW * (BM / (NW / 4)) * 37 * sizeof(float);
    constexpr size_t ks_bytes = KS ? (size_t)BM * 64 * sizeof(float) : 0;   // ratio table: up to 64 chunks (K <= 4096)
    constexpr size_t smem = (stage_bytes + ks_bytes) > epi_bytes ? (stage_bytes + ks_bytes) : epi_bytes;
    static_assert(smem <= 160 * 1024 && (NW == 8 || smem <= 80 * 1024), "LDS budget (two 4-wave workgroups per CU)");
    auto kern = gemm_tp3_kernel<BM, BN, VARIANT, EPI, NP, KS, CONV, NW, NS>;
    static int attr_dev_mask = 0;  // per device: the opt-in for > 64 KiB of dynamic LDS is a per-device function attribute
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    if (!(attr_dev_mask & (1 << dev))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
        attr_dev_mask |= 1 << dev;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)p.tiles_m * p.tiles_n)), dim3(NW * 64), smem, s, p);
    return 0;
}

// the specialised epilogues of the vision tower's eight GEMMs (ops.py EncoderLayerFn); anything else runs the generic one
template <int BM, int VARIANT>
int launch_epi(const Tp3Params& p, int epi, hipStream_t s) {
    switch (epi) {
        case E_F32: return launch<BM, 256, VARIANT, E_F32>(p, s);                                           // data gradients
        case E_BIAS | E_F32: return launch<BM, 256, VARIANT, E_BIAS | E_F32>(p, s);                         // qkv
        case E_BIAS | E_RES | E_F32: return launch<BM, 256, VARIANT, E_BIAS | E_RES | E_F32>(p, s);         // out_proj, fc2
        case E_BIAS | E_QGELU | E_PRE | E_TP3: return launch<BM, 256, VARIANT, E_BIAS | E_QGELU | E_PRE | E_TP3>(p, s);   // fc1 (training)
        case E_BIAS | E_QGELU | E_TP3: return launch<BM, 256, VARIANT, E_BIAS | E_QGELU | E_TP3>(p, s);     // fc1 (no tape)
        case E_DQGELU | E_TP3: return launch<BM, 256, VARIANT, E_DQGELU | E_TP3>(p, s);                     // dz of the backward
        case E_BIAS | E_TP3: return launch<BM, 256, VARIANT, E_BIAS | E_TP3>(p, s);                         // qkv for the tp3 attention
        case E_TP3: return launch<BM, 256, VARIANT, E_TP3>(p, s);                                           // dO for the tp3 attention
        default: return launch<BM, 256, VARIANT, -1>(p, s);
    }
}

// compile-time epilogue code of a call, or -1 (generic) when it uses anything the specialised set does not cover
inline int epi_code(const Tp3Params& p) {
    if ((p.alpha != 1.0f && !p.a_scale && !p.a_kscale) || (p.act & TVL_ACT_POST_RESIDUAL)) return -1;
    const int act = p.act & 0xff;
    if ((act != TVL_ACT_NONE && act != TVL_ACT_QUICK_GELU && act != TVL_ACT_RELU) || (p.dact != TVL_ACT_NONE && p.dact != TVL_ACT_QUICK_GELU)) return -1;
    return (p.bias ? E_BIAS : 0) | (p.residual ? E_RES : 0) | (act == TVL_ACT_QUICK_GELU ? E_QGELU : 0) | (act == TVL_ACT_RELU ? E_RELU : 0) | (p.dact ? E_DQGELU : 0) | (p.pre_out ? E_PRE : 0) |
           (p.C ? E_F32 : 0) | (p.Cp ? E_TP3 : 0) | ((p.a_scale || p.a_kscale) ? E_RSCALE : 0) | (p.Ch2 ? E_H2OUT : 0);
}

}  // namespace

// per-tile translation units (parallel make): Tp3Params is TU-local, hence the opaque pointer
int tvl_gemm_tp3_t128(const void* params, int epi, hipStream_t s);
int tvl_gemm_tp3_t256(const void* params, int epi, hipStream_t s);
int tvl_gemm_h2_w4(const void* params, int bm, int epi, hipStream_t s);   // gemm_h2_w4.hip: 4-wave workgroups, two per CU
int tvl_gemm_h2_ns4(const void* params, int bm, int epi, hipStream_t s);  // gemm_h2_ns4.hip / _ns5.hip: deeper DMA rings
int tvl_gemm_h2_ns5(const void* params, int bm, int epi, hipStream_t s);
