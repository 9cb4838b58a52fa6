// Vision-tower attention (d_h = 64, no masks), forward and backward, on TWO-piece fp16 operands ("h2", tp3.h / gemm_h2.hip): the same
// kernels as attention_tp3.hip -- same tiles (32 keys / queries from the sample's first row, LDS-DMA, double-buffered, one barrier per
// tile), same fragment mappings, same flash structure -- with every product as THREE MFMAs (h0 h0 + h0 h1 + h1 h0 on
// v_mfma_f32_32x32x16_f16) instead of six, and two thirds of the operand bytes.  Both families are power-limited (all-zero operands:
// +18 % forward, +34 % backward), so the MFMA count is what sets their time (profiles/r2_attention_experiments.md).
//
// Scales (exact powers of two; every accumulator is unscaled in fp32 before it meets another quantity):
//   QKV image   one scale for the tensor, from the QKV GEMM epilogue's bound (tvl_gemm_h2_out, per-tensor mode): qkv_inv[0]
//   P           in [0, 1]: the static 2^13
//   dO image    one scale for the tensor, from the out-projection data gradient's epilogue: do_inv[0]
//   dS (dQ)     per query row q -- the contraction of dQ = dS K runs over keys -- from |dS[q, k]| <= ||dO_q|| max_k ||V_k|| + |delta_q|
//               with ||dO_q|| exact (the lane holds the row) and ||V_k|| <= 8 * 2^14 * qkv_inv
//   dS (dK)     one scale per (sample, head) -- the contraction of dK = dS^T Q runs over queries -- from the largest ||dO_q|| of the
//               (sample, head), which the dQ kernel leaves in dnorm_max by atomicMax on the float's bits
// Outputs: O as an h2 image that shares the QKV scale (|O| <= max |V|) and dQ | dK | dV as an h2 image with one exact scale per
// (row, 64-column block) -- each such block is written by one lane pair that holds all of it (store_block) -- for tvl_gemm_h2 /
// tvl_gemm_h2_ks; or both as tp3 images (three bf16 pieces) for the tp3 consumers.
#include <type_traits>
#include <utility>
#include "common.h"
#include "tp3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int PIECE = 1024, BLK2 = 2 * PIECE;
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int DH = 64;
constexpr float P_SCALE = 8192.0f, P_INV = 1.0f / 8192.0f;
constexpr int VCH = BLK2 + 128;   // forward: LDS stride of a V d-chunk (odd chunks start half a bank row later: conflict-free transposed reads)

__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(size_t)lds_byte, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, unsigned lds_byte) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(size_t)lds_byte, 4, 0, 0);
}
template <int OFF>
__device__ __forceinline__ f16x8 lds_b128(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_f4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ f16x8 frag_of(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(f16x8, make_uint4(a, b, c, d)); }

// counted LDS waits carrying the fragments they guard (attention_tp3.hip)
template <int N>
__device__ __forceinline__ void wait_k(f16x8 (&f)[2]) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_v(u32x2 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N)); }
template <int OFF>
__device__ __forceinline__ void read_k(f16x8 (&f)[2], unsigned a) { f[0] = lds_b128<OFF>(a); f[1] = lds_b128<OFF + PIECE>(a); }
// transposed fragments {piece 0 lo, hi, piece 1 lo, hi} of one 16-row step of a 32-column block; HI = byte distance of "rows + 8"
template <int OFF, int HI>
__device__ __forceinline__ void read_t(u32x2 (&f)[4], unsigned a) {
    f[0] = lds_tr<OFF>(a); f[1] = lds_tr<OFF + HI>(a); f[2] = lds_tr<OFF + PIECE>(a); f[3] = lds_tr<OFF + PIECE + HI>(a);
}

// h0 h1 + h1 h0 + h0 h0 (smallest products first)
__device__ __forceinline__ f32x16 mma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma3_t(const u32x2 (&v)[4], const f16x8 (&b)[2], f32x16 acc) {
    const f16x8 a[2] = {frag_of(v[0][0], v[0][1], v[1][0], v[1][1]), frag_of(v[2][0], v[2][1], v[3][0], v[3][1])};
    return mma3(a, b, acc);
}

// 16 accumulator-layout values (already scaled) -> fragments of rows 0-15 / 16-31: h0 by truncation (v_cvt_pkrtz), h1 = fp16(x - h0)
__device__ __forceinline__ void pieces_of(const float (&x)[16], f16x8 (&f0)[2], f16x8 (&f1)[2]) {
    unsigned p0[8], p1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const auto h = __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]);
        p0[i] = __builtin_bit_cast(unsigned, h);
        const auto r = __builtin_amdgcn_cvt_pkrtz(x[2 * i] - (float)h[0], x[2 * i + 1] - (float)h[1]);
        p1[i] = __builtin_bit_cast(unsigned, r);
    }
    f0[0] = frag_of(p0[0], p0[1], p0[2], p0[3]); f0[1] = frag_of(p1[0], p1[1], p1[2], p1[3]);
    f1[0] = frag_of(p0[4], p0[5], p0[6], p0[7]); f1[1] = frag_of(p1[4], p1[5], p1[6], p1[7]);
}

// fragment (row m, 16 columns of column block kb) of an h2 image straight into registers: 2 x 16-byte loads
__device__ __forceinline__ void row_frags(const unsigned char* __restrict__ img, int kblocks, long m, int kb, int hh, f16x8 (&out)[2]) {
    const unsigned char* src = img + ((m >> 5) * kblocks + kb) * (long)BLK2 + (hh * 32 + (int)(m & 31)) * 16;
    out[0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(src));
    out[1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(src + PIECE));
}
// the 8 fp32 values a lane's two pieces encode (still scaled)
__device__ __forceinline__ void join8(const f16x8 (&f)[2], float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)f[0][e] + (float)f[1][e];
}
__device__ __forceinline__ const unsigned char* unit_at(const unsigned char* colblk, long blk_stride, long m, int hh) {
    return colblk + (m >> 5) * blk_stride + (hh * 32 + (int)(m & 31)) * 16;
}
__device__ __forceinline__ bool prio_of_workgroup() { return ((blockIdx.x >> 3) >> 5) & 1; }
__device__ __forceinline__ int xcd_vid() {
    const int total = (int)gridDim.x, L = (int)blockIdx.x, per = total / 8;
    return L < per * 8 ? (L % 8) * per + L / 8 : L;
}
__device__ __forceinline__ float half_sum(float x) { return xor32_sum(x); }

// A . B accumulated over the four 16-k steps of d_h = 64: A fragments double-buffered out of LDS (piece blocks BLK2 apart), B in registers
__device__ __forceinline__ f32x16 mma_rows(unsigned a_rd, const f16x8 (&bq)[4][2]) {
    const f32x16 zero = {};
    f16x8 ka[2], kb[2];
    read_k<0>(ka, a_rd); read_k<BLK2>(kb, a_rd);
    wait_k<2>(ka); f32x16 acc = mma3(ka, bq[0], zero);
    read_k<2 * BLK2>(ka, a_rd); wait_k<2>(kb); acc = mma3(kb, bq[1], acc);
    read_k<3 * BLK2>(kb, a_rd); wait_k<2>(ka); acc = mma3(ka, bq[2], acc);
    wait_k<0>(kb); acc = mma3(kb, bq[3], acc);
    return acc;
}
// acc[d] += T^T . X for both 32-column blocks d of a 32-row tile T in the image's own piece layout (transposed reads at t_rd)
__device__ __forceinline__ void mma_cols(unsigned t_rd, const f16x8 (&x0)[2], const f16x8 (&x1)[2], f32x16 (&acc)[2]) {
    u32x2 va[4], vb[4];
    read_t<0, 128>(va, t_rd); read_t<0, 128>(vb, t_rd + 256);
    wait_v<4>(va); acc[0] = mma3_t(va, x0, acc[0]);
    read_t<2 * BLK2, 128>(va, t_rd); wait_v<4>(vb); acc[0] = mma3_t(vb, x1, acc[0]);
    read_t<2 * BLK2, 128>(vb, t_rd + 256); wait_v<4>(va); acc[1] = mma3_t(va, x0, acc[1]);
    wait_v<0>(vb); acc[1] = mma3_t(vb, x1, acc[1]);
}

struct FwdP {
    const unsigned char* qkv; int kb; const float* qkv_inv;   // h2 image of [B*T, 3*H*64], kb = 3*H*64/16; its inverse scale [1]
    unsigned char* o_tp3; int o_kb;                            // image of O [B*T, H*64]: tp3, or (o_h2 != 0) h2 with the QKV image's scale
    int o_h2;
    float* lse;
    int B, H, T; float scale;
};

constexpr int K_STAGE = 8 * PIECE, V_STAGE = 4 * VCH;
constexpr int FWD_LDS = 2 * K_STAGE + 2 * V_STAGE;

__global__ __launch_bounds__(256, 2) void attn_fwd_h2_kernel(FwdP p) {
    TVL_KERNEL_ENTRY();
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = p.T, D = p.H * DH;
    const int nqb = (T + 127) / 128;
    const int vid = xcd_vid();
    const int qb = vid % nqb, head = (vid / nqb) % p.H, b = vid / (nqb * p.H);
    const int qi = qb * 128 + wave * 32 + l31;
    const long m_q = (long)b * T + (qi < T ? qi : T - 1);
    if (prio_of_workgroup()) __builtin_amdgcn_s_setprio(2);

    f16x8 qf[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) row_frags(p.qkv, p.kb, m_q, head * 4 + s, h, qf[s]);
    f32x16 acc_o[2] = {};
    float m_run = NEG_BIG, l_run = 0.f;
    const float inv_q = p.qkv_inv[0];
    const float sc2 = p.scale * LOG2E * inv_q * inv_q;   // raw S = s^2 times the true score

    const long row_lo = (long)b * T;
    const int nkt = (T + 31) / 32;
    const long tile_stride = (long)p.kb * BLK2;
    const unsigned char* k_col = p.qkv + (long)((D + head * DH) / 16) * BLK2;
    const unsigned char* v_col = p.qkv + (long)((2 * D + head * DH) / 16) * BLK2;
    const unsigned k_lane = (unsigned)(unit_at(k_col, tile_stride, row_lo + (l31 < T ? l31 : T - 1), h) - k_col);
    // V pieces land key-interleaved: LDS unit L (16 B) of a piece = (key L >> 1, d-half L & 1)
    const unsigned v_lane = (unsigned)(unit_at(v_col, tile_stride, row_lo + ((lane >> 1) < T ? (lane >> 1) : T - 1), lane & 1) - v_col);
    auto issue = [&](int kt) {
        const unsigned kd = lds0 + (kt & 1) * K_STAGE, vd = lds0 + 2 * K_STAGE + (kt & 1) * V_STAGE;
        const unsigned char* ks = k_col + (k_lane + (32 * kt + l31 < T ? (unsigned)kt * (unsigned)tile_stride : 0u));
        const unsigned char* vs = v_col + (v_lane + (32 * kt + (lane >> 1) < T ? (unsigned)kt * (unsigned)tile_stride : 0u));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + 4 * i;   // 0..7: (d-chunk pc / 2, piece pc % 2), consecutive in the image
            glds16(ks + pc * PIECE, kd + pc * PIECE);
            glds16(vs + pc * PIECE, vd + (pc / 2) * VCH + (pc % 2) * PIECE);
        }
    };
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = (4 * h + (li >> 2)) * 32 + (li & 3) * 8 + g1 * VCH;
    const unsigned k_rd = lds0 + lane * 16, v_rd = lds0 + 2 * K_STAGE + tr_off;

    // One key tile.  Vector work is what bounds this kernel next to its 24 MFMAs (stamps: profiles/r2_attention_experiments.md), so:
    // no clamps (v_exp_f32 of a hugely negative argument is 0, of -1e30 * sc2 too), the key mask exists only in the peeled last tile, and
    // the rescaling of O by alpha is skipped whenever no lane of the wave saw a new maximum (alpha = 1 exactly: most tiles after the first
    // few).  P's static scale 2^13 stays a multiplication: folded into the exponent it would move the argument of the dominant terms
    // from ~0 to ~13 and cost them three bits (the argument's rounding error becomes P's relative error).
    auto tile = [&]<bool LAST>(std::integral_constant<bool, LAST>, int kt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (!LAST) issue(kt + 1);
        const unsigned kb_ = k_rd + (kt & 1) * K_STAGE, vb_ = v_rd + (kt & 1) * V_STAGE;
        f32x16 sc = mma_rows(kb_, qf);                 // S^T [key (register), query (lane)], scaled by s^2
        u32x2 va[4], vb[4];
        read_t<0, 256>(va, vb_); read_t<0, 256>(vb, vb_ + 512);
        if constexpr (LAST) {
            const int key0 = 32 * kt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = (key0 + (r & 3) + 8 * (r >> 2) < T) ? sc[r] : NEG_BIG;
        }
        // raw-unit running maximum (sc2 > 0 commutes with max); masked entries: -1e30 * sc2 is still hugely negative for any sane scale
        float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
        mx = fmaxf(mx, sc[15]);
        mx = xor32_max(mx);
        const bool grew = mx > m_run;
        if (__builtin_amdgcn_ballot_w64(grew)) {       // wave-uniform: some query of this wave has a new maximum
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc2);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_o[0][r] *= alpha; acc_o[1][r] *= alpha; }
        }
        const float nm = -m_run * sc2;
        float pv[16];
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float e0 = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, nm)), e1 = __builtin_amdgcn_exp2f(fmaf(sc[r + 1], sc2, nm));
            rs0 += e0; rs1 += e1;
            pv[r] = e0 * P_SCALE; pv[r + 1] = e1 * P_SCALE;
        }
        l_run += rs0 + rs1;
        f16x8 pf0[2], pf1[2];
        pieces_of(pv, pf0, pf1);
        wait_v<4>(va); acc_o[0] = mma3_t(va, pf0, acc_o[0]);
        read_t<2 * VCH, 256>(va, vb_); wait_v<4>(vb); acc_o[0] = mma3_t(vb, pf1, acc_o[0]);
        read_t<2 * VCH, 256>(vb, vb_ + 512); wait_v<4>(va); acc_o[1] = mma3_t(va, pf0, acc_o[1]);
        wait_v<0>(vb); acc_o[1] = mma3_t(vb, pf1, acc_o[1]);
    };
    issue(0);
    for (int kt = 0; kt < nkt - 1; ++kt) tile(std::false_type{}, kt);
    tile(std::true_type{}, nkt - 1);
    const float l_tot = xor32_sum(l_run);
    // P's scale and the softmax denominator; V's scale is undone for a tp3 image and KEPT for an h2 image: O is a convex combination of
    // V rows, so |O s| <= max |V s| < 2^14 -- the QKV image's scale is a valid scale for O as well
    const float inv = (p.o_h2 ? 1.0f : inv_q) * P_INV / l_tot;
    if (qi < T) {
        const long m = (long)b * T + qi;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {acc_o[d][4 * g] * inv, acc_o[d][4 * g + 1] * inv, acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv};
                const int col = head * DH + d * 32 + 8 * g + 4 * h;
                if (p.o_h2) h2::store4(p.o_tp3, p.o_kb, m, col, v);
                else tp3::store4(p.o_tp3, p.o_kb, m, col, v);
            }
        if (h == 0 && p.lse) p.lse[((long)b * p.H + head) * T + qi] = (m_run * sc2 + log2f(l_tot)) * LN2;
    }
}

// ---- backward --------------------------------------------------------------------------------------------------------------------
struct BwdP {
    const unsigned char* qkv; int kb; const float* qkv_inv;     // h2 image of packed QKV + its inverse scale [1]
    const unsigned char* o_img; int o_h2;                        // image of O [B*T, H*64] (the forward's output): tp3, or h2 with the QKV scale
    const unsigned char* do_img; int o_kb; const float* do_inv;  // h2 image of dO [B*T, H*64] + its inverse scale [1]; o_kb = H*64/16
    const float* lse;
    float* delta;                  // [B, H, T]: written by the dQ kernel, read by the dK/dV kernel
    unsigned* dnorm_max;           // [B, H]: bits of max_q ||dO_q|| (true units) per (sample, head); zeroed by the entry point
    unsigned char* g_img;          // image of dQ | dK | dV [B*T, 3*H*64]: tp3, or (g_h2 != 0) h2 with one scale per (row, 64-column block):
    int g_h2; float* g_kscale;     // g_kscale[m * 3H + part * H + head] = that block's inverse scale (exact: the writer holds the block)
    int B, H, T; float scale;
    int only_blk;                  // >= 0: only the 128-row block `only_blk` of every sample gets its dQ / dK / dV (the caller needs no other rows: the
                                   // first layer under the prompts); the dQ kernel's other workgroups still leave delta and the dO norms behind
};

constexpr int BWD_STAGE = 16 * PIECE + 256;   // two 8-piece tiles + 64 floats (log-sum-exp | delta of a query tile; dK/dV kernel only)
constexpr int BWD_LDS = 2 * BWD_STAGE;

// one (row, 64-column block) of the packed gradient: the lane pair (l31, l31 + 32) holds its 64 values (factor f already applied)
__device__ __forceinline__ void store_block(const BwdP& p, long m, int col0, int chunk, int h, const f32x16 (&acc)[2], float f) {
    if (p.g_h2) {
        float amax = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(acc[d][r] * f));
        amax = xor32_max(amax);
        const float inv = h2::inv_scale_of(amax);
        if (h == 0) p.g_kscale[m * (3L * p.H) + chunk] = inv;
        f *= 1.0f / inv;
    }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v[4] = {acc[d][4 * g] * f, acc[d][4 * g + 1] * f, acc[d][4 * g + 2] * f, acc[d][4 * g + 3] * f};
            const int col = col0 + d * 32 + 8 * g + 4 * h;
            if (p.g_h2) h2::store4(p.g_img, p.kb, m, col, v);
            else tp3::store4(p.g_img, p.kb, m, col, v);
        }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_h2_kernel(BwdP p) {
    TVL_KERNEL_ENTRY();
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = p.T, D = p.H * DH;
    const int nqb = (T + 127) / 128;
    const int vid = xcd_vid();
    const int qb = vid % nqb, head = (vid / nqb) % p.H, b = vid / (nqb * p.H);
    const int qi = qb * 128 + wave * 32 + l31;
    const int qrow = qi < T ? qi : T - 1;
    const long m_q = (long)b * T + qrow;
    const float inv_q = p.qkv_inv[0], inv_do = p.do_inv[0];

    f16x8 qf[4][2], dof[4][2];
    float dl = 0.f, dn2 = 0.f;   // delta = sum dO * O and ||dO_q, head||^2 (true units)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        row_frags(p.qkv, p.kb, m_q, head * 4 + s, h, qf[s]);
        row_frags(p.do_img, p.o_kb, m_q, head * 4 + s, h, dof[s]);
        float x[8], y[8];
        join8(dof[s], x);
        if (p.o_h2) {
            f16x8 of[2];
            row_frags(p.o_img, p.o_kb, m_q, head * 4 + s, h, of);
            join8(of, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= inv_q;
        } else {
            const unsigned char* osrc = p.o_img + ((m_q >> 5) * p.o_kb + head * 4 + s) * (long)tp3::BLK + (h * 32 + (int)(m_q & 31)) * 16;
            const uint4 c[3] = {*reinterpret_cast<const uint4*>(osrc), *reinterpret_cast<const uint4*>(osrc + tp3::PIECE),
                                *reinterpret_cast<const uint4*>(osrc + 2 * tp3::PIECE)};
            tp3::join8(c, y);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] *= inv_do; dl += x[e] * y[e]; dn2 += x[e] * x[e]; }
    }
    dl = half_sum(dl);
    const float dn = sqrtf(half_sum(dn2));
    const long stat = ((long)b * p.H + head) * T + qrow;
    if (h == 0 && qi < T) p.delta[stat] = dl;
    {   // the (sample, head)'s largest ||dO_q||, for the dK/dV kernel's dS scale
        float w = qi < T ? dn : 0.f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) w = fmaxf(w, __shfl_xor(w, o, 64));
        if (lane == 0) atomicMax(p.dnorm_max + (long)b * p.H + head, __builtin_bit_cast(unsigned, w));
    }
    if (p.only_blk >= 0 && qb != p.only_blk) return;   // (workgroup-uniform, before any barrier) delta and dnorm_max are all this block owed
    const float sc2 = p.scale * LOG2E * inv_q * inv_q;
    const float nlse2 = -p.lse[stat] * LOG2E;
    const float dp_unscale = inv_q * inv_do;
    // |dS[q, k]| <= P (|dP| + |delta|) <= ||dO_q|| max_k ||V_k|| + |delta_q|, ||V_k|| <= 8 * 2^14 * inv_q
    const float ds_inv = h2::inv_scale_of(dn * (8.0f * 16384.0f) * inv_q + fabsf(dl));
    const float ds_scale = 1.0f / ds_inv;

    f32x16 acc_dq[2] = {};
    const long row_lo = (long)b * T;
    const int nkt = (T + 31) / 32;
    const long tile_stride = (long)p.kb * BLK2;
    const unsigned char* k_col = p.qkv + (long)((D + head * DH) / 16) * BLK2;
    const unsigned char* v_col = p.qkv + (long)((2 * D + head * DH) / 16) * BLK2;
    const unsigned k_lane = (unsigned)(unit_at(k_col, tile_stride, row_lo + (l31 < T ? l31 : T - 1), h) - k_col);
    const long v_minus_k = v_col - k_col;
    auto issue = [&](int kt) {
        const unsigned dst = lds0 + (kt & 1) * BWD_STAGE;
        const unsigned char* ks = k_col + (k_lane + (32 * kt + l31 < T ? (unsigned)kt * (unsigned)tile_stride : 0u));
        const unsigned char* vs = ks + v_minus_k;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + 4 * i;
            glds16(ks + pc * PIECE, dst + pc * PIECE);
            glds16(vs + pc * PIECE, dst + (8 + pc) * PIECE);
        }
    };
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = ((li & 3) >> 1) * 512 + (4 * h + (li >> 2)) * 16 + (li & 1) * 8 + g1 * BLK2;

    // dS = P (dP - delta), scaled for its fp16 pieces: the factors that do not depend on the key are folded once per lane
    const float c_dp = dp_unscale * ds_scale, c_dl = dl * ds_scale;
    auto tile = [&]<bool LAST>(std::integral_constant<bool, LAST>, int kt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (!LAST) issue(kt + 1);
        const unsigned st = lds0 + (kt & 1) * BWD_STAGE;
        f32x16 sc = mma_rows(st + lane * 16, qf);                  // S^T  [key (register), query (lane)]
        f32x16 dp = mma_rows(st + 8 * PIECE + lane * 16, dof);     // dP^T
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, nlse2)) * fmaf(dp[r], c_dp, -c_dl);
        if constexpr (LAST) {   // keys beyond the sample exist only in the last tile
            const int key0 = 32 * kt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = (key0 + (r & 3) + 8 * (r >> 2) < T) ? ds[r] : 0.f;
        }
        f16x8 x0[2], x1[2];
        pieces_of(ds, x0, x1);
        mma_cols(st + tr_off, x0, x1, acc_dq);                     // dQ^T += K^T . dS^T
    };
    issue(0);
    for (int kt = 0; kt < nkt - 1; ++kt) tile(std::false_type{}, kt);
    tile(std::true_type{}, nkt - 1);
    if (qi < T) store_block(p, (long)b * T + qi, head * DH, head, h, acc_dq, p.scale * inv_q * ds_inv);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_h2_kernel(BwdP p) {
    TVL_KERNEL_ENTRY();
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = p.T, D = p.H * DH;
    const int nkb = (T + 127) / 128;
    const int vid = xcd_vid();
    const int kblk = vid % nkb, head = (vid / nkb) % p.H, b = vid / (nkb * p.H);
    if (p.only_blk >= 0 && kblk != p.only_blk) return;
    const int ki = kblk * 128 + wave * 32 + l31;
    const bool key_ok = ki < T;
    const long m_k = (long)b * T + (key_ok ? ki : T - 1);
    const float inv_q = p.qkv_inv[0], inv_do = p.do_inv[0];

    f16x8 kf[4][2], vf[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        row_frags(p.qkv, p.kb, m_k, (D + head * DH) / 16 + s, h, kf[s]);
        row_frags(p.qkv, p.kb, m_k, (2 * D + head * DH) / 16 + s, h, vf[s]);
    }
    f32x16 acc_dk[2] = {}, acc_dv[2] = {};
    const float sc2 = p.scale * LOG2E * inv_q * inv_q;
    const float dp_unscale = inv_q * inv_do;
    // one dS scale per (sample, head): |dS| <= 2 max_q ||dO_q|| max_k ||V_k||
    const float dn_max = __builtin_bit_cast(float, p.dnorm_max[(long)b * p.H + head]);
    const float ds_inv = h2::inv_scale_of(2.0f * dn_max * (8.0f * 16384.0f) * inv_q);
    const float ds_scale = 1.0f / ds_inv;

    const long row_lo = (long)b * T;
    const int nqt = (T + 31) / 32;
    const long q_stride = (long)p.kb * BLK2, d_stride = (long)p.o_kb * BLK2;
    const unsigned char* q_col = p.qkv + (long)((head * DH) / 16) * BLK2;
    const unsigned char* d_col = p.do_img + (long)((head * DH) / 16) * BLK2;
    const float* stat_base = (lane < 32 ? p.lse : p.delta) + ((long)b * p.H + head) * T;
    const unsigned q_lane = (unsigned)(unit_at(q_col, q_stride, row_lo + (l31 < T ? l31 : T - 1), h) - q_col);
    const unsigned d_lane = (unsigned)(unit_at(d_col, d_stride, row_lo + (l31 < T ? l31 : T - 1), h) - d_col);
    auto issue = [&](int qt) {
        const unsigned dst = lds0 + (qt & 1) * BWD_STAGE;
        const bool in_sample = 32 * qt + l31 < T;
        const unsigned char* qs = q_col + (q_lane + (in_sample ? (unsigned)qt * (unsigned)q_stride : 0u));
        const unsigned char* ds_ = d_col + (d_lane + (in_sample ? (unsigned)qt * (unsigned)d_stride : 0u));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + 4 * i;
            glds16(qs + pc * PIECE, dst + pc * PIECE);
            glds16(ds_ + pc * PIECE, dst + (8 + pc) * PIECE);
        }
        if (wave == 0) {
            const int q = 32 * qt + l31;
            glds4(stat_base + (q < T ? q : T - 1), dst + 16 * PIECE);
        }
    };
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = ((li & 3) >> 1) * 512 + (4 * h + (li >> 2)) * 16 + (li & 1) * 8 + g1 * BLK2;

    // dS = P (dP - delta) with the key-independent factors folded once.  Keys are on the LANE here, so a lane of a key beyond the sample
    // only ever pollutes its own (never stored) column: the one mask left is the query mask of the peeled last tile.
    const float c_ds = ds_scale, c_dp = dp_unscale * c_ds;
    auto tile = [&]<bool LAST>(std::integral_constant<bool, LAST>, int qt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (!LAST) issue(qt + 1);
        const unsigned st = lds0 + (qt & 1) * BWD_STAGE;
        f32x16 sc = mma_rows(st + lane * 16, kf);                  // S   [query (register), key (lane)]
        f32x16 dp = mma_rows(st + 8 * PIECE + lane * 16, vf);      // dP
        f32x4 l4[4], d4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) l4[g] = lds_f4<0>(st + 16 * PIECE + (8 * g + 4 * h) * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) d4[g] = lds_f4<128>(st + 16 * PIECE + (8 * g + 4 * h) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l4[0]), "+v"(l4[1]), "+v"(l4[2]), "+v"(l4[3]), "+v"(d4[0]), "+v"(d4[1]), "+v"(d4[2]), "+v"(d4[3]));
        float pv[16], dsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, -LOG2E * l4[r >> 2][r & 3]));
            pv[r] = pr * P_SCALE;
            dsv[r] = pr * fmaf(dp[r], c_dp, -c_ds * d4[r >> 2][r & 3]);
        }
        if constexpr (LAST) {
            const int q0 = 32 * qt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = q0 + (r & 3) + 8 * (r >> 2) < T;
                pv[r] = ok ? pv[r] : 0.f;
                dsv[r] = ok ? dsv[r] : 0.f;
            }
        }
        f16x8 x0[2], x1[2];
        pieces_of(pv, x0, x1);
        mma_cols(st + 8 * PIECE + tr_off, x0, x1, acc_dv);         // dV^T += dO^T . P
        pieces_of(dsv, x0, x1);
        mma_cols(st + tr_off, x0, x1, acc_dk);                     // dK^T += Q^T . dS
    };
    issue(0);
    for (int qt = 0; qt < nqt - 1; ++qt) tile(std::false_type{}, qt);
    tile(std::true_type{}, nqt - 1);
    if (key_ok) {
        const long m = (long)b * T + ki;
        store_block(p, m, D + head * DH, p.H + head, h, acc_dk, p.scale * inv_q * ds_inv);
        store_block(p, m, 2 * D + head * DH, 2 * p.H + head, h, acc_dv, inv_do * P_INV);
    }
}

// rows b*T + row0 .. + n - 1 of the packed-gradient image (two fp16 pieces, one scale per (row, 64-column block)) back to fp32 [B*n, K]
__global__ __launch_bounds__(256) void h2k_gather_rows_kernel(const unsigned char* __restrict__ img, const float* __restrict__ kscale, int K, int B, int T,
                                                              int row0, int n, float* __restrict__ out) {
    TVL_KERNEL_ENTRY();
    const int units = K >> 3;
    const long total = (long)B * n * units;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int u = (int)(i % units);
        const long rj = i / units;
        const long m = (rj / n) * T + row0 + (rj % n);
        const unsigned char* src = img + ((m >> 5) * (K >> 4) + (u >> 1)) * (long)BLK2 + (((u & 1) * 32 + (int)(m & 31)) * 16);
        const f16x8 h0 = *reinterpret_cast<const f16x8*>(src), h1 = *reinterpret_cast<const f16x8*>(src + PIECE);
        const float inv = kscale[m * (K >> 6) + (u >> 3)];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((float)h0[e] + (float)h1[e]) * inv;
        float* o = out + rj * K + u * 8;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace

extern "C" int tvl_h2k_gather_rows(const void* img, const float* kscale, int32_t K, int32_t B, int32_t T, int32_t row0, int32_t n, float* out, tvlStream_t stream) {
    TVL_REQUIRE(img && kscale && out && K > 0 && K % 64 == 0 && B > 0 && T > 0 && n > 0 && row0 >= 0 && row0 + n <= T, "tvl_h2k_gather_rows: bad arguments");
    TVL_REQUIRE(tvl_aligned16(img) && tvl_aligned16(out), "tvl_h2k_gather_rows: operands must be 16-byte aligned");
    const long total = (long)B * n * (K / 8);
    long nb = (total + 255) / 256;
    nb = nb > 4096 ? 4096 : nb;
    hipLaunchKernelGGL(h2k_gather_rows_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned char*>(img), kscale,
                       K, B, T, row0, n, out);
    TVL_LAUNCH_CHECK("tvl_h2k_gather_rows");
    return 0;
}

// O = softmax(Q K^T * scale) V on the h2 image of packed QKV (one tensor scale: qkv_inv[1]); O as a tp3 image or (o_as_h2 != 0) as an h2 image
// that shares the QKV image's scale; lse [B, H, T].
extern "C" int tvl_attn_h2_fwd(const void* qkv_h2, const float* qkv_inv, void* o_tp3, int32_t o_as_h2, float* lse, int32_t B, int32_t H, int32_t T,
                               float scale, tvlStream_t stream) {
    TVL_REQUIRE(qkv_h2 && qkv_inv && o_tp3, "tvl_attn_h2_fwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && T > 0 && scale > 0.f, "tvl_attn_h2_fwd: bad shape / scale");
    TVL_REQUIRE(tvl_aligned16(qkv_h2) && tvl_aligned16(o_tp3), "tvl_attn_h2_fwd: images must be 16-byte aligned");
    TVL_REQUIRE((long)((T + 127) / 128) * H * B < (1L << 31), "tvl_attn_h2_fwd: grid too large");
    FwdP p;
    p.qkv = reinterpret_cast<const unsigned char*>(qkv_h2); p.kb = 3 * H * DH / 16; p.qkv_inv = qkv_inv;
    p.o_tp3 = reinterpret_cast<unsigned char*>(o_tp3); p.o_kb = H * DH / 16; p.o_h2 = o_as_h2; p.lse = lse;
    p.B = B; p.H = H; p.T = T; p.scale = scale;
    dim3 grid((unsigned)((T + 127) / 128 * H * B));
    // occupancy experiment knob: extra dynamic LDS (KiB, <= 64 KiB in total without an opt-in) lowers the workgroups per CU
    static const int lds_pad_f = getenv("TVL_ATTN_LDS_PAD_KB") ? atoi(getenv("TVL_ATTN_LDS_PAD_KB")) * 1024 : 0;
    hipLaunchKernelGGL(attn_fwd_h2_kernel, grid, dim3(256), FWD_LDS + lds_pad_f, reinterpret_cast<hipStream_t>(stream), p);
    TVL_LAUNCH_CHECK("tvl_attn_h2_fwd");
    return 0;
}

// Backward: dQ | dK | dV as the tp3 image of the packed gradient, or (g_as_h2 != 0) as an h2 image with one exact scale per (row, 64-column
// block) in g_kscale [B*T, 3*H] (the A operand of tvl_gemm_h2_ks).  delta: [B, H, T] fp32 workspace; dnorm_ws: [B, H] 4-byte workspace.
extern "C" int tvl_attn_h2_bwd(const void* qkv_h2, const float* qkv_inv, const void* o_tp3, int32_t o_is_h2, const void* do_h2, const float* do_inv, const float* lse,
                               float* delta, void* dnorm_ws, void* dqkv_tp3, int32_t g_as_h2, float* g_kscale, int32_t B, int32_t H, int32_t T, float scale,
                               int32_t only_block, tvlStream_t stream) {
    TVL_REQUIRE(only_block < (T + 127) / 128, "tvl_attn_h2_bwd: only_block %d beyond the %d row blocks of a sample", only_block, (T + 127) / 128);
    TVL_REQUIRE(qkv_h2 && qkv_inv && o_tp3 && do_h2 && do_inv && lse && delta && dnorm_ws && dqkv_tp3, "tvl_attn_h2_bwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && T > 0 && scale > 0.f, "tvl_attn_h2_bwd: bad shape / scale");
    TVL_REQUIRE(tvl_aligned16(qkv_h2) && tvl_aligned16(o_tp3) && tvl_aligned16(do_h2) && tvl_aligned16(dqkv_tp3), "tvl_attn_h2_bwd: images must be 16-byte aligned");
    TVL_REQUIRE((long)((T + 127) / 128) * H * B < (1L << 31), "tvl_attn_h2_bwd: grid too large");
    TVL_REQUIRE(!g_as_h2 || g_kscale, "tvl_attn_h2_bwd: an h2 gradient image needs g_kscale [B*T, 3*H]");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dnorm_ws, 0, sizeof(unsigned) * (size_t)B * H, s);
    TVL_REQUIRE(e == hipSuccess, "tvl_attn_h2_bwd: memset failed: %s", hipGetErrorString(e));
    BwdP p{};
    p.only_blk = only_block < 0 ? -1 : only_block;
    p.qkv = reinterpret_cast<const unsigned char*>(qkv_h2); p.kb = 3 * H * DH / 16; p.qkv_inv = qkv_inv;
    p.o_img = reinterpret_cast<const unsigned char*>(o_tp3); p.o_h2 = o_is_h2; p.do_img = reinterpret_cast<const unsigned char*>(do_h2); p.o_kb = H * DH / 16; p.do_inv = do_inv;
    p.lse = lse; p.delta = delta; p.dnorm_max = reinterpret_cast<unsigned*>(dnorm_ws); p.g_img = reinterpret_cast<unsigned char*>(dqkv_tp3);
    p.g_h2 = g_as_h2; p.g_kscale = g_kscale;
    p.B = B; p.H = H; p.T = T; p.scale = scale;
    dim3 grid((unsigned)((T + 127) / 128 * H * B));
    static const int lds_pad_b = getenv("TVL_ATTN_LDS_PAD_KB") ? atoi(getenv("TVL_ATTN_LDS_PAD_KB")) * 1024 : 0;
    hipLaunchKernelGGL(attn_bwd_dq_h2_kernel, grid, dim3(256), BWD_LDS + lds_pad_b, s, p);      // writes delta and dnorm_max ...
    hipLaunchKernelGGL(attn_bwd_dkdv_h2_kernel, grid, dim3(256), BWD_LDS + lds_pad_b, s, p);    // ... which this one reads (same stream)
    TVL_LAUNCH_CHECK("tvl_attn_h2_bwd");
    return 0;
}
