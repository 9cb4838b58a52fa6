// Vision-tower attention (d_h = 64, no masks) on operands that arrive PRE-SPLIT: Q, K, V are read from the tp3 image of the packed
// QKV matrix [B*T, 3*H*64] that the QKV GEMM's epilogue wrote (three bf16 pieces per element in MFMA-fragment order, tp3.h).
//
// Same arithmetic as attention_bf16s.hip (flash structure, 6-MFMA piece products, fp32 accumulate):
//     S^T = K . Q^T       A = K fragment = one tp3 piece (32 keys x 16 d), straight from LDS by ds_read_b128 at lane * 16
//     O^T += V^T . P^T    A = V^T fragment: the tp3 V piece read with ds_read_b64_tr_b16, B = P^T built from the S^T accumulators
// What changed is how operands reach the matrix cores.  attention_bf16s.hip re-splits every fp32 K / V tile in every workgroup
// (~6 VALU per MFMA: the kernels ran at the vector rate, 0.23-0.36 of the MFMA ceiling) and fills LDS with ds_write.  Here a key
// tile is one 32-row block of the QKV image -- for one head its K (and V) pieces are 12 KiB CONTIGUOUS in HBM -- so the fill is 24
// LDS-DMA pieces per tile (global_load_lds_dwordx4, double-buffered, counted vmcnt, one barrier per tile), and Q fragments are
// plain 16-byte loads.  Key tiles follow the image's row blocks (not the sample's own 32-key grid): the first / last tile of a
// sample share their block with the neighbouring sample, whose keys are masked.
//
// LDS reads are inline asm: hipcc would drain the DMA ring (vmcnt(0)) in front of every LDS read it knows about (gemm_tp3_kernel.h).
#include "common.h"
#include "tp3.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

using tp3::BLK;
using tp3::PIECE;

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int DH = 64;
constexpr int VCH = BLK + 128;        // LDS stride of a V d-chunk: odd chunks start half a bank row later (see the transposed reads)

struct FwdP {
    const unsigned char* qkv; int kb;     // tp3 image of [B*T, 3*H*64]; kb = 3*H*64/16
    unsigned char* o_tp3; int o_kb;       // tp3 image of O [B*T, H*64]
    float* lse;                           // [B, H, T] natural log-sum-exp of the scaled scores (may be null)
    int B, H, T; float scale;
    long long* stamps;                    // diagnostics only (variant 16): per workgroup {clock start, clock end, wall start, wall end}
};

__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(size_t)lds_byte, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // keep every MFMA that consumes an asm read below the wait (guide §5.4 rule 18)
}
__device__ __forceinline__ bf16x8 frag_of(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d)); }

// x[0..N) -> three planes of N/2 dwords: round-to-nearest pieces of the running residual (tp3.h)
template <int N>
__device__ __forceinline__ void split3(float (&x)[N], unsigned (&p0)[N / 2], unsigned (&p1)[N / 2], unsigned (&p2)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        p0[i] = tp3::pack_rn(x[2 * i], x[2 * i + 1]);
        x[2 * i] -= tp3::bfloat(p0[i] << 16); x[2 * i + 1] -= tp3::bfloat(p0[i] & 0xFFFF0000u);
        p1[i] = tp3::pack_rn(x[2 * i], x[2 * i + 1]);
        x[2 * i] -= tp3::bfloat(p1[i] << 16); x[2 * i + 1] -= tp3::bfloat(p1[i] & 0xFFFF0000u);
        p2[i] = tp3::pack_rn(x[2 * i], x[2 * i + 1]);
    }
}
__device__ __forceinline__ f32x16 mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
#pragma unroll
    for (int order = 2; order >= 0; --order)
#pragma unroll
        for (int i = 0; i <= order; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[order - i], acc, 0, 0, 0);
    return acc;
}

// fragment (rows m0..m0+31 by lane, 16 columns of column block kb) of a tp3 image, straight into registers: 3 x 16-byte loads
__device__ __forceinline__ void row_frags(const unsigned char* __restrict__ img, int kblocks, long m, int kb, int hh, bf16x8 (&out)[3]) {
    const unsigned char* src = img + ((m >> 5) * kblocks + kb) * (long)BLK + (hh * 32 + (int)(m & 31)) * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p) out[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(src + p * PIECE));
}

// Address of the 16-byte unit (row m, k-half hh) of piece 0 of a column block of a tp3 image (colblk = image + column block * BLK,
// blk_stride = bytes between row blocks).  Key / query tiles start at the SAMPLE's first row, not at a 32-row block of the image: a
// tile's 32 rows straddle two row blocks, which costs the LDS-DMA nothing (its global addresses are per lane) and saves the 17th tile
// and the first-tile mask that block-aligned tiles needed at T = 495.
__device__ __forceinline__ const unsigned char* unit_at(const unsigned char* colblk, long blk_stride, long m, int hh) {
    return colblk + (m >> 5) * blk_stride + (hh * 32 + (int)(m & 31)) * 16;
}

// Two workgroups share a CU and their waves a SIMD.  With equal priority the co-resident waves fall into step (both in the MFMA
// stretch, then both in the vector stretch) and the two stretches add.  Every second workgroup the dispatcher places on a CU -- it
// deals one workgroup to each of an XCD's 32 CUs before the second -- therefore runs at a higher static priority: it takes the
// matrix pipe whenever it wants it, its partner fills the gaps (MI355X_MICROARCH.md, 'Two waves per SIMD', items 2 and 4).
__device__ __forceinline__ bool prio_of_workgroup() { return ((blockIdx.x >> 3) >> 5) & 1; }

// counted LDS waits that carry the fragments they guard as "+v" operands: the MFMAs that consume those registers depend on the asm
// and stay below it, while unrelated vector arithmetic remains free to move across (a sched_barrier would pin everything)
template <int N>
__device__ __forceinline__ void wait_k(bf16x8 (&f)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_v(u32x2 (&f)[6]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N));
}
template <int OFF>
__device__ __forceinline__ void read_k(bf16x8 (&f)[3], unsigned a) {
    f[0] = lds_b128<OFF>(a); f[1] = lds_b128<OFF + PIECE>(a); f[2] = lds_b128<OFF + 2 * PIECE>(a);
}
// one 16-key step of a 32-d block of V^T: {piece 0 lo, hi, piece 1 lo, hi, piece 2 lo, hi}; "hi" = keys + 8
template <int OFF>
__device__ __forceinline__ void read_v(u32x2 (&f)[6], unsigned a) {
    f[0] = lds_tr<OFF>(a); f[1] = lds_tr<OFF + 256>(a); f[2] = lds_tr<OFF + PIECE>(a); f[3] = lds_tr<OFF + PIECE + 256>(a);
    f[4] = lds_tr<OFF + 2 * PIECE>(a); f[5] = lds_tr<OFF + 2 * PIECE + 256>(a);
}
__device__ __forceinline__ f32x16 mma6_v(const u32x2 (&v)[6], const bf16x8 (&pf)[3], f32x16 acc) {
    const bf16x8 vf[3] = {frag_of(v[0][0], v[0][1], v[1][0], v[1][1]), frag_of(v[2][0], v[2][1], v[3][0], v[3][1]),
                          frag_of(v[4][0], v[4][1], v[5][0], v[5][1])};
    return mma6(vf, pf, acc);
}

constexpr int K_STAGE = 12 * PIECE, V_STAGE = 4 * VCH;   // LDS: K ring [2][K_STAGE], then V ring [2][V_STAGE]
constexpr int FWD_LDS = 2 * K_STAGE + 2 * V_STAGE;

// Four waves, 32 queries each; key tiles are the image's 32-row blocks, double-buffered in LDS by LDS-DMA, one barrier per tile.
// Two workgroups share a CU (176 VGPRs).  What bounds it (profiles/r2_attention_experiments.md): per key tile a wave has 48 MFMAs (1,536 cycles
// of its SIMD's matrix pipe) and ~170 vector instructions (~770 cycles of the SIMD's vector issue), and co-resident waves fall into
// step -- all in the matrix stretch, then all in the softmax -- so the two ADD: 2,300 cycles per wave-tile measured, however the
// vector work was trimmed (mask only on edge tiles, scale folded into the exp2 argument), with conflict-free transposed reads, with
// the next tile's S^T issued ahead of this tile's softmax, or with two 4-wave halves held in anti-phase by barriers (5,650 cycles
// per tile pair: every phase start exposes an LDS round trip), or with a hand-placed stream (every MFMA followed by the <= 5 vector
// instructions that fit in its shadow, the softmax of tile k under the S MFMAs of tile k+1: same time).  The kernel is POWER-limited:
// the identical instruction stream on all-zero operands runs 18 % faster (backward: 34 %), so what pays is fewer MFMAs and fewer bytes
// moved per result (profiles/r2_attention_experiments.md), not a denser schedule.
// V = 0 is the product; 1 (no DMA after the prologue), 2 (no static priority) and 16 (workgroup clock stamps) are diagnostics for
// tools/bench_attn.py.
constexpr int FWD_THREADS = 256, FWD_Q = 128;
template <int V>
__global__ __launch_bounds__(FWD_THREADS, 2) void attn_fwd_tp3_kernel(FwdP p) {
    TVL_KERNEL_ENTRY();
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: hardware deals workgroup L to XCD L % 8.  The query blocks of one (sample, head) read the same K / V, so
    // they are given consecutive VIRTUAL ids inside one XCD's share of the grid and meet in that XCD's L2.
    const int T = p.T, D = p.H * DH;
    const int nqb = (T + FWD_Q - 1) / FWD_Q, total = (int)gridDim.x;
    const int L = (int)blockIdx.x, per = total / 8;
    const int vid = L < per * 8 ? (L % 8) * per + L / 8 : L;   // the last partial round keeps its ids
    const int qb = vid % nqb, head = (vid / nqb) % p.H, b = vid / (nqb * p.H);
    const int qi = qb * FWD_Q + wave * 32 + l31;
    const long m_q = (long)b * T + (qi < T ? qi : T - 1);

    long long t0 = 0, w0 = 0;
    if constexpr (V & 16) { t0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    if constexpr (!(V & 2)) {
        if (prio_of_workgroup()) __builtin_amdgcn_s_setprio(2);
    }
    bf16x8 qf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) row_frags(p.qkv, p.kb, m_q, head * 4 + s, h, qf[s]);

    f32x16 acc_o[2], zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_o[0][r] = 0.f; acc_o[1][r] = 0.f; zero[r] = 0.f; }
    float m_run = NEG_BIG, l_run = 0.f;
    const float sc2 = p.scale * LOG2E;

    // key tile kt = keys [32 kt, 32 kt + 32) of this sample (image rows b*T + ...; rows past the image's last row are clamped, their
    // keys masked).  DMA: a tile is 12 K pieces + 12 V pieces; wave w moves pieces w, w + 4, w + 8 of both.
    const long row_lo = (long)b * T;
    const int nkt = (T + 31) / 32;
    const long tile_stride = (long)p.kb * BLK;
    const unsigned char* k_col = p.qkv + (long)((D + head * DH) / 16) * BLK;
    const unsigned char* v_col = p.qkv + (long)((2 * D + head * DH) / 16) * BLK;
    const unsigned k_lane = (unsigned)(unit_at(k_col, tile_stride, row_lo + (l31 < T ? l31 : T - 1), h) - k_col);   // tile 0 (a row of the sample); tile kt is kt row blocks further
    // V pieces land key-interleaved: LDS unit L (16 B) of a piece = (key L >> 1, d-half L & 1), i.e. a key's 16 d are 32 contiguous bytes
    const unsigned v_lane = (unsigned)(unit_at(v_col, tile_stride, row_lo + ((lane >> 1) < T ? (lane >> 1) : T - 1), lane & 1) - v_col);
    auto issue = [&](int kt) {
        const unsigned kd = lds0 + (kt & 1) * K_STAGE, vd = lds0 + 2 * K_STAGE + (kt & 1) * V_STAGE;
        // rows past the sample's end (last tile only) are masked anyway: those lanes re-read their tile-0 row, so no address leaves the image
        const unsigned char* ks = k_col + (k_lane + (32 * kt + l31 < T ? (unsigned)kt * (unsigned)tile_stride : 0u));
        const unsigned char* vs = v_col + (v_lane + (32 * kt + (lane >> 1) < T ? (unsigned)kt * (unsigned)tile_stride : 0u));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wave + 4 * i;   // 0..11: (d-chunk pc / 3, piece pc % 3), consecutive in the image
            glds16(ks + pc * PIECE, kd + pc * PIECE);
            glds16(vs + pc * PIECE, vd + (pc / 3) * VCH + (pc % 3) * PIECE);
        }
    };
    // transposed-read addressing of a V piece (32 keys x 16 d, element (key, d) at key * 32 + d * 2): a 16-lane group reads
    // 4 keys x 16 d = 128 contiguous bytes; lane 4q + pp supplies key row q, columns 4 pp .. 4 pp + 3.  Lanes 16-31 read the next
    // d-chunk, VCH = 3200 bytes on = the other half of the 256-byte bank row: each 32-lane group covers all 64 banks once.
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = (4 * h + (li >> 2)) * 32 + (li & 3) * 8 + g1 * VCH;
    const unsigned k_rd = lds0 + lane * 16, v_rd = lds0 + 2 * K_STAGE + tr_off;

    // variant 32: s_memtime at the phase boundaries of every tile, summed per wave (the value is consumed at once, so no scalar
    // load is outstanding when the counted LDS waits run)
    long long ph[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int i) {
        if constexpr (V & 32) {
            __builtin_amdgcn_sched_barrier(0);
            const long long now = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ph[i] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue(0);
    if constexpr (V & 32) tprev = __builtin_readcyclecounter();
    for (int kt = 0; kt < nkt; ++kt) {
        wait_vm<0>();                       // this wave's pieces of tile kt have landed ...
        mark(4);
        __builtin_amdgcn_s_barrier();       // ... and everybody's; everybody is also done reading the other stage (tile kt-1)
        mark(0);
        if (kt + 1 < nkt && (!(V & 1) || kt < 1)) issue(kt + 1);
        const unsigned kb_ = k_rd + (kt & 1) * K_STAGE, vb_ = v_rd + (kt & 1) * V_STAGE;
        // ---- S^T = K . Q^T: fragments double-buffered in registers, counted LDS waits ----
        f32x16 sc;
        bf16x8 ka[3], kb[3];
        read_k<0>(ka, kb_); read_k<BLK>(kb, kb_);
        wait_k<3>(ka); sc = mma6(ka, qf[0], zero);
        read_k<2 * BLK>(ka, kb_); wait_k<3>(kb); sc = mma6(kb, qf[1], sc);
        read_k<3 * BLK>(kb, kb_); wait_k<3>(ka); sc = mma6(ka, qf[2], sc);
        wait_k<0>(kb); sc = mma6(kb, qf[3], sc);
        mark(1);
        // the first V^T fragments are requested now: they land underneath the softmax arithmetic
        u32x2 va[6], vb[6];
        read_v<0>(va, vb_); read_v<0>(vb, vb_ + 512);
        // ---- online softmax (query on the lane; register r <-> key (r & 3) + 8 (r >> 2) + 4 h of the tile) ----
        // The running maximum is kept in RAW score units (scale > 0 commutes with max); the scale rides on the exp2 argument's fma.
        if (kt == nkt - 1) {   // only the last tile reaches past the sample
            const int key0 = 32 * kt + 4 * h;   // key index of register 0
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = (key0 + (r & 3) + 8 * (r >> 2) < T) ? sc[r] : NEG_BIG;
        }
        float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) mx = fmaxf(mx, sc[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc2);
        const float nm = -m_new * sc2;
        float pv[16];
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, nm));
            rs += pv[r];
        }
        l_run = l_run * alpha + rs;
        m_run = m_new;
        unsigned pp0[8], pp1[8], pp2[8];
        split3<16>(pv, pp0, pp1, pp2);
        const bf16x8 pf0[3] = {frag_of(pp0[0], pp0[1], pp0[2], pp0[3]), frag_of(pp1[0], pp1[1], pp1[2], pp1[3]), frag_of(pp2[0], pp2[1], pp2[2], pp2[3])};
        const bf16x8 pf1[3] = {frag_of(pp0[4], pp0[5], pp0[6], pp0[7]), frag_of(pp1[4], pp1[5], pp1[6], pp1[7]), frag_of(pp2[4], pp2[5], pp2[6], pp2[7])};
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_o[0][r] *= alpha; acc_o[1][r] *= alpha; }
        mark(2);
        wait_v<6>(va); acc_o[0] = mma6_v(va, pf0, acc_o[0]);
        read_v<2 * VCH>(va, vb_); wait_v<6>(vb); acc_o[0] = mma6_v(vb, pf1, acc_o[0]);
        read_v<2 * VCH>(vb, vb_ + 512); wait_v<6>(va); acc_o[1] = mma6_v(va, pf0, acc_o[1]);
        wait_v<0>(vb); acc_o[1] = mma6_v(vb, pf1, acc_o[1]);
        mark(3);
    }
    if constexpr (V & 32) {
        if (threadIdx.x == 0 && p.stamps) {
            long long* o = p.stamps + 5L * blockIdx.x;
            for (int i = 0; i < 5; ++i) o[i] = ph[i];
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < T) {
        const long m = (long)b * T + qi;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {acc_o[d][4 * g] * inv, acc_o[d][4 * g + 1] * inv, acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv};
                tp3::store4(p.o_tp3, p.o_kb, m, head * DH + d * 32 + 8 * g + 4 * h, v);
            }
        if (h == 0 && p.lse) p.lse[((long)b * p.H + head) * T + qi] = (m_run * sc2 + log2f(l_tot)) * LN2;
    }
    if constexpr (V & 16) {
        if (threadIdx.x == 0 && p.stamps) {
            long long* o = p.stamps + 4L * blockIdx.x;
            o[0] = t0; o[1] = __builtin_readcyclecounter();
            // wall clock (100 MHz) in the low 40 bits, XCC id and the HW_ID's cu / sh / se bits above them
            const long long where = ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 255);
            o[2] = w0; o[3] = (wall_clock64() & ((1LL << 40) - 1)) | (where << 40);
        }
    }
}

// virtual workgroup id: blocks of one (sample, head) consecutive inside one XCD's share of the grid (as in the forward)
__device__ __forceinline__ int xcd_vid() {
    const int total = (int)gridDim.x, L = (int)blockIdx.x, per = total / 8;
    return L < per * 8 ? (L % 8) * per + L / 8 : L;
}

// ---- backward -------------------------------------------------------------------------------------------------------------
// Same arithmetic as attn_bwd_dq / dkdv_bf16s_kernel (attention_bf16s.hip): P is recomputed from Q, K and the forward's log-sum-exp,
// dS = P o (dP - delta), every product is the 6-MFMA piece sum.  Operands arrive as tp3 images: packed QKV [B*T, 3*H*64] (the QKV
// GEMM's epilogue), O (the forward) and dO (the out-projection data gradient's epilogue); dQ | dK | dV leave as the tp3 image of the
// packed gradient, the A operand of the QKV data-gradient GEMM.  Nothing is re-split per workgroup and tiles are filled by LDS-DMA.
struct BwdP {
    const unsigned char* qkv; int kb;     // tp3 image of packed QKV; kb = 3*H*64/16
    const unsigned char* o_img;           // tp3 image of O  [B*T, H*64]
    const unsigned char* do_img; int o_kb;   // tp3 image of dO [B*T, H*64]; o_kb = H*64/16
    const float* lse;                     // [B, H, T] from the forward
    float* delta;                         // [B, H, T]: written by the dQ kernel (delta = sum_d dO * O), read by the dK/dV kernel
    unsigned char* g_img;                 // tp3 image of dQ | dK | dV [B*T, 3*H*64]
    int B, H, T; float scale;
};

// transposed fragments out of a piece in the image's own layout (32 rows x 16 k, element (row, k) at (k / 8) * 512 + row * 16 + (k % 8) * 2):
// {piece 0 lo, hi, piece 1 lo, hi, piece 2 lo, hi} of one 16-row step of a 32-k block; "hi" = rows + 8
template <int OFF>
__device__ __forceinline__ void read_t(u32x2 (&f)[6], unsigned a) {
    f[0] = lds_tr<OFF>(a); f[1] = lds_tr<OFF + 128>(a); f[2] = lds_tr<OFF + PIECE>(a); f[3] = lds_tr<OFF + PIECE + 128>(a);
    f[4] = lds_tr<OFF + 2 * PIECE>(a); f[5] = lds_tr<OFF + 2 * PIECE + 128>(a);
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_f4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void glds4(const void* g, unsigned lds_byte) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(size_t)lds_byte, 4, 0, 0);
}
__device__ __forceinline__ void pieces_of(float (&x)[16], bf16x8 (&f0)[3], bf16x8 (&f1)[3]) {
    unsigned p0[8], p1[8], p2[8];
    split3<16>(x, p0, p1, p2);
    f0[0] = frag_of(p0[0], p0[1], p0[2], p0[3]); f0[1] = frag_of(p1[0], p1[1], p1[2], p1[3]); f0[2] = frag_of(p2[0], p2[1], p2[2], p2[3]);
    f1[0] = frag_of(p0[4], p0[5], p0[6], p0[7]); f1[1] = frag_of(p1[4], p1[5], p1[6], p1[7]); f1[2] = frag_of(p2[4], p2[5], p2[6], p2[7]);
}
// A . B accumulated over the four 16-k steps of d_h = 64: A fragments double-buffered out of LDS (piece blocks BLK apart), B in registers
__device__ __forceinline__ f32x16 mma_rows(unsigned a_rd, const bf16x8 (&bq)[4][3]) {
    const f32x16 zero = {};   // an inline constant as the first MFMA's accumulator input, not 16 registers
    bf16x8 ka[3], kb[3];
    read_k<0>(ka, a_rd); read_k<BLK>(kb, a_rd);
    wait_k<3>(ka); f32x16 acc = mma6(ka, bq[0], zero);
    read_k<2 * BLK>(ka, a_rd); wait_k<3>(kb); acc = mma6(kb, bq[1], acc);
    read_k<3 * BLK>(kb, a_rd); wait_k<3>(ka); acc = mma6(ka, bq[2], acc);
    wait_k<0>(kb); acc = mma6(kb, bq[3], acc);
    return acc;
}
// acc[d] += T^T . X for both 32-column blocks d of a 32-row tile T (transposed reads at t_rd), X = the piece fragments of a 32 x 32
// accumulator-layout matrix (rows 0-15 / 16-31)
__device__ __forceinline__ void mma_cols(unsigned t_rd, const bf16x8 (&x0)[3], const bf16x8 (&x1)[3], f32x16 (&acc)[2]) {
    u32x2 va[6], vb[6];
    read_t<0>(va, t_rd); read_t<0>(vb, t_rd + 256);
    wait_v<6>(va); acc[0] = mma6_v(va, x0, acc[0]);
    read_t<2 * BLK>(va, t_rd); wait_v<6>(vb); acc[0] = mma6_v(vb, x1, acc[0]);
    read_t<2 * BLK>(vb, t_rd + 256); wait_v<6>(va); acc[1] = mma6_v(va, x0, acc[1]);
    wait_v<0>(vb); acc[1] = mma6_v(vb, x1, acc[1]);
}

constexpr int BWD_STAGE = 24 * PIECE + 256;   // two 12-piece tiles + 64 floats (log-sum-exp | delta of a query tile; dK/dV kernel only)
constexpr int BWD_LDS = 2 * BWD_STAGE;

// dQ: query-stationary.  Per key tile: S^T = K.Q^T, dP^T = V.dO^T (K, V fragments straight out of the DMA image), dS^T on the
// accumulators, dQ^T += K^T . dS^T (K^T by transposed reads of the same K pieces).  Also writes delta for the dK/dV kernel.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_tp3_kernel(BwdP p) {
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

    bf16x8 qf[4][3], dof[4][3];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        row_frags(p.qkv, p.kb, m_q, head * 4 + s, h, qf[s]);
        row_frags(p.do_img, p.o_kb, m_q, head * 4 + s, h, dof[s]);
        bf16x8 of[3];
        row_frags(p.o_img, p.o_kb, m_q, head * 4 + s, h, of);
        const uint4 a[3] = {__builtin_bit_cast(uint4, dof[s][0]), __builtin_bit_cast(uint4, dof[s][1]), __builtin_bit_cast(uint4, dof[s][2])};
        const uint4 c[3] = {__builtin_bit_cast(uint4, of[0]), __builtin_bit_cast(uint4, of[1]), __builtin_bit_cast(uint4, of[2])};
        float x[8], y[8];
        tp3::join8(a, x);
        tp3::join8(c, y);
        dl += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]) + (x[4] * y[4] + x[5] * y[5]) + (x[6] * y[6] + x[7] * y[7]);
    }
    dl += __shfl_xor(dl, 32, 64);   // the two lane halves hold the two 8-column halves of every 16-column step
    const long stat = ((long)b * p.H + head) * T + qrow;
    if (h == 0 && qi < T) p.delta[stat] = dl;
    const float sc2 = p.scale * LOG2E;
    const float nlse2 = -p.lse[stat] * LOG2E;

    f32x16 acc_dq[2] = {};

    const long row_lo = (long)b * T;
    const int nkt = (T + 31) / 32;
    const long tile_stride = (long)p.kb * BLK;
    const unsigned char* k_col = p.qkv + (long)((D + head * DH) / 16) * BLK;
    const unsigned char* v_col = p.qkv + (long)((2 * D + head * DH) / 16) * BLK;
    const unsigned k_lane = (unsigned)(unit_at(k_col, tile_stride, row_lo + (l31 < T ? l31 : T - 1), h) - k_col);   // tile 0 (a row of the sample); tile kt is kt row blocks further
    const long v_minus_k = v_col - k_col;
    auto issue = [&](int kt) {   // key tile kt = keys [32 kt, 32 kt + 32) of the sample (unit_at)
        const unsigned dst = lds0 + (kt & 1) * BWD_STAGE;
        // rows past the sample's end (last tile only) are masked anyway: those lanes re-read their tile-0 row, so no address leaves the image
        const unsigned char* ks = k_col + (k_lane + (32 * kt + l31 < T ? (unsigned)kt * (unsigned)tile_stride : 0u));
        const unsigned char* vs = ks + v_minus_k;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wave + 4 * i;
            glds16(ks + pc * PIECE, dst + pc * PIECE);
            glds16(vs + pc * PIECE, dst + (12 + pc) * PIECE);
        }
    };
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = ((li & 3) >> 1) * 512 + (4 * h + (li >> 2)) * 16 + (li & 1) * 8 + g1 * BLK;

    issue(0);
    for (int kt = 0; kt < nkt; ++kt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) issue(kt + 1);
        const unsigned st = lds0 + (kt & 1) * BWD_STAGE;
        f32x16 sc = mma_rows(st + lane * 16, qf);                  // S^T  [key (register), query (lane)]
        f32x16 dp = mma_rows(st + 12 * PIECE + lane * 16, dof);    // dP^T
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, nlse2)) * (dp[r] - dl);
        if (kt == nkt - 1) {   // only the last tile reaches past the sample
            const int key0 = 32 * kt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = (key0 + (r & 3) + 8 * (r >> 2) < T) ? ds[r] : 0.f;
        }
        bf16x8 x0[3], x1[3];
        pieces_of(ds, x0, x1);
        mma_cols(st + tr_off, x0, x1, acc_dq);                           // dQ^T += K^T . dS^T
    }
    if (qi < T) {
        const long m = (long)b * T + qi;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {acc_dq[d][4 * g] * p.scale, acc_dq[d][4 * g + 1] * p.scale, acc_dq[d][4 * g + 2] * p.scale,
                                    acc_dq[d][4 * g + 3] * p.scale};
                tp3::store4(p.g_img, p.kb, m, head * DH + d * 32 + 8 * g + 4 * h, v);
            }
    }
}

// dK, dV: key-stationary (K, V fragments of the wave's 32 keys in registers).  Per query tile (a 32-row block of the images):
// S = Q.K^T, dP = dO.V^T (Q, dO fragments straight out of the DMA image; query on the register, key on the lane),
// dV^T += dO^T . P, dK^T += Q^T . dS (transposed reads of the same pieces).
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_tp3_kernel(BwdP p) {
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
    const int ki = kblk * 128 + wave * 32 + l31;
    const bool key_ok = ki < T;
    const long m_k = (long)b * T + (key_ok ? ki : T - 1);

    bf16x8 kf[4][3], vf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        row_frags(p.qkv, p.kb, m_k, (D + head * DH) / 16 + s, h, kf[s]);
        row_frags(p.qkv, p.kb, m_k, (2 * D + head * DH) / 16 + s, h, vf[s]);
    }
    f32x16 acc_dk[2] = {}, acc_dv[2] = {};
    const float sc2 = p.scale * LOG2E;

    const long row_lo = (long)b * T;
    const int nqt = (T + 31) / 32;
    const long q_stride = (long)p.kb * BLK, d_stride = (long)p.o_kb * BLK;
    const unsigned char* q_col = p.qkv + (long)((head * DH) / 16) * BLK;
    const unsigned char* d_col = p.do_img + (long)((head * DH) / 16) * BLK;
    const float* stat_base = (lane < 32 ? p.lse : p.delta) + ((long)b * p.H + head) * T;
    const unsigned q_lane = (unsigned)(unit_at(q_col, q_stride, row_lo + (l31 < T ? l31 : T - 1), h) - q_col);   // tile 0 (a row of the sample); tile qt is qt row blocks further
    const unsigned d_lane = (unsigned)(unit_at(d_col, d_stride, row_lo + (l31 < T ? l31 : T - 1), h) - d_col);
    auto issue = [&](int qt) {   // query tile qt = queries [32 qt, 32 qt + 32) of the sample (unit_at)
        const unsigned dst = lds0 + (qt & 1) * BWD_STAGE;
        // uniform base + 32-bit lane offset (images are < 4 GB).  Rows past the sample's end (last tile only) are masked anyway: those
        // lanes re-read their tile-0 row instead, so that no address leaves the image
        const bool in_sample = 32 * qt + l31 < T;
        const unsigned char* qs = q_col + (q_lane + (in_sample ? (unsigned)qt * (unsigned)q_stride : 0u));
        const unsigned char* ds_ = d_col + (d_lane + (in_sample ? (unsigned)qt * (unsigned)d_stride : 0u));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wave + 4 * i;
            glds16(qs + pc * PIECE, dst + pc * PIECE);
            glds16(ds_ + pc * PIECE, dst + (12 + pc) * PIECE);
        }
        if (wave == 0) {   // lanes 0-31: log-sum-exp of the tile's 32 queries, lanes 32-63: their delta (clamped inside the sample)
            const int q = 32 * qt + l31;
            glds4(stat_base + (q < T ? q : T - 1), dst + 24 * PIECE);
        }
    };
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = ((li & 3) >> 1) * 512 + (4 * h + (li >> 2)) * 16 + (li & 1) * 8 + g1 * BLK;

    issue(0);
    for (int qt = 0; qt < nqt; ++qt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (qt + 1 < nqt) issue(qt + 1);
        const unsigned st = lds0 + (qt & 1) * BWD_STAGE;
        f32x16 sc = mma_rows(st + lane * 16, kf);                  // S   [query (register), key (lane)]
        f32x16 dp = mma_rows(st + 12 * PIECE + lane * 16, vf);     // dP
        // rows of S / dP are the tile's queries (r & 3) + 8 (r >> 2) + 4 h: their statistics come as four float4 each
        f32x4 l4[4], d4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) l4[g] = lds_f4<0>(st + 24 * PIECE + (8 * g + 4 * h) * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) d4[g] = lds_f4<128>(st + 24 * PIECE + (8 * g + 4 * h) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l4[0]), "+v"(l4[1]), "+v"(l4[2]), "+v"(l4[3]), "+v"(d4[0]), "+v"(d4[1]), "+v"(d4[2]), "+v"(d4[3]));
        float pv[16], dsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, -LOG2E * l4[r >> 2][r & 3]));
            dsv[r] = pv[r] * (dp[r] - d4[r >> 2][r & 3]);
        }
        if (qt == nqt - 1 || !key_ok) {   // only the last tile reaches past the sample
            const int q0 = 32 * qt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = key_ok && q0 + (r & 3) + 8 * (r >> 2) < T;
                pv[r] = ok ? pv[r] : 0.f;
                dsv[r] = ok ? dsv[r] : 0.f;
            }
        }
        bf16x8 x0[3], x1[3];
        pieces_of(pv, x0, x1);
        mma_cols(st + 12 * PIECE + tr_off, x0, x1, acc_dv);              // dV^T += dO^T . P
        pieces_of(dsv, x0, x1);
        mma_cols(st + tr_off, x0, x1, acc_dk);                           // dK^T += Q^T . dS
    }
    if (key_ok) {
        const long m = (long)b * T + ki;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = head * DH + d * 32 + 8 * g + 4 * h;
                const float vk[4] = {acc_dk[d][4 * g] * p.scale, acc_dk[d][4 * g + 1] * p.scale, acc_dk[d][4 * g + 2] * p.scale,
                                     acc_dk[d][4 * g + 3] * p.scale};
                const float vv[4] = {acc_dv[d][4 * g], acc_dv[d][4 * g + 1], acc_dv[d][4 * g + 2], acc_dv[d][4 * g + 3]};
                tp3::store4(p.g_img, p.kb, m, D + col, vk);
                tp3::store4(p.g_img, p.kb, m, 2 * D + col, vv);
            }
    }
}

}  // namespace

static int attn_tp3_fwd_launch(const void* qkv_tp3, void* o_tp3, float* lse, int32_t B, int32_t H, int32_t T, float scale, int variant,
                               long long* stamps, tvlStream_t stream) {
    TVL_REQUIRE(qkv_tp3 && o_tp3, "tvl_attn_tp3_fwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && T > 0, "tvl_attn_tp3_fwd: bad shape");
    TVL_REQUIRE(tvl_aligned16(qkv_tp3) && tvl_aligned16(o_tp3), "tvl_attn_tp3_fwd: tp3 images must be 16-byte aligned");
    TVL_REQUIRE((long)((T + FWD_Q - 1) / FWD_Q) * H * B < (1L << 31), "tvl_attn_tp3_fwd: grid too large");
    FwdP p;
    p.qkv = reinterpret_cast<const unsigned char*>(qkv_tp3); p.kb = 3 * H * DH / 16;
    p.o_tp3 = reinterpret_cast<unsigned char*>(o_tp3); p.o_kb = H * DH / 16; p.lse = lse;
    p.B = B; p.H = H; p.T = T; p.scale = scale; p.stamps = stamps;
    dim3 grid((unsigned)((T + FWD_Q - 1) / FWD_Q * H * B));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (variant) {
        case 0: hipLaunchKernelGGL(attn_fwd_tp3_kernel<0>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
        case 1: hipLaunchKernelGGL(attn_fwd_tp3_kernel<1>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
        case 2: hipLaunchKernelGGL(attn_fwd_tp3_kernel<2>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
        case 16: hipLaunchKernelGGL(attn_fwd_tp3_kernel<16>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
        case 32: hipLaunchKernelGGL(attn_fwd_tp3_kernel<32>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
        case 34: hipLaunchKernelGGL(attn_fwd_tp3_kernel<34>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
        default: TVL_REQUIRE(false, "tvl_attn_tp3_fwd: unknown variant %d", variant);
    }
    TVL_LAUNCH_CHECK("tvl_attn_tp3_fwd");
    return 0;
}

extern "C" int tvl_attn_tp3_fwd(const void* qkv_tp3, void* o_tp3, float* lse, int32_t B, int32_t H, int32_t T, float scale, tvlStream_t stream) {
    return attn_tp3_fwd_launch(qkv_tp3, o_tp3, lse, B, H, T, scale, 0, nullptr, stream);
}

// Diagnostics for tools/bench_attn.py: ablation variants (outputs wrong by construction) and per-workgroup clock stamps
// (stamps: int64 [workgroups][4], variant 16).  Not part of the drop-in boundary.
extern "C" int tvl_attn_tp3_fwd_diag(const void* qkv_tp3, void* o_tp3, float* lse, int32_t B, int32_t H, int32_t T, float scale, int32_t variant,
                                     int64_t* stamps, tvlStream_t stream) {
    return attn_tp3_fwd_launch(qkv_tp3, o_tp3, lse, B, H, T, scale, variant, reinterpret_cast<long long*>(stamps), stream);
}

// Backward of tvl_attn_tp3_fwd: dQ | dK | dV as the tp3 image of the packed gradient.  delta is a [B, H, T] fp32 workspace.
extern "C" int tvl_attn_tp3_bwd(const void* qkv_tp3, const void* o_tp3, const void* do_tp3, const float* lse, float* delta, void* dqkv_tp3,
                                int32_t B, int32_t H, int32_t T, float scale, tvlStream_t stream) {
    TVL_REQUIRE(qkv_tp3 && o_tp3 && do_tp3 && lse && delta && dqkv_tp3, "tvl_attn_tp3_bwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && T > 0, "tvl_attn_tp3_bwd: bad shape");
    TVL_REQUIRE(tvl_aligned16(qkv_tp3) && tvl_aligned16(o_tp3) && tvl_aligned16(do_tp3) && tvl_aligned16(dqkv_tp3),
                "tvl_attn_tp3_bwd: tp3 images must be 16-byte aligned");
    TVL_REQUIRE((long)((T + 127) / 128) * H * B < (1L << 31), "tvl_attn_tp3_bwd: grid too large");
    BwdP p;
    p.qkv = reinterpret_cast<const unsigned char*>(qkv_tp3); p.kb = 3 * H * DH / 16;
    p.o_img = reinterpret_cast<const unsigned char*>(o_tp3); p.do_img = reinterpret_cast<const unsigned char*>(do_tp3); p.o_kb = H * DH / 16;
    p.lse = lse; p.delta = delta; p.g_img = reinterpret_cast<unsigned char*>(dqkv_tp3);
    p.B = B; p.H = H; p.T = T; p.scale = scale;
    dim3 grid((unsigned)((T + 127) / 128 * H * B));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(attn_bwd_dq_tp3_kernel, grid, dim3(256), BWD_LDS, s, p);      // also writes delta ...
    hipLaunchKernelGGL(attn_bwd_dkdv_tp3_kernel, grid, dim3(256), BWD_LDS, s, p);    // ... which this one reads (same stream)
    TVL_LAUNCH_CHECK("tvl_attn_tp3_bwd");
    return 0;
}
