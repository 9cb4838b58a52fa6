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
// per tile pair: every phase start exposes an LDS round trip).  The way on is a hand-placed stream with <= 5 vector instructions in
// every MFMA gap (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'); hipcc's scheduler does not produce one around inline asm.
// V = 0 is the product; 1 (no DMA after the prologue) and 16 (workgroup clock stamps) are diagnostics for tools/bench_attn.py.
constexpr int FWD_THREADS = 256, FWD_Q = 128;
template <int V>
__global__ __launch_bounds__(FWD_THREADS, 2) void attn_fwd_tp3_kernel(FwdP p) {
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
    bf16x8 qf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) row_frags(p.qkv, p.kb, m_q, head * 4 + s, h, qf[s]);

    f32x16 acc_o[2], zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_o[0][r] = 0.f; acc_o[1][r] = 0.f; zero[r] = 0.f; }
    float m_run = NEG_BIG, l_run = 0.f;
    const float sc2 = p.scale * LOG2E;

    // key tiles = the image's row blocks that overlap this sample's rows [b*T, (b+1)*T)
    const long row_lo = (long)b * T, row_hi = row_lo + T;
    const int rb_lo = (int)(row_lo >> 5);
    const int nkt = (int)((row_hi - 1) >> 5) - rb_lo + 1;
    const int lo_in_blk = (int)(row_lo & 31);
    // DMA: a tile is 12 K pieces + 12 V pieces (each group contiguous in HBM); wave w moves pieces w, w + 4, w + 8 of both
    const unsigned char* k_src = p.qkv + ((long)rb_lo * p.kb + (D + head * DH) / 16) * BLK + lane * 16;
    // V pieces land key-interleaved: LDS unit L (16 B) of a piece = (key L >> 1, d-half L & 1), i.e. a key's 16 d are 32 contiguous bytes
    const unsigned char* v_src = p.qkv + ((long)rb_lo * p.kb + (2 * D + head * DH) / 16) * BLK + ((lane & 1) * 32 + (lane >> 1)) * 16;
    const long tile_stride = (long)p.kb * BLK;
    auto issue = [&](int kt) {
        const unsigned kd = lds0 + (kt & 1) * K_STAGE, vd = lds0 + 2 * K_STAGE + (kt & 1) * V_STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wave + 4 * i;   // 0..11
            glds16(k_src + kt * tile_stride + pc * PIECE, kd + pc * PIECE);
            glds16(v_src + kt * tile_stride + pc * PIECE, vd + (pc / 3) * VCH + (pc % 3) * PIECE);
        }
    };
    // transposed-read addressing of a V piece (32 keys x 16 d, element (key, d) at key * 32 + d * 2): a 16-lane group reads
    // 4 keys x 16 d = 128 contiguous bytes; lane 4q + pp supplies key row q, columns 4 pp .. 4 pp + 3.  Lanes 16-31 read the next
    // d-chunk, VCH = 3200 bytes on = the other half of the 256-byte bank row: each 32-lane group covers all 64 banks once.
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = (4 * h + (li >> 2)) * 32 + (li & 3) * 8 + g1 * VCH;
    const unsigned k_rd = lds0 + lane * 16, v_rd = lds0 + 2 * K_STAGE + tr_off;

    issue(0);
    for (int kt = 0; kt < nkt; ++kt) {
        wait_vm<0>();                       // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();       // ... and everybody's; everybody is also done reading the other stage (tile kt-1)
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
        // the first V^T fragments are requested now: they land underneath the softmax arithmetic
        u32x2 va[6], vb[6];
        read_v<0>(va, vb_); read_v<0>(vb, vb_ + 512);
        // ---- online softmax (query on the lane; register r <-> key (r & 3) + 8 (r >> 2) + 4 h of the tile) ----
        // The running maximum is kept in RAW score units (scale > 0 commutes with max); the scale rides on the exp2 argument's fma.
        if (kt == 0 || kt == nkt - 1) {   // only the first / last tile of a sample can hold a neighbour's keys
            const int key0 = 32 * kt + 4 * h - lo_in_blk;   // key index, relative to the sample, of register 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + (r & 3) + 8 * (r >> 2);
                sc[r] = (key >= 0 && key < T) ? sc[r] : NEG_BIG;
            }
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
        wait_v<6>(va); acc_o[0] = mma6_v(va, pf0, acc_o[0]);
        read_v<2 * VCH>(va, vb_); wait_v<6>(vb); acc_o[0] = mma6_v(vb, pf1, acc_o[0]);
        read_v<2 * VCH>(vb, vb_ + 512); wait_v<6>(va); acc_o[1] = mma6_v(va, pf0, acc_o[1]);
        wait_v<0>(vb); acc_o[1] = mma6_v(vb, pf1, acc_o[1]);
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
        case 16: hipLaunchKernelGGL(attn_fwd_tp3_kernel<16>, grid, dim3(FWD_THREADS), FWD_LDS, s, p); break;
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
