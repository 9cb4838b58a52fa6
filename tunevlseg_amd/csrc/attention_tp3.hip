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
constexpr int KV_TILE = 24 * PIECE;   // one key tile in LDS: K pieces [4 d-chunks][3], then V pieces [4][3]

struct FwdP {
    const unsigned char* qkv; int kb;     // tp3 image of [B*T, 3*H*64]; kb = 3*H*64/16
    unsigned char* o_tp3; int o_kb;       // tp3 image of O [B*T, H*64]
    float* lse;                           // [B, H, T] natural log-sum-exp of the scaled scores (may be null)
    int B, H, T; float scale;
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

__global__ __launch_bounds__(256, 3) void attn_fwd_tp3_kernel(FwdP p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 stages x KV_TILE
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T, D = p.H * DH;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const long m_q = (long)b * T + (qi < T ? qi : T - 1);

    bf16x8 qf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) row_frags(p.qkv, p.kb, m_q, head * 4 + s, h, qf[s]);

    f32x16 acc_o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;
    const float sc2 = p.scale * LOG2E;

    // key tiles = the image's row blocks that overlap this sample's rows [b*T, (b+1)*T)
    const long row_lo = (long)b * T, row_hi = row_lo + T;
    const int rb_lo = (int)(row_lo >> 5);
    const int nkt = (int)((row_hi - 1) >> 5) - rb_lo + 1;
    const int lo_in_blk = (int)(row_lo & 31);
    // DMA: per tile 12 K pieces + 12 V pieces (each group contiguous in HBM); wave w takes pieces w, w+4, ... (6 per wave)
    const unsigned char* k_src = p.qkv + ((long)rb_lo * p.kb + (D + head * DH) / 16) * BLK + lane * 16;
    const unsigned char* v_src = p.qkv + ((long)rb_lo * p.kb + (2 * D + head * DH) / 16) * BLK + lane * 16;
    const long tile_stride = (long)p.kb * BLK;
    auto issue = [&](int kt, int stage) {
        const unsigned dst = lds0 + stage * KV_TILE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wave + 4 * i;   // 0..11
            glds16(k_src + kt * tile_stride + pc * PIECE, dst + pc * PIECE);
            glds16(v_src + kt * tile_stride + pc * PIECE, dst + (12 + pc) * PIECE);
        }
    };
    // transposed-read addressing of a V piece (32 keys x 16 d, element (key, d) at (d / 8) * 512 + key * 16 + (d % 8) * 2): a 16-lane
    // group reads 4 keys x 16 d; lane 4q + pp supplies key row q, columns 4 pp .. 4 pp + 3
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const unsigned tr_off = ((li & 3) >> 1) * 512 + (4 * h + (li >> 2)) * 16 + (li & 1) * 8 + g1 * BLK;
    const unsigned k_rd = lds0 + lane * 16, v_rd = lds0 + 12 * PIECE + tr_off;

    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        wait_vm<0>();                       // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();       // ... and everybody's; everybody is also done reading the other stage (tile kt-1)
        if (kt + 1 < nkt) issue(kt + 1, st ^ 1);
        const unsigned kb_ = k_rd + st * KV_TILE, vb_ = v_rd + st * KV_TILE;
        // ---- S^T = K . Q^T ----
        bf16x8 kf[4][3];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s][0] = lds_b128<0>(kb_ + s * BLK); kf[s][1] = lds_b128<PIECE>(kb_ + s * BLK); kf[s][2] = lds_b128<2 * PIECE>(kb_ + s * BLK);
        }
        wait_lds();
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) sc = mma6(kf[s], qf[s], sc);
        // V^T fragments of d-block 0 are requested now: they land underneath the softmax arithmetic
        u32x2 vt[2][3][2];   // [s2][piece][lo / hi]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            vt[s2][0][0] = lds_tr<0>(vb_ + s2 * 256); vt[s2][0][1] = lds_tr<128>(vb_ + s2 * 256);
            vt[s2][1][0] = lds_tr<PIECE>(vb_ + s2 * 256); vt[s2][1][1] = lds_tr<PIECE + 128>(vb_ + s2 * 256);
            vt[s2][2][0] = lds_tr<2 * PIECE>(vb_ + s2 * 256); vt[s2][2][1] = lds_tr<2 * PIECE + 128>(vb_ + s2 * 256);
        }
        // ---- online softmax (query on the lane; register r <-> key (r & 3) + 8 (r >> 2) + 4 h of the tile) ----
        // the running maximum is kept in RAW score units (scale > 0 commutes with max); the scale rides on the exp2 argument's fma.
        // Only the first / last tile of a sample can hold a neighbour's keys: interior tiles skip the mask arithmetic (uniform branch).
        const int key0 = 32 * kt + 4 * h - lo_in_blk;   // key index relative to the sample of register 0
        if (kt == 0 || kt == nkt - 1) {
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
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            if (d == 1) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    vt[s2][0][0] = lds_tr<2 * BLK>(vb_ + s2 * 256); vt[s2][0][1] = lds_tr<2 * BLK + 128>(vb_ + s2 * 256);
                    vt[s2][1][0] = lds_tr<2 * BLK + PIECE>(vb_ + s2 * 256); vt[s2][1][1] = lds_tr<2 * BLK + PIECE + 128>(vb_ + s2 * 256);
                    vt[s2][2][0] = lds_tr<2 * BLK + 2 * PIECE>(vb_ + s2 * 256); vt[s2][2][1] = lds_tr<2 * BLK + 2 * PIECE + 128>(vb_ + s2 * 256);
                }
            }
            wait_lds();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 vf[3] = {frag_of(vt[s2][0][0][0], vt[s2][0][0][1], vt[s2][0][1][0], vt[s2][0][1][1]),
                                      frag_of(vt[s2][1][0][0], vt[s2][1][0][1], vt[s2][1][1][0], vt[s2][1][1][1]),
                                      frag_of(vt[s2][2][0][0], vt[s2][2][0][1], vt[s2][2][1][0], vt[s2][2][1][1])};
                const bf16x8 pf[3] = {frag_of(pp0[4 * s2], pp0[4 * s2 + 1], pp0[4 * s2 + 2], pp0[4 * s2 + 3]),
                                      frag_of(pp1[4 * s2], pp1[4 * s2 + 1], pp1[4 * s2 + 2], pp1[4 * s2 + 3]),
                                      frag_of(pp2[4 * s2], pp2[4 * s2 + 1], pp2[4 * s2 + 2], pp2[4 * s2 + 3])};
                acc_o[d] = mma6(vf, pf, acc_o[d]);
            }
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
}

}  // namespace

extern "C" int tvl_attn_tp3_fwd(const void* qkv_tp3, void* o_tp3, float* lse, int32_t B, int32_t H, int32_t T, float scale, tvlStream_t stream) {
    TVL_REQUIRE(qkv_tp3 && o_tp3, "tvl_attn_tp3_fwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && T > 0 && B <= 65535 && H <= 65535, "tvl_attn_tp3_fwd: bad shape");
    TVL_REQUIRE(tvl_aligned16(qkv_tp3) && tvl_aligned16(o_tp3), "tvl_attn_tp3_fwd: tp3 images must be 16-byte aligned");
    FwdP p;
    p.qkv = reinterpret_cast<const unsigned char*>(qkv_tp3); p.kb = 3 * H * DH / 16;
    p.o_tp3 = reinterpret_cast<unsigned char*>(o_tp3); p.o_kb = H * DH / 16; p.lse = lse;
    p.B = B; p.H = H; p.T = T; p.scale = scale;
    dim3 grid((T + 127) / 128, H, B);
    hipLaunchKernelGGL(attn_fwd_tp3_kernel, grid, dim3(256), 2 * KV_TILE, reinterpret_cast<hipStream_t>(stream), p);
    TVL_LAUNCH_CHECK("tvl_attn_tp3_fwd");
    return 0;
}
