// Flash attention forward for d_h = 64 on the bf16 matrix cores with 3-way bf16 operand splitting (fp32-equivalent,
// see gemm_bf16s.hip): Q, K, V and the probabilities P are each carried as three bf16 pieces and every product is the
// 6-MFMA sum over piece pairs, accumulated in fp32.  Same structure as attn_fwd_kernel (attention.hip):
//     S^T = K . Q^T       A = K rows  (LDS [key][d], ds_read_b128),                B = Q (registers, pre-scaled)
//     O^T += V^T . P^T    A = V^T     (LDS [key][d] read with ds_read_b64_tr_b16),  B = P^T built from the S^T accumulators
// The k order of the second product is the accumulator order (guide §3): element j of lane half h of k-step s is key
// 16s + 8(j>>2) + 4h + (j&3); the transposed LDS read delivers V in exactly that order (two 4-key blocks per fragment).
// Used for the unmasked vision-tower attention (95 % of the attention FLOPs); masked / small-head cases stay on the
// exact-fp32 kernel.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int DH = 64;
constexpr int LDKB = DH + 8;   // K rows: 144 B -> ds_read_b128 conflict free
constexpr int LDVB = DH + 32;  // V rows: 192 B -> the 4 rows of a transposed-read block land on disjoint bank windows
constexpr int S = 3;

struct Params {
    const float *q, *k, *v; long q_bs, k_bs, v_bs; int q_ts, k_ts, v_ts;
    float* o; int ldo; float* lse;
    int B, H, T; float scale;
};

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned pack_trunc(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}
// x[0..N) -> three planes of N/2 dwords each (pieces 1,2 by truncation of the running residual, piece 3 rounded)
template <int N>
__device__ __forceinline__ void split3(float (&x)[N], unsigned (&p0)[N / 2], unsigned (&p1)[N / 2], unsigned (&p2)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) p0[i] = pack_trunc(fbits(x[2 * i]), fbits(x[2 * i + 1]));
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] - bfloat(fbits(x[i]) & 0xFFFF0000u);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) p1[i] = pack_trunc(fbits(x[2 * i]), fbits(x[2 * i + 1]));
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] - bfloat(fbits(x[i]) & 0xFFFF0000u);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) p2[i] = pack_rn(x[2 * i], x[2 * i + 1]);
}
__device__ __forceinline__ bf16x8 frag_of(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}

// stage a [32][64] fp32 tile (rows clamped to T-1) as three bf16 planes [3][32][LD]
struct TileRegs { float4 v[2]; };
__device__ __forceinline__ void tile_gload(TileRegs& s, const float* __restrict__ base, int ts, int row0, int T) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        int r = row0 + (idx >> 4);
        r = r < T ? r : T - 1;
        s.v[i] = *reinterpret_cast<const float4*>(base + (long)r * ts + 4 * (idx & 15));
    }
}
template <int LD>
__device__ __forceinline__ void tile_sstore(const TileRegs& s, __bf16* __restrict__ lds) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        float x[4] = {s.v[i].x, s.v[i].y, s.v[i].z, s.v[i].w};
        unsigned p0[2], p1[2], p2[2];
        split3<4>(x, p0, p1, p2);
        const int off = (idx >> 4) * LD + 4 * (idx & 15);
        *reinterpret_cast<uint2*>(&lds[off]) = make_uint2(p0[0], p0[1]);
        *reinterpret_cast<uint2*>(&lds[32 * LD + off]) = make_uint2(p1[0], p1[1]);
        *reinterpret_cast<uint2*>(&lds[64 * LD + off]) = make_uint2(p2[0], p2[1]);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_bf16s_kernel(Params p) {
    constexpr int KSZ = S * 32 * LDKB, VSZ = S * 32 * LDVB;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[2][KSZ];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[2][VSZ];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const int qrow = qi < T ? qi : T - 1;
    const float* qb = p.q + b * p.q_bs + head * DH;
    const float* kb = p.k + b * p.k_bs + head * DH;
    const float* vb = p.v + b * p.v_bs + head * DH;

    // Q fragments: 4 k16-steps x 3 planes, element j of half h = Q[q][16s + 8h + j], pre-scaled by scale*log2(e)
    bf16x8 qf[4][S];
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qb + (long)qrow * p.q_ts + 16 * s + 8 * h);
            const float4 c = *reinterpret_cast<const float4*>(qb + (long)qrow * p.q_ts + 16 * s + 8 * h + 4);
            float x[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, c.x * sc, c.y * sc, c.z * sc, c.w * sc};
            unsigned p0[4], p1[4], p2[4];
            split3<8>(x, p0, p1, p2);
            qf[s][0] = frag_of(p0[0], p0[1], p0[2], p0[3]);
            qf[s][1] = frag_of(p1[0], p1[1], p1[2], p1[3]);
            qf[s][2] = frag_of(p2[0], p2[1], p2[2], p2[3]);
        }
    }

    f32x16 acc_o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    // transposed-read addressing: 16-lane group = 4 keys x 16 d; lane 4q+p supplies row q, columns 4p..4p+3
    const int li = lane & 15;
    const int tr_row = li >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (li & 3);

    const int nkt = (T + 31) / 32;
    TileRegs sk, sv;
    tile_gload(sk, kb, p.k_ts, 0, T);
    tile_gload(sv, vb, p.v_ts, 0, T);
    tile_sstore<LDKB>(sk, Ks[0]);
    tile_sstore<LDVB>(sv, Vs[0]);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            tile_gload(sk, kb, p.k_ts, (kt + 1) * 32, T);
            tile_gload(sv, vb, p.v_ts, (kt + 1) * 32, T);
        }
        const __bf16* ks = Ks[cur];
        const __bf16* vs = Vs[cur];

        // ---- S^T = K . Q^T ----
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 kf[S];
#pragma unroll
            for (int pl = 0; pl < S; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(&ks[(pl * 32 + l31) * LDKB + 16 * s + 8 * h]);
#pragma unroll
            for (int order = S - 1; order >= 0; --order)
#pragma unroll
                for (int a = 0; a <= order; ++a) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[a], qf[s][order - a], sc, 0, 0, 0);
        }
        // ---- online softmax (query on the lane, keys kappa(r,h) in the registers) ----
        float mx = NEG_BIG;
        const int kbase = kt * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            const float v = key < T ? sc[r] : NEG_BIG;
            sc[r] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float pv[16];
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = sc[r] > 0.5f * NEG_BIG ? __builtin_amdgcn_exp2f(sc[r] - m_new) : 0.f;
            rs += pv[r];
        }
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // P^T fragments: k-step s2 takes registers 8*s2 .. 8*s2+7 (element j <-> key 16 s2 + 8 (j>>2) + 4 h + (j&3))
        unsigned pp0[8], pp1[8], pp2[8];
        split3<16>(pv, pp0, pp1, pp2);
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 vf[S];
#pragma unroll
                for (int pl = 0; pl < S; ++pl) {
                    const __bf16* base = vs + (pl * 32 + 16 * s2 + 4 * h + tr_row) * LDVB + d * 32 + tr_col;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * LDVB));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    vf[pl] = frag_of(l2.x, l2.y, h2.x, h2.y);
                }
                const bf16x8 pf[S] = {frag_of(pp0[4 * s2], pp0[4 * s2 + 1], pp0[4 * s2 + 2], pp0[4 * s2 + 3]),
                                      frag_of(pp1[4 * s2], pp1[4 * s2 + 1], pp1[4 * s2 + 2], pp1[4 * s2 + 3]),
                                      frag_of(pp2[4 * s2], pp2[4 * s2 + 1], pp2[4 * s2 + 2], pp2[4 * s2 + 3])};
#pragma unroll
                for (int order = S - 1; order >= 0; --order)
#pragma unroll
                    for (int a = 0; a <= order; ++a)
                        acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[a], pf[order - a], acc_o[d], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) {
            tile_sstore<LDKB>(sk, Ks[cur ^ 1]);
            tile_sstore<LDVB>(sv, Vs[cur ^ 1]);
        }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < T) {
        float* ob = p.o + ((long)b * T + qi) * p.ldo + head * DH;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(ob + d * 32 + 8 * g + 4 * h) = make_float4(acc_o[d][4 * g] * inv, acc_o[d][4 * g + 1] * inv,
                                                                                   acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv);
        if (h == 0 && p.lse) p.lse[((long)b * p.H + head) * T + qi] = (m_run + log2f(l_tot)) * LN2;
    }
}

}  // namespace

// internal entry (dispatched from tvl_attn_fwd): d_h = 64, no causal / key mask
int tvl_attn_fwd_bf16s_impl(const tvlAttnFwdArgs* a, hipStream_t s) {
    Params p;
    p.q = a->q; p.k = a->k; p.v = a->v; p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs;
    p.q_ts = a->q_ts; p.k_ts = a->k_ts; p.v_ts = a->v_ts; p.o = a->o; p.ldo = a->ldo; p.lse = a->lse;
    p.B = a->B; p.H = a->H; p.T = a->T; p.scale = a->scale;
    dim3 grid((a->T + 127) / 128, a->H, a->B);
    hipLaunchKernelGGL(attn_fwd_bf16s_kernel, grid, dim3(256), 0, s, p);
    return 0;
}
