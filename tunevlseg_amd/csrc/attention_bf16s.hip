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
#include "tp3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
// DH = 64: the vision tower (fp32 operands: layers below the image-format threshold, CRIS).  DH = 16 (round 3): the CLIPSeg decoder's
// 4 x 16 heads (reduce_dim 64) -- one k-step for S, half of each 32-row output block unused, still 18 MFMAs of 32 cycles per key tile
// against 24 of 64 cycles on the fp32 MFMA the decoder ran on before.
constexpr int S = 3;
template <int DH> struct Geo {
    static constexpr int LDKB = DH + 8;    // K rows: 144 B (DH = 64) -> ds_read_b128 conflict free
    static constexpr int LDVB = DH + 32;   // V rows: 192 B (DH = 64) -> the 4 rows of a transposed-read block land on disjoint bank windows
    static constexpr int KS = DH / 16;     // k-steps of a product that contracts over d
    static constexpr int NDB = (DH + 31) / 32;   // 32-row blocks of an output whose rows are d
    static constexpr int F4 = 32 * DH / 4;       // float4 of a [32][DH] tile
    static constexpr int NLD = (F4 + 255) / 256; // ... per thread
};

struct Params {
    const float *q, *k, *v; long q_bs, k_bs, v_bs; int q_ts, k_ts, v_ts;
    float* o; int ldo; float* lse;
    int B, H, T; float scale;
    unsigned char* o_tp3; int o_kb;   // optional: O as the tp3 image of [B*T, H*64] (o_kb = H*64/16), the out_proj GEMM's A operand
};

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned pack_trunc(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}
// x[0..N) -> three planes of N/2 dwords each: every piece is the round-to-nearest bf16 of the running residual (tp3.h)
template <int N>
__device__ __forceinline__ void split3(float (&x)[N], unsigned (&p0)[N / 2], unsigned (&p1)[N / 2], unsigned (&p2)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        p0[i] = pack_rn(x[2 * i], x[2 * i + 1]);
        x[2 * i] -= bfloat(p0[i] << 16); x[2 * i + 1] -= bfloat(p0[i] & 0xFFFF0000u);
        p1[i] = pack_rn(x[2 * i], x[2 * i + 1]);
        x[2 * i] -= bfloat(p1[i] << 16); x[2 * i + 1] -= bfloat(p1[i] & 0xFFFF0000u);
        p2[i] = pack_rn(x[2 * i], x[2 * i + 1]);
    }
}
__device__ __forceinline__ bf16x8 frag_of(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}

// stage a [32][DH] fp32 tile (rows clamped to T-1) as three bf16 planes [3][32][LD]
template <int DH> struct TileRegs { float4 v[Geo<DH>::NLD]; };
template <int DH>
__device__ __forceinline__ void tile_gload(TileRegs<DH>& s, const float* __restrict__ base, int ts, int row0, int T) {
#pragma unroll
    for (int i = 0; i < Geo<DH>::NLD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < Geo<DH>::F4) {
            int r = row0 + idx / (DH / 4);
            r = r < T ? r : T - 1;
            s.v[i] = *reinterpret_cast<const float4*>(base + (long)r * ts + 4 * (idx % (DH / 4)));
        }
    }
}
template <int DH, int LD>
__device__ __forceinline__ void tile_sstore(const TileRegs<DH>& s, __bf16* __restrict__ lds) {
#pragma unroll
    for (int i = 0; i < Geo<DH>::NLD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < Geo<DH>::F4) {
            float x[4] = {s.v[i].x, s.v[i].y, s.v[i].z, s.v[i].w};
            unsigned p0[2], p1[2], p2[2];
            split3<4>(x, p0, p1, p2);
            const int off = (idx / (DH / 4)) * LD + 4 * (idx % (DH / 4));
            *reinterpret_cast<uint2*>(&lds[off]) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(&lds[32 * LD + off]) = make_uint2(p1[0], p1[1]);
            *reinterpret_cast<uint2*>(&lds[64 * LD + off]) = make_uint2(p2[0], p2[1]);
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_bf16s_kernel(Params p) {
    TVL_KERNEL_ENTRY();
    constexpr int LDKB = Geo<DH>::LDKB, LDVB = Geo<DH>::LDVB, KS = Geo<DH>::KS, NDB = Geo<DH>::NDB;
    (void)LDVB;
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
    bf16x8 qf[KS][S];
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
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

    f32x16 acc_o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    // transposed-read addressing: 16-lane group = 4 keys x 16 d; lane 4q+p supplies row q, columns 4p..4p+3
    const int li = lane & 15;
    const int tr_row = li >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (li & 3);

    const int nkt = (T + 31) / 32;
    TileRegs<DH> sk, sv;
    tile_gload(sk, kb, p.k_ts, 0, T);
    tile_gload(sv, vb, p.v_ts, 0, T);
    tile_sstore<DH, LDKB>(sk, Ks[0]);
    tile_sstore<DH, LDVB>(sv, Vs[0]);
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
        for (int s = 0; s < KS; ++s) {
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
        mx = xor32_max(mx);
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
        for (int d = 0; d < NDB; ++d) {
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
            tile_sstore<DH, LDKB>(sk, Ks[cur ^ 1]);
            tile_sstore<DH, LDVB>(sv, Vs[cur ^ 1]);
        }
        __syncthreads();
    }

    const float l_tot = xor32_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (qi < T) {
        const long m = (long)b * T + qi;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {acc_o[d][4 * g] * inv, acc_o[d][4 * g + 1] * inv, acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv};
                const int col = head * DH + d * 32 + 8 * g + 4 * h;
                if (d * 32 + 8 * g + 4 * h >= DH) continue;
                if (p.o) *reinterpret_cast<float4*>(p.o + m * p.ldo + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (p.o_tp3) tp3::store4(p.o_tp3, p.o_kb, m, col, v);   // the two lane halves fill one 16-byte chunk per row
            }
        if (h == 0 && p.lse) p.lse[((long)b * p.H + head) * T + qi] = (m_run + log2f(l_tot)) * LN2;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// backward, same two-kernel split as attention.hip (dQ: query on the lane; dK/dV: key on the lane), P recomputed from
// Q, K and the forward LSE.  Tiles that are read both by rows (first products) and transposed (second products) use
// the 144-byte row stride: conflict free for ds_read_b128, 2-way for half of the transposed reads.
// ------------------------------------------------------------------------------------------------------------------
struct BwdParams {
    const float *q, *k, *v; long q_bs, k_bs, v_bs; int q_ts, k_ts, v_ts;
    const float* d_o; int ldo; const float* lse; const float* delta;
    float *dq, *dk, *dv; long dq_bs, dk_bs, dv_bs; int dq_ts, dk_ts, dv_ts;
    int B, H, T; float scale;
    unsigned char* g_tp3; int g_kb;   // optional: dQ | dK | dV as the tp3 image of the packed gradient [B*T, 3*H*64] (g_kb = 3*H*64/16)
};

// 8 consecutive fp32 of one row -> three bf16x8 fragments (optionally scaled)
__device__ __forceinline__ void row_frags(const float* __restrict__ src, float sc, bf16x8 (&out)[S]) {
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 c = *reinterpret_cast<const float4*>(src + 4);
    float x[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, c.x * sc, c.y * sc, c.z * sc, c.w * sc};
    unsigned p0[4], p1[4], p2[4];
    split3<8>(x, p0, p1, p2);
    out[0] = frag_of(p0[0], p0[1], p0[2], p0[3]);
    out[1] = frag_of(p1[0], p1[1], p1[2], p1[3]);
    out[2] = frag_of(p2[0], p2[1], p2[2], p2[3]);
}
// row fragments of k16-step s from an LDS tile [3][32][LDKB]
template <int LDKB>
__device__ __forceinline__ void lds_row_frags(const __bf16* __restrict__ t, int l31, int h, int s, bf16x8 (&out)[S]) {
#pragma unroll
    for (int pl = 0; pl < S; ++pl) out[pl] = *reinterpret_cast<const bf16x8*>(&t[(pl * 32 + l31) * LDKB + 16 * s + 8 * h]);
}
// transposed fragments (A[i = column d][k = row]) of k-step s2 / 32-column block d from an LDS tile [3][32][LD]
template <int LD>
__device__ __forceinline__ void lds_tr_frags(const __bf16* __restrict__ t, int tr_row, int tr_col, int h, int s2, int d, bf16x8 (&out)[S]) {
#pragma unroll
    for (int pl = 0; pl < S; ++pl) {
        const __bf16* base = t + (pl * 32 + 16 * s2 + 4 * h + tr_row) * LD + d * 32 + tr_col;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * LD));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        out[pl] = frag_of(l2.x, l2.y, h2.x, h2.y);
    }
}
__device__ __forceinline__ f32x16 mma6(const bf16x8 (&a)[S], const bf16x8 (&b)[S], f32x16 acc) {
#pragma unroll
    for (int order = S - 1; order >= 0; --order)
#pragma unroll
        for (int i = 0; i <= order; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[order - i], acc, 0, 0, 0);
    return acc;
}

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16s_kernel(BwdParams p) {
    TVL_KERNEL_ENTRY();
    constexpr int LDKB = Geo<DH>::LDKB, LDVB = Geo<DH>::LDVB, KS = Geo<DH>::KS, NDB = Geo<DH>::NDB;
    (void)LDVB;
    constexpr int TSZ = S * 32 * LDKB;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[2][TSZ];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[2][TSZ];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const int qrow = qi < T ? qi : T - 1;
    const float* qb = p.q + b * p.q_bs + head * DH;
    const float* kb = p.k + b * p.k_bs + head * DH;
    const float* vb = p.v + b * p.v_bs + head * DH;
    const float* dob = p.d_o + ((long)b * T) * p.ldo + head * DH;

    bf16x8 qf[KS][S], dof[KS][S];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        row_frags(qb + (long)qrow * p.q_ts + 16 * s + 8 * h, p.scale * LOG2E, qf[s]);
        row_frags(dob + (long)qrow * p.ldo + 16 * s + 8 * h, 1.0f, dof[s]);
    }
    const long stat = ((long)b * p.H + head) * T + qrow;
    const float lse2 = p.lse[stat] * LOG2E;
    const float dl = p.delta[stat];

    f32x16 acc_dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[d][r] = 0.f;
    const int li = lane & 15;
    const int tr_row = li >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (li & 3);

    const int nkt = (T + 31) / 32;
    TileRegs<DH> sk, sv;
    tile_gload(sk, kb, p.k_ts, 0, T);
    tile_gload(sv, vb, p.v_ts, 0, T);
    tile_sstore<DH, LDKB>(sk, Ks[0]);
    tile_sstore<DH, LDKB>(sv, Vs[0]);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            tile_gload(sk, kb, p.k_ts, (kt + 1) * 32, T);
            tile_gload(sv, vb, p.v_ts, (kt + 1) * 32, T);
        }
        const __bf16* ks = Ks[cur];
        const __bf16* vs = Vs[cur];
        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 kf[S], vf[S];
            lds_row_frags<LDKB>(ks, l31, h, s, kf);
            lds_row_frags<LDKB>(vs, l31, h, s, vf);
            sc = mma6(kf, qf[s], sc);
            dp = mma6(vf, dof[s], dp);
        }
        float ds[16];
        const int kbase = kt * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            const float pv = key < T ? __builtin_amdgcn_exp2f(sc[r] - lse2) : 0.f;
            ds[r] = pv * (dp[r] - dl);
        }
        unsigned d0[8], d1[8], d2[8];
        split3<16>(ds, d0, d1, d2);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 df[S] = {frag_of(d0[4 * s2], d0[4 * s2 + 1], d0[4 * s2 + 2], d0[4 * s2 + 3]),
                                  frag_of(d1[4 * s2], d1[4 * s2 + 1], d1[4 * s2 + 2], d1[4 * s2 + 3]),
                                  frag_of(d2[4 * s2], d2[4 * s2 + 1], d2[4 * s2 + 2], d2[4 * s2 + 3])};
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                bf16x8 ktf[S];
                lds_tr_frags<LDKB>(ks, tr_row, tr_col, h, s2, d, ktf);
                acc_dq[d] = mma6(ktf, df, acc_dq[d]);
            }
        }
        if (kt + 1 < nkt) {
            tile_sstore<DH, LDKB>(sk, Ks[cur ^ 1]);
            tile_sstore<DH, LDKB>(sv, Vs[cur ^ 1]);
        }
        __syncthreads();
    }
    if (qi < T) {
        const long m = (long)b * T + qi;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {acc_dq[d][4 * g] * p.scale, acc_dq[d][4 * g + 1] * p.scale, acc_dq[d][4 * g + 2] * p.scale,
                                    acc_dq[d][4 * g + 3] * p.scale};
                const int col = head * DH + d * 32 + 8 * g + 4 * h;
                if (d * 32 + 8 * g + 4 * h >= DH) continue;
                if (p.dq) *reinterpret_cast<float4*>(p.dq + b * p.dq_bs + (long)qi * p.dq_ts + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (p.g_tp3) tp3::store4(p.g_tp3, p.g_kb, m, col, v);
            }
    }
}

template <int DH>
__global__ __launch_bounds__(256, DH == 64 ? 1 : 2) void attn_bwd_dkdv_bf16s_kernel(BwdParams p) {
    TVL_KERNEL_ENTRY();
    constexpr int LDKB = Geo<DH>::LDKB, LDVB = Geo<DH>::LDVB, KS = Geo<DH>::KS, NDB = Geo<DH>::NDB;
    (void)LDVB;
    constexpr int TSZ = S * 32 * LDKB;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[2][TSZ];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[2][TSZ];
    __shared__ __attribute__((aligned(16))) float lse_s[2][32];
    __shared__ __attribute__((aligned(16))) float del_s[2][32];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T;
    const int ki = blockIdx.x * 128 + wave * 32 + l31;
    const int krow = ki < T ? ki : T - 1;
    const bool key_ok = ki < T;
    const float* qb = p.q + b * p.q_bs + head * DH;
    const float* kb = p.k + b * p.k_bs + head * DH;
    const float* vb = p.v + b * p.v_bs + head * DH;
    const float* dob = p.d_o + ((long)b * T) * p.ldo + head * DH;
    const float* lseb = p.lse + ((long)b * p.H + head) * T;
    const float* delb = p.delta + ((long)b * p.H + head) * T;

    bf16x8 kf[KS][S], vf[KS][S];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        row_frags(kb + (long)krow * p.k_ts + 16 * s + 8 * h, p.scale * LOG2E, kf[s]);
        row_frags(vb + (long)krow * p.v_ts + 16 * s + 8 * h, 1.0f, vf[s]);
    }
    f32x16 acc_dk[NDB], acc_dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_dk[d][r] = 0.f; acc_dv[d][r] = 0.f; }
    const int li = lane & 15;
    const int tr_row = li >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (li & 3);

    const int nqt = (T + 31) / 32;
    TileRegs<DH> sq, sd;
    float lse_reg = 0.f, del_reg = 0.f;
    auto gload = [&](int qt) {
        tile_gload(sq, qb, p.q_ts, qt * 32, T);
        tile_gload(sd, dob, p.ldo, qt * 32, T);
        if (threadIdx.x < 32) {
            int q = qt * 32 + threadIdx.x;
            q = q < T ? q : T - 1;
            lse_reg = lseb[q] * LOG2E;
            del_reg = delb[q];
        }
    };
    auto sstore = [&](int buf) {
        tile_sstore<DH, LDKB>(sq, Qs[buf]);
        tile_sstore<DH, LDKB>(sd, Ds[buf]);
        if (threadIdx.x < 32) { lse_s[buf][threadIdx.x] = lse_reg; del_s[buf][threadIdx.x] = del_reg; }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        const int cur = qt & 1;
        if (qt + 1 < nqt) gload(qt + 1);
        const __bf16* qs = Qs[cur];
        const __bf16* ds_t = Ds[cur];
        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 qrf[S], drf[S];
            lds_row_frags<LDKB>(qs, l31, h, s, qrf);
            lds_row_frags<LDKB>(ds_t, l31, h, s, drf);
            sc = mma6(qrf, kf[s], sc);
            dp = mma6(drf, vf[s], dp);
        }
        // rows of S / dP are queries kappa(r,h)
        float pv[16], dsv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[cur][8 * g + 4 * h]);
            const float4 d4 = *reinterpret_cast<const float4*>(&del_s[cur][8 * g + 4 * h]);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
            const float dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const bool ok = key_ok && (qt * 32 + 8 * g + 4 * h + i) < T;
                pv[r] = ok ? __builtin_amdgcn_exp2f(sc[r] - lv[i]) : 0.f;
                dsv[r] = pv[r] * (dp[r] - dv4[i]);
            }
        }
        unsigned a0[8], a1[8], a2[8];
        split3<16>(pv, a0, a1, a2);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf[S] = {frag_of(a0[4 * s2], a0[4 * s2 + 1], a0[4 * s2 + 2], a0[4 * s2 + 3]),
                                  frag_of(a1[4 * s2], a1[4 * s2 + 1], a1[4 * s2 + 2], a1[4 * s2 + 3]),
                                  frag_of(a2[4 * s2], a2[4 * s2 + 1], a2[4 * s2 + 2], a2[4 * s2 + 3])};
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                bf16x8 dtf[S];
                lds_tr_frags<LDKB>(ds_t, tr_row, tr_col, h, s2, d, dtf);
                acc_dv[d] = mma6(dtf, pf, acc_dv[d]);
            }
        }
        split3<16>(dsv, a0, a1, a2);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 sf[S] = {frag_of(a0[4 * s2], a0[4 * s2 + 1], a0[4 * s2 + 2], a0[4 * s2 + 3]),
                                  frag_of(a1[4 * s2], a1[4 * s2 + 1], a1[4 * s2 + 2], a1[4 * s2 + 3]),
                                  frag_of(a2[4 * s2], a2[4 * s2 + 1], a2[4 * s2 + 2], a2[4 * s2 + 3])};
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                bf16x8 qtf[S];
                lds_tr_frags<LDKB>(qs, tr_row, tr_col, h, s2, d, qtf);
                acc_dk[d] = mma6(qtf, sf, acc_dk[d]);
            }
        }
        if (qt + 1 < nqt) sstore(cur ^ 1);
        __syncthreads();
    }
    if (ki < T) {
        const long m = (long)b * T + ki;
        const int D = p.H * DH;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = head * DH + d * 32 + 8 * g + 4 * h;
                if (d * 32 + 8 * g + 4 * h >= DH) continue;
                const float vk[4] = {acc_dk[d][4 * g] * p.scale, acc_dk[d][4 * g + 1] * p.scale, acc_dk[d][4 * g + 2] * p.scale,
                                     acc_dk[d][4 * g + 3] * p.scale};
                const float vv[4] = {acc_dv[d][4 * g], acc_dv[d][4 * g + 1], acc_dv[d][4 * g + 2], acc_dv[d][4 * g + 3]};
                if (p.dk) *reinterpret_cast<float4*>(p.dk + b * p.dk_bs + (long)ki * p.dk_ts + col) = make_float4(vk[0], vk[1], vk[2], vk[3]);
                if (p.dv) *reinterpret_cast<float4*>(p.dv + b * p.dv_bs + (long)ki * p.dv_ts + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                if (p.g_tp3) {
                    tp3::store4(p.g_tp3, p.g_kb, m, D + col, vk);
                    tp3::store4(p.g_tp3, p.g_kb, m, 2 * D + col, vv);
                }
            }
    }
}

// delta[b,h,t] = sum_d dO * O with O read from its tp3 image: one wave per (32-row block, head); lane (r, hh) rebuilds 8 columns
// of 4 k-blocks (32 of the head's 64 columns), the two lane halves are summed by one cross-half shuffle
__global__ __launch_bounds__(256) void attn_delta_tp3_kernel(const unsigned char* __restrict__ o_tp3, int o_kb, const float* __restrict__ d_o,
                                                             int ldo, float* __restrict__ delta, int B, int H, int T) {
    TVL_KERNEL_ENTRY();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long M = (long)B * T;
    const long rbs = (M + 31) / 32;
    const long item = (long)blockIdx.x * 4 + wave;   // (row block, head)
    if (item >= rbs * H) return;
    const long rb = item / H;
    const int head = (int)(item % H);
    const int r = lane & 31, hh = lane >> 5;
    const long m = rb * 32 + r;
    float acc = 0.f;
    if (m < M) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int kb = head * 4 + kq;
            const unsigned char* src = o_tp3 + (rb * o_kb + kb) * (long)tp3::BLK + lane * 16;
            uint4 pl[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) pl[s] = *reinterpret_cast<const uint4*>(src + s * tp3::PIECE);
            float ov[8];
            tp3::join8(pl, ov);
            const float* g = d_o + m * ldo + kb * 16 + hh * 8;
            const float4 a = *reinterpret_cast<const float4*>(g), c = *reinterpret_cast<const float4*>(g + 4);
            acc += (ov[0] * a.x + ov[1] * a.y) + (ov[2] * a.z + ov[3] * a.w) + (ov[4] * c.x + ov[5] * c.y) + (ov[6] * c.z + ov[7] * c.w);
        }
    }
    acc = xor32_sum(acc);
    if (m < M && hh == 0) {
        const long bb = m / T, t = m % T;
        delta[(bb * H + head) * T + t] = acc;
    }
}

}  // namespace

// internal entry (dispatched from tvl_attn_fwd): d_h = 64, no causal / key mask
int tvl_attn_fwd_bf16s_impl(const tvlAttnFwdArgs* a, void* o_tp3, hipStream_t s) {
    Params p;
    p.o_tp3 = reinterpret_cast<unsigned char*>(o_tp3); p.o_kb = a->H * a->dh / 16;
    if (a->dh != 64 && (a->dh != 16 || o_tp3)) return 1;
    p.q = a->q; p.k = a->k; p.v = a->v; p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs;
    p.q_ts = a->q_ts; p.k_ts = a->k_ts; p.v_ts = a->v_ts; p.o = a->o; p.ldo = a->ldo; p.lse = a->lse;
    p.B = a->B; p.H = a->H; p.T = a->T; p.scale = a->scale;
    dim3 grid((a->T + 127) / 128, a->H, a->B);
    if (a->dh == 64) hipLaunchKernelGGL(attn_fwd_bf16s_kernel<64>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_fwd_bf16s_kernel<16>, grid, dim3(256), 0, s, p);
    return 0;
}

// dQ + dK/dV kernels (delta is computed by the caller's pre-pass)
int tvl_attn_bwd_bf16s_impl(const tvlAttnBwdArgs* a, const void* o_tp3, void* dqkv_tp3, hipStream_t s) {
    BwdParams p;
    p.g_tp3 = reinterpret_cast<unsigned char*>(dqkv_tp3); p.g_kb = 3 * a->H * a->dh / 16;
    if (a->dh != 64 && (a->dh != 16 || o_tp3 || dqkv_tp3)) return 1;
    if (o_tp3) {  // delta from the tp3 image of O (the fp32 O was never written)
        const long items = (((long)a->B * a->T + 31) / 32) * a->H;
        hipLaunchKernelGGL(attn_delta_tp3_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, reinterpret_cast<const unsigned char*>(o_tp3),
                           a->H * 64 / 16, a->d_o, a->ldo, a->delta, a->B, a->H, a->T);
    }
    p.q = a->q; p.k = a->k; p.v = a->v; p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs;
    p.q_ts = a->q_ts; p.k_ts = a->k_ts; p.v_ts = a->v_ts; p.d_o = a->d_o; p.ldo = a->ldo; p.lse = a->lse; p.delta = a->delta;
    p.dq = a->dq; p.dk = a->dk; p.dv = a->dv; p.dq_bs = a->dq_bs; p.dk_bs = a->dk_bs; p.dv_bs = a->dv_bs;
    p.dq_ts = a->dq_ts; p.dk_ts = a->dk_ts; p.dv_ts = a->dv_ts; p.B = a->B; p.H = a->H; p.T = a->T; p.scale = a->scale;
    dim3 grid((a->T + 127) / 128, a->H, a->B);
    if (a->dh == 64) {
        hipLaunchKernelGGL(attn_bwd_dq_bf16s_kernel<64>, grid, dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dkdv_bf16s_kernel<64>, grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_bf16s_kernel<16>, grid, dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dkdv_bf16s_kernel<16>, grid, dim3(256), 0, s, p);
    }
    return 0;
}
