// Flash-style multi-head softmax attention on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Forward (one wave = 32 query rows, 4 waves per workgroup share the K/V tiles in LDS):
//   S^T = K . Q^T          A = K tile (LDS, ds_read_b128, k-permuted), B = Q (registers)
//   online softmax         the query index sits on the LANE (C/D layout: col = lane&31), the 32 keys of
//                          the tile sit in the 16 accumulator registers x 2 lane halves -> row max/sum are
//                          register reductions + one cross-half shuffle
//   O^T += V^T . P^T       A = V rows (LDS, ds_read_b32, conflict free), B = P^T straight from the
//                          S^T accumulators ("accumulator tile as the next MFMA's operand", guide §3)
// so neither the TxT scores nor P ever leave registers.
//
// Backward = delta pre-pass + two kernels with the same structure (deterministic, no atomics):
//   dQ   kernel: wave owns 32 queries (query on the lane), loops over key tiles
//   dKdV kernel: wave owns 32 keys    (key   on the lane), loops over query tiles
// Both recompute P from Q, K and the forward's LSE.
//
// kappa(r,h) = (r&3) + 8*(r>>2) + 4*h : row of a 32x32 accumulator held in register r of lane half h.
#include <stdlib.h>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int kappa(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnParams {
    const float *q, *k, *v; long q_bs, k_bs, v_bs; int q_ts, k_ts, v_ts;
    float* o; const float* o_in; const float* d_o; int ldo;
    float* lse; const float* lse_in; float* delta;
    float *dq, *dk, *dv; long dq_bs, dk_bs, dv_bs; int dq_ts, dk_ts, dv_ts;
    const int* key_mask;
    int B, H, T, Tk, causal;  // T = queries, Tk = keys (== T for self-attention)
    float scale;
};

// cooperative global -> register -> LDS staging of a [32][DH] tile into [32][DH+4]
template <int DH>
struct TileRegs {
    static constexpr int N = (32 * DH / 4 + 255) / 256;
    // a native vector, not float4: HIP's float4 is a struct whose copy is a memcpy, and memcpys in and out of a two-element array keep the
    // array a stack object (d_h = 64: 80 B of scratch per lane in all three kernels) -- vector loads / stores are promoted to registers
    typedef float vec4 __attribute__((ext_vector_type(4)));
    vec4 v[N];
};
template <int DH>
__device__ __forceinline__ void tile_gload(TileRegs<DH>& s, const float* __restrict__ base, int ts, int row0, int T) {
#pragma unroll
    for (int i = 0; i < TileRegs<DH>::N; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < 32 * DH / 4) {
            int r = row0 + idx / (DH / 4);
            r = r < T ? r : T - 1;
            s.v[i] = *reinterpret_cast<const typename TileRegs<DH>::vec4*>(base + (long)r * ts + 4 * (idx % (DH / 4)));
        }
    }
}
template <int DH>
__device__ __forceinline__ void tile_sstore(const TileRegs<DH>& s, float* __restrict__ lds) {
#pragma unroll
    for (int i = 0; i < TileRegs<DH>::N; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < 32 * DH / 4) *reinterpret_cast<typename TileRegs<DH>::vec4*>(&lds[(idx / (DH / 4)) * (DH + 4) + 4 * (idx % (DH / 4))]) = s.v[i];
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    TVL_KERNEL_ENTRY();
    constexpr int LD = DH + 4;
    constexpr int NC = DH / 8;            // k-chunks of 8
    constexpr int NDB = (DH + 31) / 32;   // 32-wide d blocks
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float kbias[2][32];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T, Tk = p.Tk;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qi = q0 + l31;
    const int qrow = qi < T ? qi : T - 1;

    const float* qb = p.q + b * p.q_bs + head * DH;
    const float* kb = p.k + b * p.k_bs + head * DH;
    const float* vb = p.v + b * p.v_bs + head * DH;
    const int* mb = p.key_mask ? p.key_mask + (long)b * Tk : nullptr;

    // Q fragment, pre-scaled so that exp2 can be used: s' = (q.k) * scale * log2(e)
    float qreg[NC * 4];
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(qb + (long)qrow * p.q_ts + 8 * c + 4 * h);
            qreg[4 * c + 0] = v.x * sc; qreg[4 * c + 1] = v.y * sc; qreg[4 * c + 2] = v.z * sc; qreg[4 * c + 3] = v.w * sc;
        }
    }

    f32x16 acc_o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    const int nkt = (Tk + 31) / 32;
    TileRegs<DH> sk, sv;
    float bias_reg = 0.f;
    auto gload = [&](int kt) {
        tile_gload<DH>(sk, kb, p.k_ts, kt * 32, Tk);
        tile_gload<DH>(sv, vb, p.v_ts, kt * 32, Tk);
        if (threadIdx.x < 32) {
            const int key = kt * 32 + threadIdx.x;
            bias_reg = (key < Tk && (!mb || mb[key] != 0)) ? 0.f : NEG_BIG;
        }
    };
    auto sstore = [&](int buf) {
        tile_sstore<DH>(sk, Ks[buf]);
        tile_sstore<DH>(sv, Vs[buf]);
        if (threadIdx.x < 32) kbias[buf][threadIdx.x] = bias_reg;
    };
    gload(0);
    sstore(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float* ks = Ks[cur];
        const float* vs = Vs[cur];

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(&ks[l31 * LD + 8 * c + 4 * h]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[4 * c + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[4 * c + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[4 * c + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[4 * c + 3], s, 0, 0, 0);
        }
        // mask: additive key bias (tail + padding) and causal cut
        float mx = NEG_BIG;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 kb4 = *reinterpret_cast<const float4*>(&kbias[cur][8 * g + 4 * h]);
            const float kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                float v = s[r] + kbv[i];
                if (p.causal && (kt * 32 + 8 * g + 4 * h + i) > qi) v = NEG_BIG;
                v = fmaxf(v, NEG_BIG);
                s[r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = s[r] > 0.5f * NEG_BIG ? fast_exp2(s[r] - m_new) : 0.f;
            s[r] = pv;
            rs += pv;
        }
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            const int col = d * 32 + l31;
            const bool colok = col < DH;
            const int colc = colok ? col : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = vs[kappa(r, h) * LD + colc];
                a = colok ? a : 0.f;
                acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], acc_o[d], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < T) {
        float* ob = p.o + ((long)b * T + qi) * p.ldo + head * DH;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = d * 32 + 8 * g + 4 * h;
                if (col < DH) {
                    float4 o4 = make_float4(acc_o[d][4 * g] * inv, acc_o[d][4 * g + 1] * inv, acc_o[d][4 * g + 2] * inv,
                                            acc_o[d][4 * g + 3] * inv);
                    *reinterpret_cast<float4*>(ob + col) = o4;
                }
            }
        if (h == 0 && p.lse) p.lse[((long)b * p.H + head) * T + qi] = (m_run + log2f(l_tot)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------
// backward pre-pass: delta[b,h,t] = sum_d dO * O
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, int ldo,
                                                         float* __restrict__ delta, int B, int H, int T) {
    TVL_KERNEL_ENTRY();
    constexpr int G = DH / 4;  // lanes per (row, head)
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long pair = gid / G;
    const int sub = (int)(gid % G);
    const long total = (long)B * T * H;
    float acc = 0.f;
    long row = 0; int head = 0;
    if (pair < total) {
        row = pair / H;
        head = (int)(pair % H);
        const float4 a = *reinterpret_cast<const float4*>(o + row * ldo + head * DH + 4 * sub);
        const float4 c = *reinterpret_cast<const float4*>(d_o + row * ldo + head * DH + 4 * sub);
        acc = (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (pair < total && sub == 0) {
        const long bb = row / T, t = row % T;
        delta[(bb * H + head) * T + t] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// backward dQ: query on the lane
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
    TVL_KERNEL_ENTRY();
    constexpr int LD = DH + 4;
    constexpr int NC = DH / 8;
    constexpr int NDB = (DH + 31) / 32;
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float kbias[2][32];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T, Tk = p.Tk;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const int qrow = qi < T ? qi : T - 1;

    const float* qb = p.q + b * p.q_bs + head * DH;
    const float* kb = p.k + b * p.k_bs + head * DH;
    const float* vb = p.v + b * p.v_bs + head * DH;
    const float* dob = p.d_o + ((long)b * T) * p.ldo + head * DH;
    const int* mb = p.key_mask ? p.key_mask + (long)b * Tk : nullptr;

    float qreg[NC * 4], doreg[NC * 4];
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(qb + (long)qrow * p.q_ts + 8 * c + 4 * h);
            qreg[4 * c + 0] = v.x * sc; qreg[4 * c + 1] = v.y * sc; qreg[4 * c + 2] = v.z * sc; qreg[4 * c + 3] = v.w * sc;
            const float4 w = *reinterpret_cast<const float4*>(dob + (long)qrow * p.ldo + 8 * c + 4 * h);
            doreg[4 * c + 0] = w.x; doreg[4 * c + 1] = w.y; doreg[4 * c + 2] = w.z; doreg[4 * c + 3] = w.w;
        }
    }
    const long stat = ((long)b * p.H + head) * T + qrow;
    const float lse2 = p.lse_in[stat] * LOG2E;
    const float dl = p.delta[stat];

    f32x16 acc_dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[d][r] = 0.f;

    const int nkt = (Tk + 31) / 32;
    TileRegs<DH> sk, sv;
    float bias_reg = 0.f;
    auto gload = [&](int kt) {
        tile_gload<DH>(sk, kb, p.k_ts, kt * 32, Tk);
        tile_gload<DH>(sv, vb, p.v_ts, kt * 32, Tk);
        if (threadIdx.x < 32) {
            const int key = kt * 32 + threadIdx.x;
            bias_reg = (key < Tk && (!mb || mb[key] != 0)) ? 0.f : NEG_BIG;
        }
    };
    auto sstore = [&](int buf) {
        tile_sstore<DH>(sk, Ks[buf]);
        tile_sstore<DH>(sv, Vs[buf]);
        if (threadIdx.x < 32) kbias[buf][threadIdx.x] = bias_reg;
    };
    gload(0);
    sstore(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float* ks = Ks[cur];
        const float* vs = Vs[cur];

        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(&ks[l31 * LD + 8 * c + 4 * h]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[4 * c + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[4 * c + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[4 * c + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[4 * c + 3], s, 0, 0, 0);
            const float4 vf = *reinterpret_cast<const float4*>(&vs[l31 * LD + 8 * c + 4 * h]);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, doreg[4 * c + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, doreg[4 * c + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, doreg[4 * c + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, doreg[4 * c + 3], dp, 0, 0, 0);
        }
        // dS^T = P^T * (dP^T - delta)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 kb4 = *reinterpret_cast<const float4*>(&kbias[cur][8 * g + 4 * h]);
            const float kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                bool ok = kbv[i] == 0.f;
                if (p.causal && (kt * 32 + 8 * g + 4 * h + i) > qi) ok = false;
                const float pv = ok ? fast_exp2(s[r] - lse2) : 0.f;
                s[r] = pv * (dp[r] - dl);
            }
        }
        // dQ^T += K^T . dS^T
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const int col = d * 32 + l31;
            const bool colok = col < DH;
            const int colc = colok ? col : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = ks[kappa(r, h) * LD + colc];
                a = colok ? a : 0.f;
                acc_dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], acc_dq[d], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    if (qi < T) {
        float* ob = p.dq + b * p.dq_bs + (long)qi * p.dq_ts + head * DH;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = d * 32 + 8 * g + 4 * h;
                if (col < DH) {
                    float4 o4 = make_float4(acc_dq[d][4 * g] * p.scale, acc_dq[d][4 * g + 1] * p.scale,
                                            acc_dq[d][4 * g + 2] * p.scale, acc_dq[d][4 * g + 3] * p.scale);
                    *reinterpret_cast<float4*>(ob + col) = o4;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// backward dK, dV: key on the lane
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnParams p) {
    TVL_KERNEL_ENTRY();
    constexpr int LD = DH + 4;
    constexpr int NC = DH / 8;
    constexpr int NDB = (DH + 31) / 32;
    __shared__ __attribute__((aligned(16))) float Qs[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float Ds[2][32 * LD];
    __shared__ __attribute__((aligned(16))) float lse_s[2][32];
    __shared__ __attribute__((aligned(16))) float del_s[2][32];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int T = p.T, Tk = p.Tk;
    const int ki = blockIdx.x * 128 + wave * 32 + l31;
    const int krow = ki < Tk ? ki : Tk - 1;

    const float* qb = p.q + b * p.q_bs + head * DH;
    const float* kb = p.k + b * p.k_bs + head * DH;
    const float* vb = p.v + b * p.v_bs + head * DH;
    const float* dob = p.d_o + ((long)b * T) * p.ldo + head * DH;
    const float* lseb = p.lse_in + ((long)b * p.H + head) * T;
    const float* delb = p.delta + ((long)b * p.H + head) * T;
    const bool key_ok = ki < Tk && (!p.key_mask || p.key_mask[(long)b * Tk + ki] != 0);

    float kreg[NC * 4], vreg[NC * 4];
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(kb + (long)krow * p.k_ts + 8 * c + 4 * h);
            kreg[4 * c + 0] = v.x * sc; kreg[4 * c + 1] = v.y * sc; kreg[4 * c + 2] = v.z * sc; kreg[4 * c + 3] = v.w * sc;
            const float4 w = *reinterpret_cast<const float4*>(vb + (long)krow * p.v_ts + 8 * c + 4 * h);
            vreg[4 * c + 0] = w.x; vreg[4 * c + 1] = w.y; vreg[4 * c + 2] = w.z; vreg[4 * c + 3] = w.w;
        }
    }

    f32x16 acc_dk[NDB], acc_dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_dk[d][r] = 0.f; acc_dv[d][r] = 0.f; }

    const int nqt = (T + 31) / 32;
    TileRegs<DH> sq, sd;
    float lse_reg = 0.f, del_reg = 0.f;
    auto gload = [&](int qt) {
        tile_gload<DH>(sq, qb, p.q_ts, qt * 32, T);
        tile_gload<DH>(sd, dob, p.ldo, qt * 32, T);
        if (threadIdx.x < 32) {
            int q = qt * 32 + threadIdx.x;
            q = q < T ? q : T - 1;
            lse_reg = lseb[q] * LOG2E;
            del_reg = delb[q];
        }
    };
    auto sstore = [&](int buf) {
        tile_sstore<DH>(sq, Qs[buf]);
        tile_sstore<DH>(sd, Ds[buf]);
        if (threadIdx.x < 32) { lse_s[buf][threadIdx.x] = lse_reg; del_s[buf][threadIdx.x] = del_reg; }
    };
    gload(0);
    sstore(0);
    __syncthreads();

    for (int qt = 0; qt < nqt; ++qt) {
        const int cur = qt & 1;
        if (qt + 1 < nqt) gload(qt + 1);
        const float* qs = Qs[cur];
        const float* ds = Ds[cur];

        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 qf = *reinterpret_cast<const float4*>(&qs[l31 * LD + 8 * c + 4 * h]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.x, kreg[4 * c + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.y, kreg[4 * c + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.z, kreg[4 * c + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.w, kreg[4 * c + 3], s, 0, 0, 0);
            const float4 df = *reinterpret_cast<const float4*>(&ds[l31 * LD + 8 * c + 4 * h]);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(df.x, vreg[4 * c + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(df.y, vreg[4 * c + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(df.z, vreg[4 * c + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(df.w, vreg[4 * c + 3], dp, 0, 0, 0);
        }
        // rows of S / dP are queries kappa(r,h); P = exp2(S' - lse2[q]); dS = P * (dP - delta[q])
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[cur][8 * g + 4 * h]);
            const float4 d4 = *reinterpret_cast<const float4*>(&del_s[cur][8 * g + 4 * h]);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
            const float dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const int q = qt * 32 + 8 * g + 4 * h + i;
                bool ok = key_ok && q < T;
                if (p.causal && ki > q) ok = false;
                const float pv = ok ? fast_exp2(s[r] - lv[i]) : 0.f;
                s[r] = pv;                       // P
                dp[r] = pv * (dp[r] - dv4[i]);   // dS
            }
        }
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const int col = d * 32 + l31;
            const bool colok = col < DH;
            const int colc = colok ? col : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = ds[kappa(r, h) * LD + colc];
                a = colok ? a : 0.f;
                acc_dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], acc_dv[d], 0, 0, 0);
                float a2 = qs[kappa(r, h) * LD + colc];
                a2 = colok ? a2 : 0.f;
                acc_dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, dp[r], acc_dk[d], 0, 0, 0);
            }
        }
        if (qt + 1 < nqt) sstore(cur ^ 1);
        __syncthreads();
    }

    if (ki < Tk) {
        float* okb = p.dk + b * p.dk_bs + (long)ki * p.dk_ts + head * DH;
        float* ovb = p.dv + b * p.dv_bs + (long)ki * p.dv_ts + head * DH;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = d * 32 + 8 * g + 4 * h;
                if (col < DH) {
                    *reinterpret_cast<float4*>(okb + col) = make_float4(acc_dk[d][4 * g] * p.scale, acc_dk[d][4 * g + 1] * p.scale,
                                                                        acc_dk[d][4 * g + 2] * p.scale, acc_dk[d][4 * g + 3] * p.scale);
                    *reinterpret_cast<float4*>(ovb + col) = make_float4(acc_dv[d][4 * g], acc_dv[d][4 * g + 1], acc_dv[d][4 * g + 2],
                                                                        acc_dv[d][4 * g + 3]);
                }
            }
    }
}

bool strides_ok(const void* ptr, long bs, int ts) { return tvl_aligned16(ptr) && bs % 4 == 0 && ts % 4 == 0; }

}  // namespace

#define TVL_DH_DISPATCH(dh, CALL)              \
    switch (dh) {                              \
        case 8: { constexpr int DH = 8; CALL; } break;   \
        case 16: { constexpr int DH = 16; CALL; } break; \
        case 32: { constexpr int DH = 32; CALL; } break; \
        case 64: { constexpr int DH = 64; CALL; } break; \
        default: tvl_set_error("attention: unsupported head dim %d (8/16/32/64)", dh); return 1; \
    }

extern "C" int tvl_attn_fwd(const tvlAttnFwdArgs* a, tvlStream_t stream) {
    TVL_REQUIRE(a && a->q && a->k && a->v && a->o, "tvl_attn_fwd: null pointer");
    TVL_REQUIRE(a->B > 0 && a->H > 0 && a->T > 0, "tvl_attn_fwd: bad shape");
    TVL_REQUIRE(a->B <= 65535 && a->H <= 65535, "tvl_attn_fwd: B/H exceed grid limits");
    TVL_REQUIRE(strides_ok(a->q, a->q_bs, a->q_ts) && strides_ok(a->k, a->k_bs, a->k_ts) && strides_ok(a->v, a->v_bs, a->v_ts) &&
                    tvl_aligned16(a->o) && a->ldo % 4 == 0,
                "tvl_attn_fwd: operands must be 16-byte aligned with strides divisible by 4");
    TVL_REQUIRE(a->ldo >= a->H * a->dh && a->q_ts >= a->H * a->dh, "tvl_attn_fwd: row stride smaller than H*dh");
    AttnParams p = {};
    p.q = a->q; p.k = a->k; p.v = a->v; p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs;
    p.q_ts = a->q_ts; p.k_ts = a->k_ts; p.v_ts = a->v_ts;
    p.o = a->o; p.ldo = a->ldo; p.lse = a->lse; p.key_mask = a->key_mask;
    p.B = a->B; p.H = a->H; p.T = a->T; p.Tk = a->Tk > 0 ? a->Tk : a->T; p.causal = a->causal; p.scale = a->scale;
    TVL_REQUIRE(!a->causal || p.Tk == p.T, "tvl_attn_fwd: causal needs Tk == T");
    dim3 grid((a->T + 127) / 128, a->H, a->B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((a->dh == 64 || a->dh == 16) && !a->causal && !a->key_mask && p.Tk == p.T && tvl_attn_mode_bf16s()) {
        TVL_REQUIRE(tvl_attn_fwd_bf16s_impl(a, nullptr, s) == 0, "tvl_attn_fwd: split-bf16 launch failed");
        TVL_LAUNCH_CHECK("tvl_attn_fwd(bf16s)");
        return 0;
    }
    TVL_DH_DISPATCH(a->dh, hipLaunchKernelGGL(attn_fwd_kernel<DH>, grid, dim3(256), 0, s, p));
    TVL_LAUNCH_CHECK("tvl_attn_fwd");
    return 0;
}

// TVL_ATTN_MODE=f32 keeps every attention on the exact-fp32 MFMA kernels; default: 3xbf16-split kernels where they apply
int tvl_attn_mode_bf16s(void) {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("TVL_ATTN_MODE");
        mode = (e && e[0] == 'f') ? 0 : 1;
    }
    return mode;
}

extern "C" int tvl_attn_bwd(const tvlAttnBwdArgs* a, tvlStream_t stream) {
    TVL_REQUIRE(a && a->q && a->k && a->v && a->o && a->d_o && a->lse && a->delta && a->dq && a->dk && a->dv,
                "tvl_attn_bwd: null pointer");
    TVL_REQUIRE(a->B > 0 && a->H > 0 && a->T > 0, "tvl_attn_bwd: bad shape");
    TVL_REQUIRE(a->B <= 65535 && a->H <= 65535, "tvl_attn_bwd: B/H exceed grid limits");
    TVL_REQUIRE(strides_ok(a->q, a->q_bs, a->q_ts) && strides_ok(a->k, a->k_bs, a->k_ts) && strides_ok(a->v, a->v_bs, a->v_ts) &&
                    strides_ok(a->dq, a->dq_bs, a->dq_ts) && strides_ok(a->dk, a->dk_bs, a->dk_ts) &&
                    strides_ok(a->dv, a->dv_bs, a->dv_ts) && tvl_aligned16(a->o) && tvl_aligned16(a->d_o) && a->ldo % 4 == 0,
                "tvl_attn_bwd: operands must be 16-byte aligned with strides divisible by 4");
    AttnParams p = {};
    p.q = a->q; p.k = a->k; p.v = a->v; p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs;
    p.q_ts = a->q_ts; p.k_ts = a->k_ts; p.v_ts = a->v_ts;
    p.o_in = a->o; p.d_o = a->d_o; p.ldo = a->ldo; p.lse_in = a->lse; p.delta = a->delta;
    p.dq = a->dq; p.dk = a->dk; p.dv = a->dv; p.dq_bs = a->dq_bs; p.dk_bs = a->dk_bs; p.dv_bs = a->dv_bs;
    p.dq_ts = a->dq_ts; p.dk_ts = a->dk_ts; p.dv_ts = a->dv_ts;
    p.key_mask = a->key_mask; p.B = a->B; p.H = a->H; p.T = a->T; p.Tk = a->Tk > 0 ? a->Tk : a->T; p.causal = a->causal;
    p.scale = a->scale;
    TVL_REQUIRE(!a->causal || p.Tk == p.T, "tvl_attn_bwd: causal needs Tk == T");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {
        const long pairs = (long)a->B * a->T * a->H;
        const long threads = pairs * (a->dh / 4);
        const unsigned grid = (unsigned)((threads + 255) / 256);
        TVL_DH_DISPATCH(a->dh, hipLaunchKernelGGL(attn_delta_kernel<DH>, dim3(grid), dim3(256), 0, s, a->o, a->d_o, a->ldo,
                                                  a->delta, a->B, a->H, a->T));
        TVL_LAUNCH_CHECK("tvl_attn_bwd(delta)");
    }
    if ((a->dh == 64 || a->dh == 16) && !a->causal && !a->key_mask && p.Tk == p.T && tvl_attn_mode_bf16s()) {
        TVL_REQUIRE(tvl_attn_bwd_bf16s_impl(a, nullptr, nullptr, s) == 0, "tvl_attn_bwd: split-bf16 launch failed");
        TVL_LAUNCH_CHECK("tvl_attn_bwd(bf16s)");
        return 0;
    }
    dim3 grid((a->T + 127) / 128, a->H, a->B);
    TVL_DH_DISPATCH(a->dh, hipLaunchKernelGGL(attn_bwd_dq_kernel<DH>, grid, dim3(256), 0, s, p));
    TVL_LAUNCH_CHECK("tvl_attn_bwd(dq)");
    dim3 gridk((p.Tk + 127) / 128, a->H, a->B);
    TVL_DH_DISPATCH(a->dh, hipLaunchKernelGGL(attn_bwd_dkdv_kernel<DH>, gridk, dim3(256), 0, s, p));
    TVL_LAUNCH_CHECK("tvl_attn_bwd(dkdv)");
    return 0;
}

// ---- vision-tower attention (d_h = 64, no masks) between tp3 GEMMs ---------------------------------------------------------------
extern "C" int tvl_attn_fwd_tp3(const tvlAttnFwdArgs* a, void* o_tp3, tvlStream_t stream) {
    TVL_REQUIRE(a && a->q && a->k && a->v && o_tp3, "tvl_attn_fwd_tp3: null pointer");
    TVL_REQUIRE(a->B > 0 && a->H > 0 && a->T > 0 && a->B <= 65535 && a->H <= 65535, "tvl_attn_fwd_tp3: bad shape");
    TVL_REQUIRE(a->dh == 64 && !a->causal && !a->key_mask && (a->Tk == 0 || a->Tk == a->T), "tvl_attn_fwd_tp3: d_h = 64 without masks only");
    TVL_REQUIRE(tvl_attn_mode_bf16s(), "tvl_attn_fwd_tp3: needs the split-bf16 attention kernels (TVL_ATTN_MODE=f32 is set)");
    TVL_REQUIRE(strides_ok(a->q, a->q_bs, a->q_ts) && strides_ok(a->k, a->k_bs, a->k_ts) && strides_ok(a->v, a->v_bs, a->v_ts) && tvl_aligned16(o_tp3) &&
                    (!a->o || (tvl_aligned16(a->o) && a->ldo % 4 == 0 && a->ldo >= a->H * 64)),
                "tvl_attn_fwd_tp3: operands must be 16-byte aligned with strides divisible by 4");
    TVL_REQUIRE(tvl_attn_fwd_bf16s_impl(a, o_tp3, reinterpret_cast<hipStream_t>(stream)) == 0, "tvl_attn_fwd_tp3: launch failed");
    TVL_LAUNCH_CHECK("tvl_attn_fwd_tp3");
    return 0;
}

extern "C" int tvl_attn_bwd_tp3(const tvlAttnBwdArgs* a, const void* o_tp3, void* dqkv_tp3, tvlStream_t stream) {
    TVL_REQUIRE(a && a->q && a->k && a->v && a->d_o && a->lse && a->delta && o_tp3 && dqkv_tp3, "tvl_attn_bwd_tp3: null pointer");
    TVL_REQUIRE(a->B > 0 && a->H > 0 && a->T > 0 && a->B <= 65535 && a->H <= 65535, "tvl_attn_bwd_tp3: bad shape");
    TVL_REQUIRE(a->dh == 64 && !a->causal && !a->key_mask && (a->Tk == 0 || a->Tk == a->T), "tvl_attn_bwd_tp3: d_h = 64 without masks only");
    TVL_REQUIRE(tvl_attn_mode_bf16s(), "tvl_attn_bwd_tp3: needs the split-bf16 attention kernels (TVL_ATTN_MODE=f32 is set)");
    TVL_REQUIRE(strides_ok(a->q, a->q_bs, a->q_ts) && strides_ok(a->k, a->k_bs, a->k_ts) && strides_ok(a->v, a->v_bs, a->v_ts) && tvl_aligned16(a->d_o) &&
                    a->ldo % 4 == 0 && tvl_aligned16(o_tp3) && tvl_aligned16(dqkv_tp3),
                "tvl_attn_bwd_tp3: operands must be 16-byte aligned with strides divisible by 4");
    TVL_REQUIRE(!a->dq && !a->dk && !a->dv, "tvl_attn_bwd_tp3: the gradient leaves as the tp3 image only (dq/dk/dv must be NULL)");
    TVL_REQUIRE(tvl_attn_bwd_bf16s_impl(a, o_tp3, dqkv_tp3, reinterpret_cast<hipStream_t>(stream)) == 0, "tvl_attn_bwd_tp3: launch failed");
    TVL_LAUNCH_CHECK("tvl_attn_bwd_tp3");
    return 0;
}
