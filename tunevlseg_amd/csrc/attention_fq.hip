// Cross-attention with FEW queries and many keys, no masks: the K class embeddings of DenseCLIP's ContextDecoder attending over the 1 + H*W visual tokens
// (reference src/models/components/denseclip/models.py:463-481,520-524: 20 queries x 1601 keys per (sample, head) at 640 x 640).  The flash kernels put queries on
// lanes and walk the keys serially: with 20 queries that is 51 key tiles in a row on B * H = 64 workgroups, 190-200 us per launch.  Here the KEY dimension carries
// the parallelism and the probabilities are materialised (B * H * Tq * Tk floats: 8 MB): five small fp32 kernels, every sum in a fixed order.
//   fq_qk      S[b,h,q,k]  = alpha * <A[b,q,h,:], Bm[b,k,h,:]>            (S = Q K^T * scale;  dP = dO V^T)
//   fq_softmax P = softmax_k(S) in place, lse[b,h,q] (natural log)
//   fq_pk      O[b,q,h,:]  = alpha * sum_k P[b,h,q,k] * Bm[b,k,h,:]      (O = P V;  dQ = scale * dS K)
//   fq_ds      dS = P * (dP - delta[b,h,q]),  delta = <dO[b,q,h,:], O[b,q,h,:]>
//   fq_tk      G[b,k,h,:]  = alpha * sum_q W[b,h,q,k] * A[b,q,h,:]        (dV = P^T dO;  dK = scale * dS^T Q)
// Matrices are [B*T, ld] row-major with head h in columns h*DH ..; Tq <= 32.
#include "common.h"

namespace {

constexpr int FQ_MAXQ = 32;

template <int DH>
__global__ __launch_bounds__(256) void fq_qk_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb, float* __restrict__ S, int H, int Tq,
                                                    int Tk, float alpha) {
    TVL_KERNEL_ENTRY();
    __shared__ __attribute__((aligned(16))) float As[FQ_MAXQ * DH];
    const int b = blockIdx.z, h = blockIdx.y;
    for (int i = threadIdx.x; i < Tq * (DH / 4); i += 256) {
        const int q = i / (DH / 4), c = (i % (DH / 4)) * 4;
        *reinterpret_cast<float4*>(&As[q * DH + c]) = *reinterpret_cast<const float4*>(A + ((long)b * Tq + q) * lda + h * DH + c);
    }
    __syncthreads();
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= Tk) return;
    float4 row[DH / 4];
    const float* src = Bm + ((long)b * Tk + key) * ldb + h * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) row[c] = *reinterpret_cast<const float4*>(src + 4 * c);
    float* out = S + (((long)b * H + h) * Tq) * Tk + key;
    for (int q = 0; q < Tq; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(&As[q * DH + 4 * c]);
            acc = fmaf(a.x, row[c].x, acc); acc = fmaf(a.y, row[c].y, acc); acc = fmaf(a.z, row[c].z, acc); acc = fmaf(a.w, row[c].w, acc);
        }
        out[(long)q * Tk] = acc * alpha;
    }
}

// one workgroup per row (b, h, q): max, sum of exp, normalise; fixed reduction order (wave DPP, then the four waves in order)
__global__ __launch_bounds__(256) void fq_softmax_kernel(float* __restrict__ S, float* __restrict__ lse, int Tk) {
    TVL_KERNEL_ENTRY();
    __shared__ float red[4];
    float* row = S + (long)blockIdx.x * Tk;
    const int w = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int k = threadIdx.x; k < Tk; k += 256) m = fmaxf(m, row[k]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int k = threadIdx.x; k < Tk; k += 256) {
        const float e = expf(row[k] - m);
        row[k] = e;
        s += e;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[w] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / s;
    for (int k = threadIdx.x; k < Tk; k += 256) row[k] *= inv;
    if (threadIdx.x == 0 && lse) lse[blockIdx.x] = m + logf(s);
}

// O[b, q, h, :] = alpha * sum_k P[b, h, q, k] Bm[b, k, h, :].  Workgroup (query group g of QG queries, h, b); 256 threads = DH/4 column quads x KG key groups:
// P tiles [QG x 64 keys] staged in LDS, each thread walks its keys of the tile; the key groups meet in LDS and are added in a fixed order.
// P is addressed as P + b * p_bs + h * p_hs + q * p_qs + k * p_ks and Bm's samples are bm_bs elements apart, so the same kernel serves the score map's
// text-side gradient (dT[b] = dS[b]^T V[b]: the "queries" are the K classes, the "keys" the H*W pixels of dS [B*H*W, K], one P for all column blocks of V).
template <int DH, int QG>
__global__ __launch_bounds__(256) void fq_pk_kernel(const float* __restrict__ P, long p_bs, long p_hs, long p_qs, long p_ks, const float* __restrict__ Bm, long bm_bs, int ldb,
                                                    float* __restrict__ O, int ldo, int H, int Tq, int Tk, float alpha) {
    TVL_KERNEL_ENTRY();
    constexpr int CQ = DH / 4, KG = 256 / CQ;   // 16 column quads x 16 key groups at DH = 64
    __shared__ float Ps[QG][64];
    __shared__ __attribute__((aligned(16))) float part[KG][QG][DH];
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QG;
    const int cq = threadIdx.x % CQ, kg = threadIdx.x / CQ;
    float4 acc[QG];
#pragma unroll
    for (int q = 0; q < QG; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Pb = P + (long)b * p_bs + (long)h * p_hs + (long)q0 * p_qs;
    const float* Bb = Bm + (long)b * bm_bs + h * DH;
    for (int k0 = 0; k0 < Tk; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < QG * 64; i += 256) {
            // (q fastest when P's query stride is the unit one: adjacent threads then read adjacent addresses)
            const int q = p_qs == 1 ? i % QG : i >> 6, kk = p_qs == 1 ? i / QG : i & 63;
            Ps[q][kk] = (q0 + q < Tq && k0 + kk < Tk) ? Pb[(long)q * p_qs + (long)(k0 + kk) * p_ks] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 64 / KG; ++j) {
            const int kk = kg + KG * j, key = k0 + kk;
            if (key < Tk) {
                const float4 v = *reinterpret_cast<const float4*>(Bb + (long)key * ldb + 4 * cq);
#pragma unroll
                for (int q = 0; q < QG; ++q) {
                    const float p = Ps[q][kk];
                    acc[q].x = fmaf(p, v.x, acc[q].x); acc[q].y = fmaf(p, v.y, acc[q].y); acc[q].z = fmaf(p, v.z, acc[q].z); acc[q].w = fmaf(p, v.w, acc[q].w);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QG; ++q) *reinterpret_cast<float4*>(&part[kg][q][4 * cq]) = acc[q];
    __syncthreads();
    for (int i = threadIdx.x; i < QG * DH; i += 256) {
        const int q = i / DH, d = i % DH;
        if (q0 + q >= Tq) continue;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) s += part[g][q][d];
        O[((long)b * Tq + q0 + q) * ldo + h * DH + d] = s * alpha;
    }
}

// dS = P * (dP - delta), delta[b, h, q] = <dO[b, q, h, :], O[b, q, h, :]>; one workgroup per row (b, h, q); dS overwrites dP
template <int DH>
__global__ __launch_bounds__(256) void fq_ds_kernel(const float* __restrict__ P, float* __restrict__ dP, const float* __restrict__ dO, int lddo, const float* __restrict__ O,
                                                    int ldo, int H, int Tq, int Tk) {
    TVL_KERNEL_ENTRY();
    __shared__ float delta_s;
    const int row = blockIdx.x;
    const int q = row % Tq, h = (row / Tq) % H, b = row / (Tq * H);
    if (threadIdx.x < 64) {
        float d = 0.f;
        for (int c = threadIdx.x; c < DH; c += 64) d = fmaf(dO[((long)b * Tq + q) * lddo + h * DH + c], O[((long)b * Tq + q) * ldo + h * DH + c], d);
        d = wave_sum(d);
        if (threadIdx.x == 0) delta_s = d;
    }
    __syncthreads();
    const float delta = delta_s;
    const float* p = P + (long)row * Tk;
    float* g = dP + (long)row * Tk;
    for (int k = threadIdx.x; k < Tk; k += 256) g[k] = p[k] * (g[k] - delta);
}

// G[b, k, h, :] = alpha * sum_q W[b, h, q, k] * A[b, q, h, :]: one key per thread, the Tq rows of A in LDS
template <int DH>
__global__ __launch_bounds__(256) void fq_tk_kernel(const float* __restrict__ Wt, const float* __restrict__ A, int lda, float* __restrict__ G, int ldg, int H, int Tq, int Tk,
                                                    float alpha) {
    TVL_KERNEL_ENTRY();
    __shared__ __attribute__((aligned(16))) float As[FQ_MAXQ * DH];
    const int b = blockIdx.z, h = blockIdx.y;
    for (int i = threadIdx.x; i < Tq * (DH / 4); i += 256) {
        const int q = i / (DH / 4), c = (i % (DH / 4)) * 4;
        *reinterpret_cast<float4*>(&As[q * DH + c]) = *reinterpret_cast<const float4*>(A + ((long)b * Tq + q) * lda + h * DH + c);
    }
    __syncthreads();
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= Tk) return;
    float4 acc[DH / 4];
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* w = Wt + (((long)b * H + h) * Tq) * Tk + key;
    for (int q = 0; q < Tq; ++q) {
        const float wq = w[(long)q * Tk];
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(&As[q * DH + 4 * c]);
            acc[c].x = fmaf(wq, a.x, acc[c].x); acc[c].y = fmaf(wq, a.y, acc[c].y); acc[c].z = fmaf(wq, a.z, acc[c].z); acc[c].w = fmaf(wq, a.w, acc[c].w);
        }
    }
    float* dst = G + ((long)b * Tk + key) * ldg + h * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) *reinterpret_cast<float4*>(dst + 4 * c) = make_float4(acc[c].x * alpha, acc[c].y * alpha, acc[c].z * alpha, acc[c].w * alpha);
}

bool fq_ok(const void* p, int ld, int H, int dh) { return p && tvl_aligned16(p) && ld % 4 == 0 && ld >= H * dh; }

}  // namespace

#define FQ_DH_DISPATCH(dh, STMT)                                                       \
    do {                                                                               \
        if ((dh) == 64) { constexpr int DH = 64; STMT; }                               \
        else if ((dh) == 32) { constexpr int DH = 32; STMT; }                          \
        else if ((dh) == 16) { constexpr int DH = 16; STMT; }                          \
        else { tvl_set_error("few-query attention: d_h must be 16, 32 or 64 (got %d)", (int)(dh)); return 1; } \
    } while (0)

extern "C" int tvl_fq_qk(const float* A, int32_t lda, const float* Bm, int32_t ldb, float* S, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dh, float alpha,
                         tvlStream_t stream) {
    TVL_REQUIRE(S && B > 0 && H > 0 && Tq > 0 && Tq <= FQ_MAXQ && Tk > 0 && B <= 65535 && H <= 65535, "tvl_fq_qk: bad shape (Tq = %d must be <= %d)", Tq, FQ_MAXQ);
    TVL_REQUIRE(fq_ok(A, lda, H, dh) && fq_ok(Bm, ldb, H, dh), "tvl_fq_qk: operands must be 16-byte aligned with row strides divisible by 4 and >= H*dh");
    FQ_DH_DISPATCH(dh, hipLaunchKernelGGL(fq_qk_kernel<DH>, dim3((Tk + 255) / 256, H, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A, lda, Bm, ldb, S, H, Tq, Tk, alpha));
    TVL_LAUNCH_CHECK("tvl_fq_qk");
    return 0;
}

extern "C" int tvl_fq_softmax(float* S, float* lse, int64_t rows, int32_t Tk, tvlStream_t stream) {
    TVL_REQUIRE(S && rows > 0 && rows < (1ll << 31) && Tk > 0, "tvl_fq_softmax: bad shape");
    hipLaunchKernelGGL(fq_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, lse, Tk);
    TVL_LAUNCH_CHECK("tvl_fq_softmax");
    return 0;
}

extern "C" int tvl_fq_pk(const float* P, int64_t p_bs, int64_t p_hs, int64_t p_qs, int64_t p_ks, const float* Bm, int64_t bm_bs, int32_t ldb, float* O, int32_t ldo, int32_t B,
                         int32_t H, int32_t Tq, int32_t Tk, int32_t dh, float alpha, tvlStream_t stream) {
    TVL_REQUIRE(P && B > 0 && H > 0 && Tq > 0 && Tq <= FQ_MAXQ && Tk > 0 && B <= 65535 && H <= 65535, "tvl_fq_pk: bad shape (Tq = %d must be <= %d)", Tq, FQ_MAXQ);
    TVL_REQUIRE(fq_ok(Bm, ldb, H, dh) && fq_ok(O, ldo, H, dh) && bm_bs % 4 == 0, "tvl_fq_pk: operands must be 16-byte aligned with row / sample strides divisible by 4 and rows >= H*dh");
    constexpr int QG = 4;
    FQ_DH_DISPATCH(dh, hipLaunchKernelGGL((fq_pk_kernel<DH, QG>), dim3((Tq + QG - 1) / QG, H, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, (long)p_bs, (long)p_hs, (long)p_qs,
                                          (long)p_ks, Bm, (long)bm_bs, ldb, O, ldo, H, Tq, Tk, alpha));
    TVL_LAUNCH_CHECK("tvl_fq_pk");
    return 0;
}

extern "C" int tvl_fq_ds(const float* P, float* dP, const float* dO, int32_t lddo, const float* O, int32_t ldo, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dh,
                         tvlStream_t stream) {
    TVL_REQUIRE(P && dP && dO && O && B > 0 && H > 0 && Tq > 0 && Tk > 0 && (long)B * H * Tq < (1ll << 31), "tvl_fq_ds: bad shape");
    TVL_REQUIRE(lddo >= H * dh && ldo >= H * dh, "tvl_fq_ds: row stride smaller than H*dh");
    FQ_DH_DISPATCH(dh, hipLaunchKernelGGL(fq_ds_kernel<DH>, dim3((unsigned)((long)B * H * Tq)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, dP, dO, lddo, O, ldo, H, Tq, Tk));
    TVL_LAUNCH_CHECK("tvl_fq_ds");
    return 0;
}

extern "C" int tvl_fq_tk(const float* W, const float* A, int32_t lda, float* G, int32_t ldg, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dh, float alpha,
                         tvlStream_t stream) {
    TVL_REQUIRE(W && B > 0 && H > 0 && Tq > 0 && Tq <= FQ_MAXQ && Tk > 0 && B <= 65535 && H <= 65535, "tvl_fq_tk: bad shape (Tq = %d must be <= %d)", Tq, FQ_MAXQ);
    TVL_REQUIRE(fq_ok(A, lda, H, dh) && fq_ok(G, ldg, H, dh), "tvl_fq_tk: operands must be 16-byte aligned with row strides divisible by 4 and >= H*dh");
    FQ_DH_DISPATCH(dh, hipLaunchKernelGGL(fq_tk_kernel<DH>, dim3((Tk + 255) / 256, H, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, A, lda, G, ldg, H, Tq, Tk, alpha));
    TVL_LAUNCH_CHECK("tvl_fq_tk");
    return 0;
}
