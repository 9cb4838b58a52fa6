// DenseCLIP (BASELINE configs[4]) pieces that are not GEMMs or attention: the ViT-B/16 FPN taps of the reference's CLIPVisionTransformer
// (src/models/components/denseclip/models.py:581-600,693-701) over NHWC pixel matrices, and the gamma-scaled residual of
// DenseCLIP.after_extract_feat (denseclip.py:157).  All HBM-bound, one pass each.
#include "common.h"

namespace {

inline int nblk(long n, int per = 256) {
    long b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

// ---- GroupNorm(num_groups = 1): one mean / variance per SAMPLE over all (pixel, channel) elements ---------------------------------
// stage 1: grid (P, B); workgroup p of sample b reduces its share of the rows to (sum, sum of squares) in double -> work[(b*P + p)*2 ..]
template <int V>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, long bs, int ldx, int rows, int C, double* __restrict__ work, int P) {
    TVL_KERNEL_ENTRY();
    __shared__ double red[2][4];
    const int b = blockIdx.y, p = blockIdx.x;
    const int per = (rows + P - 1) / P;
    const int r0 = p * per, r1 = min(rows, r0 + per);
    const int CV = C / V;
    const float* xb = x + (long)b * bs;
    double s = 0.0, ss = 0.0;
    const long total = (long)max(0, r1 - r0) * CV;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int c = (int)(i % CV) * V;
        const int r = r0 + (int)(i / CV);
        const float* src = xb + (long)r * ldx + c;
        if constexpr (V == 4) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            // the four products in float are exact enough only as doubles: the sum of squares meets mean^2 in a cancellation
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        } else {
            const double v = src[0];
            s += v;
            ss += v * v;
        }
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s; red[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        work[((long)b * P + p) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        work[((long)b * P + p) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
// stage 2: one wave per sample adds the P partials in a fixed order -> stats[b] = (mean, rstd)
__global__ __launch_bounds__(64) void gn_finalize_kernel(const double* __restrict__ work, int P, double count, float eps, float* __restrict__ stats) {
    TVL_KERNEL_ENTRY();
    const int b = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int p = lane; p < P; p += 64) { s += work[((long)b * P + p) * 2]; ss += work[((long)b * P + p) * 2 + 1]; }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    if (lane == 0) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        stats[2 * b] = (float)mean;
        stats[2 * b + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
// y[b, pixel, c] = (x - mean_b) * rstd_b * gamma_c + beta_c ; POOL = 2: followed by MaxPool2d(2, 2) (models.py:598-600) in the same pass
template <int V, int POOL>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, long bs, int ldx, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                       int B, int H, int W, int C) {
    TVL_KERNEL_ENTRY();
    const int Ho = H / POOL, Wo = W / POOL, CV = C / V;
    const long total = (long)B * Ho * Wo * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        const long r = i / CV;
        const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
        const float mean = stats[2 * b], rstd = stats[2 * b + 1];
        float sc[V], sh[V], out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            sc[e] = rstd * gamma[c + e];
            sh[e] = beta[c + e] - mean * sc[e];
            out[e] = -INFINITY;
        }
#pragma unroll
        for (int dy = 0; dy < POOL; ++dy)
#pragma unroll
            for (int dx = 0; dx < POOL; ++dx) {
                const float* src = x + (long)b * bs + ((long)(oy * POOL + dy) * W + ox * POOL + dx) * ldx + c;
                float v[V];
                if constexpr (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
                    v[0] = src[0];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) out[e] = fmaxf(out[e], fmaf(v[e], sc[e], sh[e]));
            }
        if constexpr (V == 4) *reinterpret_cast<float4*>(y + r * ldy + c) = make_float4(out[0], out[1], out[2], out[3]);
        else y[r * ldy + c] = out[0];
    }
}

// ---- ConvTranspose2d(k = 2, s = 2) tail: the GEMM's [pixels, (dy, dx, co)] output back to a raster NHWC map ----------------------------
// in  : [B*H*W * 4^(L-1), 4*C] contiguous = flat (((b*H + y)*W + x) * 4^L + sub) * C + c, sub = (dy1*2 + dx1) [* 4 + (dy2*2 + dx2)]
// out : [B, H*2^L, W*2^L, C] with row stride ldo; pixel (y*2^L + (dy1 [*2 + dy2]), x*2^L + (dx1 [*2 + dx2]))
template <int V>
__global__ __launch_bounds__(256) void tconv_unshuffle_kernel(const float* __restrict__ in, float* __restrict__ out, int ldo, int B, int H, int W, int C, int L) {
    TVL_KERNEL_ENTRY();
    const int CV = C / V, S = 1 << (2 * L), F = 1 << L;
    const long total = (long)B * H * W * S * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        long r = i / CV;
        const int sub = (int)(r % S);
        r /= S;
        const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((long)W * H));
        int oy, ox;
        if (L == 1) {
            oy = sub >> 1; ox = sub & 1;
        } else {
            const int s1 = sub >> 2, s2 = sub & 3;
            oy = (s1 >> 1) * 2 + (s2 >> 1);
            ox = (s1 & 1) * 2 + (s2 & 1);
        }
        const long orow = ((long)b * H * F + (long)y * F + oy) * ((long)W * F) + (long)x * F + ox;
        if constexpr (V == 4) *reinterpret_cast<float4*>(out + orow * ldo + c) = *reinterpret_cast<const float4*>(in + (i / CV) * C + c);
        else out[orow * ldo + c] = in[(i / CV) * C + c];
    }
}

// ---- out = a + g[c] * b  (text_embeddings + gamma * text_diff, denseclip.py:157) ---------------------------------------------------
__global__ void colscale_add_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g, float* __restrict__ out, long rows, int cols) {
    TVL_KERNEL_ENTRY();
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) out[i] = fmaf(g[i % cols], b[i], a[i]);
}
// db = d * g[c] ; dg[c] = sum_r d[r, c] * b[r, c].  A workgroup owns 64 columns; its 16 row groups take every 16th row and meet in LDS, summed in a fixed
// order (bitwise reproducible).  (One thread per column walking all rows took 153 us for 320 x 512.)
__global__ __launch_bounds__(1024) void colscale_bwd_kernel(const float* __restrict__ d, const float* __restrict__ b, const float* __restrict__ g, float* __restrict__ db,
                                                            float* __restrict__ dg, long rows, int cols) {
    TVL_KERNEL_ENTRY();
    __shared__ float part[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < cols) {
        const float gc = g[c];
        for (long r = rg; r < rows; r += 16) {
            const float dv = d[r * cols + c];
            if (db) db[r * cols + c] = dv * gc;
            acc = fmaf(dv, b[r * cols + c], acc);
        }
    }
    part[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < cols && dg) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += part[i][cl];
        dg[c] = s;
    }
}

// out[b * rows + i, k] = full[b * T + skip + i, b * K + k]: the diagonal blocks of ONE [B*T, B*K] GEMM (every sample's pixels against every sample's class
// vectors) are the per-sample score maps; one launch instead of B skinny GEMMs of M = H*W rows each
__global__ void blockdiag_gather_kernel(const float* __restrict__ full, int ld, float* __restrict__ out, int B, int T, int skip, int rows, int K) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * rows * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long r = i / K;
        const int b = (int)(r / rows), px = (int)(r % rows);
        out[i] = full[((long)b * T + skip + px) * ld + (long)b * K + k];
    }
}

}  // namespace

extern "C" int tvl_blockdiag_gather(const float* full, int32_t ld, float* out, int32_t B, int32_t T, int32_t skip, int32_t rows, int32_t K, tvlStream_t stream) {
    TVL_REQUIRE(full && out && B > 0 && T > 0 && skip >= 0 && rows > 0 && skip + rows <= T && K > 0 && ld >= B * K, "tvl_blockdiag_gather: bad shape");
    hipLaunchKernelGGL(blockdiag_gather_kernel, dim3(nblk((long)B * rows * K)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), full, ld, out, B, T, skip, rows, K);
    TVL_LAUNCH_CHECK("tvl_blockdiag_gather");
    return 0;
}

extern "C" int64_t tvl_groupnorm_work_doubles(int32_t B, int32_t rows) {
    const int P = rows < 256 ? 1 : (rows / 64 > 256 ? 256 : rows / 64);
    return (int64_t)B * P * 2;
}

extern "C" int tvl_groupnorm_stats(const float* x, int64_t batch_stride, int32_t ldx, int32_t B, int32_t rows, int32_t C, float eps, float* stats,
                                   double* work, tvlStream_t stream) {
    TVL_REQUIRE(x && stats && work, "tvl_groupnorm_stats: null pointer");
    TVL_REQUIRE(B > 0 && rows > 0 && C > 0 && ldx >= C && batch_stride >= (int64_t)rows * ldx, "tvl_groupnorm_stats: bad shape B=%d rows=%d C=%d ldx=%d", B, rows, C, ldx);
    const int P = (int)(tvl_groupnorm_work_doubles(1, rows) / 2);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (C % 4 == 0 && ldx % 4 == 0 && batch_stride % 4 == 0 && tvl_aligned16(x))
        hipLaunchKernelGGL(gn_partial_kernel<4>, dim3(P, B), dim3(256), 0, s, x, (long)batch_stride, ldx, rows, C, work, P);
    else
        hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(P, B), dim3(256), 0, s, x, (long)batch_stride, ldx, rows, C, work, P);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, s, work, P, (double)rows * C, eps, stats);
    TVL_LAUNCH_CHECK("tvl_groupnorm_stats");
    return 0;
}

extern "C" int tvl_groupnorm_apply(const float* x, int64_t batch_stride, int32_t ldx, const float* stats, const float* gamma, const float* beta, float* y,
                                   int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pool, tvlStream_t stream) {
    TVL_REQUIRE(x && stats && gamma && beta && y, "tvl_groupnorm_apply: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C, "tvl_groupnorm_apply: bad shape");
    TVL_REQUIRE(pool == 1 || (pool == 2 && H % 2 == 0 && W % 2 == 0), "tvl_groupnorm_apply: pool must be 1, or 2 with even H, W (got %d, %dx%d)", pool, H, W);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && batch_stride % 4 == 0 && tvl_aligned16(x) && tvl_aligned16(y) && tvl_aligned16(gamma) && tvl_aligned16(beta);
    const long total = (long)B * (H / pool) * (W / pool) * (v4 ? C / 4 : C);
#define TVL_GN_LAUNCH(V, POOL) \
    hipLaunchKernelGGL((gn_apply_kernel<V, POOL>), dim3(nblk(total)), dim3(256), 0, s, x, (long)batch_stride, ldx, stats, gamma, beta, y, ldy, B, H, W, C)
    if (v4 && pool == 1) TVL_GN_LAUNCH(4, 1);
    else if (v4) TVL_GN_LAUNCH(4, 2);
    else if (pool == 1) TVL_GN_LAUNCH(1, 1);
    else TVL_GN_LAUNCH(1, 2);
#undef TVL_GN_LAUNCH
    TVL_LAUNCH_CHECK("tvl_groupnorm_apply");
    return 0;
}

extern "C" int tvl_tconv2x2_unshuffle(const float* in, float* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t C, int32_t levels, tvlStream_t stream) {
    TVL_REQUIRE(in && out, "tvl_tconv2x2_unshuffle: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && ldo >= C && (levels == 1 || levels == 2), "tvl_tconv2x2_unshuffle: bad shape (levels=%d)", levels);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool v4 = C % 4 == 0 && ldo % 4 == 0 && tvl_aligned16(in) && tvl_aligned16(out);
    const long total = (long)B * H * W * (1 << (2 * levels)) * (v4 ? C / 4 : C);
    if (v4) hipLaunchKernelGGL(tconv_unshuffle_kernel<4>, dim3(nblk(total)), dim3(256), 0, s, in, out, ldo, B, H, W, C, levels);
    else hipLaunchKernelGGL(tconv_unshuffle_kernel<1>, dim3(nblk(total)), dim3(256), 0, s, in, out, ldo, B, H, W, C, levels);
    TVL_LAUNCH_CHECK("tvl_tconv2x2_unshuffle");
    return 0;
}

extern "C" int tvl_colscale_add(const float* a, const float* b, const float* g, float* out, int64_t rows, int32_t cols, tvlStream_t stream) {
    TVL_REQUIRE(a && b && g && out && rows > 0 && cols > 0, "tvl_colscale_add: bad arguments");
    hipLaunchKernelGGL(colscale_add_kernel, dim3(nblk(rows * cols)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, g, out, (long)rows, cols);
    TVL_LAUNCH_CHECK("tvl_colscale_add");
    return 0;
}

extern "C" int tvl_colscale_bwd(const float* d, const float* b, const float* g, float* db, float* dg, int64_t rows, int32_t cols, tvlStream_t stream) {
    TVL_REQUIRE(d && b && g && (db || dg) && rows > 0 && cols > 0, "tvl_colscale_bwd: bad arguments");
    hipLaunchKernelGGL(colscale_bwd_kernel, dim3((cols + 63) / 64), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), d, b, g, db, dg, (long)rows, cols);
    TVL_LAUNCH_CHECK("tvl_colscale_bwd");
    return 0;
}
