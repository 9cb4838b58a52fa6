// LayerNorm forward / backward over the last dimension, one wave64 per row, values held in
// registers (float4 per lane), two-pass variance like torch.  HBM-bound: each row is read once
// and written once (fwd), or dy/x read once and dx written once (bwd).
#include "common.h"

namespace {

// LN_MAXV float4 per lane held in registers: 4 -> cols <= 1024, 8 -> cols <= 2048 (the CRIS decoder's LayerNorm(2048))

template <bool VEC, int LN_MAXV = 4>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    const float inv_n = 1.0f / (float)cols;
    if (VEC) {
        const int nv = cols >> 2;  // float4 count
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[i] = reinterpret_cast<const float4*>(xr)[c];
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (beta) b = reinterpret_cast<const float4*>(beta)[c];
                float4 o;
                o.x = (v[i].x - mean) * rstd * g.x + b.x;
                o.y = (v[i].y - mean) * rstd * g.y + b.y;
                o.z = (v[i].z - mean) * rstd * g.z + b.z;
                o.w = (v[i].w - mean) * rstd * g.w + b.w;
                reinterpret_cast<float4*>(yr)[c] = o;
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    } else {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += xr[c];
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float d = xr[c] - mean;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mean) * rstd * gamma[c] + (beta ? beta[c] : 0.f);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma, xhat = (x-mean)*rstd ; [+ dres]
template <bool VEC, int LN_MAXV = 4>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    const float* dyr = dy + row * cols;
    float* dxr = dx + row * cols;
    const float* rr = dres ? dres + row * cols : nullptr;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const float inv_n = 1.0f / (float)cols;
    if (VEC) {
        const int nv = cols >> 2;
        float4 g[LN_MAXV], xh[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 d = reinterpret_cast<const float4*>(dyr)[c];
                const float4 xv = reinterpret_cast<const float4*>(xr)[c];
                const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
                xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                g[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                if (dgamma) {
                    atomicAdd(&dgamma[4 * c + 0], d.x * xh[i].x); atomicAdd(&dgamma[4 * c + 1], d.y * xh[i].y);
                    atomicAdd(&dgamma[4 * c + 2], d.z * xh[i].z); atomicAdd(&dgamma[4 * c + 3], d.w * xh[i].w);
                }
                if (dbeta) {
                    atomicAdd(&dbeta[4 * c + 0], d.x); atomicAdd(&dbeta[4 * c + 1], d.y);
                    atomicAdd(&dbeta[4 * c + 2], d.z); atomicAdd(&dbeta[4 * c + 3], d.w);
                }
            }
        }
        const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o;
                o.x = rstd * (g[i].x - m1 - xh[i].x * m2);
                o.y = rstd * (g[i].y - m1 - xh[i].y * m2);
                o.z = rstd * (g[i].z - m1 - xh[i].z * m2);
                o.w = rstd * (g[i].w - m1 - xh[i].w * m2);
                if (rr) {
                    const float4 r4 = reinterpret_cast<const float4*>(rr)[c];
                    o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w;
                }
                reinterpret_cast<float4*>(dxr)[c] = o;
            }
        }
    } else {
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = dyr[c] * gamma[c];
            s1 += gg;
            s2 += gg * xh;
            if (dgamma) atomicAdd(&dgamma[c], dyr[c] * xh);
            if (dbeta) atomicAdd(&dbeta[c], dyr[c]);
        }
        const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = dyr[c] * gamma[c];
            float o = rstd * (gg - m1 - xh * m2);
            if (rr) o += rr[c];
            dxr[c] = o;
        }
    }
}

}  // namespace

extern "C" int tvl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int64_t rows, int32_t cols, float eps, tvlStream_t stream) {
    TVL_REQUIRE(x && gamma && y, "tvl_layernorm_fwd: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0, "tvl_layernorm_fwd: bad shape rows=%ld cols=%d", (long)rows, cols);
    const bool vec = (cols % 4 == 0) && cols <= 256 * 8 && tvl_aligned16(x) && tvl_aligned16(y) && tvl_aligned16(gamma) &&
                     (!beta || tvl_aligned16(beta));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec && cols <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<true, 4>), dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps);
    else if (vec) hipLaunchKernelGGL((ln_fwd_kernel<true, 8>), dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps);
    else hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps);
    TVL_LAUNCH_CHECK("tvl_layernorm_fwd");
    return 0;
}

extern "C" int tvl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 const float* dres, float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t cols,
                                 tvlStream_t stream) {
    TVL_REQUIRE(dy && x && gamma && mean && rstd && dx, "tvl_layernorm_bwd: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0, "tvl_layernorm_bwd: bad shape rows=%ld cols=%d", (long)rows, cols);
    const bool vec = (cols % 4 == 0) && cols <= 256 * 8 && tvl_aligned16(x) && tvl_aligned16(dy) && tvl_aligned16(dx) &&
                     tvl_aligned16(gamma) && (!dres || tvl_aligned16(dres));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec && cols <= 1024) hipLaunchKernelGGL((ln_bwd_kernel<true, 4>), dim3(grid), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols);
    else if (vec) hipLaunchKernelGGL((ln_bwd_kernel<true, 8>), dim3(grid), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols);
    else hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols);
    TVL_LAUNCH_CHECK("tvl_layernorm_bwd");
    return 0;
}
