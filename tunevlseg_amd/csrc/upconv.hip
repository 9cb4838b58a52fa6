// The reference's optional "new last layer" (base_clipseg.py:58-71):
//     Upsample(scale=ps, bilinear, align_corners=False) -> Conv2d(C -> 1, k x k, padding=same, replicate)
// Both stages are linear and the upsample acts per channel, so the channel contraction is done
// FIRST (a [B*G*G, C] x [C, k*k] GEMM by the caller -> "taps"), and this file evaluates
//     out[b,y,x] = bias + sum_{ky,kx} bilinear(taps[b,:,:,ky,kx]) at (clamp(y+ky-pl), clamp(x+kx-pl))
// directly on the G x G maps: the C x (G*ps)^2 upsampled tensor (31.7 MB/img at C=64, 352^2) is
// never materialised.  The backward uses the separability of (shift+clamp+bilinear) in y and x.
#include "common.h"

namespace {

struct Lerp { int i0, i1; float w0, w1; };

// torch area_pixel_compute_source_index(align_corners=False) + upsample_bilinear2d index rule
__device__ __forceinline__ Lerp lerp_of(int dst, int G, float inv_scale) {
    float src = inv_scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 < G - 1 ? i0 : G - 1;
    const int i1 = i0 < G - 1 ? i0 + 1 : i0;
    float l1 = src - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    return {i0, i1, 1.0f - l1, l1};
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

constexpr int KMAX = 7;

__global__ __launch_bounds__(256) void upconv_fwd_kernel(const float* __restrict__ taps, int ldg, const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int G, int ps, int k) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const int pl = (k - 1) / 2;
    const float inv_scale = 1.0f / (float)ps;
    const long total = (long)B * S * S;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % S), y = (int)((idx / S) % S), b = (int)(idx / ((long)S * S));
        Lerp lx[KMAX], ly[KMAX];
#pragma unroll
        for (int t = 0; t < KMAX; ++t)
            if (t < k) {
                lx[t] = lerp_of(clampi(x + t - pl, 0, S - 1), G, inv_scale);
                ly[t] = lerp_of(clampi(y + t - pl, 0, S - 1), G, inv_scale);
            }
        const float* tb = taps + (long)b * G * G * ldg;
        float acc = bias ? bias[0] : 0.f;
#pragma unroll
        for (int ky = 0; ky < KMAX; ++ky)
            if (ky < k) {
#pragma unroll
                for (int kx = 0; kx < KMAX; ++kx)
                    if (kx < k) {
                        const int tap = ky * k + kx;
                        const float v00 = tb[((long)ly[ky].i0 * G + lx[kx].i0) * ldg + tap];
                        const float v01 = tb[((long)ly[ky].i0 * G + lx[kx].i1) * ldg + tap];
                        const float v10 = tb[((long)ly[ky].i1 * G + lx[kx].i0) * ldg + tap];
                        const float v11 = tb[((long)ly[ky].i1 * G + lx[kx].i1) * ldg + tap];
                        acc += ly[ky].w0 * (lx[kx].w0 * v00 + lx[kx].w1 * v01) + ly[ky].w1 * (lx[kx].w0 * v10 + lx[kx].w1 * v11);
                    }
            }
        out[idx] = acc;
    }
}

// The same sum in two separable passes (work != null): the x pass folds the k column taps and the bilinear x weights once per
// (source row i, ky, x) -- B * G * k * S values -- and the y pass reads 2 k of them per output pixel, coalesced along x: 10 + 10 loads
// per pixel row / pixel instead of 100 gathered loads per pixel (179 -> ~40 us at B = 32, 352^2).
//   R[b][i][ky][x] = sum_kx sum_j wx(clamp(x + kx - pl), j) taps[b, i, j, ky * k + kx]
__global__ __launch_bounds__(256) void upconv_fwd_x_kernel(const float* __restrict__ taps, int ldg, float* __restrict__ work, int B, int G, int ps, int k) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const int pl = (k - 1) / 2;
    const float inv_scale = 1.0f / (float)ps;
    const long total = (long)B * G * k * S;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % S), ky = (int)((idx / S) % k), i = (int)((idx / ((long)S * k)) % G), b = (int)(idx / ((long)S * k * G));
        const float* tb = taps + ((long)b * G + i) * G * ldg + ky * k;
        float acc = 0.f;
        for (int kx = 0; kx < k; ++kx) {
            const Lerp l = lerp_of(clampi(x + kx - pl, 0, S - 1), G, inv_scale);
            acc += l.w0 * tb[(long)l.i0 * ldg + kx] + l.w1 * tb[(long)l.i1 * ldg + kx];
        }
        work[idx] = acc;
    }
}
//   out[b][y][x] = bias + sum_ky sum_i wy(clamp(y + ky - pl), i) R[b][i][ky][x]
__global__ __launch_bounds__(256) void upconv_fwd_y_kernel(const float* __restrict__ work, const float* __restrict__ bias, float* __restrict__ out,
                                                           int B, int G, int ps, int k) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const int pl = (k - 1) / 2;
    const float inv_scale = 1.0f / (float)ps;
    const long total = (long)B * S * S;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % S), y = (int)((idx / S) % S), b = (int)(idx / ((long)S * S));
        const float* wb = work + (long)b * G * k * S + x;
        float acc = bias ? bias[0] : 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const Lerp l = lerp_of(clampi(y + ky - pl, 0, S - 1), G, inv_scale);
            acc += l.w0 * wb[((long)l.i0 * k + ky) * S] + l.w1 * wb[((long)l.i1 * k + ky) * S];
        }
        out[idx] = acc;
    }
}

// weight that U-coordinate X (already clamped) puts on source cell j
__device__ __forceinline__ float cell_weight(int X, int j, int G, float inv_scale) {
    const Lerp l = lerp_of(X, G, inv_scale);
    return (l.i0 == j ? l.w0 : 0.f) + (l.i1 == j ? l.w1 : 0.f);
}

// pass 1: work[b][kx][y][j] = sum_x dout[b][y][x] * wx(clamp(x+kx-pl), j)
__global__ __launch_bounds__(256) void upconv_bwd_x_kernel(const float* __restrict__ dout, float* __restrict__ work, int B, int G, int ps, int k) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const int pl = (k - 1) / 2;
    const float inv_scale = 1.0f / (float)ps;
    const long total = (long)B * k * S * G;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % G), y = (int)((idx / G) % S), kx = (int)((idx / ((long)G * S)) % k), b = (int)(idx / ((long)G * S * k));
        const int xlo = clampi(ps * j - ps / 2 - 1 - k, 0, S - 1), xhi = clampi(ps * j + ps + ps / 2 + k, 0, S - 1);
        const float* row = dout + ((long)b * S + y) * S;
        float acc = 0.f;
        for (int x = xlo; x <= xhi; ++x) acc += row[x] * cell_weight(clampi(x + kx - pl, 0, S - 1), j, G, inv_scale);
        work[idx] = acc;
    }
}
// pass 2: dtaps[b][i][j][ky*k+kx] = sum_y wy(clamp(y+ky-pl), i) * work[b][kx][y][j]
__global__ __launch_bounds__(256) void upconv_bwd_y_kernel(const float* __restrict__ work, float* __restrict__ dtaps, int ldg, int B, int G, int ps, int k) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const int pl = (k - 1) / 2;
    const float inv_scale = 1.0f / (float)ps;
    const long total = (long)B * G * G * k * k;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int tap = (int)(idx % (k * k));
        const int kx = tap % k, ky = tap / k;
        const long cell = idx / (k * k);
        const int j = (int)(cell % G), i = (int)((cell / G) % G), b = (int)(cell / ((long)G * G));
        const int ylo = clampi(ps * i - ps / 2 - 1 - k, 0, S - 1), yhi = clampi(ps * i + ps + ps / 2 + k, 0, S - 1);
        const float* wb = work + (((long)b * k + kx) * S) * G + j;
        float acc = 0.f;
        for (int y = ylo; y <= yhi; ++y) acc += wb[(long)y * G] * cell_weight(clampi(y + ky - pl, 0, S - 1), i, G, inv_scale);
        dtaps[cell * ldg + tap] = acc;
    }
}

}  // namespace

extern "C" int tvl_upconv_taps_fwd(const float* taps, int32_t ldg, const float* bias, float* out, float* work, int32_t B, int32_t G, int32_t ps,
                                   int32_t k, tvlStream_t stream) {
    TVL_REQUIRE(taps && out && B > 0 && G > 0 && ps > 0, "tvl_upconv_taps_fwd: bad arguments");
    TVL_REQUIRE(k >= 1 && k <= KMAX && ldg >= k * k, "tvl_upconv_taps_fwd: kernel size %d unsupported (1..%d) or ldg too small", k, KMAX);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)B * G * ps * G * ps;
    long grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    if (work) {   // two separable passes through work [B, G, k, G * ps]
        const long tx = (long)B * G * k * G * ps;
        long gx = (tx + 255) / 256;
        if (gx > 16384) gx = 16384;
        hipLaunchKernelGGL(upconv_fwd_x_kernel, dim3((unsigned)gx), dim3(256), 0, s, taps, ldg, work, B, G, ps, k);
        TVL_LAUNCH_CHECK("tvl_upconv_taps_fwd(x)");
        hipLaunchKernelGGL(upconv_fwd_y_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const float*)work, bias, out, B, G, ps, k);
    } else {
        hipLaunchKernelGGL(upconv_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, s, taps, ldg, bias, out, B, G, ps, k);
    }
    TVL_LAUNCH_CHECK("tvl_upconv_taps_fwd");
    return 0;
}

extern "C" int tvl_upconv_taps_bwd(const float* dout, float* dtaps, int32_t ldg, float* work, int32_t B, int32_t G, int32_t ps, int32_t k,
                                   tvlStream_t stream) {
    TVL_REQUIRE(dout && dtaps && work && B > 0 && G > 0 && ps > 0, "tvl_upconv_taps_bwd: bad arguments");
    TVL_REQUIRE(k >= 1 && k <= KMAX && ldg >= k * k, "tvl_upconv_taps_bwd: kernel size %d unsupported (1..%d) or ldg too small", k, KMAX);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {
        const long total = (long)B * k * G * ps * G;
        long grid = (total + 255) / 256;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(upconv_bwd_x_kernel, dim3((unsigned)grid), dim3(256), 0, s, dout, work, B, G, ps, k);
        TVL_LAUNCH_CHECK("tvl_upconv_taps_bwd(x)");
    }
    {
        const long total = (long)B * G * G * k * k;
        long grid = (total + 255) / 256;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(upconv_bwd_y_kernel, dim3((unsigned)grid), dim3(256), 0, s, work, dtaps, ldg, B, G, ps, k);
        TVL_LAUNCH_CHECK("tvl_upconv_taps_bwd(y)");
    }
    return 0;
}
