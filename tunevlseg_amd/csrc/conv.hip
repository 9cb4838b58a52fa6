// CRIS conv path (reference cris_model/clip.py:18-274, layers.py:15-119,359-445, coop_cris.py:235): the HBM-bound pieces
// around the GEMMs.  Feature maps are NHWC "pixel matrices": element (b, y, x, c) at ptr[((b*H + y)*W + x)*ld + c], so a
// 1x1 conv is a plain GEMM over the map, a 3x3 conv a GEMM over the im2col matrix built here, and channel concatenation
// is a column offset into a wider matrix (ld > C).  Every kernel is one coalesced pass (channels on consecutive lanes).
#include "common.h"
#include "tp3.h"

namespace {

inline unsigned nblk(long n, int per = 256) {
    const long b = (n + per - 1) / per;
    return (unsigned)(b < 1 ? 1 : (b > 1048576 ? 1048576 : b));
}

// ---- im2col for 3x3 / pad 1 / stride s ------------------------------------------------------------------------------
// cols[(b,oy,ox), (ky*3+kx)*C + c] = x[b, oy*s+ky-1, ox*s+kx-1, c]   (0 outside; columns 9C..ldc-1 are zero-filled)
template <bool VEC>
__global__ void im2col3x3_kernel(const float* __restrict__ x, long sb, long sy, long sx, long sc, float* __restrict__ cols, int ldc,
                                 int B, int H, int W, int C, int stride, int Ho, int Wo) {
    TVL_KERNEL_ENTRY();
    if (VEC) {  // NHWC, C % 4 == 0: one float4 of channels per thread
        const int c4 = C >> 2;
        const long total = (long)B * Ho * Wo * 9 * c4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)(i % c4) * 4;
            long r = i / c4;
            const int tap = (int)(r % 9);
            r /= 9;
            const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
            const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(x + b * sb + iy * sy + ix * sx + c);
            *reinterpret_cast<float4*>(cols + r * ldc + tap * C + c) = v;
        }
    } else {
        const long total = (long)B * Ho * Wo * ldc;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int col = (int)(i % ldc);
            const long r = i / ldc;
            float v = 0.f;
            if (col < 9 * C) {
                const int tap = col / C, c = col % C;
                const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
                const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[b * sb + iy * sy + ix * sx + c * sc];
            }
            cols[i] = v;
        }
    }
}

// ---- nn.AvgPool2d(k) -------------------------------------------------------------------------------------------------
// V = channels per thread (4: float4 over the channel dimension of the NHWC map; 1: any C / alignment)
template <int V>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int B, int H, int W, int C, int k) {
    TVL_KERNEL_ENTRY();
    const int Ho = H / k, Wo = W / k, CV = C / V;
    const long total = (long)B * Ho * Wo * CV;
    const float inv = 1.0f / (float)(k * k);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        const long r = i / CV;
        const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const float* src = x + (((long)b * H + oy * k + dy) * W + ox * k + dx) * ldx + c;
                if constexpr (V == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                } else {
                    acc[0] += src[0];
                }
            }
        if constexpr (V == 4) *reinterpret_cast<float4*>(y + r * ldy + c) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
        else y[r * ldy + c] = acc[0] * inv;
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx, int B, int H, int W, int C, int k) {
    TVL_KERNEL_ENTRY();
    const int Ho = H / k, Wo = W / k;
    const long total = (long)B * H * W * C;
    const float inv = 1.0f / (float)(k * k);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;
        const int ix = (int)(r % W), iy = (int)((r / W) % H), b = (int)(r / ((long)W * H));
        const int oy = iy / k, ox = ix / k;
        float v = 0.f;
        if (oy < Ho && ox < Wo) v = dy[(((long)b * Ho + oy) * Wo + ox) * lddy + c] * inv;
        dx[r * lddx + c] = v;
    }
}

// ---- bilinear upsample by an integer factor, align_corners=False ---------------------------------------------------------
// PyTorch: src = max(0, (dst + 0.5)/s - 0.5); i0 = floor(src); i1 = i0 + (i0 < n-1); w1 = src - i0
__device__ __forceinline__ void bil_src(int dst, float inv_s, int n, int& i0, int& i1, float& w1) {
    float src = ((float)dst + 0.5f) * inv_s - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 < n - 1 ? i0 : n - 1;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    w1 = src - (float)i0;
}
// V = 4: one float4 of channels per thread (C, ldx, ldy multiples of 4, 16-byte aligned); V = 1: scalar
template <int V>
struct VecT { typedef float type; };
template <>
struct VecT<4> { typedef float4 type; };
__device__ __forceinline__ float vmul(float a, float w) { return a * w; }
__device__ __forceinline__ float4 vmul(float4 a, float w) { return make_float4(a.x * w, a.y * w, a.z * w, a.w * w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <int V>
__device__ __forceinline__ typename VecT<V>::type vzero();
template <>
__device__ __forceinline__ float vzero<1>() { return 0.f; }
template <>
__device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int V>
__global__ void bilinear_up_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int B, int H, int W, int C, int s) {
    TVL_KERNEL_ENTRY();
    typedef typename VecT<V>::type T;
    const int Ho = H * s, Wo = W * s;
    const int cv = C / V;
    const float inv_s = 1.0f / (float)s;
    const long total = (long)B * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const long r = i / cv;
        const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(oy, inv_s, H, y0, y1, ly);
        bil_src(ox, inv_s, W, x0, x1, lx);
        const float* xb = x + (long)b * H * W * ldx + c;
        const T v00 = *reinterpret_cast<const T*>(xb + ((long)y0 * W + x0) * ldx), v01 = *reinterpret_cast<const T*>(xb + ((long)y0 * W + x1) * ldx);
        const T v10 = *reinterpret_cast<const T*>(xb + ((long)y1 * W + x0) * ldx), v11 = *reinterpret_cast<const T*>(xb + ((long)y1 * W + x1) * ldx);
        const T top = vadd(vmul(v00, 1.f - lx), vmul(v01, lx)), bot = vadd(vmul(v10, 1.f - lx), vmul(v11, lx));
        *reinterpret_cast<T*>(y + r * ldy + c) = vadd(vmul(top, 1.f - ly), vmul(bot, ly));
    }
}
// The same upsample written directly as the h2 image of the result (csrc/tp3.h; the next 3x3 conv's operand, tvl_conv3x3_h2): one thread
// = 8 consecutive channels of one output pixel = one 16-byte store per piece, lane order as in h2_pack_kernel.  Bilinear weights are a convex
// combination, so the input's maximum (bits, from tvl_h2_absmax) bounds the output: the image carries the input's scale.
__global__ __launch_bounds__(256) void bilinear_up_h2_kernel(const float* __restrict__ x, int ldx, const unsigned* __restrict__ bits,
                                                             float* __restrict__ inv_scale, unsigned char* __restrict__ out, int B, int H, int W, int C,
                                                             int s, long rows_padded) {
    TVL_KERNEL_ENTRY();
    const int Ho = H * s, Wo = W * s;
    const unsigned KB = (unsigned)(C >> 4);
    const float inv_s = 1.0f / (float)s;
    const long rows = (long)B * Ho * Wo;
    const float inv_all = h2::inv_scale_of(__builtin_bit_cast(float, bits[0]));
    const float sc = 1.0f / inv_all;   // a power of two: exact
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[0] = inv_all;
    const long total = rows_padded * (C >> 3);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const unsigned blk = (unsigned)(t >> 6);
        const unsigned rb = blk / KB, kb = blk - rb * KB;
        const long r = (long)rb * 32 + (lane & 31);
        const int c = (int)kb * 16 + (lane >> 5) * 8;
        uint4 pl[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (r < rows) {
            const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
            int y0, y1, x0, x1;
            float ly, lx;
            bil_src(oy, inv_s, H, y0, y1, ly);
            bil_src(ox, inv_s, W, x0, x1, lx);
            const float* xb = x + (long)b * H * W * ldx + c;
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 v00 = *reinterpret_cast<const float4*>(xb + ((long)y0 * W + x0) * ldx + 4 * q), v01 = *reinterpret_cast<const float4*>(xb + ((long)y0 * W + x1) * ldx + 4 * q);
                const float4 v10 = *reinterpret_cast<const float4*>(xb + ((long)y1 * W + x0) * ldx + 4 * q), v11 = *reinterpret_cast<const float4*>(xb + ((long)y1 * W + x1) * ldx + 4 * q);
                const float4 top = vadd(vmul(v00, 1.f - lx), vmul(v01, lx)), bot = vadd(vmul(v10, 1.f - lx), vmul(v11, lx));
                const float4 o = vadd(vmul(top, 1.f - ly), vmul(bot, ly));   // the same operation order as bilinear_up_fwd_kernel: identical fp32 values
                v[4 * q] = o.x * sc; v[4 * q + 1] = o.y * sc; v[4 * q + 2] = o.z * sc; v[4 * q + 3] = o.w * sc;
            }
            h2::split8(v, pl);
        }
        unsigned char* o = out + (long)blk * h2::BLK + lane * 16;
        *reinterpret_cast<uint4*>(o) = pl[0];
        *reinterpret_cast<uint4*>(o + h2::PIECE) = pl[1];
    }
}
// gather form of the transpose: input pixel i receives from the outputs whose (i0, i1) touch it
template <int V>
__global__ void bilinear_up_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx, int B, int H, int W, int C, int s) {
    TVL_KERNEL_ENTRY();
    typedef typename VecT<V>::type T;
    const int Ho = H * s, Wo = W * s;
    const int cv = C / V;
    const float inv_s = 1.0f / (float)s;
    const long total = (long)B * H * W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const long r = i / cv;
        const int ix = (int)(r % W), iy = (int)((r / W) % H), b = (int)(r / ((long)W * H));
        const int oy_lo = max(0, s * iy - s), oy_hi = min(Ho - 1, s * iy + 2 * s - 1);
        const int ox_lo = max(0, s * ix - s), ox_hi = min(Wo - 1, s * ix + 2 * s - 1);
        const float* db = dy + (long)b * Ho * Wo * lddy + c;
        T acc = vzero<V>();
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float ly;
            bil_src(oy, inv_s, H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            T row = vzero<V>();
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float lx;
                bil_src(ox, inv_s, W, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx != 0.f) row = vadd(row, vmul(*reinterpret_cast<const T*>(db + ((long)oy * Wo + ox) * lddy), wx));
            }
            acc = vadd(acc, vmul(row, wy));
        }
        *reinterpret_cast<T*>(dx + r * lddx + c) = acc;
    }
}

// ---- bicubic resize of single-channel maps, align_corners=True (A = -0.75) ----------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
    const float A = -0.75f;
    const float t1 = t + 1.f, t2 = 1.f - t, t3 = 2.f - t;
    w[0] = ((A * t1 - 5.f * A) * t1 + 8.f * A) * t1 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * t2 - (A + 3.f)) * t2 * t2 + 1.f;
    w[3] = ((A * t3 - 5.f * A) * t3 + 8.f * A) * t3 - 4.f * A;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void bicubic_ac_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ extra, float a, float r,
                                      int B, int Hi, int Wi, int Ho, int Wo, float sy, float sx) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), b = (int)(i / ((long)Wo * Ho));
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_coeffs(fy - (float)iy, wy);
        cubic_coeffs(fx - (float)ix, wx);
        const float* xb = x + (long)b * Hi * Wi;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* row = xb + (long)clampi(iy - 1 + j, 0, Hi - 1) * Wi;
            float rv = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) rv += wx[k] * row[clampi(ix - 1 + k, 0, Wi - 1)];
            acc += wy[j] * rv;
        }
        float v = a * acc;
        if (extra) v += r * extra[i];
        y[i] = v;
    }
}
__global__ void bicubic_ac_bwd_kernel(const float* __restrict__ dy, float a, float* __restrict__ dx, int B, int Hi, int Wi, int Ho, int Wo,
                                      float sy, float sx) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * Hi * Wi;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % Wi), iy = (int)((i / Wi) % Hi), b = (int)(i / ((long)Wi * Hi));
        // outputs whose 4-tap window [floor(src)-1, floor(src)+2] (clamped) can touch this pixel
        const int oy_lo = sy > 0.f ? max(0, (int)floorf((float)(iy - 2) / sy) - 1) : 0;
        const int oy_hi = sy > 0.f ? min(Ho - 1, (int)ceilf((float)(iy + 2) / sy) + 1) : Ho - 1;
        const int ox_lo = sx > 0.f ? max(0, (int)floorf((float)(ix - 2) / sx) - 1) : 0;
        const int ox_hi = sx > 0.f ? min(Wo - 1, (int)ceilf((float)(ix + 2) / sx) + 1) : Wo - 1;
        const float* db = dy + (long)b * Ho * Wo;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float fy = sy * (float)oy;
            const int jy = (int)floorf(fy);
            float w4[4];
            cubic_coeffs(fy - (float)jy, w4);
            float wy = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) wy += clampi(jy - 1 + j, 0, Hi - 1) == iy ? w4[j] : 0.f;
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float fx = sx * (float)ox;
                const int jx = (int)floorf(fx);
                float v4[4];
                cubic_coeffs(fx - (float)jx, v4);
                float wx = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) wx += clampi(jx - 1 + k, 0, Wi - 1) == ix ? v4[k] : 0.f;
                if (wx != 0.f) row += wx * db[(long)oy * Wo + ox];
            }
            acc += wy * row;
        }
        dx[i] = a * acc;
    }
}

// ---- predict tail: probability map -> bicubic resize (align_corners=False, no antialias) -> 8-bit grey level -----------------
// TF.resize(pred, mask_shape, BICUBIC, antialias=False) then torchvision.utils.save_image's x*255 + 0.5, clamp, uint8
// (reference src/utils/save_utils.py:93-101)
__global__ void bicubic_resize_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int Hi, int Wi, int Ho, int Wo,
                                         float sy, float sx) {
    TVL_KERNEL_ENTRY();
    const long total = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)(i / Wo);
        const float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_coeffs(fy - (float)iy, wy);
        cubic_coeffs(fx - (float)ix, wx);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* row = x + (long)clampi(iy - 1 + j, 0, Hi - 1) * Wi;
            float rv = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) rv += wx[k] * row[clampi(ix - 1 + k, 0, Wi - 1)];
            acc += wy[j] * rv;
        }
        float v = acc * 255.f + 0.5f;
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        out[i] = (unsigned char)v;
    }
}

// ---- Projector tail: per-sample 3x3 conv C -> 1 whose kernel + bias come from the text state --------------------------------
// word[b, c*9 + t] = weight of channel c, tap t = ky*3+kx; word[b, 9C] = bias  (layers.py:106-118)
// forward: taps[b,p,t] = sum_c x[b,p,c] * w[b,c,t]   then   out[b,y,x] = bias[b] + sum_t taps[b,(y+ky-1, x+kx-1), t]
constexpr int DC_PIX = 64;   // pixels per block (forward taps kernel)
constexpr int DC_CH = 64;    // channel chunk staged through LDS
__global__ __launch_bounds__(256) void dynconv_taps_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ word, int ldw,
                                                           float* __restrict__ taps, int HW, int C) {
    TVL_KERNEL_ENTRY();
    __shared__ float xs[DC_PIX][DC_CH + 1];
    __shared__ float ws[DC_CH][9];
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * DC_PIX;
    const int p = threadIdx.x & 63, g = threadIdx.x >> 6;  // pixel, tap group: taps g, g+4, g+8
    float acc[3] = {0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += DC_CH) {
        const int cn = min(DC_CH, C - c0);
        for (int i = threadIdx.x; i < DC_PIX * DC_CH; i += 256) {
            const int pp = i / DC_CH, cc = i % DC_CH;
            float v = 0.f;
            if (p0 + pp < HW && cc < cn) v = x[((long)b * HW + p0 + pp) * ldx + c0 + cc];
            xs[pp][cc] = v;
        }
        for (int i = threadIdx.x; i < DC_CH * 9; i += 256) {
            const int cc = i / 9;
            ws[cc][i % 9] = cc < cn ? word[(long)b * ldw + (long)(c0 + cc) * 9 + i % 9] : 0.f;
        }
        __syncthreads();
        for (int cc = 0; cc < DC_CH; ++cc) {
            const float xv = xs[p][cc];
            acc[0] += xv * ws[cc][g];
            acc[1] += xv * ws[cc][g + 4];
            if (g == 0) acc[2] += xv * ws[cc][8];
        }
        __syncthreads();
    }
    if (p0 + p < HW) {
        float* t = taps + ((long)b * HW + p0 + p) * 9;
        t[g] = acc[0];
        t[g + 4] = acc[1];
        if (g == 0) t[8] = acc[2];
    }
}
__global__ void dynconv_gather_kernel(const float* __restrict__ taps, const float* __restrict__ word, int ldw, float* __restrict__ out,
                                      int B, int H, int W, int C) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        float acc = word[(long)b * ldw + 9L * C];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int py = yy + t / 3 - 1, px = xx + t % 3 - 1;
            if (py >= 0 && py < H && px >= 0 && px < W) acc += taps[(((long)b * H + py) * W + px) * 9 + t];
        }
        out[i] = acc;
    }
}
// backward: dtaps[b,p,t] = dout[b, py-ky+1, px-kx+1];  dx[b,p,c] = sum_t dtaps*w[b,c,t];
//           dword[b,c*9+t] = sum_p x[b,p,c]*dtaps[b,p,t];  dword[b,9C] = sum_p dout[b,p]   (two-stage, deterministic)
constexpr int DCB_PIX = 128;
__global__ __launch_bounds__(256) void dynconv_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ word, int ldw, float* __restrict__ dx, int lddx,
                                                          float* __restrict__ work, int H, int W, int C, int nchunk) {
    TVL_KERNEL_ENTRY();
    __shared__ float dts[DCB_PIX][9];
    __shared__ float red[256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int HW = H * W;
    const int p0 = chunk * DCB_PIX;
    float dsum = 0.f;
    for (int i = threadIdx.x; i < DCB_PIX * 9; i += 256) {
        const int pp = i / 9, t = i % 9;
        const int pix = p0 + pp;
        float v = 0.f;
        if (pix < HW) {
            const int py = pix / W, px = pix % W;
            const int oy = py - (t / 3) + 1, ox = px - (t % 3) + 1;
            if (oy >= 0 && oy < H && ox >= 0 && ox < W) v = dout[((long)b * H + oy) * W + ox];
        }
        dts[pp][t] = v;
    }
    for (int i = threadIdx.x; i < DCB_PIX; i += 256)
        if (p0 + i < HW) dsum += dout[(long)b * HW + p0 + i];
    red[threadIdx.x] = dsum;
    __syncthreads();
    float* wk = work + ((long)b * nchunk + chunk) * (9L * C + 1);
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < 256; ++i) s += red[i];
        wk[9L * C] = s;
    }
    const int np = min(DCB_PIX, HW - p0);
    for (int c = threadIdx.x; c < C; c += 256) {
        float w9[9], acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { w9[t] = word[(long)b * ldw + (long)c * 9 + t]; acc[t] = 0.f; }
        for (int pp = 0; pp < np; ++pp) {
            const long row = (long)b * HW + p0 + pp;
            const float xv = x[row * ldx + c];
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float dt = dts[pp][t];
                d += dt * w9[t];
                acc[t] += xv * dt;
            }
            if (dx) dx[row * lddx + c] = d;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[(long)c * 9 + t] = acc[t];
    }
}
__global__ void dynconv_reduce_kernel(const float* __restrict__ work, float* __restrict__ dword, int ldw, int B, int n, int nchunk) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % n), b = (int)(i / n);
        float s = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) s += work[((long)b * nchunk + ch) * n + j];
        dword[(long)b * ldw + j] = s;
    }
}

// ---- strided matrix copy (channel concat / split of pixel matrices) --------------------------------------------------------
__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, long rows, int cols, int vec) {
    TVL_KERNEL_ENTRY();
    if (vec) {
        const int c4 = cols >> 2;
        const long total = rows * c4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / c4;
            const int c = (int)(i % c4) * 4;
            *reinterpret_cast<float4*>(dst + r * ldd + c) = *reinterpret_cast<const float4*>(src + r * lds + c);
        }
    } else {
        const long total = rows * cols;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / cols;
            const int c = (int)(i % cols);
            dst[r * ldd + c] = src[r * lds + c];
        }
    }
}

}  // namespace

extern "C" int tvl_copy2d(const float* src, int32_t lds, float* dst, int32_t ldd, int64_t rows, int32_t cols, tvlStream_t stream) {
    TVL_REQUIRE(src && dst, "tvl_copy2d: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "tvl_copy2d: bad shape rows=%ld cols=%d lds=%d ldd=%d", (long)rows, cols, lds, ldd);
    const int vec = cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && tvl_aligned16(src) && tvl_aligned16(dst);
    hipLaunchKernelGGL(copy2d_kernel, dim3(nblk(vec ? rows * (cols / 4) : rows * cols)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, lds,
                       dst, ldd, (long)rows, cols, vec);
    TVL_LAUNCH_CHECK("tvl_copy2d");
    return 0;
}

extern "C" int tvl_im2col3x3(const float* x, int64_t sb, int64_t sy, int64_t sx, int64_t sc, float* cols, int32_t ldc, int32_t B,
                             int32_t H, int32_t W, int32_t C, int32_t stride, tvlStream_t stream) {
    TVL_REQUIRE(x && cols, "tvl_im2col3x3: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2), "tvl_im2col3x3: bad shape B=%d H=%d W=%d C=%d stride=%d", B, H, W, C, stride);
    TVL_REQUIRE(ldc >= 9 * C, "tvl_im2col3x3: ldc (%d) < 9*C (%d)", ldc, 9 * C);
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = sc == 1 && C % 4 == 0 && ldc == 9 * C && sx % 4 == 0 && sy % 4 == 0 && sb % 4 == 0 && tvl_aligned16(x) && tvl_aligned16(cols);
    if (vec)
        hipLaunchKernelGGL(im2col3x3_kernel<true>, dim3(nblk((long)B * Ho * Wo * 9 * (C / 4))), dim3(256), 0, s, x, sb, sy, sx, sc, cols, ldc, B,
                           H, W, C, stride, Ho, Wo);
    else
        hipLaunchKernelGGL(im2col3x3_kernel<false>, dim3(nblk((long)B * Ho * Wo * ldc)), dim3(256), 0, s, x, sb, sy, sx, sc, cols, ldc, B, H, W,
                           C, stride, Ho, Wo);
    TVL_LAUNCH_CHECK("tvl_im2col3x3");
    return 0;
}

extern "C" int tvl_avgpool_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                               tvlStream_t stream) {
    TVL_REQUIRE(x && y, "tvl_avgpool_fwd: null pointer");
    TVL_REQUIRE(B > 0 && C > 0 && k > 0 && H >= k && W >= k && H % k == 0 && W % k == 0, "tvl_avgpool_fwd: H=%d W=%d not divisible by k=%d", H, W, k);
    TVL_REQUIRE(ldx >= C && ldy >= C, "tvl_avgpool_fwd: leading dimension too small");
    if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && tvl_aligned16(x) && tvl_aligned16(y))
        hipLaunchKernelGGL(avgpool_fwd_kernel<4>, dim3(nblk((long)B * (H / k) * (W / k) * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, y,
                           ldy, B, H, W, C, k);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel<1>, dim3(nblk((long)B * (H / k) * (W / k) * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, y,
                           ldy, B, H, W, C, k);
    TVL_LAUNCH_CHECK("tvl_avgpool_fwd");
    return 0;
}
extern "C" int tvl_avgpool_bwd(const float* dy, int32_t lddy, float* dx, int32_t lddx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                               tvlStream_t stream) {
    TVL_REQUIRE(dy && dx, "tvl_avgpool_bwd: null pointer");
    TVL_REQUIRE(B > 0 && C > 0 && k > 0 && H >= k && W >= k && H % k == 0 && W % k == 0, "tvl_avgpool_bwd: H=%d W=%d not divisible by k=%d", H, W, k);
    TVL_REQUIRE(lddx >= C && lddy >= C, "tvl_avgpool_bwd: leading dimension too small");
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(nblk((long)B * H * W * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, lddy, dx, lddx, B,
                       H, W, C, k);
    TVL_LAUNCH_CHECK("tvl_avgpool_bwd");
    return 0;
}

extern "C" int tvl_bilinear_up_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t s,
                                   tvlStream_t stream) {
    TVL_REQUIRE(x && y, "tvl_bilinear_up_fwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && s >= 1 && s <= 16, "tvl_bilinear_up_fwd: bad shape");
    TVL_REQUIRE(ldx >= C && ldy >= C, "tvl_bilinear_up_fwd: leading dimension too small");
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && tvl_aligned16(x) && tvl_aligned16(y);
    if (vec)
        hipLaunchKernelGGL(bilinear_up_fwd_kernel<4>, dim3(nblk((long)B * H * s * W * s * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                           ldx, y, ldy, B, H, W, C, s);
    else
        hipLaunchKernelGGL(bilinear_up_fwd_kernel<1>, dim3(nblk((long)B * H * s * W * s * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                           y, ldy, B, H, W, C, s);
    TVL_LAUNCH_CHECK("tvl_bilinear_up_fwd");
    return 0;
}
// x [B*H*W, C] -> the h2 image of F.interpolate(x, scale_factor=s, mode="bilinear") as a pixel matrix [B*(sH)*(sW), C], one scale for the
// tensor (inv_scale[0]); amax_bits: the bit pattern of max |x| (tvl_h2_absmax).  C % 16 == 0.  The caller appends the zero block
// tvl_conv3x3_h2 wants (this kernel writes tvl_h2_bytes(B*sH*sW, C) bytes).
extern "C" int tvl_bilinear_up_h2(const float* x, int32_t ldx, const void* amax_bits, void* out, float* inv_scale, int32_t B, int32_t H, int32_t W,
                                  int32_t C, int32_t s, tvlStream_t stream) {
    TVL_REQUIRE(x && amax_bits && out && inv_scale, "tvl_bilinear_up_h2: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0 && s >= 1 && s <= 16, "tvl_bilinear_up_h2: bad shape (C %% 16 == 0)");
    TVL_REQUIRE(ldx >= C && ldx % 4 == 0 && tvl_aligned16(x) && tvl_aligned16(out), "tvl_bilinear_up_h2: ldx %% 4 == 0 and 16-byte aligned operands");
    const long rows = (long)B * H * s * W * s, rp = (rows + 31) / 32 * 32;
    TVL_REQUIRE(rp / 32 * (C / 16) < (1ll << 31), "tvl_bilinear_up_h2: image too large");
    hipLaunchKernelGGL(bilinear_up_h2_kernel, dim3(nblk(rp * (C / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                       reinterpret_cast<const unsigned*>(amax_bits), inv_scale, reinterpret_cast<unsigned char*>(out), B, H, W, C, s, rp);
    TVL_LAUNCH_CHECK("tvl_bilinear_up_h2");
    return 0;
}
extern "C" int tvl_bilinear_up_bwd(const float* dy, int32_t lddy, float* dx, int32_t lddx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t s,
                                   tvlStream_t stream) {
    TVL_REQUIRE(dy && dx, "tvl_bilinear_up_bwd: null pointer");
    TVL_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && s >= 1 && s <= 16, "tvl_bilinear_up_bwd: bad shape");
    TVL_REQUIRE(lddx >= C && lddy >= C, "tvl_bilinear_up_bwd: leading dimension too small");
    const bool vec = C % 4 == 0 && lddx % 4 == 0 && lddy % 4 == 0 && tvl_aligned16(dx) && tvl_aligned16(dy);
    if (vec)
        hipLaunchKernelGGL(bilinear_up_bwd_kernel<4>, dim3(nblk((long)B * H * W * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, lddy, dx,
                           lddx, B, H, W, C, s);
    else
        hipLaunchKernelGGL(bilinear_up_bwd_kernel<1>, dim3(nblk((long)B * H * W * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, lddy, dx, lddx,
                           B, H, W, C, s);
    TVL_LAUNCH_CHECK("tvl_bilinear_up_bwd");
    return 0;
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int tvl_bicubic_ac_fwd(const float* x, float* y, const float* extra, float a, float r, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho,
                                  int32_t Wo, tvlStream_t stream) {
    TVL_REQUIRE(x && y, "tvl_bicubic_ac_fwd: null pointer");
    TVL_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "tvl_bicubic_ac_fwd: bad shape");
    hipLaunchKernelGGL(bicubic_ac_fwd_kernel, dim3(nblk((long)B * Ho * Wo)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, extra, a, r, B,
                       Hi, Wi, Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo));
    TVL_LAUNCH_CHECK("tvl_bicubic_ac_fwd");
    return 0;
}
extern "C" int tvl_bicubic_ac_bwd(const float* dy, float a, float* dx, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                  tvlStream_t stream) {
    TVL_REQUIRE(dy && dx, "tvl_bicubic_ac_bwd: null pointer");
    TVL_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "tvl_bicubic_ac_bwd: bad shape");
    hipLaunchKernelGGL(bicubic_ac_bwd_kernel, dim3(nblk((long)B * Hi * Wi)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, a, dx, B, Hi, Wi,
                       Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo));
    TVL_LAUNCH_CHECK("tvl_bicubic_ac_bwd");
    return 0;
}

extern "C" int tvl_bicubic_resize_u8(const float* x, uint8_t* out, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, tvlStream_t stream) {
    TVL_REQUIRE(x && out, "tvl_bicubic_resize_u8: null pointer");
    TVL_REQUIRE(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "tvl_bicubic_resize_u8: bad shape");
    hipLaunchKernelGGL(bicubic_resize_u8_kernel, dim3(nblk((long)Ho * Wo)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    TVL_LAUNCH_CHECK("tvl_bicubic_resize_u8");
    return 0;
}

extern "C" int tvl_dynconv_fwd(const float* x, int32_t ldx, const float* word, int32_t ldw, float* taps, float* out, int32_t B, int32_t H,
                               int32_t W, int32_t C, tvlStream_t stream) {
    TVL_REQUIRE(x && word && taps && out, "tvl_dynconv_fwd: null pointer");
    TVL_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0, "tvl_dynconv_fwd: bad shape");
    TVL_REQUIRE(ldx >= C && ldw >= 9 * C + 1, "tvl_dynconv_fwd: leading dimension too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int HW = H * W;
    hipLaunchKernelGGL(dynconv_taps_kernel, dim3((HW + DC_PIX - 1) / DC_PIX, B), dim3(256), 0, s, x, ldx, word, ldw, taps, HW, C);
    TVL_LAUNCH_CHECK("tvl_dynconv_fwd(taps)");
    hipLaunchKernelGGL(dynconv_gather_kernel, dim3(nblk((long)B * HW)), dim3(256), 0, s, taps, word, ldw, out, B, H, W, C);
    TVL_LAUNCH_CHECK("tvl_dynconv_fwd(gather)");
    return 0;
}
extern "C" int64_t tvl_dynconv_bwd_work_floats(int32_t B, int32_t H, int32_t W, int32_t C) {
    const long nchunk = ((long)H * W + DCB_PIX - 1) / DCB_PIX;
    return (long)B * nchunk * (9L * C + 1);
}
extern "C" int tvl_dynconv_bwd(const float* dout, const float* x, int32_t ldx, const float* word, int32_t ldw, float* dx, int32_t lddx,
                               float* dword, float* work, int32_t B, int32_t H, int32_t W, int32_t C, tvlStream_t stream) {
    TVL_REQUIRE(dout && x && word && dword && work, "tvl_dynconv_bwd: null pointer");
    TVL_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0, "tvl_dynconv_bwd: bad shape");
    TVL_REQUIRE(ldx >= C && ldw >= 9 * C + 1 && (!dx || lddx >= C), "tvl_dynconv_bwd: leading dimension too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = (H * W + DCB_PIX - 1) / DCB_PIX;
    hipLaunchKernelGGL(dynconv_bwd_kernel, dim3(nchunk, B), dim3(256), 0, s, dout, x, ldx, word, ldw, dx, lddx, work, H, W, C, nchunk);
    TVL_LAUNCH_CHECK("tvl_dynconv_bwd");
    hipLaunchKernelGGL(dynconv_reduce_kernel, dim3(nblk((long)B * (9 * C + 1))), dim3(256), 0, s, work, dword, ldw, B, 9 * C + 1, nchunk);
    TVL_LAUNCH_CHECK("tvl_dynconv_bwd(reduce)");
    return 0;
}
