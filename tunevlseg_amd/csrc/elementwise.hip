// HBM-bound token plumbing, FiLM, pixel shuffle, optimiser and small reductions.
// Everything here is one pass over its operands with 16-byte accesses where shapes allow.
#include <stdarg.h>
#include "common.h"

// ---------------------------------------------------------------------------------------------
// error channel
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void tvl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* tvl_last_error(void) { return g_err; }
extern "C" int tvl_abi_version(void) { return TVL_ABI_VERSION; }
extern "C" int tvl_build_flags(void) {
    int f = 0;
#ifdef TVL_EXPERIMENTS
    f |= 1;
#endif
#ifdef TVL_DIAGNOSTIC_KERNELS
    f |= 2;
#endif
#ifdef TVL_POISON_LDS
    f |= 4;
#endif
    return f;
}

namespace {

inline unsigned nblk(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// ---- im2col for the patch conv --------------------------------------------------------------
__global__ void im2col_kernel(const float* __restrict__ img, float* __restrict__ cols, int B, int C, int H, int W, int ps) {
    TVL_KERNEL_ENTRY();
    const int gh = H / ps, gw = W / ps;
    const int kdim = C * ps * ps;
    const int k4 = kdim >> 2;
    const long total = (long)B * gh * gw * k4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % k4) * 4;
        const long m = i / k4;
        const int gx = (int)(m % gw), gy = (int)((m / gw) % gh), b = (int)(m / ((long)gw * gh));
        const int px = kk % ps, py = (kk / ps) % ps, c = kk / (ps * ps);
        const float4 v = *reinterpret_cast<const float4*>(img + (((long)b * C + c) * H + gy * ps + py) * W + gx * ps + px);
        *reinterpret_cast<float4*>(cols + m * kdim + kk) = v;
    }
}

// ---- vision token assembly ------------------------------------------------------------------
__global__ void vision_assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                       const float* __restrict__ ctx, long ctx_bs, float* __restrict__ x0, int B, int P, int n, int D) {
    TVL_KERNEL_ENTRY();
    const int T = 1 + P + n;
    const long total = (long)B * T * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const long bt = i / D;
        const int t = (int)(bt % T), b = (int)(bt / T);
        float v;
        if (t == 0) v = cls[c] + pos[c];
        else if (t <= P) v = patch[((long)b * P + (t - 1)) * D + c] + pos[(long)t * D + c];
        else v = ctx[b * ctx_bs + (long)(t - 1 - P) * D + c];
        x0[i] = v;
    }
}

// ---- text token assembly --------------------------------------------------------------------
__global__ void text_assemble_kernel(const long long* __restrict__ ids, int L, const int* __restrict__ map,
                                     const float* __restrict__ table, const float* __restrict__ ctx, long ctx_bs,
                                     const float* __restrict__ pos, float* __restrict__ out, int B, int T, int D, long vocab) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * T * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const long bt = i / D;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int mp = map[t];
        float v;
        if (mp >= 0) {
            // an id outside the embedding table (a tokenizer that does not belong to this backbone) must not become a wild read: the
            // row turns into NaN and the loss says so (HF's nn.Embedding raises a device assert there)
            const long long id = ids[(long)b * L + mp];
            v = (id >= 0 && id < vocab) ? table[id * D + c] : __builtin_nanf("");
        } else v = ctx[b * ctx_bs + (long)(-mp - 1) * D + c];
        out[i] = v + pos[(long)t * D + c];
    }
}

// ---- generic row splice (API-compat learner.forward paths) -----------------------------------
__global__ void splice_rows_kernel(const float* __restrict__ x, int L, const int* __restrict__ map, const float* __restrict__ ctx, long ctx_bs,
                                   float* __restrict__ out, int B, int T, int D) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * T * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const long bt = i / D;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int mp = map[t];
        out[i] = mp >= 0 ? x[((long)b * L + mp) * D + c] : ctx[b * ctx_bs + (long)(-mp - 1) * D + c];
    }
}

// ---- in-place row overwrite / its gradient --------------------------------------------------
__global__ void rows_overwrite_kernel(float* __restrict__ x, const float* __restrict__ src, long src_bs, int B, int T, int D, int row0, int n) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * n * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int j = (int)((i / D) % n);
        const int b = (int)(i / ((long)D * n));
        x[((long)b * T + row0 + j) * D + c] = src[b * src_bs + (long)j * D + c];
    }
}
__global__ void rows_grad_kernel(float* __restrict__ g, float* __restrict__ dst, int B, int T, int D, int row0, int n,
                                 int reduce_batch, int zero_src, int accumulate) {
    TVL_KERNEL_ENTRY();
    const long total = (long)n * D * (reduce_batch ? 1 : B);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int j = (int)((i / D) % n);
        if (reduce_batch) {
            float acc = 0.f;
            for (int b = 0; b < B; ++b) {
                float* p = g + ((long)b * T + row0 + j) * D + c;
                acc += *p;
                if (zero_src) *p = 0.f;
            }
            float* d = dst + (long)j * D + c;
            *d = accumulate ? *d + acc : acc;
        } else {
            const int b = (int)(i / ((long)D * n));
            float* p = g + ((long)b * T + row0 + j) * D + c;
            float* d = dst + ((long)b * n + j) * D + c;
            *d = accumulate ? *d + *p : *p;
            if (zero_src) *p = 0.f;
        }
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx, float* __restrict__ out, int B, int T, int D) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D), b = (int)(i / D);
        out[i] = x[((long)b * T + idx[b]) * D + c];
    }
}
__global__ void scatter_rows_add_kernel(const float* __restrict__ dout, const int* __restrict__ idx, float* __restrict__ dx, int B, int T, int D) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D), b = (int)(i / D);
        dx[((long)b * T + idx[b]) * D + c] += dout[i];
    }
}

// ---- FiLM -----------------------------------------------------------------------------------
__global__ void film_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mul, const float* __restrict__ add,
                                float* __restrict__ y, int B, int T, int C) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long)T * C));
        y[i] = mul[(long)b * C + c] * x[i] + add[(long)b * C + c];
    }
}
// grid (B, ceil(C/64)); 1024 threads = 64 columns x 16 token groups (B = 32 workgroups only: with 4 groups each thread walked 121 tokens one
// dependent load after the other, 44 us for 4 MB); the groups meet in LDS in a fixed order
constexpr int FILM_TG = 16;
__global__ __launch_bounds__(64 * FILM_TG) void film_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mul,
                                                                float* __restrict__ dx, float* __restrict__ dmul, float* __restrict__ dadd,
                                                                int B, int T, int C) {
    TVL_KERNEL_ENTRY();
    __shared__ float s1[FILM_TG][64], s2[FILM_TG][64];
    const int b = blockIdx.x;
    const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        const float m = mul[(long)b * C + c];
#pragma unroll 4
        for (int t = tg; t < T; t += FILM_TG) {
            const long i = ((long)b * T + t) * C + c;
            const float d = dy[i];
            a1 += d * x[i];
            a2 += d;
            dx[i] = m * d;
        }
    }
    s1[tg][cl] = a1;
    s2[tg][cl] = a2;
    __syncthreads();
    if (tg == 0 && c < C) {
        float r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int g = 0; g < FILM_TG; ++g) { r1 += s1[g][cl]; r2 += s2[g][cl]; }
        if (dmul) dmul[(long)b * C + c] = r1;
        if (dadd) dadd[(long)b * C + c] = r2;
    }
}

// ---- ConvTranspose2d(k = s = ps) tail: column block -> image --------------------------------
__global__ void pixel_shuffle_kernel(const float* __restrict__ cols, const float* __restrict__ bias, const float* __restrict__ extra,
                                     float a, float r, float* __restrict__ logits, int B, int G, int ps) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const long total = (long)B * S * S;
    const float bv = bias ? bias[0] : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S), b = (int)(i / ((long)S * S));
        const int gx = x / ps, px = x % ps, gy = y / ps, py = y % ps;
        float v = a * (cols[(((long)b * G + gy) * G + gx) * (ps * ps) + py * ps + px] + bv);
        if (extra) v += r * extra[i];
        logits[i] = v;
    }
}
__global__ void pixel_unshuffle_kernel(const float* __restrict__ dlogits, float a, float* __restrict__ dcols, int B, int G, int ps) {
    TVL_KERNEL_ENTRY();
    const int S = G * ps;
    const long total = (long)B * S * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes dcols: (b, gy, gx, py, px)
        const int px = (int)(i % ps), py = (int)((i / ps) % ps);
        const long m = i / (ps * ps);
        const int gx = (int)(m % G), gy = (int)((m / G) % G), b = (int)(m / ((long)G * G));
        dcols[i] = a * dlogits[((long)b * S + gy * ps + py) * S + gx * ps + px];
    }
}

// ---- input side: decoded uint8 sample -> network input (albumentations Normalize + ToTensorV2 of the reference's transforms) ----
// image [B,H,W,3] uint8 -> [B,3,H,W] float, (x/255 - mean[c]) / std[c]; one thread = 4 consecutive pixels of one (b, c) plane row
__global__ void normalize_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, long pixels_per_image, int B,
                                    float m0, float m1, float m2, float s0, float s1, float s2) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * pixels_per_image;
    const float mean[3] = {m0, m1, m2}, inv[3] = {1.0f / (255.0f * s0), 1.0f / (255.0f * s1), 1.0f / (255.0f * s2)};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / pixels_per_image, p = i - b * pixels_per_image;
        const uint8_t* px = img + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(b * 3 + c) * pixels_per_image + p] = ((float)px[c] - 255.0f * mean[c]) * inv[c];
    }
}
// mask [B,H,W] uint8 -> [B,1,H,W] float = x / 255 (image_text_mask_dataset.py:66-71); the metrics' mask.long() keeps only 255
__global__ void mask_u8_kernel(const uint8_t* __restrict__ m, float* __restrict__ out, long n) {
    TVL_KERNEL_ENTRY();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)m[i] / 255.0f;
}

// ---- (1 - r) * main + r * extra with r read on the device (trainable residual_ratio: no host round trip) ----
__global__ void mix_kernel(const float* __restrict__ main_, const float* __restrict__ extra, const float* __restrict__ ratio,
                           float* __restrict__ out, long n) {
    TVL_KERNEL_ENTRY();
    const float r = ratio[0], a = 1.0f - r;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a * main_[i] + r * extra[i];
}
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ ratio, int one_minus, float* __restrict__ y, long n) {
    TVL_KERNEL_ENTRY();
    const float s = one_minus ? 1.0f - ratio[0] : ratio[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = s * x[i];
}

// ---- optimiser / misc -----------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale, int* __restrict__ nonfinite) {
    TVL_KERNEL_ENTRY();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        if (nonfinite && !(fabsf(gr) <= 3.4028235e38f)) nonfinite[0] = 1;   // sticky flag (every writer stores the same value): a non-finite gradient reached the update
        float pv = p[i] * (1.0f - lr * wd);
        const float mv = b1 * m[i] + (1.0f - b1) * gr;
        const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pv -= (lr / bc1) * (mv / denom);
        p[i] = pv;
    }
}
__global__ void fill_kernel(float* __restrict__ p, float val, long n) {
    TVL_KERNEL_ENTRY();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = val;
}
__global__ void axpby_kernel(const float* __restrict__ x, float a, float* __restrict__ y, float b, long n) {
    TVL_KERNEL_ENTRY();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}
__global__ void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y, long rows, int cols, int act) {
    TVL_KERNEL_ENTRY();
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (bias) v += bias[i % cols];
        y[i] = act_f(v, act);
    }
}
// one wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm, int rows, int cols) {
    TVL_KERNEL_ENTRY();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[(long)row * cols + c]; s += v * v; }
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int c = lane; c < cols; c += 64) y[(long)row * cols + c] = x[(long)row * cols + c] * inv;
    if (lane == 0 && inv_norm) inv_norm[row] = inv;
}
// y = x/||x||: dx = (dy - y * <dy,y>) / ||x||
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv_norm,
                                                         float* __restrict__ dx, int rows, int cols) {
    TVL_KERNEL_ENTRY();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += dy[(long)row * cols + c] * y[(long)row * cols + c];
    s = wave_sum(s);
    const float inv = inv_norm[row];
    for (int c = lane; c < cols; c += 64) dx[(long)row * cols + c] = (dy[(long)row * cols + c] - y[(long)row * cols + c] * s) * inv;
}
// column sums; grid (ceil(cols/64), row_chunks); 256 threads = 64 columns x 4 row groups; atomics across chunks
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int cols, long rows_per_chunk) {
    TVL_KERNEL_ENTRY();
    __shared__ float s[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    const long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    float a = 0.f;
    if (c < cols)
        for (long r = r0 + rg; r < r1; r += 4) a += x[r * cols + c];
    s[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < cols) atomicAdd(&out[c], (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]));
}

__global__ void dact_mul_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ out, long n, int act) {
    TVL_KERNEL_ENTRY();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = dy[i] * dact_f(pre[i], act);
}
// out[b,j,:] = bias[b,:] + cvec[j,:]
__global__ void outer_add_kernel(const float* __restrict__ bias, const float* __restrict__ cvec, float* __restrict__ out, int B, int n, int D) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * n * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D), j = (int)((i / D) % n), b = (int)(i / ((long)D * n));
        out[i] = bias[(long)b * D + c] + cvec[(long)j * D + c];
    }
}
// dbias[b,:] = sum_j dout[b,j,:] ; dcvec[j,:] = sum_b dout[b,j,:]
__global__ void outer_add_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dbias, float* __restrict__ dcvec, int B, int n, int D) {
    TVL_KERNEL_ENTRY();
    const long total = (long)(B + n) * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int row = (int)(i / D);
        float acc = 0.f;
        if (row < B) {
            for (int j = 0; j < n; ++j) acc += dout[((long)row * n + j) * D + c];
            dbias[(long)row * D + c] = acc;
        } else {
            const int j = row - B;
            for (int b = 0; b < B; ++b) acc += dout[((long)b * n + j) * D + c];
            dcvec[(long)j * D + c] = acc;
        }
    }
}

// counter-based dropout: keep(i) is a pure function of (seed, i), so the backward pass regenerates the same mask
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p, float scale, unsigned long long seed) {
    TVL_KERNEL_ENTRY();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long r = mix64(seed ^ mix64((unsigned long long)i));
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
        y[i] = u >= p ? x[i] * scale : 0.f;
    }
}

// out[0] (+)= sum_i x[i]*y[i]  (double accumulation inside the block, one float atomic per block)
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, long n) {
    TVL_KERNEL_ENTRY();
    __shared__ double s[4];
    double a = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += (double)x[i] * (double)(y ? y[i] : 1.0f);
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (float)((s[0] + s[1]) + (s[2] + s[3])));
}

}  // namespace

#define S_(stream) reinterpret_cast<hipStream_t>(stream)
#define GRID_FOR(total) dim3(nblk((total)) < 8192u ? (nblk((total)) ? nblk((total)) : 1u) : 8192u)

extern "C" int tvl_im2col_patch(const float* img, float* cols, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ps, tvlStream_t stream) {
    TVL_REQUIRE(img && cols, "tvl_im2col_patch: null pointer");
    TVL_REQUIRE(B > 0 && C > 0 && ps > 0 && H % ps == 0 && W % ps == 0 && ps % 4 == 0 && W % 4 == 0,
                "tvl_im2col_patch: bad shape B=%d C=%d H=%d W=%d ps=%d", B, C, H, W, ps);
    TVL_REQUIRE(tvl_aligned16(img) && tvl_aligned16(cols), "tvl_im2col_patch: unaligned");
    const long total = (long)B * (H / ps) * (W / ps) * (C * ps * ps / 4);
    hipLaunchKernelGGL(im2col_kernel, GRID_FOR(total), dim3(256), 0, S_(stream), img, cols, B, C, H, W, ps);
    TVL_LAUNCH_CHECK("tvl_im2col_patch");
    return 0;
}

extern "C" int tvl_vision_assemble(const float* patch, const float* cls, const float* pos, const float* ctx, int64_t ctx_bs, float* x0,
                                   int32_t B, int32_t P, int32_t n, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(patch && cls && pos && x0 && (n == 0 || ctx), "tvl_vision_assemble: null pointer");
    TVL_REQUIRE(B > 0 && P > 0 && n >= 0 && D > 0, "tvl_vision_assemble: bad shape");
    const long total = (long)B * (1 + P + n) * D;
    hipLaunchKernelGGL(vision_assemble_kernel, GRID_FOR(total), dim3(256), 0, S_(stream), patch, cls, pos, ctx, (long)ctx_bs, x0, B, P, n, D);
    TVL_LAUNCH_CHECK("tvl_vision_assemble");
    return 0;
}

extern "C" int tvl_text_assemble(const int64_t* ids, int32_t L, const int32_t* map, const float* table, int64_t vocab, const float* ctx, int64_t ctx_bs,
                                 const float* pos, float* out, int32_t B, int32_t T, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(ids && map && table && pos && out, "tvl_text_assemble: null pointer");
    TVL_REQUIRE(B > 0 && T > 0 && D > 0 && L > 0 && vocab > 0, "tvl_text_assemble: bad shape");
    const long total = (long)B * T * D;
    hipLaunchKernelGGL(text_assemble_kernel, GRID_FOR(total), dim3(256), 0, S_(stream), reinterpret_cast<const long long*>(ids), L, map,
                       table, ctx, (long)ctx_bs, pos, out, B, T, D, (long)vocab);
    TVL_LAUNCH_CHECK("tvl_text_assemble");
    return 0;
}

extern "C" int tvl_rows_overwrite(float* x, const float* src, int64_t src_bs, int32_t B, int32_t T, int32_t D, int32_t row0, int32_t n,
                                  tvlStream_t stream) {
    TVL_REQUIRE(x && src, "tvl_rows_overwrite: null pointer");
    TVL_REQUIRE(B > 0 && n > 0 && row0 >= 0 && row0 + n <= T && D > 0, "tvl_rows_overwrite: rows [%d,%d) outside T=%d", row0, row0 + n, T);
    const long total = (long)B * n * D;
    hipLaunchKernelGGL(rows_overwrite_kernel, GRID_FOR(total), dim3(256), 0, S_(stream), x, src, (long)src_bs, B, T, D, row0, n);
    TVL_LAUNCH_CHECK("tvl_rows_overwrite");
    return 0;
}

extern "C" int tvl_rows_grad(float* g, float* dst, int32_t B, int32_t T, int32_t D, int32_t row0, int32_t n, int32_t reduce_batch,
                             int32_t zero_src, int32_t accumulate, tvlStream_t stream) {
    TVL_REQUIRE(g && dst, "tvl_rows_grad: null pointer");
    TVL_REQUIRE(B > 0 && n > 0 && row0 >= 0 && row0 + n <= T && D > 0, "tvl_rows_grad: rows [%d,%d) outside T=%d", row0, row0 + n, T);
    const long total = (long)n * D * (reduce_batch ? 1 : B);
    hipLaunchKernelGGL(rows_grad_kernel, GRID_FOR(total), dim3(256), 0, S_(stream), g, dst, B, T, D, row0, n, reduce_batch, zero_src, accumulate);
    TVL_LAUNCH_CHECK("tvl_rows_grad");
    return 0;
}

extern "C" int tvl_gather_rows(const float* x, const int32_t* idx, float* out, int32_t B, int32_t T, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(x && idx && out && B > 0 && T > 0 && D > 0, "tvl_gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, GRID_FOR((long)B * D), dim3(256), 0, S_(stream), x, idx, out, B, T, D);
    TVL_LAUNCH_CHECK("tvl_gather_rows");
    return 0;
}
extern "C" int tvl_scatter_rows_add(const float* dout, const int32_t* idx, float* dx, int32_t B, int32_t T, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(dout && idx && dx && B > 0 && T > 0 && D > 0, "tvl_scatter_rows_add: bad arguments");
    hipLaunchKernelGGL(scatter_rows_add_kernel, GRID_FOR((long)B * D), dim3(256), 0, S_(stream), dout, idx, dx, B, T, D);
    TVL_LAUNCH_CHECK("tvl_scatter_rows_add");
    return 0;
}

extern "C" int tvl_film_fwd(const float* x, const float* mul, const float* add, float* y, int32_t B, int32_t T, int32_t C, tvlStream_t stream) {
    TVL_REQUIRE(x && mul && add && y && B > 0 && T > 0 && C > 0, "tvl_film_fwd: bad arguments");
    hipLaunchKernelGGL(film_fwd_kernel, GRID_FOR((long)B * T * C), dim3(256), 0, S_(stream), x, mul, add, y, B, T, C);
    TVL_LAUNCH_CHECK("tvl_film_fwd");
    return 0;
}
extern "C" int tvl_film_bwd(const float* dy, const float* x, const float* mul, float* dx, float* dmul, float* dadd, int32_t B, int32_t T,
                            int32_t C, tvlStream_t stream) {
    TVL_REQUIRE(dy && x && mul && dx && B > 0 && T > 0 && C > 0, "tvl_film_bwd: bad arguments");
    TVL_REQUIRE(B <= 65535 * 32768, "tvl_film_bwd: batch too large");
    hipLaunchKernelGGL(film_bwd_kernel, dim3(B, (C + 63) / 64), dim3(64 * FILM_TG), 0, S_(stream), dy, x, mul, dx, dmul, dadd, B, T, C);
    TVL_LAUNCH_CHECK("tvl_film_bwd");
    return 0;
}

extern "C" int tvl_pixel_shuffle_fwd(const float* cols, const float* bias, const float* extra, float a, float r, float* logits, int32_t B,
                                     int32_t G, int32_t ps, tvlStream_t stream) {
    TVL_REQUIRE(cols && logits && B > 0 && G > 0 && ps > 0, "tvl_pixel_shuffle_fwd: bad arguments");
    hipLaunchKernelGGL(pixel_shuffle_kernel, GRID_FOR((long)B * G * ps * G * ps), dim3(256), 0, S_(stream), cols, bias, extra, a, r, logits, B, G, ps);
    TVL_LAUNCH_CHECK("tvl_pixel_shuffle_fwd");
    return 0;
}
extern "C" int tvl_pixel_unshuffle_bwd(const float* dlogits, float a, float* dcols, int32_t B, int32_t G, int32_t ps, tvlStream_t stream) {
    TVL_REQUIRE(dlogits && dcols && B > 0 && G > 0 && ps > 0, "tvl_pixel_unshuffle_bwd: bad arguments");
    hipLaunchKernelGGL(pixel_unshuffle_kernel, GRID_FOR((long)B * G * ps * G * ps), dim3(256), 0, S_(stream), dlogits, a, dcols, B, G, ps);
    TVL_LAUNCH_CHECK("tvl_pixel_unshuffle_bwd");
    return 0;
}

extern "C" int tvl_normalize_u8(const uint8_t* img, float* out, int32_t B, int32_t H, int32_t W, const float* mean3, const float* std3,
                                tvlStream_t stream) {
    TVL_REQUIRE(img && out && mean3 && std3 && B > 0 && H > 0 && W > 0, "tvl_normalize_u8: bad arguments");
    TVL_REQUIRE(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "tvl_normalize_u8: std must be positive");
    hipLaunchKernelGGL(normalize_u8_kernel, GRID_FOR((long)B * H * W), dim3(256), 0, S_(stream), img, out, (long)H * W, B, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2]);
    TVL_LAUNCH_CHECK("tvl_normalize_u8");
    return 0;
}
extern "C" int tvl_mask_u8(const uint8_t* mask, float* out, int64_t n, tvlStream_t stream) {
    TVL_REQUIRE(mask && out && n > 0, "tvl_mask_u8: bad arguments");
    hipLaunchKernelGGL(mask_u8_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), mask, out, (long)n);
    TVL_LAUNCH_CHECK("tvl_mask_u8");
    return 0;
}

extern "C" int tvl_mix(const float* main_, const float* extra, const float* ratio, float* out, int64_t n, tvlStream_t stream) {
    TVL_REQUIRE(main_ && extra && ratio && out && n > 0, "tvl_mix: bad arguments");
    hipLaunchKernelGGL(mix_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), main_, extra, ratio, out, (long)n);
    TVL_LAUNCH_CHECK("tvl_mix");
    return 0;
}
extern "C" int tvl_scale_dev(const float* x, const float* ratio, int32_t one_minus, float* y, int64_t n, tvlStream_t stream) {
    TVL_REQUIRE(x && ratio && y && n > 0, "tvl_scale_dev: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), x, ratio, one_minus, y, (long)n);
    TVL_LAUNCH_CHECK("tvl_scale_dev");
    return 0;
}

extern "C" int tvl_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int32_t step_t, float grad_scale, int32_t* nonfinite, tvlStream_t stream) {
    TVL_REQUIRE(p && g && m && v && n > 0 && step_t >= 1, "tvl_adamw: bad arguments");
    // bias corrections in double on the host, as torch.optim.AdamW computes them (fp32 powf is ~1e-5 off in the step size at small t)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step_t));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step_t));
    hipLaunchKernelGGL(adamw_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay,
                       bc1, bc2_sqrt, grad_scale, nonfinite);
    TVL_LAUNCH_CHECK("tvl_adamw");
    return 0;
}
extern "C" int tvl_fill(float* p, float val, int64_t n, tvlStream_t stream) {
    TVL_REQUIRE(p && n > 0, "tvl_fill: bad arguments");
    hipLaunchKernelGGL(fill_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), p, val, (long)n);
    TVL_LAUNCH_CHECK("tvl_fill");
    return 0;
}
extern "C" int tvl_axpby(const float* x, float a, float* y, float b, int64_t n, tvlStream_t stream) {
    TVL_REQUIRE(x && y && n > 0, "tvl_axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), x, a, y, b, (long)n);
    TVL_LAUNCH_CHECK("tvl_axpby");
    return 0;
}
extern "C" int tvl_bias_act(const float* x, const float* bias, float* y, int64_t rows, int32_t cols, int32_t act, tvlStream_t stream) {
    TVL_REQUIRE(x && y && rows > 0 && cols > 0, "tvl_bias_act: bad arguments");
    hipLaunchKernelGGL(bias_act_kernel, GRID_FOR((long)rows * cols), dim3(256), 0, S_(stream), x, bias, y, (long)rows, cols, act);
    TVL_LAUNCH_CHECK("tvl_bias_act");
    return 0;
}
extern "C" int tvl_l2norm_fwd(const float* x, float* y, float* inv_norm, int32_t rows, int32_t cols, tvlStream_t stream) {
    TVL_REQUIRE(x && y && rows > 0 && cols > 0, "tvl_l2norm_fwd: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, S_(stream), x, y, inv_norm, rows, cols);
    TVL_LAUNCH_CHECK("tvl_l2norm_fwd");
    return 0;
}
extern "C" int tvl_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int32_t rows, int32_t cols, tvlStream_t stream) {
    TVL_REQUIRE(dy && y && inv_norm && dx && rows > 0 && cols > 0, "tvl_l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, S_(stream), dy, y, inv_norm, dx, rows, cols);
    TVL_LAUNCH_CHECK("tvl_l2norm_bwd");
    return 0;
}
extern "C" int tvl_colsum(const float* x, float* out, int64_t rows, int32_t cols, int32_t accumulate, tvlStream_t stream) {
    TVL_REQUIRE(x && out && rows > 0 && cols > 0, "tvl_colsum: bad arguments");
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * cols, S_(stream));
        TVL_REQUIRE(e == hipSuccess, "tvl_colsum: memset failed: %s", hipGetErrorString(e));
    }
    long chunks = (rows + 1023) / 1024;
    if (chunks > 1024) chunks = 1024;
    const long rpc = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, (unsigned)chunks), dim3(256), 0, S_(stream), x, out, (long)rows, cols, rpc);
    TVL_LAUNCH_CHECK("tvl_colsum");
    return 0;
}

extern "C" int tvl_dot(const float* x, const float* y, float* out, int64_t n, int32_t accumulate, tvlStream_t stream) {
    TVL_REQUIRE(x && out && n > 0, "tvl_dot: bad arguments");
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float), S_(stream));
        TVL_REQUIRE(e == hipSuccess, "tvl_dot: memset failed: %s", hipGetErrorString(e));
    }
    long grid = (n + 256 * 16 - 1) / (256 * 16);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(dot_kernel, dim3((unsigned)grid), dim3(256), 0, S_(stream), x, y, out, (long)n);
    TVL_LAUNCH_CHECK("tvl_dot");
    return 0;
}

extern "C" int tvl_dact_mul(const float* dy, const float* pre, float* out, int64_t n, int32_t act, tvlStream_t stream) {
    TVL_REQUIRE(dy && pre && out && n > 0, "tvl_dact_mul: bad arguments");
    hipLaunchKernelGGL(dact_mul_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), dy, pre, out, (long)n, act);
    TVL_LAUNCH_CHECK("tvl_dact_mul");
    return 0;
}
extern "C" int tvl_outer_add(const float* bias, const float* cvec, float* out, int32_t B, int32_t n, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(bias && cvec && out && B > 0 && n > 0 && D > 0, "tvl_outer_add: bad arguments");
    hipLaunchKernelGGL(outer_add_kernel, GRID_FOR((long)B * n * D), dim3(256), 0, S_(stream), bias, cvec, out, B, n, D);
    TVL_LAUNCH_CHECK("tvl_outer_add");
    return 0;
}
extern "C" int tvl_outer_add_bwd(const float* dout, float* dbias, float* dcvec, int32_t B, int32_t n, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(dout && dbias && dcvec && B > 0 && n > 0 && D > 0, "tvl_outer_add_bwd: bad arguments");
    hipLaunchKernelGGL(outer_add_bwd_kernel, GRID_FOR((long)(B + n) * D), dim3(256), 0, S_(stream), dout, dbias, dcvec, B, n, D);
    TVL_LAUNCH_CHECK("tvl_outer_add_bwd");
    return 0;
}

extern "C" int tvl_splice_rows(const float* x, int32_t L, const int32_t* map, const float* ctx, int64_t ctx_bs, float* out, int32_t B,
                               int32_t T, int32_t D, tvlStream_t stream) {
    TVL_REQUIRE(x && map && ctx && out && B > 0 && T > 0 && D > 0 && L > 0, "tvl_splice_rows: bad arguments");
    hipLaunchKernelGGL(splice_rows_kernel, GRID_FOR((long)B * T * D), dim3(256), 0, S_(stream), x, L, map, ctx, (long)ctx_bs, out, B, T, D);
    TVL_LAUNCH_CHECK("tvl_splice_rows");
    return 0;
}

extern "C" int tvl_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, tvlStream_t stream) {
    TVL_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "tvl_dropout: bad arguments (p=%f)", (double)p);
    hipLaunchKernelGGL(dropout_kernel, GRID_FOR((long)n), dim3(256), 0, S_(stream), x, y, (long)n, p, 1.0f / (1.0f - p), (unsigned long long)seed);
    TVL_LAUNCH_CHECK("tvl_dropout");
    return 0;
}
