// Input-side image transforms on the device: what the reference does per sample on the host with
// albumentations / OpenCV (configs/experiment/coop/clipseg.yaml:78-120) runs here as two launches over the whole (ragged) batch.
//
//   tvl_resize_u8    ragged uint8 images (each its own height x width, packed back to back) -> [B, H, W, C] uint8;
//                    INTER_CUBIC with OpenCV's 8-bit fixed-point definition (half-pixel centres, Keys A = -0.75 in float, weights
//                    quantised to 11 fractional bits, int32 passes, (v + 2^21) >> 22, replicate border) or INTER_NEAREST
//                    (floor(dx * scale), no half-pixel shift) -- albumentations.Resize for the image / the mask
//   tvl_augment_u8   [B, H, W, 3] uint8 + [B, H, W] uint8 mask -> normalised fp32 [B, 3, H, W] + fp32 mask / 255 [B, 1, H, W]:
//                    per-sample affine warp (cubic / nearest, BORDER_REPLICATE; identity when the sample's flag is clear),
//                    brightness / contrast as albumentations' uint8 table, A.Normalize + ToTensorV2.  One pass, one write.
//
// Memory-bound gathers: one thread per output pixel, neighbouring threads read neighbouring source pixels (L1 / L2 reuse of the
// 4 x 4 footprint).  The float steps use the non-contracting intrinsics so that they round exactly like the CPU restatement
// (oracle/augment_oracle.py), which is what the tests hold them to; cv2 / albumentations are absent from the image ("unpinned").
#include "common.h"

#define S_(stream) reinterpret_cast<hipStream_t>(stream)
static inline unsigned nblk_(long total) { return (unsigned)((total + 255) / 256); }
#define GRID_FOR(total) dim3(nblk_((total)) < 16384u ? (nblk_((total)) ? nblk_((total)) : 1u) : 16384u)

namespace {

__device__ __forceinline__ void cubic_w(float x, float (&w)[4]) {   // cv::interpolateCubic, A = -0.75, float, no FMA contraction
    const float A = -0.75f;
    const float x1 = __fadd_rn(x, 1.0f);
    w[0] = __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(A, x1), 5.0f * A), x1), 8.0f * A), x1), 4.0f * A);
    w[1] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2.0f, x), A + 3.0f), x), x), 1.0f);
    const float y = __fsub_rn(1.0f, x);
    w[2] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2.0f, y), A + 3.0f), y), y), 1.0f);
    w[3] = __fsub_rn(__fsub_rn(__fsub_rn(1.0f, w[0]), w[1]), w[2]);
}

// taps and 11-bit weights of one destination coordinate
__device__ __forceinline__ void axis_taps(int d, int dst, int src, int (&idx)[4], int (&wq)[4]) {
    const double scale = (double)src / (double)dst;
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    const float fl = floorf(f);
    const int s = (int)fl;
    float w[4];
    cubic_w(__fsub_rn(f, fl), w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int q = (int)rint((double)w[k] * 2048.0);   // cvRound: half to even
        wq[k] = q < -32768 ? -32768 : (q > 32767 ? 32767 : q);
        int i = s + k - 1;
        idx[k] = i < 0 ? 0 : (i > src - 1 ? src - 1 : i);
    }
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ packed, const int64_t* __restrict__ offs,
                                                        const int32_t* __restrict__ hw, int B, int C, int H, int W, int cubic,
                                                        uint8_t* __restrict__ out) {
    TVL_KERNEL_ENTRY();
    const long total = (long)B * H * W;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int x = (int)(t % W), y = (int)((t / W) % H), b = (int)(t / ((long)W * H));
        const int h = hw[2 * b], w = hw[2 * b + 1];
        const uint8_t* src = packed + offs[b];
        uint8_t* o = out + t * C;
        if (!cubic) {
            int sy = (int)floor((double)y * ((double)h / (double)H)), sx = (int)floor((double)x * ((double)w / (double)W));
            sy = sy > h - 1 ? h - 1 : sy;
            sx = sx > w - 1 ? w - 1 : sx;
            for (int c = 0; c < C; ++c) o[c] = src[((long)sy * w + sx) * C + c];
            continue;
        }
        int xi[4], xw[4], yi[4], yw[4];
        axis_taps(x, W, w, xi, xw);
        axis_taps(y, H, h, yi, yw);
        for (int c = 0; c < C; ++c) {
            int v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint8_t* row = src + (long)yi[j] * w * C + c;
                const int r = row[xi[0] * C] * xw[0] + row[xi[1] * C] * xw[1] + row[xi[2] * C] * xw[2] + row[xi[3] * C] * xw[3];
                v += r * yw[j];
            }
            v = (v + (1 << 21)) >> 22;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

struct AugParams {   // per sample, 8 floats: inverse affine matrix (dst -> src) row-major 2 x 3, alpha, beta; flags in a separate int
    float m[6], alpha, beta;
};

__global__ __launch_bounds__(256) void augment_u8_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                         const float* __restrict__ params, const int32_t* __restrict__ flags, float m0, float m1,
                                                         float m2, float s0, float s1, float s2, float* __restrict__ out_img,
                                                         float* __restrict__ out_mask, int B, int H, int W) {
    TVL_KERNEL_ENTRY();
    const long hwp = (long)H * W, total = (long)B * hwp;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int x = (int)(t % W), y = (int)((t / W) % H), b = (int)(t / hwp);
        const float* p = params + 8 * b;
        const int fl = flags[b];
        const uint8_t* im = img + (long)b * hwp * 3;
        int px[3];
        int mk = 0;
        if (fl & 1) {
            const float fx = (float)x, fy = (float)y;
            const float sx = __fadd_rn(__fadd_rn(__fmul_rn(p[0], fx), __fmul_rn(p[1], fy)), p[2]);
            const float sy = __fadd_rn(__fadd_rn(__fmul_rn(p[3], fx), __fmul_rn(p[4], fy)), p[5]);
            const float x0 = floorf(sx), y0 = floorf(sy);
            float wx[4], wy[4];
            cubic_w(__fsub_rn(sx, x0), wx);
            cubic_w(__fsub_rn(sy, y0), wy);
            float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int yy = (int)y0 + j - 1;
                yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int xx = (int)x0 + i - 1;
                    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                    const uint8_t* s = im + ((long)yy * W + xx) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) row[c] = __fadd_rn(row[c], __fmul_rn((float)s[c], wx[i]));
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(row[c], wy[j]));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r = rintf(acc[c]);
                px[c] = (int)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
            }
            if (mask) {
                int mx = (int)rintf(sx), my = (int)rintf(sy);
                mx = mx < 0 ? 0 : (mx > W - 1 ? W - 1 : mx);
                my = my < 0 ? 0 : (my > H - 1 ? H - 1 : my);
                mk = mask[(long)b * hwp + (long)my * W + mx];
            }
        } else {
            const uint8_t* s = im + ((long)y * W + x) * 3;
            px[0] = s[0]; px[1] = s[1]; px[2] = s[2];
            if (mask) mk = mask[t];
        }
        if (fl & 2) {   // albumentations' uint8 table: clip(v * alpha + beta * 255, 0, 255) truncated
            const float al = p[6], be = p[7];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = (float)px[c];
                if (al != 1.0f) v = __fmul_rn(v, al);
                if (be != 0.0f) v = __fadd_rn(v, __fmul_rn(be, 255.0f));
                v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                px[c] = (int)v;
            }
        }
        const long o = (long)b * 3 * hwp + (long)y * W + x;
        out_img[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[0], 255.0f), m0), s0);
        out_img[o + hwp] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[1], 255.0f), m1), s1);
        out_img[o + 2 * hwp] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[2], 255.0f), m2), s2);
        if (mask) out_mask[t] = __fdiv_rn((float)mk, 255.0f);
    }
}

}  // namespace

// Ragged batch resize.  packed: the B images back to back, image b = hw[2b] rows x hw[2b + 1] columns x C interleaved channels starting at
// byte offs[b] (device arrays).  mode: 0 = INTER_NEAREST, 2 = INTER_CUBIC (cv2's values).  out: [B, H, W, C] uint8.
// Replaces albumentations.Resize (reference configs/experiment/coop/clipseg.yaml:80-84) for the whole batch.
extern "C" int tvl_resize_u8(const uint8_t* packed, const int64_t* offs, const int32_t* hw, int32_t B, int32_t C, int32_t H, int32_t W,
                             int32_t mode, uint8_t* out, tvlStream_t stream) {
    TVL_REQUIRE(packed && offs && hw && out && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0, "tvl_resize_u8: bad arguments");
    TVL_REQUIRE(mode == 0 || mode == 2, "tvl_resize_u8: mode must be 0 (nearest) or 2 (cubic), got %d", mode);
    hipLaunchKernelGGL(resize_u8_kernel, GRID_FOR((long)B * H * W), dim3(256), 0, S_(stream), packed, offs, hw, B, C, H, W, mode == 2 ? 1 : 0, out);
    TVL_LAUNCH_CHECK("tvl_resize_u8");
    return 0;
}

// Fused per-sample augmentation + normalisation.  img [B, H, W, 3] uint8; mask [B, H, W] uint8 or null; params [B, 8] = inverse affine
// (dst -> src, 2 x 3 row-major), alpha, beta; flags [B]: bit 0 = warp, bit 1 = brightness / contrast.  out_img [B, 3, H, W] fp32 =
// (v / 255 - mean) / std; out_mask [B, 1, H, W] fp32 = m / 255.  Replaces Affine + RandomBrightnessContrast + Normalize + ToTensorV2
// (reference configs/experiment/coop/clipseg.yaml:85-120); PadIfNeeded / CropNonEmptyMaskIfExists are identities at the resized size.
extern "C" int tvl_augment_u8(const uint8_t* img, const uint8_t* mask, const float* params, const int32_t* flags, const float* mean3,
                              const float* std3, float* out_img, float* out_mask, int32_t B, int32_t H, int32_t W, tvlStream_t stream) {
    TVL_REQUIRE(img && params && flags && mean3 && std3 && out_img && B > 0 && H > 0 && W > 0, "tvl_augment_u8: bad arguments");
    TVL_REQUIRE(!mask == !out_mask, "tvl_augment_u8: mask and out_mask go together");
    TVL_REQUIRE(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "tvl_augment_u8: std must be positive");
    hipLaunchKernelGGL(augment_u8_kernel, GRID_FOR((long)B * H * W), dim3(256), 0, S_(stream), img, mask, params, flags, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2], out_img, out_mask, B, H, W);
    TVL_LAUNCH_CHECK("tvl_augment_u8");
    return 0;
}
