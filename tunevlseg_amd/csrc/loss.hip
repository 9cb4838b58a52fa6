// DiceCE loss statistics + integer confusion counts in one pass over logits/target, and the
// loss gradient in a second pass.  HBM-bound: 8 bytes read per pixel (fwd), 8 read + 4 written (bwd).
//
// Semantics restated from monai.losses.DiceCELoss(sigmoid=True, lambda_dice, lambda_ce) for one
// channel and torchmetrics Dice(average="samples") / JaccardIndex(task="binary")
// (call sites: reference configs/model/vpt_clipseg.yaml:21-25, image_text_mask_module.py:87-107,284-302).
#include "common.h"

namespace {

// grid (chunks, B).  Float sums: every workgroup writes its four partial sums to `part` [B][chunks][4]; dicece_finish_kernel adds
// them in chunk order, so the loss is bitwise reproducible (an atomicAdd(double) here made its last bits depend on arrival order).
// The integer counts stay on atomics: integer addition is exact in any order.
__global__ __launch_bounds__(256) void dicece_stats_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                           double* __restrict__ part, long long* __restrict__ isum,
                                                           uint8_t* __restrict__ label, long N, float thr) {
    TVL_KERNEL_ENTRY();
    __shared__ double sf[4][4];
    __shared__ long long si[4][4];
    const int b = blockIdx.y;
    const float* lg = logits + (long)b * N;
    const float* tg = target + (long)b * N;
    float s_pt = 0.f, s_p = 0.f, s_t = 0.f, s_bce = 0.f;
    int tp = 0, fp = 0, fn = 0, tn = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
        const float x = lg[i], t = tg[i];
        const float e = expf(-fabsf(x));
        // sigmoid without overflow: x >= 0 -> 1/(1+e), x < 0 -> e/(1+e)
        const float pr = (x >= 0.f ? 1.0f : e) / (1.0f + e);
        s_pt += pr * t;
        s_p += pr;
        s_t += t;
        s_bce += fmaxf(x, 0.f) - x * t + log1pf(e);
        const bool lab = pr > thr;
        const bool pos = ((long long)t) != 0;  // targets = mask.long()
        tp += lab && pos;
        fp += lab && !pos;
        fn += !lab && pos;
        tn += !lab && !pos;
        if (label) label[(long)b * N + i] = lab ? 1 : 0;
    }
    double d0 = wave_sum_d((double)s_pt), d1 = wave_sum_d((double)s_p), d2 = wave_sum_d((double)s_t), d3 = wave_sum_d((double)s_bce);
    long long i0 = wave_sum_ll(tp), i1 = wave_sum_ll(fp), i2 = wave_sum_ll(fn), i3 = wave_sum_ll(tn);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sf[w][0] = d0; sf[w][1] = d1; sf[w][2] = d2; sf[w][3] = d3;
        si[w][0] = i0; si[w][1] = i1; si[w][2] = i2; si[w][3] = i3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        part[((long)b * gridDim.x + blockIdx.x) * 4 + k] = (sf[0][k] + sf[1][k]) + (sf[2][k] + sf[3][k]);
        atomicAdd(reinterpret_cast<unsigned long long*>(&isum[b * 4 + k]),
                  (unsigned long long)((si[0][k] + si[1][k]) + (si[2][k] + si[3][k])));
    }
}

// one wave per (sample, quantity): the chunks side by side on the lanes, then a fixed-order wave reduction (a thread per output walked up to 256
// partial sums one dependent add after the other: 18.5 us for 128 outputs)
__global__ __launch_bounds__(64) void dicece_finish_kernel(const double* __restrict__ part, double* __restrict__ fsum, int B, int chunks) {
    TVL_KERNEL_ENTRY();
    const int i = blockIdx.x;  // (b, k)
    const int b = i >> 2, k = i & 3, lane = threadIdx.x;
    double s = 0.0;
    for (int c = lane; c < chunks; c += 64) s += part[((long)b * chunks + c) * 4 + k];
    s = wave_sum_d(s);
    if (lane == 0) fsum[i] = s;
}

__global__ __launch_bounds__(256) void dicece_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                         const double* __restrict__ fsum, float* __restrict__ dlogits, int B, long N,
                                                         float lambda_dice, float lambda_ce, float snr, float sdr,
                                                         const float* __restrict__ gscale) {
    TVL_KERNEL_ENTRY();
    const int b = blockIdx.y;
    const float gs = gscale ? gscale[0] : 1.0f;
    const double I = fsum[b * 4 + 0], D = fsum[b * 4 + 1] + fsum[b * 4 + 2];
    // d/dp [1 - (2I+snr)/(D+sdr)] = ((2I+snr) - 2t(D+sdr)) / (D+sdr)^2
    const float num = (float)(2.0 * I + snr), den = (float)(D + sdr);
    const float inv_den2 = 1.0f / (den * den);
    const float kd = gs * lambda_dice / (float)B;
    const float kc = gs * lambda_ce / ((float)B * (float)N);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
        const long j = (long)b * N + i;
        const float x = logits[j], t = target[j];
        const float e = expf(-fabsf(x));
        const float pr = (x >= 0.f ? 1.0f : e) / (1.0f + e);
        const float dp = pr * (1.0f - pr);
        dlogits[j] = kd * dp * (num - 2.0f * t * den) * inv_den2 + kc * (pr - t);
    }
}

// loss = lambda_dice * mean_b [1 - (2 I_b + snr) / (D_b + sdr)] + lambda_ce * (sum_b bce_b) / (B N), float64 inside, one workgroup, fixed
// summation order (thread-strided partial sums, then a shared-memory tree): bitwise reproducible like fsum itself
__global__ __launch_bounds__(256) void dicece_loss_kernel(const double* __restrict__ fsum, float* __restrict__ loss, int B, double n_pix,
                                                          double lambda_dice, double lambda_ce, double snr, double sdr, int* __restrict__ nonfinite) {
    TVL_KERNEL_ENTRY();
    __shared__ double sd[256], sb[256];
    double d = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        d += 1.0 - (2.0 * fsum[b * 4 + 0] + snr) / (fsum[b * 4 + 1] + fsum[b * 4 + 2] + sdr);
        c += fsum[b * 4 + 3];
    }
    sd[threadIdx.x] = d;
    sb[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sd[threadIdx.x] += sd[threadIdx.x + w];
            sb[threadIdx.x] += sb[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l = (float)(lambda_dice * (sd[0] / (double)B) + lambda_ce * (sb[0] / ((double)B * n_pix)));
        loss[0] = l;
        if (nonfinite && !(fabsf(l) <= 3.4028235e38f)) atomicAdd(nonfinite, 1);   // sticky device-side flag: NaN / Inf losses are counted, never waited for
    }
}

}  // namespace

static long dicece_chunks(long N) {
    long chunks = (N + 256 * 8 - 1) / (256 * 8);
    return chunks > 256 ? 256 : chunks;
}

extern "C" int64_t tvl_dicece_work_doubles(int32_t B, int64_t N) { return B > 0 && N > 0 ? (int64_t)B * dicece_chunks((long)N) * 4 : -1; }

extern "C" int tvl_dicece_stats(const float* logits, const float* target, double* fsum, int64_t* isum, uint8_t* label, double* work,
                                int32_t B, int64_t N, float thr, tvlStream_t stream) {
    TVL_REQUIRE(logits && target && fsum && isum && work && B > 0 && N > 0, "tvl_dicece_stats: bad arguments");
    TVL_REQUIRE(B <= 65535, "tvl_dicece_stats: batch too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(isum, 0, sizeof(int64_t) * 4 * B, s);
    TVL_REQUIRE(e == hipSuccess, "tvl_dicece_stats: memset failed: %s", hipGetErrorString(e));
    const long chunks = dicece_chunks((long)N);
    hipLaunchKernelGGL(dicece_stats_kernel, dim3((unsigned)chunks, B), dim3(256), 0, s, logits, target, work,
                       reinterpret_cast<long long*>(isum), label, (long)N, thr);
    hipLaunchKernelGGL(dicece_finish_kernel, dim3((unsigned)(B * 4)), dim3(64), 0, s, (const double*)work, fsum, B, (int)chunks);
    TVL_LAUNCH_CHECK("tvl_dicece_stats");
    return 0;
}

extern "C" int tvl_dicece_bwd(const float* logits, const float* target, const double* fsum, float* dlogits, int32_t B, int64_t N,
                              float lambda_dice, float lambda_ce, float smooth_nr, float smooth_dr, const float* gscale,
                              tvlStream_t stream) {
    TVL_REQUIRE(logits && target && fsum && dlogits && B > 0 && N > 0, "tvl_dicece_bwd: bad arguments");
    TVL_REQUIRE(B <= 65535, "tvl_dicece_bwd: batch too large");
    long chunks = (N + 256 * 8 - 1) / (256 * 8);
    if (chunks > 256) chunks = 256;
    hipLaunchKernelGGL(dicece_bwd_kernel, dim3((unsigned)chunks, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, target,
                       fsum, dlogits, B, (long)N, lambda_dice, lambda_ce, smooth_nr, smooth_dr, gscale);
    TVL_LAUNCH_CHECK("tvl_dicece_bwd");
    return 0;
}

extern "C" int tvl_dicece_loss(const double* fsum, float* loss, int32_t B, int64_t N, float lambda_dice, float lambda_ce,
                               float smooth_nr, float smooth_dr, int32_t* nonfinite, tvlStream_t stream) {
    TVL_REQUIRE(fsum && loss && B > 0 && N > 0, "tvl_dicece_loss: bad arguments");
    hipLaunchKernelGGL(dicece_loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), fsum, loss, (int)B, (double)N,
                       (double)lambda_dice, (double)lambda_ce, (double)smooth_nr, (double)smooth_dr, nonfinite);
    TVL_LAUNCH_CHECK("tvl_dicece_loss");
    return 0;
}
