// "h2": the two-piece fp16 operand format and its GEMM entry point (the kernel is gemm_tp3_kernel.h with NP = 2).
//
// x * s = h0 + h1 with h0 = fp16(x s), h1 = fp16(x s - h0): 11 + 11 significand bits; the three products h0 h0 + h0 h1 + h1 h0 on
// v_mfma_f32_32x32x16_f16 (fp16 products are exact in fp32) leave a relative error of ~2^-22 per product, below the fp32 rounding of
// the accumulation (tools/piece_schemes.py: 2.5e-9 of sum|a||b| at K = 3072; bf16 x 6: 1.8e-9; torch fp32: 1.0e-8) -- at HALF the
// MFMAs and two thirds of the bytes of the three-piece bf16 format.  fp16's exponent range is the price: the second piece of a
// scaled value below 0.125 is subnormal, so every row (or tensor) is multiplied by an exact power of two s that puts its largest
// magnitude in [2^13, 2^14) -- far from 65504, and anything down to 2^-17 of the maximum keeps both pieces normal.  The GEMM's
// epilogue undoes the scales: per-row factors of A (a_scale[m] = 1 / s_m) and alpha (= 1 / s of the B tensor), both exact.
//
// Image layout = tp3's with two pieces: block (rb, kb) at ((rb * K/16 + kb) * 2 + piece) * 1024, same lane order inside a piece.
#include "gemm_h2m_kernel.h"

namespace {

constexpr int BLK2 = 2 * PIECE;

__device__ __forceinline__ float pow2_scale_inv(float amax) { return h2::inv_scale_of(amax); }

// whole-tensor maximum -> one inverse scale (two stages: block maxima by atomicMax on the bit pattern of a non-negative float).
// Rows that follow one another in memory (ldx == K) are read as one flat array, four independent 16-byte loads per thread and trip.
__device__ __forceinline__ float amax4(float m, const float4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// MASK kernels pack x * (mask > 0): the ReLU gate of a data gradient applied on the way into the image (mask = the layer's output)
__device__ __forceinline__ float4 gate4(const float4 v, const float4 g) {
    return make_float4(g.x > 0.f ? v.x : 0.f, g.y > 0.f ? v.y : 0.f, g.z > 0.f ? v.z : 0.f, g.w > 0.f ? v.w : 0.f);
}
template <bool MASK>
__device__ __forceinline__ float4 ld4(const float* __restrict__ x, const float* __restrict__ mask, long off, long moff) {
    const float4 v = *reinterpret_cast<const float4*>(x + off);
    if constexpr (MASK) return gate4(v, *reinterpret_cast<const float4*>(mask + moff));
    return v;
}
template <bool MASK>
__global__ __launch_bounds__(256) void h2_absmax_kernel(const float* __restrict__ x, long ldx, long rows, int K, unsigned* __restrict__ bits,
                                                        const float* __restrict__ mask, long ldm) {
    TVL_KERNEL_ENTRY();
    float m = 0.f;
    const long total = rows * (K >> 2);
    const long stride = (long)gridDim.x * 256;
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (ldx == K && (!MASK || ldm == K)) {
        for (; t + 3 * stride < total; t += 4 * stride) {
            const float4 a = ld4<MASK>(x, mask, 4 * t, 4 * t), b = ld4<MASK>(x, mask, 4 * (t + stride), 4 * (t + stride)),
                         c = ld4<MASK>(x, mask, 4 * (t + 2 * stride), 4 * (t + 2 * stride)), d = ld4<MASK>(x, mask, 4 * (t + 3 * stride), 4 * (t + 3 * stride));
            m = amax4(amax4(amax4(amax4(m, a), b), c), d);
        }
        for (; t < total; t += stride) m = amax4(m, ld4<MASK>(x, mask, 4 * t, 4 * t));
    } else {
        const unsigned k4 = (unsigned)(K >> 2);
        for (; t < total; t += stride) {
            const unsigned r = (unsigned)t / k4;   // total < 2^32 (host check)
            const unsigned c = ((unsigned)t - r * k4) * 4;
            m = amax4(m, ld4<MASK>(x, mask, (long)r * ldx + c, (long)r * ldm + c));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // one atomic per workgroup, and only when it can raise the maximum: thousands of atomics on one address serialise in L2
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        const unsigned mb = __builtin_bit_cast(unsigned, m);
        if (mb > __hip_atomic_load(bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bits, mb);
    }
}
// one thread = 8 consecutive k of one row -> one 16-byte store per piece (same mapping as tp3_pack_kernel)
template <bool MASK = false>
__device__ __forceinline__ void h2_pack8(const float* __restrict__ xr, float s, unsigned char* __restrict__ o, const float* __restrict__ mr = nullptr) {
    const float4 a = ld4<MASK>(xr, mr, 0, 0), b = ld4<MASK>(xr, mr, 4, 4);
    const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
    uint4 pl[2];
    h2::split8(v, pl);
    *reinterpret_cast<uint4*>(o) = pl[0];
    *reinterpret_cast<uint4*>(o + PIECE) = pl[1];
}
// (tensor mode: `bits` = the maximum found by h2_absmax_kernel; the first thread publishes the inverse scale)
template <bool MASK>
__global__ __launch_bounds__(256) void h2_pack_kernel(const float* __restrict__ x, long ldx, long rows, int K, const unsigned* __restrict__ bits,
                                                      float* __restrict__ inv_scale, unsigned char* __restrict__ out, long rows_padded,
                                                      const float* __restrict__ mask, long ldm) {
    TVL_KERNEL_ENTRY();
    const unsigned KB = (unsigned)(K >> 4);
    const long total = rows_padded * (K >> 3);   // blocks * 64 < 2^31 * 64 (host check on the block count)
    const float inv_all = pow2_scale_inv(__builtin_bit_cast(float, bits[0]));
    const float s_all = 1.0f / inv_all;   // a power of two: exact
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[0] = inv_all;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const unsigned blk = (unsigned)(t >> 6);
        const unsigned rb = blk / KB;
        const unsigned kb = blk - rb * KB;
        const int r = lane & 31, h = lane >> 5;
        const long row = (long)rb * 32 + r;
        unsigned char* o = out + (long)blk * BLK2 + lane * 16;
        if (row < rows) {
            h2_pack8<MASK>(x + row * ldx + kb * 16 + h * 8, s_all, o, MASK ? mask + row * ldm + kb * 16 + h * 8 : nullptr);
        } else {
            *reinterpret_cast<uint4*>(o) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(o + PIECE) = make_uint4(0, 0, 0, 0);
        }
    }
}

// per-row mode in one pass over HBM: a workgroup owns one 32-row block; its four waves first reduce eight rows each (maximum -> scale,
// L2 norm), then all 256 threads pack the block, whose second read comes from L2
template <bool MASK>
__global__ __launch_bounds__(256) void h2_rowpack_kernel(const float* __restrict__ x, long ldx, long rows, int K, float* __restrict__ inv_scale,
                                                         float* __restrict__ row_norm, unsigned char* __restrict__ out, const float* __restrict__ mask,
                                                         long ldm) {
    TVL_KERNEL_ENTRY();
    __shared__ float s_mul[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rb = blockIdx.x;
    // the wave's eight rows side by side: eight independent loads in flight per trip
    float m[8], ss[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = 0.f; ss[i] = 0.f; }
    const long row0 = rb * 32 + wave * 8;
    for (int c = lane * 4; c < K; c += 256) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long row = row0 + i < rows ? row0 + i : rows - 1;
            v[i] = ld4<MASK>(x, mask, row * ldx + c, row * ldm + c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            m[i] = amax4(m[i], v[i]);
            ss[i] += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = wave_max(m[i]); ss[i] = wave_sum(ss[i]); }   // DPP path (common.h): no LDS crossbar trips
    if (lane < 8 && row0 + lane < rows) {
        float mm = m[0], s2 = ss[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) { mm = lane == i ? m[i] : mm; s2 = lane == i ? ss[i] : s2; }
        const float inv = pow2_scale_inv(mm);
        inv_scale[row0 + lane] = inv;
        if (row_norm) row_norm[row0 + lane] = sqrtf(s2) * 1.0001f;
        s_mul[wave * 8 + lane] = 1.0f / inv;
    }
    __syncthreads();
    const int KB = K >> 4;
    const int r = lane & 31, h = lane >> 5;
    const long row = rb * 32 + r;
#pragma unroll 4
    for (int kb = wave; kb < KB; kb += 4) {
        unsigned char* o = out + (rb * KB + kb) * (long)BLK2 + lane * 16;
        if (row < rows) {
            h2_pack8<MASK>(x + row * ldx + kb * 16 + h * 8, s_mul[r], o, MASK ? mask + row * ldm + kb * 16 + h * 8 : nullptr);
        } else {
            *reinterpret_cast<uint4*>(o) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(o + PIECE) = make_uint4(0, 0, 0, 0);
        }
    }
}

// TVL_GEMM_M16=0: the first-generation ring on v_mfma_f32_32x32x16_f16 (A/B switch; default = the 16x16x32 kernels, gemm_h2m_kernel.h)
bool use_m16() {
    static const bool on = !(getenv("TVL_GEMM_M16") && getenv("TVL_GEMM_M16")[0] == '0');
    return on;
}

int launch_h2(const Tp3Params& p, int bm, int epi, hipStream_t s) {
#ifdef TVL_EXPERIMENTS
    if (bm == 128) return tvl_gemm_h2_w4(&p, bm, epi, s);
#else
    if (bm == 128 || bm == 2564 || bm == 1924 || bm == 2565 || bm == 1925) { tvl_set_error("tvl_gemm_h2: tile_m = %d selects an experiment kernel (build with make EXPERIMENTS=1)", bm); return 1; }
#endif
    if ((bm == 2566 || (bm == 256 && use_m16())) && p.K >= 96) return tvl_gemm_h2m_t256(&p, epi, s);
    if ((bm == 1926 || (bm == 192 && use_m16())) && p.K >= 96) return tvl_gemm_h2m_t192(&p, epi, s);
    if (bm == 2566 || bm == 1926) bm = bm / 10;
    if (bm == 2560 || bm == 1920) bm = bm / 10;   // explicit first-generation tiles (tools/bench_layer_gemms.py)
#ifdef TVL_EXPERIMENTS
    if (bm == 2564 || bm == 1924) return tvl_gemm_h2_ns4(&p, bm / 10, epi, s);
    if (bm == 2565 || bm == 1925) return tvl_gemm_h2_ns5(&p, bm / 10, epi, s);
#endif
    // the epilogues the vision tower's forward needs (QKV -> tp3, fc1 -> QuickGELU -> tp3 + z) + the plain ones; others: generic
    if (bm == 256) {
        switch (epi) {
            case E_BIAS | E_QGELU | E_PRE | E_TP3 | E_RSCALE: return launch<256, 256, 2, E_BIAS | E_QGELU | E_PRE | E_TP3 | E_RSCALE, 2>(p, s);
            case E_BIAS | E_QGELU | E_TP3 | E_RSCALE: return launch<256, 256, 2, E_BIAS | E_QGELU | E_TP3 | E_RSCALE, 2>(p, s);
            case E_DQGELU | E_TP3 | E_RSCALE: return launch<256, 256, 2, E_DQGELU | E_TP3 | E_RSCALE, 2>(p, s);   // dz of the backward
            case E_BIAS | E_QGELU | E_PRE | E_RSCALE | E_H2OUT: return launch<256, 256, 2, E_BIAS | E_QGELU | E_PRE | E_RSCALE | E_H2OUT, 2>(p, s);   // fc1 -> h2
            case E_BIAS | E_QGELU | E_RSCALE | E_H2OUT: return launch<256, 256, 2, E_BIAS | E_QGELU | E_RSCALE | E_H2OUT, 2>(p, s);
            case E_DQGELU | E_RSCALE | E_H2OUT: return launch<256, 256, 2, E_DQGELU | E_RSCALE | E_H2OUT, 2>(p, s);                 // dz -> h2
            case E_BIAS | E_F32 | E_RSCALE: return launch<256, 256, 2, E_BIAS | E_F32 | E_RSCALE, 2>(p, s);                       // frozen Linear / 1x1 conv
            case E_BIAS | E_RELU | E_F32 | E_RSCALE: return launch<256, 256, 2, E_BIAS | E_RELU | E_F32 | E_RSCALE, 2>(p, s);     // ... + ReLU (folded BN)
            case E_F32 | E_RSCALE: return launch<256, 256, 2, E_F32 | E_RSCALE, 2>(p, s);                                         // its data gradient
            default: return launch<256, 256, 2, -1, 2>(p, s);
        }
    }
    switch (epi) {
        case E_BIAS | E_TP3 | E_RSCALE: return launch<192, 256, 3, E_BIAS | E_TP3 | E_RSCALE, 2>(p, s);
        case E_TP3 | E_RSCALE: return launch<192, 256, 3, E_TP3 | E_RSCALE, 2>(p, s);                                 // dO of the backward
        case E_BIAS | E_RSCALE | E_H2OUT: return launch<192, 256, 3, E_BIAS | E_RSCALE | E_H2OUT, 2>(p, s);           // qkv -> h2 (attention on fp16 pieces)
        case E_RSCALE | E_H2OUT: return launch<192, 256, 3, E_RSCALE | E_H2OUT, 2>(p, s);                             // dO -> h2
        case E_BIAS | E_F32 | E_RSCALE: return launch<192, 256, 3, E_BIAS | E_F32 | E_RSCALE, 2>(p, s);
        case E_F32 | E_RSCALE: return launch<192, 256, 3, E_F32 | E_RSCALE, 2>(p, s);
        case E_BIAS | E_RES | E_F32 | E_RSCALE: return launch<192, 256, 3, E_BIAS | E_RES | E_F32 | E_RSCALE, 2>(p, s);   // out_proj, fc2
        case E_BIAS | E_RELU | E_F32 | E_RSCALE: return launch<192, 256, 3, E_BIAS | E_RELU | E_F32 | E_RSCALE, 2>(p, s);  // frozen Linear / 1x1 conv + ReLU
        default: return launch<192, 256, 3, -1, 2>(p, s);
    }
}

// 3x3 conv as an implicit GEMM (the A fragments are gathered tap by tap from the pixel matrix' image): generic epilogue (bias + ReLU of the
// folded eval BatchNorm) and the plain one of the data gradient
int launch_h2_conv(const Tp3Params& p, int bm, int epi, hipStream_t s) {
    if (use_m16() && p.K >= 96) return bm == 256 ? tvl_gemm_h2m_conv_t256(&p, epi, s) : tvl_gemm_h2m_conv_t192(&p, epi, s);
    constexpr int CONV_FWD = E_BIAS | E_RELU | E_F32 | E_RSCALE;   // conv + folded BN + ReLU
    if (bm == 256) {
        if (epi == (E_F32 | E_RSCALE)) return launch<256, 256, 2, E_F32 | E_RSCALE, 2, false, true>(p, s);
        if (epi == CONV_FWD) return launch<256, 256, 2, CONV_FWD, 2, false, true>(p, s);
        return launch<256, 256, 2, -1, 2, false, true>(p, s);
    }
    if (epi == (E_F32 | E_RSCALE)) return launch<192, 256, 3, E_F32 | E_RSCALE, 2, false, true>(p, s);
    if (epi == CONV_FWD) return launch<192, 256, 3, CONV_FWD, 2, false, true>(p, s);
    return launch<192, 256, 3, -1, 2, false, true>(p, s);
}

}  // namespace

// an aux_blocked buffer covers whole 256 x 256 tiles and exists only where tvl_gemm_h2_out picks the 256-row tile of the 16x16x32 ring
static bool aux_blocked_ok(long M, long N, long K) {
    if (M <= 0 || N <= 0 || N % 256 != 0 || !use_m16() || K < 96) return false;
    const long t256 = ((M + 255) / 256) * (N / 256), t192 = ((M + 191) / 192) * (N / 256);
    return ((t256 + 255) / 256) * 256 <= ((t192 + 255) / 256) * 192;
}
extern "C" int64_t tvl_gemm_aux_floats(int64_t M, int64_t N) {
    return aux_blocked_ok(M, N, 96) ? ((M + 255) / 256) * 256 * N : -1;
}

extern "C" int64_t tvl_h2_bytes(int64_t rows, int32_t K) {
    if (rows <= 0 || K <= 0 || K % 16 != 0) return -1;
    return ((rows + 31) / 32) * (int64_t)(K / 16) * BLK2;
}

static int h2_pack_impl(const float* x, int64_t ldx, const float* mask, int64_t ldm, int64_t rows, int32_t K, void* out, float* inv_scale,
                        float* row_norm, int32_t per_row, void* work, tvlStream_t stream) {
    TVL_REQUIRE(x && out && inv_scale && (per_row || work), "tvl_h2_pack: null pointer");
    TVL_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ldx >= K && ldx % 4 == 0, "tvl_h2_pack: need K %% 16 == 0, ldx >= K, ldx %% 4 == 0");
    TVL_REQUIRE(tvl_aligned16(out) && tvl_aligned16(x), "tvl_h2_pack: operands must be 16-byte aligned");
    TVL_REQUIRE(!mask || (ldm >= K && ldm % 4 == 0 && tvl_aligned16(mask)), "tvl_h2_pack_masked: mask needs ldm >= K, ldm %% 4 == 0, 16-byte alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long rp = (rows + 31) / 32 * 32;
    TVL_REQUIRE(rp / 32 * (K / 16) < (1ll << 31) && rows * (K / 4) < (1ll << 32), "tvl_h2_pack: image too large (%ld blocks)", rp / 32 * (K / 16));
    unsigned char* o = reinterpret_cast<unsigned char*>(out);
    if (per_row) {
        if (mask) hipLaunchKernelGGL(h2_rowpack_kernel<true>, dim3((unsigned)(rp / 32)), dim3(256), 0, s, x, (long)ldx, (long)rows, K, inv_scale, row_norm, o, mask, (long)ldm);
        else hipLaunchKernelGGL(h2_rowpack_kernel<false>, dim3((unsigned)(rp / 32)), dim3(256), 0, s, x, (long)ldx, (long)rows, K, inv_scale, row_norm, o, mask, (long)ldm);
    } else {
        hipError_t e = hipMemsetAsync(work, 0, 4, s);
        TVL_REQUIRE(e == hipSuccess, "tvl_h2_pack: memset failed: %s", hipGetErrorString(e));
        long nb = (rows * (K / 4) + 1023) / 1024;
        nb = nb > 1024 ? 1024 : nb;   // (every workgroup ends in an atomicMax on ONE address: 4096 of them serialised for ~25 us)
        unsigned* bits = reinterpret_cast<unsigned*>(work);
        if (mask) hipLaunchKernelGGL(h2_absmax_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, x, (long)ldx, (long)rows, K, bits, mask, (long)ldm);
        else hipLaunchKernelGGL(h2_absmax_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, x, (long)ldx, (long)rows, K, bits, mask, (long)ldm);
        nb = (rp * (K / 8) + 255) / 256;
        nb = nb > 1048576 ? 1048576 : nb;
        if (mask) hipLaunchKernelGGL(h2_pack_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, x, (long)ldx, (long)rows, K, (const unsigned*)bits, inv_scale, o, rp, mask, (long)ldm);
        else hipLaunchKernelGGL(h2_pack_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, x, (long)ldx, (long)rows, K, (const unsigned*)bits, inv_scale, o, rp, mask, (long)ldm);
    }
    TVL_LAUNCH_CHECK("tvl_h2_pack");
    return 0;
}

// fp32 [rows, K] -> h2 image + inverse scale(s): per_row != 0 -> inv_scale[rows] (activations: A operand), else inv_scale[1]
// (a frozen weight or a conv's pixel matrix).  `work` = 4 bytes of device scratch (per-tensor mode only).
extern "C" int tvl_h2_pack(const float* x, int64_t ldx, int64_t rows, int32_t K, void* out, float* inv_scale, float* row_norm, int32_t per_row,
                           void* work, tvlStream_t stream) {
    return h2_pack_impl(x, ldx, nullptr, 0, rows, K, out, inv_scale, row_norm, per_row, work, stream);
}
// ... of x * (mask > 0): the ReLU gate of a data gradient (mask = the layer's output y, reference F.relu backward) applied while packing, so
// the gated gradient never exists in fp32
extern "C" int tvl_h2_pack_masked(const float* x, int64_t ldx, const float* mask, int64_t ldm, int64_t rows, int32_t K, void* out, float* inv_scale,
                                  float* row_norm, int32_t per_row, void* work, tvlStream_t stream) {
    TVL_REQUIRE(mask != nullptr, "tvl_h2_pack_masked: null mask");
    return h2_pack_impl(x, ldx, mask, ldm, rows, K, out, inv_scale, row_norm, per_row, work, stream);
}

// rows b * T + row0 .. + n - 1 (b = 0 .. B - 1) of an h2 image := 0 (both pieces): the gradient cut of an in-place row overwrite
// (tvl_rows_grad, zero_src) applied to the operand image that travels with the gradient, so the image need not be packed again
__global__ __launch_bounds__(256) void h2_zero_rows_kernel(unsigned char* __restrict__ img, int K, int B, int T, int row0, int n) {
    TVL_KERNEL_ENTRY();
    const int kb2 = K >> 3;                       // 16-byte units per row and piece
    const long total = (long)B * n * kb2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int u = (int)(i % kb2);
        const long rj = i / kb2;
        const long m = (rj / n) * T + row0 + (rj % n);
        unsigned char* o = img + ((m >> 5) * (K >> 4) + (u >> 1)) * (long)BLK2 + (((u & 1) * 32 + (int)(m & 31)) * 16);
        *reinterpret_cast<uint4*>(o) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(o + 1024) = make_uint4(0u, 0u, 0u, 0u);
    }
}

extern "C" int tvl_h2_zero_rows(void* img, int32_t K, int32_t B, int32_t T, int32_t row0, int32_t n, tvlStream_t stream) {
    TVL_REQUIRE(img && K > 0 && K % 16 == 0 && B > 0 && T > 0 && n > 0 && row0 >= 0 && row0 + n <= T, "tvl_h2_zero_rows: bad arguments (K=%d B=%d T=%d row0=%d n=%d)", K, B, T, row0, n);
    const long total = (long)B * n * (K / 8);
    long nb = (total + 255) / 256;
    nb = nb > 1024 ? 1024 : nb;   // (every workgroup ends in an atomicMax on ONE address: 4096 of them serialised for ~25 us)
    hipLaunchKernelGGL(h2_zero_rows_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<unsigned char*>(img), K, B, T, row0, n);
    TVL_LAUNCH_CHECK("tvl_h2_zero_rows");
    return 0;
}

// max |x| over [rows, K] as the bit pattern of a non-negative float in bits[0] (4 bytes, zeroed here): the first half of the per-tensor
// tvl_h2_pack, for producers that write their own h2 image (tvl_bilinear_up_h2)
extern "C" int tvl_h2_absmax(const float* x, int64_t ldx, int64_t rows, int32_t K, void* bits, tvlStream_t stream) {
    TVL_REQUIRE(x && bits && rows > 0 && K > 0 && K % 4 == 0 && ldx >= K && ldx % 4 == 0 && tvl_aligned16(x), "tvl_h2_absmax: need K %% 4 == 0, ldx >= K, ldx %% 4 == 0, alignment");
    TVL_REQUIRE(rows * (K / 4) < (1ll << 32), "tvl_h2_absmax: tensor too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(bits, 0, 4, s);
    TVL_REQUIRE(e == hipSuccess, "tvl_h2_absmax: memset failed: %s", hipGetErrorString(e));
    long nb = (rows * (K / 4) + 1023) / 1024;
    nb = nb > 1024 ? 1024 : nb;   // (every workgroup ends in an atomicMax on ONE address: 4096 of them serialised for ~25 us)
    hipLaunchKernelGGL(h2_absmax_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, x, (long)ldx, (long)rows, K, reinterpret_cast<unsigned*>(bits), (const float*)nullptr, 0L);
    TVL_LAUNCH_CHECK("tvl_h2_absmax");
    return 0;
}

// epilogue(alpha * a_row_scale[m] * A . B^T) over h2 operands; same argument block as tvl_gemm_tp3 (A / B are h2 images; C_tp3, if given,
// is still a tp3 image: the consumers of this round read three bf16 pieces).  a_row_scale: [M] inverse scales of A's rows (or null);
// alpha carries the inverse scale of B.
static int gemm_h2_impl(const tvlGemmTp3Args* a, const float* a_row_scale, void* c_h2, const float* out_row_norm, float out_mul, float out_add,
                        float* out_inv_scale, int out_per_tensor, tvlStream_t stream, const tvlConvGeom* conv = nullptr) {
    TVL_REQUIRE(a != nullptr, "tvl_gemm_h2: null args");
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "tvl_gemm_h2: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    TVL_REQUIRE(a->K % 32 == 0 && a->K >= 64 && a->N % 16 == 0, "tvl_gemm_h2: need K %% 32 == 0, K >= 64, N %% 16 == 0 (K=%d N=%d)", a->K, a->N);
    TVL_REQUIRE(a->A && a->B && (a->C || a->C_tp3 || c_h2), "tvl_gemm_h2: null operand");
    TVL_REQUIRE(!c_h2 || (out_row_norm && out_inv_scale && tvl_aligned16(c_h2)), "tvl_gemm_h2: an h2 output needs row norms, a place for its inverse scales and 16-byte alignment");
    TVL_REQUIRE(tvl_aligned16(a->A) && tvl_aligned16(a->B), "tvl_gemm_h2: operands must be 16-byte aligned");
    TVL_REQUIRE(a->a_rows >= a->M && a->b_rows >= a->N, "tvl_gemm_h2: operand images hold fewer rows than M / N");
    TVL_REQUIRE(!a->C || (a->ldc >= a->N && a->ldc % 4 == 0 && tvl_aligned16(a->C)), "tvl_gemm_h2: C needs ldc >= N, ldc %% 4 == 0, 16-byte alignment");
    TVL_REQUIRE(!a->pre_out || (a->ldc >= a->N && a->ldc % 4 == 0 && tvl_aligned16(a->pre_out)), "tvl_gemm_h2: pre_out shares ldc and needs 16-byte alignment");
    TVL_REQUIRE(!a->C_tp3 || tvl_aligned16(a->C_tp3), "tvl_gemm_h2: C_tp3 must be 16-byte aligned");
    TVL_REQUIRE(!a->residual || (a->ldr >= a->N && a->ldr % 4 == 0 && tvl_aligned16(a->residual)), "tvl_gemm_h2: residual needs ldr >= N, %% 4, alignment");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N && a->ld_aux % 4 == 0 && tvl_aligned16(a->dact_aux)), "tvl_gemm_h2: dact needs an aligned dact_aux");
    TVL_REQUIRE(!a->bias || tvl_aligned16(a->bias), "tvl_gemm_h2: bias must be 16-byte aligned");
    Tp3Params p = {};
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = reinterpret_cast<const unsigned char*>(a->A); p.a_rb = (int)((a->a_rows + 31) / 32);
    p.B = reinterpret_cast<const unsigned char*>(a->B); p.b_rb = (int)((a->b_rows + 31) / 32);
    p.C = a->C; p.ldc = a->ldc; p.Cp = reinterpret_cast<unsigned char*>(a->C_tp3);
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha; p.a_scale = a_row_scale; p.a_sstride = (conv || a->a_scale_one) ? 0 : 1;
    p.aux_blocked = a->aux_blocked;
    TVL_REQUIRE(!a->aux_blocked || (c_h2 && !a->C && !a->C_tp3 && !a->residual && !conv && a->tile_m == 0 && aux_blocked_ok(a->M, a->N, a->K)),
                "tvl_gemm_h2: aux_blocked needs an image-only epilogue on the automatically chosen 256-row tile (M=%d N=%d K=%d)", a->M, a->N, a->K);
    p.work = reinterpret_cast<float*>(a->workspace); p.work_bytes = a->workspace ? a->workspace_bytes : 0;
    TVL_REQUIRE(!a->workspace || tvl_aligned16(a->workspace), "tvl_gemm_h2: workspace must be 16-byte aligned");
    p.Ch2 = reinterpret_cast<unsigned char*>(c_h2); p.out_norm = out_row_norm; p.out_mul = out_mul; p.out_add = out_add; p.out_inv = out_inv_scale; p.out_stride = out_per_tensor ? 0 : 1;
    int bm = a->tile_m;
    static const int stagger_us = getenv("TVL_GEMM_STAGGER_US") ? atoi(getenv("TVL_GEMM_STAGGER_US")) : 0;   // experiment knob
    p.stagger_ticks = stagger_us * 100;
    static const int f32_direct = getenv("TVL_GEMM_F32_DIRECT") ? atoi(getenv("TVL_GEMM_F32_DIRECT")) : 1;   // A/B switch (0 = through the LDS scratch)
    p.f32_direct = f32_direct;
    if (bm == 128 && !conv) {
    } else if (bm != 256 && bm != 192 && bm != 2564 && bm != 2565 && bm != 1924 && bm != 1925 && bm != 2566 && bm != 1926 && bm != 2560 && bm != 1920) {
        const long t256 = ((long)(a->M + 255) / 256) * ((a->N + 255) / 256), t192 = ((long)(a->M + 191) / 192) * ((a->N + 255) / 256);
        bm = ((t256 + 255) / 256) * 256 <= ((t192 + 255) / 256) * 192 ? 256 : 192;   // rounds x rows per tile; ties go to the larger tile
    }
    if (conv) {
        p.cH = conv->H; p.cW = conv->W; p.cC16 = conv->C / 16;
        p.a_rb = (int)((a->a_rows + 31) / 32);   // = the index of the appended zero block row
    }
#ifdef TVL_DIAGNOSTIC_KERNELS   // `make DIAG=1`: timing-only ablations of the plain fp32-output GEMM (WRONG results: bit 2 no DMA in the loop,
                               // bit 3 every DMA re-reads slab 0 -- L2-resident operands --, bit 4 no stores); tools/bench_gemm_ablate.py
    if (!conv && a->variant >= 32) p.pre_out = nullptr;   // (the stamp buffer travels in a->pre_out)
    if (!conv && a->variant && epi_code(p) == (E_F32 | E_RSCALE)) {
        if (a->variant >= 32) p.pre_out = a->pre_out;
        hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
        int drc = 1;
        if (bm == 256) {
            if (a->variant == 4) drc = launch<256, 256, 2 | 4, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 8) drc = launch<256, 256, 2 | 8, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 16) drc = launch<256, 256, 2 | 16, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 24) drc = launch<256, 256, 2 | 8 | 16, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 32) drc = launch<256, 256, 2 | 32, E_F32 | E_RSCALE, 2>(p, hs);   // in-kernel stamps into pre_out (12 uint64 per workgroup)
            else if (a->variant == 36) drc = launch<256, 256, 2 | 32 | 4, E_F32 | E_RSCALE, 2>(p, hs);
        } else {
            if (a->variant == 4) drc = launch<192, 256, 3 | 4, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 8) drc = launch<192, 256, 3 | 8, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 16) drc = launch<192, 256, 3 | 16, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 24) drc = launch<192, 256, 3 | 8 | 16, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 32) drc = launch<192, 256, 3 | 32, E_F32 | E_RSCALE, 2>(p, hs);
            else if (a->variant == 36) drc = launch<192, 256, 3 | 32 | 4, E_F32 | E_RSCALE, 2>(p, hs);
        }
        TVL_REQUIRE(drc == 0, "tvl_gemm_h2: unknown diagnostic variant %d", a->variant);
        TVL_LAUNCH_CHECK("tvl_gemm_h2(diag)");
        return 0;
    }
#endif
    const int rc = conv ? launch_h2_conv(p, bm, epi_code(p), reinterpret_cast<hipStream_t>(stream))
                        : launch_h2(p, bm, epi_code(p), reinterpret_cast<hipStream_t>(stream));
    TVL_REQUIRE(rc == 0, "tvl_gemm_h2: launch failed (dynamic LDS opt-in?)");
    TVL_LAUNCH_CHECK("tvl_gemm_h2");
    return 0;
}

// 3x3 / pad 1 / stride 1 conv of an NHWC pixel matrix as an implicit GEMM over h2 operands.  args->A = the h2 image of the pixel matrix
// [B*H*W, C] (ONE scale for the tensor: a_scale points at that one inverse scale) followed by one all-zero 32-row block (the source of the padding taps:
// the image holds tvl_h2_bytes(B*H*W, C) + tvl_h2_bytes(32, C) bytes); args->a_rows = args->M = B*H*W; args->B = the h2 image of the
// weights [N, 9*C], columns ordered (c / 16, ky, kx, c % 16) -- the nine taps of a 16-channel block are consecutive k-slabs, so their
// gathers re-read rows that are still in L2; args->K = 9*C.  Epilogue as tvl_gemm_h2 (fp32 output).  Needs C % 32 == 0.
extern "C" int tvl_conv3x3_h2(const tvlGemmTp3Args* a, const tvlConvGeom* g, const float* a_scale, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr && g != nullptr, "tvl_conv3x3_h2: null args");
    TVL_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->H < 32768 && g->W < 32768 && g->C > 0 && g->C % 32 == 0 && g->stride == 1,
                "tvl_conv3x3_h2: need C %% 32 == 0 and stride 1 (C=%d stride=%d)", g->C, g->stride);
    TVL_REQUIRE((int64_t)g->B * g->H * g->W == a->M && a->a_rows == a->M && a->K == 9 * g->C && (int64_t)a->M + 64 < (1ll << 31),
                "tvl_conv3x3_h2: M must be B*H*W (= a_rows) and K = 9*C (M=%d K=%d)", a->M, a->K);
    TVL_REQUIRE(a->C && !a->C_tp3 && a_scale, "tvl_conv3x3_h2: fp32 output only; the map's inverse scale is required");
    return gemm_h2_impl(a, a_scale, nullptr, nullptr, 0.f, 0.f, nullptr, 0, stream, g);
}

extern "C" int tvl_gemm_h2(const tvlGemmTp3Args* a, const float* a_row_scale, tvlStream_t stream) {
    return gemm_h2_impl(a, a_row_scale, nullptr, nullptr, 0.f, 0.f, nullptr, 0, stream);
}

// The same GEMM with its result ALSO (or only) written as an h2 image, i.e. as the A operand of the next h2 GEMM.  The producer of a
// row-scaled operand needs the row's magnitude before it writes the row; a GEMM workgroup sees 256 of its columns, so the scale comes
// from a bound instead: |out[m, n]| <= out_row_norm[m] * out_mul + out_add with out_row_norm = the L2 norms of A's rows (written by A's
// own producer), out_mul = max_n ||B row n||_2 (times the Lipschitz constant of the epilogue's activation), out_add = max |bias|.
// out_inv_scale[M] receives the inverse scales the consumer passes as its a_row_scale.
// out_per_tensor != 0: ONE scale for the whole output (out_row_norm[0] = the largest row norm of A, out_inv_scale[0]): what the attention
// kernels read (tvl_attn_h2_fwd / _bwd).
extern "C" int tvl_gemm_h2_out(const tvlGemmTp3Args* a, const float* a_row_scale, void* c_h2, const float* out_row_norm, float out_mul, float out_add,
                               float* out_inv_scale, int32_t out_per_tensor, tvlStream_t stream) {
    TVL_REQUIRE(c_h2 != nullptr, "tvl_gemm_h2_out: null h2 output");
    return gemm_h2_impl(a, a_row_scale, c_h2, out_row_norm, out_mul, out_add, out_inv_scale, out_per_tensor, stream);
}

// A with one inverse scale per (row, 64-column chunk of K): a_kscale[M][K / 64] -- the packed attention gradient dQ | dK | dV, whose
// (row, head) blocks are written by different workgroups, each with the block's exact maximum (tvl_attn_h2_bwd, h2 output).  fp32 output.
extern "C" int tvl_gemm_h2_ks(const tvlGemmTp3Args* a, const float* a_kscale, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr && a_kscale != nullptr, "tvl_gemm_h2_ks: null args");
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K >= 64 && a->K % 64 == 0 && a->K <= 4096 && a->N % 16 == 0, "tvl_gemm_h2_ks: need K %% 64 == 0, 64 <= K <= 4096, N %% 16 == 0 (K=%d N=%d)", a->K, a->N);
    TVL_REQUIRE(a->A && a->B && a->C && !a->C_tp3 && !a->bias && !a->residual && !a->pre_out && !a->dact && !a->act, "tvl_gemm_h2_ks: plain fp32 output only");
    TVL_REQUIRE(tvl_aligned16(a->A) && tvl_aligned16(a->B) && tvl_aligned16(a->C) && a->ldc >= a->N && a->ldc % 4 == 0, "tvl_gemm_h2_ks: alignment / ldc");
    TVL_REQUIRE(a->a_rows >= a->M && a->b_rows >= a->N, "tvl_gemm_h2_ks: operand images hold fewer rows than M / N");
    Tp3Params p = {};
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = reinterpret_cast<const unsigned char*>(a->A); p.a_rb = (int)((a->a_rows + 31) / 32);
    p.B = reinterpret_cast<const unsigned char*>(a->B); p.b_rb = (int)((a->b_rows + 31) / 32);
    p.C = a->C; p.ldc = a->ldc; p.alpha = a->alpha; p.a_kscale = a_kscale; p.k_chunks = a->K / 64; p.a_sstride = 1;
    static const int f32_direct_ks = getenv("TVL_GEMM_F32_DIRECT") ? atoi(getenv("TVL_GEMM_F32_DIRECT")) : 1;
    p.f32_direct = f32_direct_ks;
    const int rc = (use_m16() && p.K >= 96) ? tvl_gemm_h2m_ks_t192(&p, reinterpret_cast<hipStream_t>(stream))
                                            : launch<192, 256, 3, E_F32 | E_RSCALE, 2, true>(p, reinterpret_cast<hipStream_t>(stream));
    TVL_REQUIRE(rc == 0, "tvl_gemm_h2_ks: launch failed (dynamic LDS opt-in?)");
    TVL_LAUNCH_CHECK("tvl_gemm_h2_ks");
    return 0;
}

