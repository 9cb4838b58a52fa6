// Shared device/host helpers for libtvl_hip (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tvl_hip.h"

#define TVL_WAVE 64

void tvl_set_error(const char* fmt, ...);
// split-bf16 attention (attention_bf16s.hip), dispatched from tvl_attn_fwd / tvl_attn_bwd for d_h = 64 without masks
// o_tp3 / dqkv_tp3 (may be null): additional outputs as tp3 images (include/tvl_hip.h); with o_tp3 given to the backward, delta
// is computed from it (a->o is not read)
int tvl_attn_fwd_bf16s_impl(const tvlAttnFwdArgs* a, void* o_tp3, hipStream_t s);
int tvl_attn_bwd_bf16s_impl(const tvlAttnBwdArgs* a, const void* o_tp3, void* dqkv_tp3, hipStream_t s);
int tvl_attn_mode_bf16s(void);

#define TVL_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            tvl_set_error(__VA_ARGS__);       \
            return 1;                         \
        }                                     \
    } while (0)

#define TVL_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            tvl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

static inline bool tvl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool tvl_dev_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float quick_gelu_f(float z) {
    return z / (1.0f + expf(-1.702f * z));
}
__device__ __forceinline__ float quick_gelu_grad_f(float z) {
    const float s = 1.0f / (1.0f + expf(-1.702f * z));
    return s + 1.702f * z * s * (1.0f - s);
}
// QuickGELU and its derivative on the hardware transcendentals (v_exp_f32 = 2^x, v_rcp_f32; 1 ulp each) for the GEMM epilogues, where
// the IEEE expf / division sequences of the functions above cost ~50 vector instructions per element: s = 1 / (1 + 2^(-1.702 log2(e) z))
__device__ __forceinline__ float sigmoid1702_fast(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * z));
}
__device__ __forceinline__ float quick_gelu_fast(float z) { return z * sigmoid1702_fast(z); }
__device__ __forceinline__ float quick_gelu_grad_fast(float z) {
    const float s = sigmoid1702_fast(z);
    return s + 1.702f * z * s * (1.0f - s);
}
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == TVL_ACT_QUICK_GELU) return quick_gelu_f(v);
    if (act == TVL_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TVL_ACT_SIGMOID) {
        const float e = expf(-fabsf(v));
        return (v >= 0.f ? 1.0f : e) / (1.0f + e);
    }
    return v;
}
__device__ __forceinline__ float dact_f(float z, int act) {
    if (act == TVL_ACT_QUICK_GELU) return quick_gelu_grad_f(z);
    if (act == TVL_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == TVL_ACT_SIGMOID) {
        const float e = expf(-fabsf(z));
        const float s = (z >= 0.f ? 1.0f : e) / (1.0f + e);
        return s * (1.0f - s);
    }
    return 1.f;
}
