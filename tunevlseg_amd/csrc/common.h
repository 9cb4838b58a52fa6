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

// Debug build (make POISON=1): every workgroup fills its whole LDS allocation with NaN patterns (fp32, fp16-pair and bf16-pair
// NaNs at once) before its first instruction of substance, so a kernel that reads an LDS location it did not write fails the parity suite in a
// single process instead of only when another process's leftovers sit there.  Expands to nothing in the product build.
#ifdef TVL_POISON_LDS
__device__ __forceinline__ void tvl_poison_lds() {
    const auto* pkt = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_dispatch_ptr();
    const unsigned bytes = pkt[6];   // hsa_kernel_dispatch_packet_t.group_segment_size (byte 24)
    auto* lds = (__attribute__((address_space(3))) unsigned*)0;
    const unsigned nthreads = blockDim.x * blockDim.y * blockDim.z, tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    for (unsigned i = tid; i < bytes / 4; i += nthreads) lds[i] = 0x7FC07FC0u;
    __syncthreads();
}
#define TVL_KERNEL_ENTRY() tvl_poison_lds()
#else
#define TVL_KERNEL_ENTRY() ((void)0)
#endif

static inline bool tvl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool tvl_dev_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Wave-wide float reductions on the DPP path (v_add_f32_dpp ...: ~10 cycles a step) instead of __shfl_xor, which hipcc lowers to
// ds_bpermute_b32 -- a trip through the LDS crossbar per step, six dependent trips per reduction, and a LayerNorm row needs four of them.
// Steps: xor 1, xor 2 (quad_perm), i <-> 7 - i (row_half_mirror), i <-> 15 - i (row_mirror) leave every lane of a 16-lane row with the row's
// total; row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 leave the wave's total in row 3; lane 63 hands it to everyone
// through an SGPR.  A fixed association ((quads) halves) rows -- the same in every kernel that reduces a row, run to run.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xf>(0.f, v);    // quad_perm [1, 0, 3, 2]
    v += dpp_f<0x4E, 0xf>(0.f, v);    // quad_perm [2, 3, 0, 1]
    v += dpp_f<0x141, 0xf>(0.f, v);   // row_half_mirror
    v += dpp_f<0x140, 0xf>(0.f, v);   // row_mirror
    v += dpp_f<0x142, 0xa>(0.f, v);   // row_bcast:15 -> rows 1, 3
    v += dpp_f<0x143, 0xc>(0.f, v);   // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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
// x of lane i combined with x of lane i ^ 32.  (v_permlane32_swap would do it in one VALU instruction, but this hipcc returns the FIRST result of
// __builtin_amdgcn_permlane32_swap for both elements -- measured on the device: both halves come back as the lower half's values -- so the
// exchange stays on __shfl_xor = ds_bpermute_b32.)
__device__ __forceinline__ float xor32_max(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }
__device__ __forceinline__ float xor32_sum(float x) { return x + __shfl_xor(x, 32, 64); }

__device__ __forceinline__ float wave_max(float v) {   // (lanes a masked step does not reach take their own value: max(v, v) = v)
    v = fmaxf(v, dpp_f<0xB1, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x4E, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x141, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x140, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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
    if (act == TVL_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // nn.GELU(): the exact (erf) form
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
    if (act == TVL_ACT_GELU) return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
    return 1.f;
}
