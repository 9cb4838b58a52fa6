// 256-row tile of the 16x16x32 two-piece fp16 ring GEMM (gemm_h2m_kernel.h); own translation unit for parallel make
#include "gemm_h2m_kernel.h"

int tvl_gemm_h2m_t256(const void* params, int epi, hipStream_t s) { return launch_m_layer_epi<256>(*static_cast<const Tp3Params*>(params), epi, s); }
int tvl_gemm_h2m_conv_t256(const void* params, int epi, hipStream_t s) { return launch_m_conv_epi<256>(*static_cast<const Tp3Params*>(params), epi, s); }
