// 128-row tile of the tp3 GEMM (wave tile 64x64): M that 192 quantises badly.
#include "gemm_tp3_kernel.h"

int tvl_gemm_tp3_t128(const void* params, int epi, hipStream_t s) { return launch_epi<128, 3>(*static_cast<const Tp3Params*>(params), epi, s); }
