// 256-row tile of the tp3 GEMM (wave tile 128x64).  128 accumulator + 2 x 72 fragment registers do not fit 256 VGPRs with the
// fragment reads pinned early, so this tile keeps hipcc's own read placement (variant bit 0 clear) -- it sinks the reads.
#include "gemm_tp3_kernel.h"

int tvl_gemm_tp3_t256(const void* params, int epi, hipStream_t s) { return launch_epi<256, 2>(*static_cast<const Tp3Params*>(params), epi, s); }
