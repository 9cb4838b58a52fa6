// 8-wave (512-thread) instantiations of the split-bf16 GEMM: the 192x256 tile (waves 2 x 4, wave tile 96x64 as in the
// 192x128 kernel), one workgroup per CU (105 KiB of LDS).  Same kernel body, compiled with 512 threads per workgroup.
#define TVL_GEMM_NTHREADS 512
#include "gemm_bf16s_kernel.h"

int tvl_gemm_bf16s_w8(const void* gemm_params, bool vec, bool conv, hipStream_t s) {
    const GemmParams& p = *static_cast<const GemmParams*>(gemm_params);
    if (conv) return launch_v<192, 256, 2, 3, true, 32, 1, true>(p, s);
    if (vec) return launch_v<192, 256, 2, 3, true, 32, 1, false>(p, s);
    return launch_v<192, 256, 2, 3, false, 32, 1, false>(p, s);
}
