// 1- and 2-piece instantiations of the split-bf16 GEMM (measured, documented reduced-precision modes "bf16" / "bf16x3").
#include "gemm_bf16s_kernel.h"

int tvl_gemm_bf16s_lowp(const void* gemm_params, int nsplit, int bm, bool vec, hipStream_t s) {
    const GemmParams& p = *static_cast<const GemmParams*>(gemm_params);
    if (vec) return nsplit == 1 ? launch_tile<1, true>(bm, p, s) : launch_tile<2, true>(bm, p, s);
    return nsplit == 1 ? launch_tile<1, false>(bm, p, s) : launch_tile<2, false>(bm, p, s);
}
