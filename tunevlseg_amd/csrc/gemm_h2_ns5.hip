// 8-wave two-piece fp16 GEMM with a 5-stage DMA ring (four slabs requested ahead: 160 KB of LDS on the 256-row tile).  Tile codes 2565 / 1925.
#include "gemm_h2_variants.h"

int tvl_gemm_h2_ns5(const void* params, int bm, int epi, hipStream_t s) {
    const Tp3Params& p = *static_cast<const Tp3Params*>(params);
    if (bm == 256) return launch_layer_epi<256, 2, 8, 5>(p, epi, s);
    if (bm == 192) return launch_layer_epi<192, 3, 8, 5>(p, epi, s);
    return 1;
}
