// 8-wave two-piece fp16 GEMM with a 4-stage DMA ring (three slabs requested ahead).  Tile codes 2564 / 1924 of tvl_gemm_h2's tile_m.
#include "gemm_h2_variants.h"

int tvl_gemm_h2_ns4(const void* params, int bm, int epi, hipStream_t s) {
    const Tp3Params& p = *static_cast<const Tp3Params*>(params);
    if (bm == 256) return launch_layer_epi<256, 2, 8, 4>(p, epi, s);
    if (bm == 192) return launch_layer_epi<192, 3, 8, 4>(p, epi, s);
    return 1;
}
