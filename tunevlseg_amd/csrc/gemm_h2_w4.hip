// Two-piece fp16 GEMM on 4-wave workgroups (NW = 4: 1 x 4 waves, tile 128 x 256, wave tile 128 x 64), two workgroups per CU.
// Own translation unit (parallel make).  Measured (profiles/r3_gemm_experiments.md): equal to the 8-wave 256 x 256 tile on the N >= 2304
// shapes, slower at N = 768 -- the two workgroups hold 1.5x the bytes in flight and need 1.5x the bytes per FLOP.
#include "gemm_h2_variants.h"

int tvl_gemm_h2_w4(const void* params, int bm, int epi, hipStream_t s) {
    const Tp3Params& p = *static_cast<const Tp3Params*>(params);
    if (bm == 128) return launch_layer_epi<128, 2, 4, 3>(p, epi, s);
    return 1;
}
