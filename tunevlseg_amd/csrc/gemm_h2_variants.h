// the epilogues of a vision layer's seven plain h2 GEMMs (ops.EncoderLayerTp3Fn), for one (tile, waves, ring depth) configuration
#pragma once
#include "gemm_tp3_kernel.h"

namespace {
template <int BM, int V, int NW, int NS>
int launch_layer_epi(const Tp3Params& p, int epi, hipStream_t s) {
    switch (epi) {
        case E_BIAS | E_RSCALE | E_H2OUT: return launch<BM, 256, V, E_BIAS | E_RSCALE | E_H2OUT, 2, false, false, NW, NS>(p, s);                                   // qkv -> h2
        case E_RSCALE | E_H2OUT: return launch<BM, 256, V, E_RSCALE | E_H2OUT, 2, false, false, NW, NS>(p, s);                                                     // dO -> h2
        case E_BIAS | E_RES | E_F32 | E_RSCALE: return launch<BM, 256, V, E_BIAS | E_RES | E_F32 | E_RSCALE, 2, false, false, NW, NS>(p, s);                       // out_proj, fc2
        case E_F32 | E_RSCALE: return launch<BM, 256, V, E_F32 | E_RSCALE, 2, false, false, NW, NS>(p, s);                                                         // data gradients
        case E_BIAS | E_QGELU | E_PRE | E_RSCALE | E_H2OUT: return launch<BM, 256, V, E_BIAS | E_QGELU | E_PRE | E_RSCALE | E_H2OUT, 2, false, false, NW, NS>(p, s);   // fc1 -> h2 + z
        case E_BIAS | E_QGELU | E_RSCALE | E_H2OUT: return launch<BM, 256, V, E_BIAS | E_QGELU | E_RSCALE | E_H2OUT, 2, false, false, NW, NS>(p, s);               // fc1, no tape
        case E_DQGELU | E_RSCALE | E_H2OUT: return launch<BM, 256, V, E_DQGELU | E_RSCALE | E_H2OUT, 2, false, false, NW, NS>(p, s);                               // dz -> h2
        default: return launch<BM, 256, V, -1, 2, false, false, NW, NS>(p, s);
    }
}
}  // namespace
