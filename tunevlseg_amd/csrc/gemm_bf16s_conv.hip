// Implicit-GEMM 3x3 conv entry point (CONV instantiations of the split-bf16 kernel, 3-piece split only).
#include "gemm_bf16s_kernel.h"

extern "C" int tvl_conv3x3_bf16s(const tvlGemmArgs* a, const tvlConvGeom* g, int32_t nsplit, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr && g != nullptr, "tvl_conv3x3_bf16s: null args");
    TVL_REQUIRE(nsplit == 3, "tvl_conv3x3_bf16s: only the fp32-equivalent 3-piece split is built (use tvl_im2col3x3 + tvl_gemm_bf16s otherwise)");
    TVL_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->C > 0 && (g->stride == 1 || g->stride == 2), "tvl_conv3x3_bf16s: bad geometry");
    const int Ho = (g->H - 1) / g->stride + 1, Wo = (g->W - 1) / g->stride + 1;
    TVL_REQUIRE((long)g->B * Ho * Wo == a->M && a->K == 9 * g->C && a->N > 0, "tvl_conv3x3_bf16s: M=%d K=%d do not match the geometry", a->M, a->K);
    TVL_REQUIRE(a->A && a->B && a->C, "tvl_conv3x3_bf16s: null operand");
    TVL_REQUIRE(g->C % 4 == 0 && a->lda % 4 == 0 && a->lda >= g->C && tvl_aligned16(a->A), "tvl_conv3x3_bf16s: needs C %% 4 == 0 and 16-byte aligned rows");
    TVL_REQUIRE(a->ldb >= a->K && a->ldb % 4 == 0 && tvl_aligned16(a->B) && a->ldc >= a->N, "tvl_conv3x3_bf16s: bad weight / output leading dimension");
    TVL_REQUIRE(!a->residual || a->ldr >= a->N, "tvl_conv3x3_bf16s: ldr too small");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N), "tvl_conv3x3_bf16s: dact needs dact_aux");
    TVL_REQUIRE((long)g->B * g->H * g->W * (long)a->lda < (1L << 40), "tvl_conv3x3_bf16s: map too large");

    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha;
    p.a_map = tvlRowMap{0, 0, 0}; p.c_map = a->c_map; p.tiles_m = p.tiles_n = 0;
    p.cH = g->H; p.cW = g->W; p.cC = g->C; p.cStride = g->stride; p.cHo = Ho; p.cWo = Wo;
    const int bm = choose_bm(a->M, a->N, a->K);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rc = launch_conv_tile(bm, p, s);
    TVL_REQUIRE(rc == 0, "tvl_conv3x3_bf16s: launch failed");
    TVL_LAUNCH_CHECK("tvl_conv3x3_bf16s");
    return 0;
}
