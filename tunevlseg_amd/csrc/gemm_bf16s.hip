// tvl_gemm_bf16s entry point; the kernel templates live in gemm_bf16s_kernel.h.  The template instantiations are spread over
// three translation units (this one: 3-piece split; gemm_bf16s_lowp.hip: 1- and 2-piece modes; gemm_bf16s_conv.hip: implicit
// 3x3 conv) so that `make -j` compiles them in parallel.
#include "gemm_bf16s_kernel.h"

// GemmParams lives in an anonymous namespace (identical in every TU that includes the header): passed as an opaque pointer
int tvl_gemm_bf16s_lowp(const void* gemm_params, int nsplit, int bm, bool vec, hipStream_t s);  // gemm_bf16s_lowp.hip

extern "C" int tvl_gemm_bf16s(const tvlGemmArgs* a, int32_t nsplit, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr, "tvl_gemm_bf16s: null args");
    TVL_REQUIRE(a->layout == TVL_NT, "tvl_gemm_bf16s: NT layout only (keep frozen weights in both orientations)");
    TVL_REQUIRE(nsplit >= 1 && nsplit <= 3, "tvl_gemm_bf16s: nsplit must be 1, 2 or 3");
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "tvl_gemm_bf16s: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    TVL_REQUIRE(a->A && a->B && a->C, "tvl_gemm_bf16s: null operand");
    TVL_REQUIRE(a->lda >= a->K && a->ldb >= a->K && a->ldc >= a->N, "tvl_gemm_bf16s: leading dimension too small");
    TVL_REQUIRE(!a->residual || a->ldr >= a->N, "tvl_gemm_bf16s: ldr too small");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N), "tvl_gemm_bf16s: dact needs dact_aux");

    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha;
    p.a_map = a->a_map; p.c_map = a->c_map; p.tiles_m = p.tiles_n = 0;
    p.cH = p.cW = p.cC = p.cStride = p.cHo = p.cWo = 0;

    const bool vec = tvl_aligned16(a->A) && tvl_aligned16(a->B) && (a->lda % 4 == 0) && (a->ldb % 4 == 0);
    const int bm = choose_bm(a->M, a->N, a->K);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = 1;
    if (nsplit != 3) rc = tvl_gemm_bf16s_lowp(&p, nsplit, bm, vec, s);
    else if (vec) rc = launch_tile<3, true>(bm, p, s);
    else rc = launch_tile<3, false>(bm, p, s);
    TVL_REQUIRE(rc == 0, "tvl_gemm_bf16s: launch failed");
    TVL_LAUNCH_CHECK("tvl_gemm_bf16s");
    return 0;
}


extern "C" int tvl_gemm_bf16s_splitk(const tvlGemmArgs* a, int32_t splits, float* workspace, int64_t workspace_floats, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr && workspace != nullptr, "tvl_gemm_bf16s_splitk: null args");
    TVL_REQUIRE(a->layout == TVL_NT, "tvl_gemm_bf16s_splitk: NT layout only");
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && splits >= 2 && splits <= 64, "tvl_gemm_bf16s_splitk: bad shape M=%d N=%d K=%d splits=%d", a->M, a->N, a->K, splits);
    TVL_REQUIRE(a->A && a->B && a->C, "tvl_gemm_bf16s_splitk: null operand");
    TVL_REQUIRE(a->lda >= a->K && a->ldb >= a->K && a->ldc >= a->N, "tvl_gemm_bf16s_splitk: leading dimension too small");
    TVL_REQUIRE(!a->residual || a->ldr >= a->N, "tvl_gemm_bf16s_splitk: ldr too small");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N), "tvl_gemm_bf16s_splitk: dact needs dact_aux");
    TVL_REQUIRE(workspace_floats >= (int64_t)splits * a->M * a->N, "tvl_gemm_bf16s_splitk: workspace needs splits*M*N floats");
    TVL_REQUIRE((long)((a->M + 63) / 64) * ((a->N + 63) / 64) <= 65535, "tvl_gemm_bf16s_splitk: meant for skinny problems (too many tiles)");
    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha;
    p.a_map = a->a_map; p.c_map = a->c_map; p.tiles_m = p.tiles_n = 0;
    p.cH = p.cW = p.cC = p.cStride = p.cHo = p.cWo = 0;
    const bool vec = tvl_aligned16(a->A) && tvl_aligned16(a->B) && (a->lda % 4 == 0) && (a->ldb % 4 == 0);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rc = vec ? launch_splitk<3, true>(p, workspace, splits, s) : launch_splitk<3, false>(p, workspace, splits, s);
    TVL_REQUIRE(rc == 0, "tvl_gemm_bf16s_splitk: launch failed");
    TVL_LAUNCH_CHECK("tvl_gemm_bf16s_splitk");
    return 0;
}
