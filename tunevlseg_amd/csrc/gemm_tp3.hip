// tvl_gemm_tp3 / tvl_tp3_pack / tvl_tp3_unpack entry points; the kernel template lives in gemm_tp3_kernel.h.  This TU holds the
// 192-row tile (the one M = B * T = 15840 selects) and its diagnostic variants; gemm_tp3_t128.hip / gemm_tp3_t256.hip the others.
#include "gemm_tp3_kernel.h"

namespace {

// one thread = 8 consecutive k of one row -> one 16-byte store per piece; consecutive lanes write consecutive 16-byte slots
__global__ void tp3_pack_kernel(const float* __restrict__ x, long ldx, long rows, int K, unsigned char* __restrict__ out, long rows_padded) {
    TVL_KERNEL_ENTRY();
    const int KB = K >> 4;
    const long total = rows_padded * (K >> 3);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const long blk = t >> 6;              // (rb, kb)
        const long rb = blk / KB;
        const int kb = (int)(blk - rb * KB);
        const int r = lane & 31, h = lane >> 5;
        const long row = rb * 32 + r;
        float v[8];
        if (row < rows) {
            const float* xr = x + row * ldx + kb * 16 + h * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xr[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        uint2 lo[3], hi[3];
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        split4(a, lo);
        split4(b, hi);
        unsigned char* o = out + blk * BLK + lane * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(o + s * PIECE) = make_uint4(lo[s].x, lo[s].y, hi[s].x, hi[s].y);
    }
}

__global__ void tp3_unpack_kernel(const unsigned char* __restrict__ in, long rows, int K, float* __restrict__ y, long ldy) {
    TVL_KERNEL_ENTRY();
    const int KB = K >> 4;
    const long rbs = (rows + 31) / 32;
    const long total = rbs * 32 * (K >> 3);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const long blk = t >> 6;
        const long rb = blk / KB;
        const int kb = (int)(blk - rb * KB);
        const int r = lane & 31, h = lane >> 5;
        const long row = rb * 32 + r;
        if (row >= rows) continue;
        const unsigned char* src = in + blk * BLK + lane * 16;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 2; s >= 0; --s) {
            const uint4 w = *reinterpret_cast<const uint4*>(src + s * PIECE);
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += tp3::bfloat(ww[e] << 16);
                acc[2 * e + 1] += tp3::bfloat(ww[e] & 0xFFFF0000u);
            }
        }
        float* yr = y + row * ldy + kb * 16 + h * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) yr[e] = acc[e];
    }
}

}  // namespace

extern "C" int64_t tvl_tp3_bytes(int64_t rows, int32_t K) {
    if (rows <= 0 || K <= 0 || K % 16 != 0) return -1;
    return ((rows + 31) / 32) * (int64_t)(K / 16) * BLK;
}

extern "C" int tvl_tp3_pack(const float* x, int64_t ldx, int64_t rows, int32_t K, void* out, tvlStream_t stream) {
    TVL_REQUIRE(x && out, "tvl_tp3_pack: null pointer");
    TVL_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ldx >= K, "tvl_tp3_pack: need K %% 16 == 0 and ldx >= K (rows=%ld K=%d ldx=%ld)", (long)rows, K, (long)ldx);
    TVL_REQUIRE(tvl_aligned16(out), "tvl_tp3_pack: output must be 16-byte aligned");
    const long rp = (rows + 31) / 32 * 32;
    const long total = rp * (K / 8);
    long nb = (total + 255) / 256;
    nb = nb > 1048576 ? 1048576 : nb;
    hipLaunchKernelGGL(tp3_pack_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)ldx, (long)rows, K,
                       reinterpret_cast<unsigned char*>(out), rp);
    TVL_LAUNCH_CHECK("tvl_tp3_pack");
    return 0;
}

extern "C" int tvl_tp3_unpack(const void* in, int64_t rows, int32_t K, float* y, int64_t ldy, tvlStream_t stream) {
    TVL_REQUIRE(in && y, "tvl_tp3_unpack: null pointer");
    TVL_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ldy >= K, "tvl_tp3_unpack: need K %% 16 == 0 and ldy >= K");
    const long total = (rows + 31) / 32 * 32 * (K / 8);
    long nb = (total + 255) / 256;
    nb = nb > 1048576 ? 1048576 : nb;
    hipLaunchKernelGGL(tp3_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char*>(in), (long)rows, K, y, (long)ldy);
    TVL_LAUNCH_CHECK("tvl_tp3_unpack");
    return 0;
}

extern "C" int tvl_gemm_tp3(const tvlGemmTp3Args* a, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr, "tvl_gemm_tp3: null args");
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "tvl_gemm_tp3: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    TVL_REQUIRE(a->K % 32 == 0 && a->K >= 64 && a->N % 16 == 0, "tvl_gemm_tp3: need K %% 32 == 0, K >= 64, N %% 16 == 0 (K=%d N=%d)", a->K, a->N);
    TVL_REQUIRE(a->A && a->B && (a->C || a->C_tp3), "tvl_gemm_tp3: null operand");
    TVL_REQUIRE(tvl_aligned16(a->A) && tvl_aligned16(a->B), "tvl_gemm_tp3: tp3 operands must be 16-byte aligned");
    TVL_REQUIRE(a->a_rows >= a->M && a->b_rows >= a->N, "tvl_gemm_tp3: operand images hold fewer rows than M / N");
    TVL_REQUIRE(!a->C || (a->ldc >= a->N && a->ldc % 4 == 0 && tvl_aligned16(a->C)), "tvl_gemm_tp3: C needs ldc >= N, ldc %% 4 == 0, 16-byte alignment");
    TVL_REQUIRE(!a->pre_out || (a->ldc >= a->N && a->ldc % 4 == 0 && tvl_aligned16(a->pre_out)), "tvl_gemm_tp3: pre_out shares ldc and needs 16-byte alignment");
    TVL_REQUIRE(!a->C_tp3 || tvl_aligned16(a->C_tp3), "tvl_gemm_tp3: C_tp3 must be 16-byte aligned");
    TVL_REQUIRE(!a->residual || (a->ldr >= a->N && a->ldr % 4 == 0 && tvl_aligned16(a->residual)), "tvl_gemm_tp3: residual needs ldr >= N, %% 4, alignment");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N && a->ld_aux % 4 == 0 && tvl_aligned16(a->dact_aux)), "tvl_gemm_tp3: dact needs an aligned dact_aux");
    TVL_REQUIRE(!a->bias || tvl_aligned16(a->bias), "tvl_gemm_tp3: bias must be 16-byte aligned");

    Tp3Params p = {};
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = reinterpret_cast<const unsigned char*>(a->A); p.a_rb = (int)((a->a_rows + 31) / 32);
    p.B = reinterpret_cast<const unsigned char*>(a->B); p.b_rb = (int)((a->b_rows + 31) / 32);
    p.C = a->C; p.ldc = a->ldc; p.Cp = reinterpret_cast<unsigned char*>(a->C_tp3);
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // tile rows: 192 fits M = B * T = 15840 (82.5 tiles: 249 / 747 / 996 workgroups for N = 768 / 2304 / 3072 on 256 CUs);
    // a->tile_m (0 = automatic) is the A/B switch of tools/bench_gemm.py
    int bm = a->tile_m;
    if (bm == 0) {
        double best = 1e300;
        const int cand[3] = {256, 192, 128};
        for (int c : cand) {
            const long tiles = ((long)(a->M + c - 1) / c) * ((a->N + 255) / 256);
            const long rounds = (tiles + 255) / 256;
            const double cost = (double)rounds * c * (c == 128 ? 1.08 : 1.0);   // the smaller tile re-reads B more often
            if (cost < best) { best = cost; bm = c; }
        }
    }
    int rc = 1;
    const int epi = epi_code(p);
    // variant: 0 = production schedule (pinned (read, MFMA) order, DMA requests spread over the slab); other values select the
    // diagnostic builds of the 192-row tile (tools/bench_gemm.py, tools/stamp_gemm.py): bit 0 unpinned, bit 1 DMA up front,
    // bits 2-4 timing-only ablations with WRONG results (no DMA / DMA of slab 0 only / no stores), bit 5 in-kernel stamps
    const int v = a->variant;
    if (bm == 256) rc = tvl_gemm_tp3_t256(&p, epi, s);
    else if (bm == 128) rc = tvl_gemm_tp3_t128(&p, epi, s);
    else if (bm != 192) TVL_REQUIRE(false, "tvl_gemm_tp3: tile_m must be 0, 128, 192 or 256");
    else if (v == 0) rc = launch_epi<192, 3>(p, epi, s);
    else if (v == 1) rc = launch<192, 256, 2, -1>(p, s);   // correct results, other schedules (tests/test_hip_kernels.py covers all three)
    else if (v == 2) rc = launch<192, 256, 1, -1>(p, s);
#ifdef TVL_DIAGNOSTIC_KERNELS   // `make DIAG=1`: never part of the shipped library (4, 8, 16 produce WRONG results by design)
    else if (v == 4) rc = launch<192, 256, 3 | 4, -1>(p, s);
    else if (v == 8) rc = launch<192, 256, 3 | 8, -1>(p, s);
    else if (v == 16) rc = launch<192, 256, 3 | 16, -1>(p, s);
    else if (v == 32) rc = launch<192, 256, 3 | 32, -1>(p, s);
    else if (v == 33) rc = launch<192, 256, 3 | 32, E_F32>(p, s);
#endif
    else TVL_REQUIRE(false, "tvl_gemm_tp3: unknown variant %d (diagnostic variants exist only in a `make DIAG=1` build)", v);
    TVL_REQUIRE(rc == 0, "tvl_gemm_tp3: launch failed");
    TVL_LAUNCH_CHECK("tvl_gemm_tp3");
    return 0;
}
