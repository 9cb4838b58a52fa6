// Exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel template serves the three operand layouts of an nn.Linear: forward
// (NT), data gradient (NN) and weight gradient (TN).  256 threads = 4 waves in a
// 2x2 arrangement; each wave owns a (BM/2)x(BN/2) block of 32x32 MFMA tiles.
// K is walked in BK=32 slabs, double-buffered in LDS through registers (global
// float4 -> VGPR -> ds_write_b128), one barrier per slab.
//
// LDS images
//   K-major operand (rows of A or W, k contiguous in memory): [rows][BK+4].  The
//   +4 pad makes the row stride 36 dwords = 4*9, so the 16 rows of one
//   ds_read_b128 lane group fall on 16 distinct 16-byte slots of the 256-byte
//   bank row: conflict free.  A lane (row r, half h) reads k = 8c+4h .. 8c+4h+3
//   with ONE ds_read_b128 and feeds them to four consecutive MFMA steps; both
//   operands use the same k permutation, so the sum over k is unchanged.
//   MN-major operand (k rows, m/n contiguous): [BK][cols], read with ds_read_b32
//   (32 consecutive dwords per half wave: conflict free).
//
// MFMA operand/accumulator maps (guide §3): A lane l -> A[l&31][l>>5],
// B lane l -> B[l>>5][l&31]; D reg r of lane l -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int NTHREADS = 256;

struct GemmParams {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* residual; int ldr;
    int act;
    float* pre_out;
    const float* dact_aux; int ld_aux; int dact;
    float alpha;
    tvlRowMap a_map, c_map;
    int tiles_m, tiles_n;
    int kchunk;  // split-K (weight gradients with a handful of output tiles): blockIdx.y owns k in [y*kchunk, (y+1)*kchunk), atomicAdd epilogue
};

__device__ __forceinline__ long map_row(int r, const tvlRowMap& m) {
    return m.div > 0 ? (long)(r / m.div) * m.mul + (r % m.div) + m.off : (long)r;
}

// 4 consecutive elements of a row-major matrix, zero beyond `cols`; `row` must be valid.
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ base, long row, int col, int cols, int ld) {
    const float* p = base + row * (long)ld + col;
    if (VEC) {
        if (col + 3 < cols) return *reinterpret_cast<const float4*>(p);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < cols) v.x = p[0];
    if (col + 1 < cols) v.y = p[1];
    if (col + 2 < cols) v.z = p[2];
    if (col + 3 < cols) v.w = p[3];
    return v;
}

// Stage registers for one operand tile.
template <int ROWS>
struct StageRegs {
    float4 v[ROWS / 32];
};

// ---- K-major operand: global [rows][k] -> LDS [ROWS][LDK] -------------------------------
template <int ROWS, bool VEC>
__device__ __forceinline__ void gload_kmajor(StageRegs<ROWS>& s, const float* __restrict__ base, int ld, int row0, int nrows,
                                             int k0, int K, const tvlRowMap& map) {
    const int t = threadIdx.x;
    const int c = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        int r = row0 + r0 + 32 * i;
        r = r < nrows ? r : nrows - 1;  // clamp: rows past the edge are computed and discarded
        s.v[i] = load4<VEC>(base, map_row(r, map), k0 + 4 * c, K, ld);
    }
}
template <int ROWS>
__device__ __forceinline__ void sstore_kmajor(const StageRegs<ROWS>& s, float* __restrict__ lds) {
    const int t = threadIdx.x;
    const int c = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) *reinterpret_cast<float4*>(&lds[(r0 + 32 * i) * LDK + 4 * c]) = s.v[i];
}

// ---- MN-major operand: global [k][cols] -> LDS [BK][COLS] -------------------------------
template <int COLS, bool VEC>
__device__ __forceinline__ void gload_mnmajor(StageRegs<COLS>& s, const float* __restrict__ base, int ld, int col0, int ncols,
                                              int k0, int K, const tvlRowMap& map) {
    constexpr int F4 = COLS / 4;            // float4 per k-row
    constexpr int KSTEP = NTHREADS / F4;    // k-rows covered per pass
    const int t = threadIdx.x;
    const int c = t % F4, kr0 = t / F4;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int k = k0 + kr0 + KSTEP * i;
        if (k < K)
            s.v[i] = load4<VEC>(base, map_row(k, map), col0 + 4 * c, ncols, ld);
        else
            s.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int COLS>
__device__ __forceinline__ void sstore_mnmajor(const StageRegs<COLS>& s, float* __restrict__ lds) {
    constexpr int F4 = COLS / 4;
    constexpr int KSTEP = NTHREADS / F4;
    const int t = threadIdx.x;
    const int c = t % F4, kr0 = t / F4;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) *reinterpret_cast<float4*>(&lds[(kr0 + KSTEP * i) * COLS + 4 * c]) = s.v[i];
}

template <int BM, int BN, int WGM, bool A_KMAJOR, bool B_KMAJOR, bool VEC>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmParams p) {
    TVL_KERNEL_ENTRY();
    constexpr int WGN = 4 / WGM;  // 4 waves arranged WGM x WGN over the tile
    constexpr int WM = BM / WGM, WN = BN / WGN;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_TILE = A_KMAJOR ? BM * LDK : BK * BM;
    constexpr int B_TILE = B_KMAJOR ? BN * LDK : BK * BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = A_TILE + B_TILE;  // LDS: [A0 | B0 | A1 | B1]

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give every
    // XCD a contiguous run of tiles (bijective form for grids that are not a multiple of 8).
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped order: consecutive tile ids walk GROUP_M row-tiles of one column-tile before moving to the next column, so
    // the ~64 tiles resident on an XCD cover an ~8 x 8 block: 8 A panels + 8 W slices instead of 3 A panels + all of W
    // (rocprofv3: FETCH_SIZE of the fc1 GEMM was 19x its algorithmic bytes with the row-major order)
    constexpr int GROUP_M = 8;
    const int gsize_full = GROUP_M * p.tiles_n;
    const int group = bid / gsize_full;
    const int gm0 = group * GROUP_M;
    const int gm = p.tiles_m - gm0 < GROUP_M ? p.tiles_m - gm0 : GROUP_M;
    const int in_group = bid - group * gsize_full;
    const int tile_m = gm0 + in_group % gm, tile_n = in_group / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StageRegs<BM> sa;
    StageRegs<BN> sb;
    const tvlRowMap ident = {0, 0, 0};

    const int kbeg = p.kchunk > 0 ? blockIdx.y * p.kchunk : 0;
    const int kend = p.kchunk > 0 ? (kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K) : p.K;
    auto gload = [&](int k0) {
        if (A_KMAJOR) gload_kmajor<BM, VEC>(sa, p.A, p.lda, m0, p.M, k0, kend, p.a_map);
        else gload_mnmajor<BM, VEC>(sa, p.A, p.lda, m0, p.M, k0, kend, p.a_map);
        if (B_KMAJOR) gload_kmajor<BN, VEC>(sb, p.B, p.ldb, n0, p.N, k0, kend, ident);
        else gload_mnmajor<BN, VEC>(sb, p.B, p.ldb, n0, p.N, k0, kend, ident);
    };
    auto sstore = [&](int buf) {
        float* a_dst = smem + buf * STAGE;
        float* b_dst = a_dst + A_TILE;
        if (A_KMAJOR) sstore_kmajor<BM>(sa, a_dst); else sstore_mnmajor<BM>(sa, a_dst);
        if (B_KMAJOR) sstore_kmajor<BN>(sb, b_dst); else sstore_mnmajor<BN>(sb, b_dst);
    };

    const int nk = (kend - kbeg + BK - 1) / BK;
    gload(kbeg);
    sstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);  // in flight behind this slab's MFMAs
        const float* as = smem + cur * STAGE;
        const float* bs = as + A_TILE;
#pragma unroll
        for (int c8 = 0; c8 < BK / 8; ++c8) {
            float af[TM][4], bf[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WM + i * 32 + l31;
                if (A_KMAJOR) {
                    const float4 v = *reinterpret_cast<const float4*>(&as[row * LDK + c8 * 8 + 4 * h]);
                    af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[i][j] = as[(c8 * 8 + 4 * h + j) * BM + row];
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int col = wn * WN + i * 32 + l31;
                if (B_KMAJOR) {
                    const float4 v = *reinterpret_cast<const float4*>(&bs[col * LDK + c8 * 8 + 4 * h]);
                    bf[i][0] = v.x; bf[i][1] = v.y; bf[i][2] = v.z; bf[i][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[i][j] = bs[(c8 * 8 + 4 * h + j) * BN + col];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[n][j], acc[i][n], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue straight from the accumulators ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + wn * WN + n * 32 + l31;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                const long crow = map_row(row, p.c_map);
                float v = acc[i][n][r] * p.alpha + bv;
                if (p.kchunk > 0) {  // split-K partial: C was zeroed by the host entry
                    atomicAdd(&p.C[crow * p.ldc + col], v);
                    continue;
                }
                if (p.dact) v *= dact_f(p.dact_aux[crow * p.ld_aux + col], p.dact);
                if (p.pre_out) p.pre_out[crow * p.ldc + col] = v;
                if (p.act & TVL_ACT_POST_RESIDUAL) {
                    if (p.residual) v += p.residual[crow * p.ldr + col];
                    v = act_f(v, p.act & 0xff);
                } else {
                    v = act_f(v, p.act);
                    if (p.residual) v += p.residual[crow * p.ldr + col];
                }
                p.C[crow * p.ldc + col] = v;
            }
        }
    }
}

template <int BM, int BN, int WGM, bool AK, bool BKM, bool VEC>
int launch(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    constexpr int A_TILE = AK ? BM * LDK : BK * BM;
    constexpr int B_TILE = BKM ? BN * LDK : BK * BN;
    constexpr size_t smem = 2 * (A_TILE + B_TILE) * sizeof(float);
    static bool attr_set = false;
    auto kern = gemm_f32_kernel<BM, BN, WGM, AK, BKM, VEC>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const long nwg = (long)p.tiles_m * p.tiles_n;
    const int splits = p.kchunk > 0 ? (p.K + p.kchunk - 1) / p.kchunk : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)splits), dim3(NTHREADS), smem, s, p);
    return 0;
}

template <int BM, int BN, int WGM, bool VEC>
int launch_layout(int layout, const GemmParams& p, hipStream_t s) {
    switch (layout) {
        case TVL_NT: return launch<BM, BN, WGM, true, true, VEC>(p, s);
        case TVL_NN: return launch<BM, BN, WGM, true, false, VEC>(p, s);
        default:
            if constexpr (BM % 128 == 0 || BM == 64) return launch<BM, BN, WGM, false, false, VEC>(p, s);
            else return 1;  // MN-major A staging needs BM/4 to divide 256
    }
}

// Tile choice = least per-CU time under wave quantisation: ceil(tiles / resident slots) * WGs-per-CU * BM*BN.
// 128x128 and 96x128 run 2 workgroups per CU (LDS 72 / 64.5 KiB), 64x64 runs 4.  M = 15840 = 165 * 96 makes
// 96x128 the balanced choice for the ViT-B/16 shapes (990 tiles over 512 slots at N = 768, vs 744 for 128x128).
struct TileChoice { int bm, bn; };
TileChoice choose_tile(int layout, long M, long N) {
    const long cus = 256;
    struct Cand { int bm, bn, per_cu; } cands[3] = {{128, 128, 2}, {96, 128, 2}, {64, 64, 4}};
    double best = 1e300;
    TileChoice out = {64, 64};
    for (const Cand& c : cands) {
        if (c.bm == 96 && layout == TVL_TN) continue;
        const long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const long slots = cus * c.per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        // small tiles pay more staging traffic per FLOP: 12% handicap keeps them for genuinely small problems
        const double cost = (double)rounds * c.per_cu * c.bm * c.bn * (c.bm == 64 ? 1.12 : 1.0);
        if (cost < best) { best = cost; out = {c.bm, c.bn}; }
    }
    return out;
}

}  // namespace

extern "C" int tvl_gemm_f32(const tvlGemmArgs* a, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr, "tvl_gemm_f32: null args");
    TVL_REQUIRE(a->layout >= 0 && a->layout <= 2, "tvl_gemm_f32: bad layout %d", a->layout);
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "tvl_gemm_f32: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    TVL_REQUIRE(a->A && a->B && a->C, "tvl_gemm_f32: null operand");
    const int a_cols = a->layout == TVL_TN ? a->M : a->K;
    const int b_cols = a->layout == TVL_NT ? a->K : a->N;
    TVL_REQUIRE(a->lda >= a_cols && a->ldb >= b_cols && a->ldc >= a->N, "tvl_gemm_f32: leading dimension too small");
    TVL_REQUIRE(!a->residual || a->ldr >= a->N, "tvl_gemm_f32: ldr too small");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N), "tvl_gemm_f32: dact needs dact_aux");
    TVL_REQUIRE((long)((a->M + 63) / 64) * ((a->N + 63) / 64) < (1L << 31), "tvl_gemm_f32: grid too large");

    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha;
    p.a_map = a->a_map; p.c_map = a->c_map; p.tiles_m = p.tiles_n = 0;
    p.kchunk = 0;

    const bool vec = tvl_aligned16(a->A) && tvl_aligned16(a->B) && (a->lda % 4 == 0) && (a->ldb % 4 == 0);
    const TileChoice tc = choose_tile(a->layout, a->M, a->N);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // weight gradients of the small trainable layers: a few output tiles but a reduction over every pixel/token of the batch.
    // One workgroup per tile would walk K alone; split K over the grid and combine with fp32 atomics (sum order varies).
    {
        const long tiles = (long)((a->M + tc.bm - 1) / tc.bm) * ((a->N + tc.bn - 1) / tc.bn);
        const bool plain = !a->bias && !a->residual && !a->pre_out && !a->dact && a->act == TVL_ACT_NONE && a->c_map.div <= 0 && a->ldc == a->N;
        if (a->layout == TVL_TN && plain && tiles <= 16 && a->K >= 4096) {
            int splits = (int)(512 / tiles);
            const int max_by_k = a->K / 512;
            splits = splits < max_by_k ? splits : max_by_k;
            if (splits > 1) {
                p.kchunk = ((a->K + splits - 1) / splits + BK - 1) / BK * BK;
                TVL_REQUIRE(hipMemsetAsync(a->C, 0, sizeof(float) * (size_t)a->M * a->N, s) == hipSuccess, "tvl_gemm_f32: memset failed");
            }
        }
    }
    int rc;
    if (tc.bm == 128) rc = vec ? launch_layout<128, 128, 2, true>(a->layout, p, s) : launch_layout<128, 128, 2, false>(a->layout, p, s);
    else if (tc.bm == 96) rc = vec ? launch_layout<96, 128, 1, true>(a->layout, p, s) : launch_layout<96, 128, 1, false>(a->layout, p, s);
    else rc = vec ? launch_layout<64, 64, 2, true>(a->layout, p, s) : launch_layout<64, 64, 2, false>(a->layout, p, s);
    TVL_REQUIRE(rc == 0, "tvl_gemm_f32: no kernel for this layout/tile");
    TVL_LAUNCH_CHECK("tvl_gemm_f32");
    return 0;
}
