// fp32-equivalent GEMM over PRE-SPLIT operands (gfx950, v_mfma_f32_32x32x16_bf16 + LDS-DMA).
//
// gemm_bf16s.hip splits every fp32 operand into 3 bf16 pieces while it stages the tile: ~5 VALU instructions per MFMA and
// all LDS fills through ds_write (<= 85 B/clk/CU, MI355X_MICROARCH.md §LDS), which is what bounds that kernel (its 96x128
// tile keeps the LDS path ~100 % busy).  Here the pieces already exist in HBM as three bf16 planes
//     x = p0 + p1 + p2        (p0, p1 by truncating the running residual, p2 rounded: tvl_split_planes)
// -- frozen weights are split once, activations by their producer or a one-pass split kernel -- so the tile fill is
// `global_load_lds_dwordx4` (no VGPRs, no VALU, no ds_write) and the main loop is ds_read_b128 + 6 MFMAs per piece-pair
// block, the same arithmetic (and the same fp32 result up to summation order) as gemm_bf16s with S = 3.
//
// Layout: NT.  A planes [3][M][lda], B planes [3][N][ldb], bf16, k-contiguous, K zero-padded to a multiple of 32.
// LDS image per operand plane: [rows][32 bf16] = 64-byte rows, unpadded (an LDS-DMA instruction writes 1 KiB = 16 rows
// contiguously); bank conflicts are avoided by an XOR swizzle of the four 16-byte chunks of a row, applied on the SOURCE
// address of the DMA and on the ds_read address:  chunk c of row r lives at slot c ^ ((r >> 2) & 3).
#include <stdlib.h>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NT_ = 256;
constexpr int BK = 32;

struct PlaneParams {
    int M, N, K;                       // K = logical depth; planes are zero padded to Kp = roundup(K, 32)
    const __bf16* A; long a_ps; int lda;
    const __bf16* B; long b_ps; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* residual; int ldr;
    int act;
    float* pre_out;
    const float* dact_aux; int ld_aux; int dact;
    float alpha;
    tvlRowMap a_map, c_map;
    int tiles_m, tiles_n;
    // implicit 3x3 / pad 1 conv: A planes are an NHWC pixel matrix [B*H*W][lda]; GEMM column (ky*3+kx)*C + c
    int conv, cH, cW, cC, cStride, cHo, cWo;
    const __bf16* zeros;               // >= 16 zero bytes: source of out-of-map conv taps
};

__device__ __forceinline__ long map_row(int r, const tvlRowMap& m) {
    return m.div > 0 ? (long)(r / m.div) * m.mul + (r % m.div) + m.off : (long)r;
}

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// epilogue of a transposed-product accumulator tile: lane -> row l31, registers 4g..4g+3 -> columns 8g+4h .. 8g+4h+3
template <int TM, int TN, bool POST>
__device__ __forceinline__ void epilogue_t(const PlaneParams& p, f32x16 (&acc)[TM][TN], int row_base, int col_base, int l31, int h) {
    const bool vec_c = (p.ldc % 4 == 0) && tvl_dev_aligned16(p.C) && (!p.pre_out || tvl_dev_aligned16(p.pre_out)) &&
                       (!p.residual || (p.ldr % 4 == 0 && tvl_dev_aligned16(p.residual))) &&
                       (!p.dact || (p.ld_aux % 4 == 0 && tvl_dev_aligned16(p.dact_aux))) && (!p.bias || tvl_dev_aligned16(p.bias));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row_base + i * 32 + l31;
        if (row >= p.M) continue;
        const long crow = map_row(row, p.c_map);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = col_base + j * 32 + 8 * g + 4 * h;
                if (col >= p.N) continue;
                float v[4] = {acc[i][j][4 * g] * p.alpha, acc[i][j][4 * g + 1] * p.alpha, acc[i][j][4 * g + 2] * p.alpha,
                              acc[i][j][4 * g + 3] * p.alpha};
                if (vec_c && col + 3 < p.N) {
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    }
                    if (p.dact) {
                        const float4 z4 = *reinterpret_cast<const float4*>(p.dact_aux + crow * p.ld_aux + col);
                        v[0] *= dact_f(z4.x, p.dact); v[1] *= dact_f(z4.y, p.dact); v[2] *= dact_f(z4.z, p.dact); v[3] *= dact_f(z4.w, p.dact);
                    }
                    if (p.pre_out) *reinterpret_cast<float4*>(p.pre_out + crow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                    if (POST && p.residual) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + crow * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act & 0xff);
                    if (!POST && p.residual) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + crow * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                    *reinterpret_cast<float4*>(p.C + crow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = col + e;
                        if (c >= p.N) continue;
                        float x = v[e] + (p.bias ? p.bias[c] : 0.f);
                        if (p.dact) x *= dact_f(p.dact_aux[crow * p.ld_aux + c], p.dact);
                        if (p.pre_out) p.pre_out[crow * p.ldc + c] = x;
                        if (POST && p.residual) x += p.residual[crow * p.ldr + c];
                        x = act_f(x, p.act & 0xff);
                        if (!POST && p.residual) x += p.residual[crow * p.ldr + c];
                        p.C[crow * p.ldc + c] = x;
                    }
                }
            }
        }
    }
}

// One k-slab = 3 planes x (BM + BN) rows x 64 B, filled by 1-KiB LDS-DMA pieces (16 rows each) dealt round-robin to the 4
// waves.  Piece order in LDS: A plane 0 rows 0.., A plane 1, A plane 2, B plane 0, ...
template <int BM, int BN, bool CONV>
struct Filler {
    static constexpr int PA = 3 * BM / 16, PB = 3 * BN / 16, PT = PA + PB;
    static constexpr int PER_WAVE = (PT + 3) / 4;
    const __bf16* src[PER_WAVE];   // per-lane source of this wave's pieces at k = 0 (chunk swizzle applied)
    // conv: the tap decides the source pixel, so keep the row decomposition instead
    long pix0[CONV ? PER_WAVE : 1];
    int oy[CONV ? PER_WAVE : 1], ox[CONV ? PER_WAVE : 1];
    int chunk[CONV ? PER_WAVE : 1];
};

template <int BM, int BN, bool CONV>
__device__ __forceinline__ void filler_init(Filler<BM, BN, CONV>& f, const PlaneParams& p, int m0, int n0, int wave, int lane) {
    using F = Filler<BM, BN, CONV>;
    const int r16 = lane >> 2;                         // row inside the 16-row piece
    const int cc = (lane & 3) ^ ((r16 >> 2) & 3);      // global chunk that lands in LDS slot (lane & 3)
#pragma unroll
    for (int i = 0; i < F::PER_WAVE; ++i) {
        const int pc = wave + 4 * i;
        f.src[i] = p.zeros;
        if (pc >= F::PT) continue;
        if (pc < F::PA) {
            const int plane = pc / (BM / 16), rb = pc % (BM / 16);
            int row = m0 + rb * 16 + r16;
            row = row < p.M ? row : p.M - 1;
            if constexpr (CONV) {
                const int hw = p.cHo * p.cWo;
                const int b = row / hw, rem = row - b * hw;
                const int oy = rem / p.cWo;
                f.pix0[i] = (long)b * p.cH * p.cW;
                f.oy[i] = oy * p.cStride - 1;
                f.ox[i] = (rem - oy * p.cWo) * p.cStride - 1;
                f.chunk[i] = cc;
                f.src[i] = p.A + plane * p.a_ps;
            } else {
                f.src[i] = p.A + plane * p.a_ps + map_row(row, p.a_map) * p.lda + cc * 8;
            }
        } else {
            const int q = pc - F::PA;
            const int plane = q / (BN / 16), rb = q % (BN / 16);
            int row = n0 + rb * 16 + r16;
            row = row < p.N ? row : p.N - 1;
            f.src[i] = p.B + plane * p.b_ps + (long)row * p.ldb + cc * 8;
            if constexpr (CONV) { f.pix0[i] = 0; f.oy[i] = 0; f.ox[i] = 0; f.chunk[i] = cc; }
        }
    }
}

template <int BM, int BN, bool CONV>
__device__ __forceinline__ void fill(const Filler<BM, BN, CONV>& f, const PlaneParams& p, unsigned char* smem, int wave, int k0) {
    using F = Filler<BM, BN, CONV>;
#pragma unroll
    for (int i = 0; i < F::PER_WAVE; ++i) {
        const int pc = wave + 4 * i;
        if (pc >= F::PT) continue;
        const __bf16* g;
        if (CONV && pc < F::PA) {
            const int kcol = k0 + 8 * f.chunk[i];
            const int tap = kcol / p.cC;
            const int c = kcol - tap * p.cC;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int iy = f.oy[i] + ky, ix = f.ox[i] + kx;
            const bool ok = kcol < p.K && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
            g = ok ? f.src[i] + (f.pix0[i] + (long)iy * p.cW + ix) * p.lda + c : p.zeros;
        } else {
            g = f.src[i] + k0;
        }
        glds16(g, smem + pc * 1024);
    }
}

template <int BM, int BN, int WGM, bool CONV, int NSTAGE = 1>
__global__ __launch_bounds__(NT_) void gemm_planes_kernel(PlaneParams p) {
    constexpr int STAGE_BYTES = 3 * (BM + BN) * 64;
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0 && BM % 16 == 0 && BN % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [A: 3][BM][64 B] then [B: 3][BN][64 B]

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
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

    Filler<BM, BN, CONV> filler;
    filler_init<BM, BN, CONV>(filler, p, m0, n0, wave, lane);
    const int nk = (p.K + BK - 1) / BK;
    fill<BM, BN, CONV>(filler, p, smem, wave, 0);

    // fragment addresses: row l31 of each 32-row block, 16-byte slot (2*ks + h) ^ ((l31 >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    const unsigned char* a_base = smem + (wm * WM + l31) * 64;
    const unsigned char* b_base = smem + 3 * BM * 64 + (wn * WN + l31) * 64;

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // vmcnt(0) + barrier: slab kt has landed for every wave (and, NSTAGE 2, the other stage is free)
        const int cur = NSTAGE == 2 ? (kt & 1) * STAGE_BYTES : 0;
        if (NSTAGE == 2 && kt + 1 < nk) fill<BM, BN, CONV>(filler, p, smem + (STAGE_BYTES - cur), wave, (kt + 1) * BK);
        bf16x8 af[2][TM][3], bf[2][TN][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((2 * ks + h) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < 3; ++s) af[ks][i][s] = *reinterpret_cast<const bf16x8*>(a_base + cur + (s * BM + i * 32) * 64 + slot);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 3; ++s) bf[ks][j][s] = *reinterpret_cast<const bf16x8*>(b_base + cur + (s * BN + j * 32) * 64 + slot);
        }
        if (NSTAGE == 1) {
            __syncthreads();  // every wave holds its fragments: the slab may be overwritten
            if (kt + 1 < nk) fill<BM, BN, CONV>(filler, p, smem, wave, (kt + 1) * BK);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int order = 2; order >= 0; --order)
#pragma unroll
                for (int sa_ = 0; sa_ <= order; ++sa_) {
                    const int sb_ = order - sa_;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][j][sb_], af[ks][i][sa_], acc[i][j], 0, 0, 0);
                }
    }
    if (p.act & TVL_ACT_POST_RESIDUAL) epilogue_t<TM, TN, true>(p, acc, m0 + wm * WM, n0 + wn * WN, l31, h);
    else epilogue_t<TM, TN, false>(p, acc, m0 + wm * WM, n0 + wn * WN, l31, h);
}

template <int BM, int BN, int WGM, bool CONV, int NSTAGE = 1>
int launch(const PlaneParams& p0, hipStream_t s) {
    PlaneParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    constexpr size_t smem = (size_t)NSTAGE * 3 * (BM + BN) * 64;
    static bool attr_set = false;
    auto kern = gemm_planes_kernel<BM, BN, WGM, CONV, NSTAGE>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)p.tiles_m * p.tiles_n)), dim3(NT_), smem, s, p);
    return 0;
}

int choose_bm(long M, long N) {
    const long cus = 256;
    struct Cand { int bm, bn, per_cu; } cands[3] = {{128, 128, 2}, {96, 128, 2}, {64, 64, 4}};
    double best = 1e300;
    int out = 64;
    for (const Cand& c : cands) {
        const long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const long slots = cus * c.per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * c.per_cu * c.bm * c.bn * (c.bm == 64 ? 1.12 : 1.0);
        if (cost < best) { best = cost; out = c.bm; }
    }
    return out;
}

template <bool CONV>
int launch_tile(int bm, const PlaneParams& p, hipStream_t s) {
    static const int two_stage = getenv("TVL_PLANES_2STAGE") ? atoi(getenv("TVL_PLANES_2STAGE")) : 0;  // experiment switch
    if (two_stage && !CONV) {
        if (bm == 128) return launch<128, 128, 2, false, 2>(p, s);
        if (bm == 96) return launch<96, 128, 1, false, 2>(p, s);
    }
    if (bm == 128) return launch<128, 128, 2, CONV>(p, s);
    if (bm == 96) return launch<96, 128, 1, CONV>(p, s);
    return launch<64, 64, 2, CONV>(p, s);
}

// ---- fp32 -> three bf16 planes -------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned pack_trunc(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}

// one thread: 8 consecutive columns of one row -> one 16-byte store per plane; columns >= cols are zero (K padding)
__global__ void split_planes_kernel(const float* __restrict__ x, int ldx, long rows, int cols, __bf16* __restrict__ planes, long ps, int ldp, int vec) {
    const int c8n = ldp >> 3;
    const long total = rows * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8n;
        const int c = (int)(i % c8n) * 8;
        float v[8];
        const float* xr = x + r * ldx + c;
        if (vec && c + 7 < cols) {
            const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = c + e < cols ? xr[e] : 0.f;
        }
        uint4 o0, o1, o2;
        unsigned* w0 = reinterpret_cast<unsigned*>(&o0);
        unsigned* w1 = reinterpret_cast<unsigned*>(&o1);
        unsigned* w2 = reinterpret_cast<unsigned*>(&o2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = v[2 * e], b = v[2 * e + 1];
            w0[e] = pack_trunc(fbits(a), fbits(b));
            a -= bfloat(fbits(a) & 0xFFFF0000u); b -= bfloat(fbits(b) & 0xFFFF0000u);
            w1[e] = pack_trunc(fbits(a), fbits(b));
            a -= bfloat(fbits(a) & 0xFFFF0000u); b -= bfloat(fbits(b) & 0xFFFF0000u);
            w2[e] = pack_rn(a, b);
        }
        __bf16* o = planes + r * ldp + c;
        *reinterpret_cast<uint4*>(o) = o0;
        *reinterpret_cast<uint4*>(o + ps) = o1;
        *reinterpret_cast<uint4*>(o + 2 * ps) = o2;
    }
}

__device__ __attribute__((aligned(16))) unsigned char g_zero16[64];

}  // namespace

extern "C" int tvl_split_planes(const float* x, int32_t ldx, int64_t rows, int32_t cols, void* planes, int64_t plane_stride, int32_t ldp,
                                tvlStream_t stream) {
    TVL_REQUIRE(x && planes, "tvl_split_planes: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0 && ldx >= cols, "tvl_split_planes: bad shape rows=%ld cols=%d ldx=%d", (long)rows, cols, ldx);
    TVL_REQUIRE(ldp % 32 == 0 && ldp >= cols && plane_stride >= rows * (int64_t)ldp && plane_stride % 8 == 0 && tvl_aligned16(planes),
                "tvl_split_planes: planes need ldp %% 32 == 0, ldp >= cols and 16-byte alignment");
    const int vec = tvl_aligned16(x) && ldx % 4 == 0;
    const long total = rows * (ldp / 8);
    long nb = (total + 255) / 256;
    nb = nb > 1048576 ? 1048576 : nb;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, (long)rows, cols,
                       reinterpret_cast<__bf16*>(planes), (long)plane_stride, ldp, vec);
    TVL_LAUNCH_CHECK("tvl_split_planes");
    return 0;
}

extern "C" int tvl_gemm_planes(const tvlGemmArgs* a, int64_t a_plane_stride, int64_t b_plane_stride, const tvlConvGeom* conv, tvlStream_t stream) {
    TVL_REQUIRE(a != nullptr, "tvl_gemm_planes: null args");
    TVL_REQUIRE(a->layout == TVL_NT, "tvl_gemm_planes: NT layout only");
    TVL_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "tvl_gemm_planes: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    TVL_REQUIRE(a->A && a->B && a->C, "tvl_gemm_planes: null operand");
    TVL_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0 && tvl_aligned16(a->A) && tvl_aligned16(a->B) && a_plane_stride % 8 == 0 && b_plane_stride % 8 == 0,
                "tvl_gemm_planes: planes must be 16-byte aligned with leading dimensions divisible by 8");
    const int Kp = (a->K + 31) / 32 * 32;
    TVL_REQUIRE(a->ldb >= Kp && a->ldc >= a->N, "tvl_gemm_planes: B planes must be zero padded to K %% 32 == 0 (ldb >= %d)", Kp);
    TVL_REQUIRE(!a->residual || a->ldr >= a->N, "tvl_gemm_planes: ldr too small");
    TVL_REQUIRE(!a->dact || (a->dact_aux && a->ld_aux >= a->N), "tvl_gemm_planes: dact needs dact_aux");

    PlaneParams p = {};
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = reinterpret_cast<const __bf16*>(a->A); p.a_ps = a_plane_stride; p.lda = a->lda;
    p.B = reinterpret_cast<const __bf16*>(a->B); p.b_ps = b_plane_stride; p.ldb = a->ldb;
    p.C = a->C; p.ldc = a->ldc;
    p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.act = a->act; p.pre_out = a->pre_out;
    p.dact_aux = a->dact_aux; p.ld_aux = a->ld_aux; p.dact = a->dact; p.alpha = a->alpha;
    p.a_map = a->a_map; p.c_map = a->c_map;
    static void* zp = nullptr;
    if (!zp) TVL_REQUIRE(hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) == hipSuccess, "tvl_gemm_planes: zero page lookup failed");
    p.zeros = reinterpret_cast<const __bf16*>(zp);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int bm = choose_bm(a->M, a->N);
    int rc;
    if (conv) {
        TVL_REQUIRE(conv->B > 0 && conv->H > 0 && conv->W > 0 && conv->C > 0 && conv->C % 8 == 0 && (conv->stride == 1 || conv->stride == 2),
                    "tvl_gemm_planes: conv needs C %% 8 == 0 and stride 1|2");
        const int Ho = (conv->H - 1) / conv->stride + 1, Wo = (conv->W - 1) / conv->stride + 1;
        TVL_REQUIRE((long)conv->B * Ho * Wo == a->M && a->K == 9 * conv->C && a->lda >= conv->C, "tvl_gemm_planes: M / K do not match the conv geometry");
        p.conv = 1; p.cH = conv->H; p.cW = conv->W; p.cC = conv->C; p.cStride = conv->stride; p.cHo = Ho; p.cWo = Wo;
        rc = launch_tile<true>(bm, p, s);
    } else {
        TVL_REQUIRE(a->lda >= Kp, "tvl_gemm_planes: A planes must be zero padded to K %% 32 == 0 (lda >= %d)", Kp);
        rc = launch_tile<false>(bm, p, s);
    }
    TVL_REQUIRE(rc == 0, "tvl_gemm_planes: launch failed");
    TVL_LAUNCH_CHECK("tvl_gemm_planes");
    return 0;
}
