// Two-piece fp16 ("h2") ring GEMM on v_mfma_f32_16x16x32_f16 -- the second generation of gemm_tp3_kernel.h<NP = 2>.
//
// Why another kernel.  Stamps inside the 32x32x16 ring (profiles/r3_gemm_experiments.md) show its k-loop already issues 83-86 % of
// its cycles as MFMAs -- and runs at 1.33-1.9 GHz: a dense MFMA loop is POWER-limited on this chip, cycles saved come back as a lower
// clock.  What raises throughput at a fixed power is fewer joules per FLOP, and the 16x16x32 shape delivers the same FLOPs per cycle at
// a clock the chip can hold ~10 % higher (MI355X_MICROARCH.md "DVFS give-back" item 7; our own probe in the ring's regime -- every
// operand re-read from LDS, one barrier per block, 256 workgroups: 589 vs 515 TFLOP/s fp32-equivalent, tools/mfma_shape_probe.hip).
//
// Same operand images as gemm_tp3_kernel.h (block (rb, kb) = 32 rows x 16 k, two 1-KiB pieces, lane-linear), same LDS-DMA fill, same
// epilogue contract.  What changes:
//   * a 16x16x32 operand = rows 16 m .. 16 m + 15 of TWO consecutive k blocks: lane (r = l & 15, q = l >> 4) reads its 16 bytes
//     (k = 8 q .. 8 q + 7) at block(kb + (q >> 1)) + ((q & 1) * 32 + 16 (m & 1) + r) * 16 -- sixteen-lane groups on distinct slots,
//     conflict-free.  A k-step is therefore 32 deep: slabs (2t, 2t + 1) sit in the LDS stage PAIR t & 1 (four 16-deep stages);
//   * ALL fragments of a 32-deep step live in registers (A: WM/16 x 2 pieces, B: 4 x 2 pieces; 96 VGPRs + 128 accumulators for the
//     256-row tile), so there is ONE raw barrier per 32-deep step (B_t): before it a wave waits for its own DMA pieces of the next
//     pair and for its LDS reads of the current one; after it the current pair is free -> the DMA of step t + 2 goes out, spread over
//     the first MFMAs -- and the next pair is visible -> the fragments of step t + 1 are fetched into each register as soon as its
//     last MFMA of step t has issued (A row by row; B in the last row's column sweep), so no LDS latency is exposed at a boundary;
//   * accumulators are TMo x 4 tiles of 16 x 16 (4 registers each: column = lane & 15, rows 4 (lane >> 4) .. + 3).  The product is
//     transposed as before (mfma(b, a)): a lane's 4 registers are 4 consecutive columns of C.
#pragma once
#include "gemm_tp3_kernel.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <bool DMA, bool NEXT, bool FIRST>
struct StepKind {};

template <int TMo, int EPI>
__device__ __forceinline__ void epilogue16(const Tp3Params& p, f32x4 (&acc)[TMo][4], int row_base, int col_base, int lane, float* scratch) {
    constexpr int LDS_ROW = 36;
    constexpr int ROWS = TMo * 16;
    const int m = lane & 15, q = lane >> 4;
    const int rr = lane >> 3, cc = (lane & 7) * 4;
    constexpr bool H2O = EPI < 0 || (EPI & E_H2OUT) != 0;
    float* rowsc = scratch + ROWS * LDS_ROW;
    const bool h2o = H2O && (EPI >= 0 || p.Ch2 != nullptr);
    if (h2o) {
        for (int rl = lane; rl < ROWS; rl += 64) {
            const int row = row_base + rl;
            float inv = 1.0f;
            if (row < p.M) {
                inv = h2::inv_scale_of(p.out_norm[(long)row * p.out_stride] * p.out_mul + p.out_add);
                if (col_base == 0 && (p.out_stride || row == 0)) p.out_inv[(long)row * p.out_stride] = inv;
            }
            rowsc[rl] = 1.0f / inv;   // a power of two: exact
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // two 32-column strips of the wave's 64 columns
#pragma unroll
        for (int i = 0; i < TMo; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                *reinterpret_cast<float4*>(&scratch[(i * 16 + m) * LDS_ROW + n * 16 + 4 * q]) =
                    make_float4(acc[i][2 * j + n][0], acc[i][2 * j + n][1], acc[i][2 * j + n][2], acc[i][2 * j + n][3]);
        const int col = col_base + j * 32 + cc;
        if (col + 3 < p.N) {
#pragma unroll 1
            for (int rl = rr; rl < ROWS; rl += 8) {
                const float4 v = *reinterpret_cast<const float4*>(&scratch[rl * LDS_ROW + cc]);
                const int row = row_base + rl;
                if (row < p.M) emit4<EPI, false>(p, row, col, v, h2o ? rowsc[rl] : 1.0f);
            }
        }
    }
}

// Epilogue of the image-writing GEMMs (QKV, dO, fc1, dz: E_H2OUT without an fp32 / tp3 result or a residual) straight from the
// accumulators, no LDS round trip and no workgroup barrier.  A 16 x 16 accumulator tile has its row on (lane & 15) and columns
// 4 (lane >> 4) .. + 3 in the four registers; in the h2 image those four columns are 8 bytes of slot (q >> 1) * 32 + row % 32, so the
// sixteen lanes of a (tile, q) write 256 contiguous bytes per piece -- against sixteen-byte granules when the same values leave through
// the row-major scratch (eight lanes per row, four columns each).  The pre-activation z (fc1's second output, the dz kernel's input) is
// fp32 row-major: 64 contiguous bytes per row and instruction.
// p.aux_blocked: z does not leave as a row-major matrix but in the accumulators' own order -- tile after tile, wave after wave, 1 KB per
// (16 x 16 block): a private layout between fc1's epilogue and the dz epilogue of the same (M, N, tile), both fully coalesced.
// four fp16 as they come from memory, carried in the first two lanes of a float4 WITHOUT touching them: a conversion at load time would make
// the wave wait for the load it has just issued, and the point of the z rows is that they are requested one row block ahead
__device__ __forceinline__ float4 half4_raw(const unsigned char* p8) {
    const uint2 u = *reinterpret_cast<const uint2*>(p8);
    return make_float4(__builtin_bit_cast(float, u.x), __builtin_bit_cast(float, u.y), 0.f, 0.f);
}
__device__ __forceinline__ void half4_unpack(const float4& raw, float (&g)[4]) {
    const uint2 u = make_uint2(__builtin_bit_cast(unsigned, raw.x), __builtin_bit_cast(unsigned, raw.y));
    _Float16 h[4];
    *reinterpret_cast<uint2*>(h) = u;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = (float)h[e];
}

template <int TMo, int EPI>
__device__ __forceinline__ void epilogue16_direct(const Tp3Params& p, f32x4 (&acc)[TMo][4], int row_base, int col_base, int lane, long blk_base) {
    static_assert((EPI & E_H2OUT) && !(EPI & (E_F32 | E_TP3 | E_RES | E_RELU)) && (EPI & E_RSCALE), "image-only epilogues");
    const int m = lane & 15, q = lane >> 4;
    const int kbn = p.N >> 4;
    float4 bias4[4];
    if constexpr ((EPI & E_BIAS) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col_base + 16 * j + 4 * q;
            bias4[j] = col + 3 < p.N ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // the QuickGELU' epilogue reads one float4 of z per 16 x 16 tile: row i + 1's four loads are in flight while row i is computed (issued one
    // by one behind their uses they cost a full memory latency per tile, 32 in a row -- 40 us of a 260 us launch)
    float4 zrow[2][4];
    auto load_z = [&](auto ic, auto bc) {   // row block i's four z tiles into zrow[b]
        constexpr int i = decltype(ic)::value, bsel = decltype(bc)::value;
        if constexpr ((EPI & E_DQGELU) != 0 && i < TMo) {
            const long row = row_base + 16 * i + m;
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ((zrow[bsel][J] = p.aux_blocked == 2 ? half4_raw(reinterpret_cast<const unsigned char*>(p.dact_aux) + ((blk_base + i * 4 + J) * 256 + lane * 4) * 2)
                                  : p.aux_blocked ? *reinterpret_cast<const float4*>(p.dact_aux + (blk_base + i * 4 + J) * 256 + lane * 4)
                                                : ((row < p.M && col_base + 16 * J + 4 * q + 3 < p.N)
                                                       ? *reinterpret_cast<const float4*>(p.dact_aux + row * p.ld_aux + col_base + 16 * J + 4 * q)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f))), ...);
            }(std::make_integer_sequence<int, 4>{});
        }
    };
    auto tile = [&](auto ic, auto jc, long row, float f, float sc) {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
        const int col = col_base + 16 * j + 4 * q;
        if (col + 3 >= p.N) return;
        float v[4] = {acc[i][j][0] * f, acc[i][j][1] * f, acc[i][j][2] * f, acc[i][j][3] * f};
        if constexpr ((EPI & E_BIAS) != 0) { v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w; }
        if constexpr ((EPI & E_DQGELU) != 0) {
            const float4 z4 = zrow[i & 1][j];
            if (p.aux_blocked == 2) {   // the buffer holds QuickGELU'(z) itself (one fp16 per element, written by fc1's epilogue), not z
                float g[4];
                half4_unpack(z4, g);
                v[0] *= g[0]; v[1] *= g[1]; v[2] *= g[2]; v[3] *= g[3];
            } else {
                v[0] *= quick_gelu_grad_fast(z4.x); v[1] *= quick_gelu_grad_fast(z4.y);
                v[2] *= quick_gelu_grad_fast(z4.z); v[3] *= quick_gelu_grad_fast(z4.w);
            }
        }
        if constexpr ((EPI & E_PRE) != 0) {
            if (p.aux_blocked == 2) {   // experiment (TVL_GEMM_ZHALF=1): the backward needs z only for QuickGELU'(z) -- leave that, as ONE fp16 per element
                _Float16 g[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (_Float16)quick_gelu_grad_fast(v[e]);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.pre_out) + ((blk_base + i * 4 + j) * 256 + lane * 4) * 2) = *reinterpret_cast<const uint2*>(g);
            } else {
                float* zo = p.aux_blocked ? p.pre_out + (blk_base + i * 4 + j) * 256 + lane * 4 : p.pre_out + row * p.ldc + col;
                *reinterpret_cast<float4*>(zo) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if constexpr ((EPI & E_QGELU) != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu_fast(v[e]);
        }
        _Float16 h0[4], h1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = v[e] * sc;
            h0[e] = (_Float16)w;
            h1[e] = (_Float16)(w - (float)h0[e]);
        }
        unsigned char* o = p.Ch2 + ((row >> 5) * kbn + (col >> 4)) * (long)h2::BLK + (((q >> 1) * 32 + (int)(row & 31)) * 16 + (q & 1) * 8);
        *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(h0);
        *reinterpret_cast<uint2*>(o + h2::PIECE) = *reinterpret_cast<const uint2*>(h1);
    };
    auto row_block = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        load_z(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
        const long row = row_base + 16 * i + m;
        if (row >= p.M) return;
        const float f = p.alpha * p.a_scale[row * p.a_sstride];
        const float inv = h2::inv_scale_of(p.out_norm[row * p.out_stride] * p.out_mul + p.out_add);
        if (col_base == 0 && q == 0 && (p.out_stride || row == 0)) p.out_inv[row * p.out_stride] = inv;
        const float sc = 1.0f / inv;   // a power of two: exact
        [&]<int... J>(std::integer_sequence<int, J...>) { (tile(ic, std::integral_constant<int, J>{}, row, f, sc), ...); }(std::make_integer_sequence<int, 4>{});
    };
    load_z(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    [&]<int... I>(std::integer_sequence<int, I...>) { (row_block(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, TMo>{});
}

// The fp32-output epilogues (out-projection / fc2: bias + residual; the data gradients: plain; frozen Linears / convs: bias [+ ReLU]) straight
// from the accumulators as well: a lane's four registers of a 16 x 16 tile are 16 contiguous bytes of one row, the four lane groups of a
// tile 64 contiguous bytes, so a row leaves as 64-byte segments (the scratch path: 128-byte segments, after an LDS round trip behind a
// workgroup barrier).  The residual's four tiles of row block i + 1 are requested while row block i is computed.
template <int TMo, int EPI>
__device__ __forceinline__ void epilogue16_direct_f32(const Tp3Params& p, f32x4 (&acc)[TMo][4], int row_base, int col_base, int lane) {
    static_assert((EPI & E_F32) && !(EPI & (E_TP3 | E_H2OUT | E_PRE | E_DQGELU | E_QGELU)) && (EPI & E_RSCALE), "plain fp32 epilogues");
    const int m = lane & 15, q = lane >> 4;
    float4 bias4[4];
    if constexpr ((EPI & E_BIAS) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col_base + 16 * j + 4 * q;
            bias4[j] = col + 3 < p.N ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 res[2][4];
    auto load_res = [&](auto ic, auto bc) {
        constexpr int i = decltype(ic)::value, bsel = decltype(bc)::value;
        if constexpr ((EPI & E_RES) != 0 && i < TMo) {
            const long row = row_base + 16 * i + m;
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ((res[bsel][J] = (row < p.M && col_base + 16 * J + 4 * q + 3 < p.N)
                                     ? *reinterpret_cast<const float4*>(p.residual + row * p.ldr + col_base + 16 * J + 4 * q)
                                     : make_float4(0.f, 0.f, 0.f, 0.f)), ...);
            }(std::make_integer_sequence<int, 4>{});
        }
    };
    auto tile = [&](auto ic, auto jc, long row, float f) {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
        const int col = col_base + 16 * j + 4 * q;
        if (col + 3 >= p.N) return;
        float v[4] = {acc[i][j][0] * f, acc[i][j][1] * f, acc[i][j][2] * f, acc[i][j][3] * f};
        if constexpr ((EPI & E_BIAS) != 0) { v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w; }
        if constexpr ((EPI & E_RELU) != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if constexpr ((EPI & E_RES) != 0) {
            const float4 r4 = res[i & 1][j];
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        *reinterpret_cast<float4*>(p.C + row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    };
    auto row_block = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        load_res(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
        const long row = row_base + 16 * i + m;
        if (row >= p.M) return;
        const float f = p.a_kscale ? p.alpha * p.a_kscale[row * p.k_chunks + p.k_chunks - 1] : p.alpha * p.a_scale[row * p.a_sstride];
        [&]<int... J>(std::integer_sequence<int, J...>) { (tile(ic, std::integral_constant<int, J>{}, row, f), ...); }(std::make_integer_sequence<int, 4>{});
    };
    load_res(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    [&]<int... I>(std::integer_sequence<int, I...>) { (row_block(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, TMo>{});
}

// PERSIST: one workgroup per CU walks several tiles (launches of more than 256 tiles: fc1, its data gradient's dz, QKV).  The reason is
// the epilogue: 256-640 KB of stores per tile that every CU of a lock-stepped round issues at the same moment -- a chip-wide HBM burst
// of ~130 MB with no MFMA under it, three times per launch.  De-phasing workgroups INDIVIDUALLY would break what keeps the operand
// stream in L2 (the 32 workgroups of an XCD read the same k-slabs of their 8 + 4 shared tiles at the same time), so whole XCDs are
// de-phased against each other: XCD x starts its first tile at k-step nt * x / 8, parks that partial tile (fp32 accumulators, 256 KB,
// in a workspace of its own: no other workgroup ever touches it), runs its remaining tiles, and finishes the first one last.  No
// idle time, no inter-workgroup traffic; the eight XCDs reach their epilogues an eighth of a tile apart.
// MEASURED (profiles/r3_gemm_experiments.md): 20-25 % SLOWER than one workgroup per tile -- inside one wave vmcnt counts stores and DMA
// requests in one queue, so a tile's epilogue stores gate the first DMA wait of the next tile, while a fresh workgroup starts its
// prologue beside the previous one's draining stores.  Kept as an opt-in experiment (TVL_GEMM_PERSIST=1), not on any default path.
template <int BM, int EPI, bool KS = false, bool CONV = false, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm_h2m_kernel(Tp3Params p) {
    TVL_KERNEL_ENTRY();
    constexpr int NW = 8, BN = 256, NP = 2;
    constexpr int WM = BM / 2;
    constexpr int TMo = WM / 16;                    // 16-row A operands per wave (8 or 6)
    constexpr int BLKP = NP * PIECE;
    constexpr int PA = NP * BM / 32, PB = NP * BN / 32, PT = PA + PB;   // 1-KiB pieces per 16-deep slab
    constexpr int PW = (PT + NW - 1) / NW;
    constexpr int STAGE = PT * PIECE, PAIR = 2 * STAGE;
    constexpr int NM = 3 * TMo * 4;                 // MFMAs per 32-deep step and wave
    static_assert(WM % 32 == 0 && (BM == 256 || BM == 192), "tile");
    static_assert(!PERSIST || (!KS && !CONV), "the persistent tile walk is built for the plain layer GEMMs");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int KB = p.K >> 4;
    const int nt = KB >> 1;                          // 32-deep steps (K % 32 == 0, K >= 96: nt >= 3)

    // ---- which tiles, in which order ------------------------------------------------------------------------------------------------
    // Tiles are numbered so that an XCD (hardware blocks b with equal b & 7 share one: observed placement, speed only) owns a contiguous
    // chunk, swept in groups of 8 row tiles x all column tiles.  Non-persistent: block b takes tile chunk_base(b & 7) + (b >> 3).
    // Persistent: block b takes chunk entries (b >> 3) + k * (gridDim.x >> 3), k = 0, 1, ...
    const int n_tiles = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int chunk_base, chunk_size;
    {
        const int total = PERSIST ? n_tiles : (int)gridDim.x;
        const int q = total >> 3, r = total & 7;
        chunk_base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        chunk_size = q + (xcd < r ? 1 : 0);
    }
    const int per_round = PERSIST ? (int)(gridDim.x >> 3) : 1;
    const int my_tiles = PERSIST ? (slot < chunk_size ? (chunk_size - slot + per_round - 1) / per_round : 0) : 1;
    if (my_tiles == 0) return;
    int split = 0;   // first k-step of the first tile's FIRST pass (0: the tile is not split)
    if constexpr (PERSIST) {
        split = (my_tiles >= 2 && p.work) ? (nt * xcd) >> 3 : 0;
        if (split < 3 || nt - split < 3) split = 0;
    }
    const int n_items = my_tiles + (split ? 1 : 0);

    const unsigned lane16 = lane * 16;
    const unsigned char* src[PW];
    int cv_row[2] = {-1, -1}, cv_yx[2] = {0, 0};
    int cv_dy = -1, cv_dx = -1, cv_cb = 0;
    auto conv_advance = [&]() {
        if constexpr (CONV) {
            if (++cv_dx == 2) {
                cv_dx = -1;
                if (++cv_dy == 2) { cv_dy = -1; ++cv_cb; }
            }
        }
    };
    // request piece i of this wave: 16-deep slab `slab` of the operands into LDS stage `stage`
    auto issue_piece = [&](auto idx, int slab, int stage) {
        constexpr int i = decltype(idx)::value;
        const int pc = wave + NW * i;
        const unsigned dst = lds0 + stage * STAGE + pc * PIECE;
        if constexpr (CONV && i < 2) {
            if (pc < PA) {   // wave-uniform
                const int iy = (cv_yx[i] >> 16) + cv_dy, ix = (cv_yx[i] & 0xffff) + cv_dx;
                const bool ok = cv_row[i] >= 0 && (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
                const int rin = ok ? cv_row[i] + cv_dy * p.cW + cv_dx : p.a_rb * 32;
                const unsigned long off = ((unsigned long)((rin >> 5) * p.cC16 + cv_cb)) * BLKP + (unsigned)((pc % NP) * PIECE + (((lane >> 5) * 32 + (rin & 31)) * 16));
                glds16(p.A + off, dst);
                return;
            }
        }
        if ((i + 1) * NW <= PT || pc < PT) {
            // keep the 64-bit base in SGPRs (saddr + 32-bit lane offset form of global_load_lds): left to itself hipcc strength-reduces
            // the slab offset into a VGPR pair per piece, and the 256-row tile has no registers to spare
            const unsigned long sb = (unsigned long)(src[i] + (long)slab * BLKP);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
            glds16(reinterpret_cast<const unsigned char*>(((unsigned long)hi << 32) | lo) + lane16, dst);
        }
    };
    auto issue_slab = [&](int slab, int stage) {
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue_piece(std::integral_constant<int, I>{}, slab, stage), ...); }(std::make_integer_sequence<int, PW>{});
        conv_advance();
    };

    // fragment addresses inside pair 0: lane (r, q) -> k block (q >> 1), slot (q & 1) * 32 + r
    const unsigned lane_off = (unsigned)(((lane >> 5) & 1) * STAGE + ((((lane >> 4) & 1) * 32) + (lane & 15)) * 16);
    const unsigned a_frag = lds0 + lane_off + wm * ((WM / 32) * BLKP);
    const unsigned b_frag = lds0 + lane_off + PA * PIECE + wn * (2 * BLKP);
    bf16x8 a[TMo][2], b[4][2];
    auto read_a = [&](auto mi, unsigned base) {
        constexpr int m = decltype(mi)::value;
        a[m][0] = lds_frag<(m >> 1) * BLKP + (m & 1) * 256>(base);
        a[m][1] = lds_frag<(m >> 1) * BLKP + (m & 1) * 256 + PIECE>(base);
    };
    auto read_b = [&](auto ni, unsigned base) {
        constexpr int n = decltype(ni)::value;
        b[n][0] = lds_frag<(n >> 1) * BLKP + (n & 1) * 256>(base);
        b[n][1] = lds_frag<(n >> 1) * BLKP + (n & 1) * 256 + PIECE>(base);
    };
    const unsigned ktab = lds0 + 4 * STAGE;
    float kratio[TMo];
    auto kratio_request = [&](int c) {
        if constexpr (KS) {
#pragma unroll
            for (int i = 0; i < TMo; ++i)
                asm volatile("ds_read_b32 %0, %1" : "=v"(kratio[i]) : "v"(ktab + (unsigned)(((wm * WM + i * 16 + (lane & 15)) * p.k_chunks + c) * 4)));
        }
    };
    f32x4 acc[TMo][4];
    int kbase = 0;   // first 16-deep slab of the pass being run (2 * its first k-step)

    // One 32-deep step, `t` counted from the pass's first step.  MFMA q = (i * 4 + j) * 3 + product; rows i ascending, columns j ascending.
    //   after MFMA 8 (row 0, columns 0-2 done)        the step's only synchronisation B_t (unless this is the last step)
    //   DMA && after MFMA 9 + 3 d                     piece d of the 2 PW pieces of slabs 2t + 4, 2t + 5 (into the pair this step reads)
    //   NEXT && one MFMA into row i + 1               A operand i of step t + 1 (from the other pair); in the last row, B operand j
    //                                                 one MFMA after its own three
    //   !FIRST && after MFMA 0                        the last A operand and the last B operand of THIS step (their registers were in use
    //                                                 until the previous step's last MFMA); first needed by MFMA 9, behind B_t's lgkmcnt(0)
    constexpr int SYNC_AT = 8;
    auto step = [&]<bool DMA, bool NEXT, bool FIRST>(StepKind<DMA, NEXT, FIRST>, int t) {
        const unsigned nxt_a = a_frag + ((t + 1) & 1) * PAIR, nxt_b = b_frag + ((t + 1) & 1) * PAIR;
        if constexpr (KS) {
            if ((t & 1) == 0 && t) {   // this step opens chunk t / 2 (64 columns of K): bring the accumulators to its scale
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TMo; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] *= kratio[i];
                const int cn = (t >> 1) + 1;
                kratio_request(cn < p.k_chunks ? cn : p.k_chunks - 1);
            }
        }
        // every fragment fetched so far has arrived (the inline-asm reads are invisible to hipcc's own waitcnt insertion): the prologue's
        // reads in the first step, the previous step's last refetches (B operands 1 and 2, issued 3-6 MFMAs ago) in the others
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        auto body = [&](auto idx) {
            constexpr int q = decltype(idx)::value;
            constexpr int prod = q % 3, ij = q / 3, i = ij / 4, j = ij % 4;
            // smallest piece products first: h0 h1, h1 h0, then h0 h0
            constexpr int pa = prod == 1 ? 1 : 0, pb = prod == 0 ? 1 : 0;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[j][pb]), __builtin_bit_cast(f16x8, a[i][pa]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!FIRST && q == 0) {
                read_a(std::integral_constant<int, TMo - 1>{}, a_frag + (t & 1) * PAIR);
                read_b(std::integral_constant<int, 3>{}, b_frag + (t & 1) * PAIR);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (q == SYNC_AT && !NEXT && !FIRST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (last step: only those two reads)
            if constexpr (NEXT && q == SYNC_AT) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next pair have landed (nothing younger is in flight)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // ... and its reads of the current pair are complete
                __builtin_amdgcn_s_barrier();                         // B_t: next pair visible to all, current pair free
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DMA && q > SYNC_AT && (q - SYNC_AT - 1) % 3 == 0 && (q - SYNC_AT - 1) / 3 < 2 * PW) {
                constexpr int d = (q - SYNC_AT - 1) / 3;
                issue_piece(std::integral_constant<int, d % PW>{}, kbase + 2 * t + 4 + d / PW, (2 * t + d / PW) & 3);
                if constexpr (d % PW == PW - 1) conv_advance();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NEXT) {
                // A operand i - 1 is dead once row i has started: refetch it one MFMA into the row
                if constexpr (i >= 1 && j == 0 && prod == 1) { read_a(std::integral_constant<int, i - 1>{}, nxt_a); __builtin_amdgcn_sched_barrier(0); }
                // last row: B operand j - 1 is dead after its three MFMAs; refetch it one MFMA later
                if constexpr (i == TMo - 1 && j >= 1 && prod == 1) { read_b(std::integral_constant<int, j - 1>{}, nxt_b); __builtin_amdgcn_sched_barrier(0); }
            }
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (body(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, NM>{});
    };

    float4* const park = PERSIST && p.work ? reinterpret_cast<float4*>(p.work) + ((long)blockIdx.x * NW + wave) * (TMo * 4) * 64 + lane : nullptr;

#pragma unroll 1
    for (int item = 0; item < n_items; ++item) {
        // item 0 with a split: k-steps [split, nt) of the first tile, accumulators parked; last item with a split: its k-steps [0, split)
        // on top of the parked accumulators; everything else: a whole tile
        const bool first_pass = split && item == 0, second_pass = split && item == n_items - 1;
        const int tile_no = second_pass ? 0 : item;
        const int t0 = first_pass ? split : 0, steps = first_pass ? nt - split : (second_pass ? split : nt);
        const int bid = chunk_base + slot + tile_no * per_round;
        constexpr int GROUP_M = 8;
        const int gsize_full = GROUP_M * p.tiles_n;
        const int group = bid / gsize_full;
        const int gm0 = group * GROUP_M;
        const int gm = p.tiles_m - gm0 < GROUP_M ? p.tiles_m - gm0 : GROUP_M;
        const int in_group = bid - group * gsize_full;
        int tile_m = gm0 + in_group % gm, tile_n = in_group / gm;
        if constexpr (BM == 192) if (p.row_walk) {   // (192-row tile only: any change of the 256-row instantiations, which sit on the 256-VGPR cliff, moves their spills)
            // few column tiles (N = 768: three): walk them fastest, so that all column tiles of a row tile sit side by side in ONE XCD's chunk of the
            // grid and A's rows are fetched into one L2 only -- the grouped order above puts a group's later columns into the next XCD's chunk
            // whenever a chunk (31-32 tiles) ends inside a group (traffic of out-proj / fc2: 1.43x algorithmic)
            tile_m = bid / p.tiles_n;
            tile_n = bid - tile_m * p.tiles_n;
        }
        kbase = 2 * t0;

        // this wave's DMA sources (slab 0): wave-uniform bases + one lane offset
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave + NW * i;
            if (pc < PA) {
                int rb = tile_m * (BM / 32) + pc / NP;
                rb = rb < p.a_rb ? rb : p.a_rb - 1;
                src[i] = p.A + ((long)rb * KB) * BLKP + (pc % NP) * PIECE;
            } else {
                const int q = pc < PT ? pc - PA : 0;
                int rb = tile_n * (BN / 32) + q / NP;
                rb = rb < p.b_rb ? rb : p.b_rb - 1;
                src[i] = p.B + ((long)rb * KB) * BLKP + (q % NP) * PIECE;
            }
        }
        static_assert(!CONV || PA <= 2 * NW, "conv: at most two A pieces per wave");
        if constexpr (CONV) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pc = wave + NW * i;
                if (pc < PA) {
                    const int r = tile_m * BM + (pc / NP) * 32 + (lane & 31);
                    const int hw = p.cH * p.cW;
                    const int rem = r % hw, oy = rem / p.cW;
                    cv_row[i] = r < p.M ? r : -1;
                    cv_yx[i] = (oy << 16) | (rem - oy * p.cW);
                }
            }
        }
        if (second_pass) {
            const float4* pk = park;
            asm volatile("" : "+v"(pk));   // (keeps hipcc from hoisting 32 address pairs out of the tile loop and spilling them)
#pragma unroll
            for (int i = 0; i < TMo; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = pk[(i * 4 + j) * 64];
                    acc[i][j][0] = v.x; acc[i][j][1] = v.y; acc[i][j][2] = v.z; acc[i][j][3] = v.w;
                }
        } else {
#pragma unroll
            for (int i = 0; i < TMo; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        }

        // KS: ratio table R[row][c] = inv[row][c-1] / inv[row][c] (c >= 1) behind the four stages, before any LDS-DMA is in flight
        if constexpr (KS) {
            float* tab = reinterpret_cast<float*>(smem + 4 * STAGE);
            const int nch = p.k_chunks;
            for (int e = threadIdx.x; e < BM * nch; e += 512) {
                const int rl = e / nch, c = e - rl * nch;
                long row = (long)tile_m * BM + rl;
                row = row < p.M ? row : p.M - 1;
                tab[e] = c ? p.a_kscale[row * nch + c - 1] / p.a_kscale[row * nch + c] : 1.0f;
            }
            __syncthreads();
        }

        // prologue: all four stages requested, wait for the first pair, fetch its fragments
        issue_slab(kbase + 0, 0);
        issue_slab(kbase + 1, 1);
        issue_slab(kbase + 2, 2);
        issue_slab(kbase + 3, 3);
        wait_groups_ct<PT, NW, 2>(wave < PT % NW);
        __builtin_amdgcn_s_barrier();
        [&]<int... I>(std::integer_sequence<int, I...>) { (read_a(std::integral_constant<int, I>{}, a_frag), ...); }(std::make_integer_sequence<int, TMo>{});
        [&]<int... I>(std::integer_sequence<int, I...>) { (read_b(std::integral_constant<int, I>{}, b_frag), ...); }(std::make_integer_sequence<int, 4>{});
        if constexpr (KS) kratio_request(1);

        // steps >= 3 (the host sends K < 96 to the 32x32x16 kernel; a split leaves >= 3 steps on either side): first step, steady steps,
        // the step with nothing left to request, the last
        step(StepKind<true, true, true>{}, 0);
#pragma unroll 1
        for (int t = 1; t < steps - 2; ++t) step(StepKind<true, true, false>{}, t);
        step(StepKind<false, true, false>{}, steps - 2);
        step(StepKind<false, false, false>{}, steps - 1);

        if (first_pass) {   // park the partial sums: 16 bytes per lane and instruction, 1 KB per wave-instruction, this workgroup's own slab
            float4* pk = park;
            asm volatile("" : "+v"(pk));
#pragma unroll
            for (int i = 0; i < TMo; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[(i * 4 + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            __syncthreads();   // every wave is past its last LDS read before the next pass's DMA overwrites the stages
        } else if constexpr (EPI >= 0 && (EPI & E_H2OUT) != 0 && (EPI & (E_F32 | E_TP3 | E_RES | E_RELU)) == 0 && !PERSIST) {
            epilogue16_direct<TMo, EPI>(p, acc, tile_m * BM + wm * WM, tile_n * BN + wn * 64, lane,   // registers -> images: no scratch, no barrier
                                        (((long)tile_m * p.tiles_n + tile_n) * NW + wave) * (TMo * 4));
        } else {
            if constexpr (EPI >= 0 && (EPI & E_F32) != 0 && (EPI & (E_TP3 | E_H2OUT | E_PRE | E_DQGELU | E_QGELU)) == 0 && (EPI & E_RSCALE) != 0 && !PERSIST) {
                if (p.f32_direct) {   // kernel-uniform
                    epilogue16_direct_f32<TMo, EPI>(p, acc, tile_m * BM + wm * WM, tile_n * BN + wn * 64, lane);
                    return;
                }
            }
            __syncthreads();  // every wave is past its last LDS read: the stages become epilogue scratch
            float* scratch = reinterpret_cast<float*>(smem) + wave * (TMo * 16 * 37);
            epilogue16<TMo, EPI>(p, acc, tile_m * BM + wm * WM, tile_n * BN + wn * 64, lane, scratch);
            if constexpr (PERSIST) __syncthreads();   // ... and stage memory again before the next tile's first DMA
        }
    }
}

template <int BM, int EPI, bool KS = false, bool CONV = false>
int launch_m(const Tp3Params& p0, hipStream_t s) {
    Tp3Params p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + 255) / 256;
    {
        static const int walk = getenv("TVL_GEMM_ROWWALK") ? atoi(getenv("TVL_GEMM_ROWWALK")) : 1;
        p.row_walk = (walk && !CONV && p.tiles_n <= 4) ? 1 : 0;
    }
    constexpr size_t stage_bytes = (size_t)4 * (2 * (BM + 256) / 32) * PIECE;
    constexpr size_t epi_bytes = (size_t)8 * (BM / 2) * 37 * sizeof(float);
    constexpr size_t ks_bytes = KS ? (size_t)BM * 64 * sizeof(float) : 0;
    constexpr size_t smem = (stage_bytes + ks_bytes) > epi_bytes ? (stage_bytes + ks_bytes) : epi_bytes;
    static_assert(smem <= 160 * 1024, "LDS budget");
    const long n_tiles = (long)p.tiles_m * p.tiles_n;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    // multi-round launches of the image-writing epilogues walk their tiles persistently (see the kernel): needs the caller's workspace
    // (PERSIST_GRID x 8 waves x BM/2 x 64 floats) and at least two tiles per workgroup of every XCD
#ifdef TVL_EXPERIMENTS   // (make EXPERIMENTS=1: the persistent instantiations spill 33-37 registers inside their tile loop and are not part of the product library)
    constexpr bool CAN_PERSIST = !KS && !CONV && EPI >= 0 && (EPI & (E_H2OUT | E_PRE)) != 0;
#else
    constexpr bool CAN_PERSIST = false;
#endif
    if constexpr (CAN_PERSIST) {
        constexpr int PERSIST_GRID = 256;
        constexpr size_t work_bytes = (size_t)PERSIST_GRID * 8 * (BM / 2 / 16 * 4) * 64 * sizeof(float4);
        if (p.work && p.work_bytes >= (long)work_bytes && n_tiles >= 2 * PERSIST_GRID + 8 && p.K >= 8 * 32 && getenv("TVL_GEMM_PERSIST") && getenv("TVL_GEMM_PERSIST")[0] == '1') {   // opt-in: measured SLOWER (see below)
            auto kern = gemm_h2m_kernel<BM, EPI, KS, CONV, true>;
            static int attr_mask = 0;
            if (!(attr_mask & (1 << dev))) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
                attr_mask |= 1 << dev;
            }
            hipLaunchKernelGGL(kern, dim3(PERSIST_GRID), dim3(512), smem, s, p);
            return 0;
        }
    }
    p.work = nullptr;
    auto kern = gemm_h2m_kernel<BM, EPI, KS, CONV, false>;
    static int attr_dev_mask = 0;
    if (!(attr_dev_mask & (1 << dev))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
        attr_dev_mask |= 1 << dev;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_tiles), dim3(512), smem, s, p);
    return 0;
}

template <int BM>
int launch_m_layer_epi(const Tp3Params& p, int epi, hipStream_t s) {
    switch (epi) {
        case E_BIAS | E_RSCALE | E_H2OUT: return launch_m<BM, E_BIAS | E_RSCALE | E_H2OUT>(p, s);                                       // qkv -> h2
        case E_RSCALE | E_H2OUT: return launch_m<BM, E_RSCALE | E_H2OUT>(p, s);                                                         // dO -> h2
        case E_BIAS | E_RES | E_F32 | E_RSCALE: return launch_m<BM, E_BIAS | E_RES | E_F32 | E_RSCALE>(p, s);                           // out_proj, fc2
        case E_F32 | E_RSCALE: return launch_m<BM, E_F32 | E_RSCALE>(p, s);                                                             // data gradients
        case E_BIAS | E_QGELU | E_PRE | E_RSCALE | E_H2OUT: return launch_m<BM, E_BIAS | E_QGELU | E_PRE | E_RSCALE | E_H2OUT>(p, s);   // fc1 -> h2 + z
        case E_BIAS | E_QGELU | E_RSCALE | E_H2OUT: return launch_m<BM, E_BIAS | E_QGELU | E_RSCALE | E_H2OUT>(p, s);                   // fc1, no tape
        case E_DQGELU | E_RSCALE | E_H2OUT: return launch_m<BM, E_DQGELU | E_RSCALE | E_H2OUT>(p, s);                                   // dz -> h2
        case E_BIAS | E_F32 | E_RSCALE: return launch_m<BM, E_BIAS | E_F32 | E_RSCALE>(p, s);                                           // frozen Linear / 1x1 conv
        case E_BIAS | E_RELU | E_F32 | E_RSCALE: return launch_m<BM, E_BIAS | E_RELU | E_F32 | E_RSCALE>(p, s);                         // ... + ReLU
        default: return launch_m<BM, -1>(p, s);
    }
}

// 3x3 conv as an implicit GEMM (A pieces gathered tap by tap): conv + folded BN + ReLU, the plain data gradient, anything else
template <int BM>
int launch_m_conv_epi(const Tp3Params& p, int epi, hipStream_t s) {
    constexpr int CONV_FWD = E_BIAS | E_RELU | E_F32 | E_RSCALE;
    if (epi == (E_F32 | E_RSCALE)) return launch_m<BM, E_F32 | E_RSCALE, false, true>(p, s);
    if (epi == CONV_FWD) return launch_m<BM, CONV_FWD, false, true>(p, s);
    return launch_m<BM, -1, false, true>(p, s);
}

}  // namespace

int tvl_gemm_h2m_t256(const void* params, int epi, hipStream_t s);   // gemm_h2m_t256.hip / gemm_h2m_t192.hip
int tvl_gemm_h2m_t192(const void* params, int epi, hipStream_t s);
int tvl_gemm_h2m_conv_t256(const void* params, int epi, hipStream_t s);
int tvl_gemm_h2m_conv_t192(const void* params, int epi, hipStream_t s);
int tvl_gemm_h2m_ks_t192(const void* params, hipStream_t s);         // A scaled per (row, 64-column chunk of K): the QKV data gradient
