// LayerNorm forward / backward over the last dimension, one wave64 per row, values held in
// registers (float4 per lane), two-pass variance like torch.  HBM-bound: each row is read once
// and written once (fwd), or dy/x read once and dx written once (bwd).
#include "common.h"
#include "tp3.h"

// No implicit multiply-add fusion in this file: the image-writing kernels must reproduce the fp32 kernels bit for bit
// (tests/test_hip_kernels.py::test_layernorm_tp3, ::test_layernorm_h2_forward_and_backward), and what -ffp-contract=fast fuses depends on the
// code around an expression.  Fused operations are written out (__builtin_fmaf); the kernels are HBM-bound, the lost fusions cost nothing.
#pragma clang fp contract(off)

namespace {

// y = ((x - mean) * rstd) * g + b with the last step as ONE fused multiply-add in every kernel of this file: the image-writing kernels must
// reproduce the fp32 kernel bit for bit (tests/test_hip_kernels.py::test_layernorm_tp3), which -ffp-contract's own choices do not guarantee
__device__ __forceinline__ float ln_y(float x, float mean, float rstd, float g, float b) { return __builtin_fmaf((x - mean) * rstd, g, b); }
// the same for the backward: xhat, g = dy gamma, the two row sums' terms and dx, every product either an explicit fma or a multiply that
// cannot be fused into a neighbour (__fmul_rn)
__device__ __forceinline__ float4 ln_xhat4(float4 x, float mean, float rstd) {
    return make_float4(__fmul_rn(x.x - mean, rstd), __fmul_rn(x.y - mean, rstd), __fmul_rn(x.z - mean, rstd), __fmul_rn(x.w - mean, rstd));
}
__device__ __forceinline__ float4 ln_mul4(float4 a, float4 b) { return make_float4(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y), __fmul_rn(a.z, b.z), __fmul_rn(a.w, b.w)); }
__device__ __forceinline__ float ln_sq4(float a, float b, float c, float d) {
    return __builtin_fmaf(a, a, __fmul_rn(b, b)) + __builtin_fmaf(c, c, __fmul_rn(d, d));
}
__device__ __forceinline__ float ln_sum4(float4 g) { return (g.x + g.y) + (g.z + g.w); }
__device__ __forceinline__ float ln_dot4(float4 g, float4 xh) {
    return __builtin_fmaf(g.x, xh.x, __fmul_rn(g.y, xh.y)) + __builtin_fmaf(g.z, xh.z, __fmul_rn(g.w, xh.w));
}
__device__ __forceinline__ float ln_dx1(float g, float xh, float m1, float m2, float rstd) { return __fmul_rn(rstd, __builtin_fmaf(-xh, m2, g - m1)); }
__device__ __forceinline__ float4 ln_dx4(float4 g, float4 xh, float m1, float m2, float rstd) {
    return make_float4(ln_dx1(g.x, xh.x, m1, m2, rstd), ln_dx1(g.y, xh.y, m1, m2, rstd), ln_dx1(g.z, xh.z, m1, m2, rstd), ln_dx1(g.w, xh.w, m1, m2, rstd));
}

// LN_MAXV float4 per lane held in registers: 4 -> cols <= 1024, 8 -> cols <= 2048 (the CRIS decoder's LayerNorm(2048))

template <bool VEC, int LN_MAXV = 4>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int cols, float eps) {
    TVL_KERNEL_ENTRY();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    const float inv_n = 1.0f / (float)cols;
    if (VEC) {
        const int nv = cols >> 2;  // float4 count
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[i] = reinterpret_cast<const float4*>(xr)[c];
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += ln_sq4(a, b, cc, d);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (beta) b = reinterpret_cast<const float4*>(beta)[c];
                float4 o;
                o.x = ln_y(v[i].x, mean, rstd, g.x, b.x);
                o.y = ln_y(v[i].y, mean, rstd, g.y, b.y);
                o.z = ln_y(v[i].z, mean, rstd, g.z, b.z);
                o.w = ln_y(v[i].w, mean, rstd, g.w, b.w);
                reinterpret_cast<float4*>(yr)[c] = o;
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    } else {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += xr[c];
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float d = xr[c] - mean;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        for (int c = lane; c < cols; c += 64) yr[c] = ln_y(xr[c], mean, rstd, gamma[c], (beta ? beta[c] : 0.f));
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma, xhat = (x-mean)*rstd ; [+ dres]
template <bool VEC, int LN_MAXV = 4>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols) {
    TVL_KERNEL_ENTRY();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    const float* dyr = dy + row * cols;
    float* dxr = dx + row * cols;
    const float* rr = dres ? dres + row * cols : nullptr;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const float inv_n = 1.0f / (float)cols;
    if (VEC) {
        const int nv = cols >> 2;
        float4 g[LN_MAXV], xh[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 d = reinterpret_cast<const float4*>(dyr)[c];
                const float4 xv = reinterpret_cast<const float4*>(xr)[c];
                const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
                xh[i] = ln_xhat4(xv, mean, rstd);
                g[i] = ln_mul4(d, gm);
                s1 += ln_sum4(g[i]);
                s2 += ln_dot4(g[i], xh[i]);
                if (dgamma) {
                    atomicAdd(&dgamma[4 * c + 0], d.x * xh[i].x); atomicAdd(&dgamma[4 * c + 1], d.y * xh[i].y);
                    atomicAdd(&dgamma[4 * c + 2], d.z * xh[i].z); atomicAdd(&dgamma[4 * c + 3], d.w * xh[i].w);
                }
                if (dbeta) {
                    atomicAdd(&dbeta[4 * c + 0], d.x); atomicAdd(&dbeta[4 * c + 1], d.y);
                    atomicAdd(&dbeta[4 * c + 2], d.z); atomicAdd(&dbeta[4 * c + 3], d.w);
                }
            }
        }
        const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o = ln_dx4(g[i], xh[i], m1, m2, rstd);
                if (rr) {
                    const float4 r4 = reinterpret_cast<const float4*>(rr)[c];
                    o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w;
                }
                reinterpret_cast<float4*>(dxr)[c] = o;
            }
        }
    } else {
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = dyr[c] * gamma[c];
            s1 += gg;
            s2 += gg * xh;
            if (dgamma) atomicAdd(&dgamma[c], dyr[c] * xh);
            if (dbeta) atomicAdd(&dbeta[c], dyr[c]);
        }
        const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = dyr[c] * gamma[c];
            float o = rstd * (gg - m1 - xh * m2);
            if (rr) o += rr[c];
            dxr[c] = o;
        }
    }
}

// ---- LayerNorm whose result feeds tvl_gemm_tp3 ---------------------------------------------------------------------------
// One workgroup = one 32-row block of the tp3 image.  Phase 1 is the kernel above (one wave per row, the row in registers,
// coalesced reads): it leaves the row statistics in LDS.  Phase 2 walks the block in fragment order -- lane (r, h) takes 8
// consecutive columns of row r (re-read from L2), normalises, splits into three bf16 pieces and the wave writes one 1-KiB
// piece per store instruction (fully coalesced).  The fp32 result is never materialised: its only consumer is the GEMM.
// NW waves per workgroup: 16 (two rows per wave in phase 1) keeps ~31 waves per CU in flight at 495 row blocks; with 4 the
// kernel ran at 8 waves per CU and a third of the HBM rate.
// NP = 3: tp3 output (three bf16 pieces).  NP = 2: h2 output (two fp16 pieces of the row scaled by a power of two, tp3.h): phase 1 also
// finds the row's largest |y| and writes the inverse scale to inv_scale[row] for the consuming GEMM's epilogue.
// STAGE (NP = 2, cols <= 1024): phase 1 leaves the block's y rows in LDS (32 rows x (cols + 4) floats, dynamic) and phase 2 reads its fragments
// from there instead of re-reading x through L2 in 32-byte pieces and recomputing y.
extern __shared__ __attribute__((aligned(16))) float s_stage[];
template <int LN_MAXV, int NW, int NP = 3, bool STAGE = false>
__global__ __launch_bounds__(64 * NW) void ln_fwd_tp3_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, unsigned char* __restrict__ out,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out, long rows, int cols,
                                                         float eps, float* __restrict__ inv_scale, float* __restrict__ row_norm,
                                                         unsigned long long* __restrict__ max_slot, unsigned tag) {
    TVL_KERNEL_ENTRY();
    __shared__ float s_mean[32], s_rstd[32], s_scale[32], s_norm[32];
    __shared__ float4 s_gb[2][LN_MAXV * 64];   // gamma | beta for the second phase (every lane of a half-wave reads the same 32 bytes there)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWS = STAGE ? 16 : 32;            // staged: 16 rows per workgroup (three 49 KB workgroups per CU at 768 columns)
    const long row0 = (long)blockIdx.x * ROWS;
    const long rb = row0 >> 5;                       // 32-row block of the image
    const int r_off = (int)(row0 & 31);
    const float inv_n = 1.0f / (float)cols;
    const int nv = cols >> 2;
    constexpr int RPW = ROWS / NW;   // rows per wave
    // gamma | beta: staged in LDS for phase 2 of the unstaged kernel (every lane of a half-wave reads the same 32 bytes there); the staged kernel
    // needs them in phase 1 only, where a lane's columns are fixed: straight into registers, issued with the first row's loads -- no staging
    // loop and no barrier in front of the rows (one memory latency less per workgroup)
    float4 gq[STAGE ? LN_MAXV : 1], bq[STAGE ? LN_MAXV : 1];
    if constexpr (STAGE) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i, cc = c < nv ? c : nv - 1;
            gq[i] = reinterpret_cast<const float4*>(gamma)[cc];
            bq[i] = beta ? reinterpret_cast<const float4*>(beta)[cc] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int c = threadIdx.x; c < 2 * nv; c += 64 * NW)
            s_gb[c >= nv][c >= nv ? c - nv : c] = c < nv ? reinterpret_cast<const float4*>(gamma)[c]
                                                         : (beta ? reinterpret_cast<const float4*>(beta)[c - nv] : make_float4(0.f, 0.f, 0.f, 0.f));
        __syncthreads();   // gamma | beta staged
    }
    // Every load of a row is issued before the first use of any of them (clamped indices instead of branches around the loads: a branch
    // per chunk made hipcc wait for each 16-byte load on its own -- eight HBM latencies in a row per wave, 31 us for a 97 MB pass).
    auto row_stats = [&](int rr) {
        const int rl = wave * RPW + rr;
        const long row = row0 + rl;
        const bool live_row = row < rows;   // wave-uniform
        const float4* xr = reinterpret_cast<const float4*>(x + (live_row ? row : rows - 1) * cols);
        float4 v[LN_MAXV];
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            v[i] = xr[c < nv ? c : nv - 1];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) s += (lane + 64 * i < nv) ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (lane + 64 * i < nv) ? ln_sq4(a, b, cc, d) : 0.f;
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        if (lane == 0 && live_row) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        if constexpr (NP == 2) {
            float amax = 0.f, ss = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c = lane + 64 * i, cc = c < nv ? c : nv - 1;
                const float4 g4 = STAGE ? gq[STAGE ? i : 0] : s_gb[0][cc], b4 = STAGE ? bq[STAGE ? i : 0] : s_gb[1][cc];
                const float y0 = ln_y(v[i].x, mean, rstd, g4.x, b4.x), y1 = ln_y(v[i].y, mean, rstd, g4.y, b4.y);
                const float y2 = ln_y(v[i].z, mean, rstd, g4.z, b4.z), y3 = ln_y(v[i].w, mean, rstd, g4.w, b4.w);
                amax = c < nv ? fmaxf(fmaxf(amax, fmaxf(fabsf(y0), fabsf(y1))), fmaxf(fabsf(y2), fabsf(y3))) : amax;
                ss += c < nv ? (y0 * y0 + y1 * y1) + (y2 * y2 + y3 * y3) : 0.f;
                if constexpr (STAGE) {
                    if (c < nv) *reinterpret_cast<float4*>(s_stage + rl * (cols + 4) + 4 * c) = make_float4(y0, y1, y2, y3);
                }
            }
            amax = wave_max(amax);
            ss = wave_sum(ss);
            const float inv = live_row ? h2::inv_scale_of(amax) : 1.0f;
            if (lane == 0) {
                const float nrm = live_row ? sqrtf(ss) * 1.0001f : 0.f;   // ||y row||_2, rounded up: it feeds a bound (tvl_gemm_h2_out)
                s_scale[rl] = 1.0f / inv; s_norm[rl] = nrm;
                if (live_row) {
                    inv_scale[row] = inv;
                    if (row_norm) row_norm[row] = nrm;
                }
            }
        }
        if (lane == 0) { s_mean[rl] = live_row ? mean : 0.f; s_rstd[rl] = live_row ? rstd : 0.f; }
    };
    if constexpr (LN_MAXV > 4) {   // wide rows: one row's registers at a time
#pragma unroll 1
        for (int rr = 0; rr < RPW; ++rr) row_stats(rr);
    } else {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) row_stats(rr);
    }
    __syncthreads();
    if constexpr (NP == 2) {   // largest row norm of the call: one tagged atomicMax per workgroup (include/tvl_hip.h)
        if (max_slot && wave == 0) {
            const float m = wave_max(lane < ROWS ? s_norm[lane] : 0.f);
            if (lane == 0) atomicMax(max_slot, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(m));
        }
    }
    const int r = lane & (ROWS - 1), h = (lane / ROWS) & 1, ks = lane / (2 * ROWS);   // ks: which of the wave's KPW k-blocks (ROWS = 16: two per pass)
    constexpr int KPW = 32 / ROWS;
    const long row = row0 + r;
    const bool live = row < rows;
    const float mean = s_mean[r], rstd = s_rstd[r];
    const int KB = cols >> 4;
    const float* xr = x + (live ? row : 0) * cols;
    constexpr int NIT = (LN_MAXV * 16 + NW * KPW - 1) / (NW * KPW);   // passes of this wave over its k-blocks (cols <= 256 LN_MAXV)
    constexpr int GRP = NIT < 4 ? NIT : 4;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT && (wave + NW * it0) * KPW < KB; it0 += GRP) {
    float4 pa[GRP], pb[GRP];
#pragma unroll
    for (int it = 0; it < GRP; ++it) {   // the rows of a group of the wave's blocks first (L2 hits: this workgroup has just read these rows)
        const int kb = (wave + NW * (it0 + it)) * KPW + ks, c0 = (kb < KB ? kb : KB - 1) * 16 + h * 8;
        if constexpr (STAGE) {
            pa[it] = *reinterpret_cast<const float4*>(s_stage + r * (cols + 4) + c0); pb[it] = *reinterpret_cast<const float4*>(s_stage + r * (cols + 4) + c0 + 4);
        } else {
            pa[it] = *reinterpret_cast<const float4*>(xr + c0); pb[it] = *reinterpret_cast<const float4*>(xr + c0 + 4);
        }
    }
#pragma unroll
    for (int it = 0; it < GRP; ++it) {
        const int kb = (wave + NW * (it0 + it)) * KPW + ks;
        if ((wave + NW * (it0 + it)) * KPW >= KB) break;   // wave-uniform
        if (kb >= KB) continue;                             // (ROWS = 16, odd number of k-blocks: the upper half-wave's last block)
        const int c4 = kb * 4 + h * 2;
        const float4 a = pa[it], b = pb[it], g0 = s_gb[0][c4], g1 = s_gb[0][c4 + 1], b0 = s_gb[1][c4], b1 = s_gb[1][c4 + 1];
        float v8[8];
        if constexpr (STAGE) {   // the staged rows ARE y (the same ln_y values phase 1 took the row maximum of)
            v8[0] = a.x; v8[1] = a.y; v8[2] = a.z; v8[3] = a.w; v8[4] = b.x; v8[5] = b.y; v8[6] = b.z; v8[7] = b.w;
        } else {
        v8[0] = ln_y(a.x, mean, rstd, g0.x, b0.x); v8[1] = ln_y(a.y, mean, rstd, g0.y, b0.y);
        v8[2] = ln_y(a.z, mean, rstd, g0.z, b0.z); v8[3] = ln_y(a.w, mean, rstd, g0.w, b0.w);
        v8[4] = ln_y(b.x, mean, rstd, g1.x, b1.x); v8[5] = ln_y(b.y, mean, rstd, g1.y, b1.y);
        v8[6] = ln_y(b.z, mean, rstd, g1.z, b1.z); v8[7] = ln_y(b.w, mean, rstd, g1.w, b1.w);
        }
        if (!live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = 0.f;
        }
        if constexpr (NP == 3) {
            uint4 pl[3];
            tp3::split8(v8, pl);
            unsigned char* o = out + (rb * KB + kb) * (long)tp3::BLK + (h * 32 + r_off + r) * 16;
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(o + s * tp3::PIECE) = pl[s];
        } else {
            const float sc = s_scale[r];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] *= sc;
            uint4 pl[2];
            h2::split8(v8, pl);
            unsigned char* o = out + (rb * KB + kb) * (long)h2::BLK + (h * 32 + r_off + r) * 16;
            *reinterpret_cast<uint4*>(o) = pl[0];
            *reinterpret_cast<uint4*>(o + h2::PIECE) = pl[1];
        }
    }
    }
}

// LayerNorm backward (+ residual gradient) writing dx twice: fp32 (the residual stream's gradient, read by the next
// LayerNorm backward) and tp3 (the A operand of the next data-gradient GEMM).  Phase 1 = ln_bwd_kernel's row pass; phase 2
// re-reads the block's fresh dx rows (this CU's own stores, drained and fenced) in fragment order.
template <int LN_MAXV, int NW, int NP = 3, bool STAGE = false>
__global__ __launch_bounds__(64 * NW) void ln_bwd_tp3_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const float* __restrict__ dres, float* __restrict__ dx, unsigned char* __restrict__ out,
                                                         long rows, int cols, float* __restrict__ inv_scale, float* __restrict__ row_norm,
                                                         unsigned long long* __restrict__ max_slot, unsigned tag) {
    TVL_KERNEL_ENTRY();
    __shared__ float s_scale[32], s_norm[32];
    __shared__ float4 s_g[LN_MAXV * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWS = STAGE ? 16 : 32;            // staged: 16 rows per workgroup (three 49 KB workgroups per CU at 768 columns)
    const long row0 = (long)blockIdx.x * ROWS;
    const long rb = row0 >> 5;                       // 32-row block of the image
    const int r_off = (int)(row0 & 31);
    const float inv_n = 1.0f / (float)cols;
    const int nv = cols >> 2;
    constexpr int RPW = ROWS / NW;
    for (int c = threadIdx.x; c < nv; c += 64 * NW) s_g[c] = reinterpret_cast<const float4*>(gamma)[c];
    __syncthreads();
    auto row_grad = [&](int rr) {
        const int rl = wave * RPW + rr;
        const long row = row0 + rl;
        const bool live_row = row < rows;   // wave-uniform
        const long rowc = live_row ? row : rows - 1;
        // all of the row's loads before the first use (see ln_fwd_tp3_kernel)
        float4 d[LN_MAXV], xv[LN_MAXV], r4[LN_MAXV];
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i, cc = c < nv ? c : nv - 1;
            d[i] = reinterpret_cast<const float4*>(dy + rowc * cols)[cc];
            xv[i] = reinterpret_cast<const float4*>(x + rowc * cols)[cc];
            r4[i] = dres ? reinterpret_cast<const float4*>(dres + rowc * cols)[cc] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float mean = mean_in[rowc], rstd = rstd_in[rowc];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            xv[i] = ln_xhat4(xv[i], mean, rstd);
            const bool in = lane + 64 * i < nv;
            d[i] = ln_mul4(d[i], s_g[in ? lane + 64 * i : nv - 1]);   // g
            s1 += in ? ln_sum4(d[i]) : 0.f;
            s2 += in ? ln_dot4(d[i], xv[i]) : 0.f;
        }
        const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
        float amax = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            float4 o = ln_dx4(d[i], xv[i], m1, m2, rstd);
            if (dres) { o.x += r4[i].x; o.y += r4[i].y; o.z += r4[i].z; o.w += r4[i].w; }
            if (c < nv && live_row) reinterpret_cast<float4*>(dx + row * cols)[c] = o;
            if constexpr (STAGE) {
                if (c < nv) *reinterpret_cast<float4*>(s_stage + rl * (cols + 4) + 4 * c) = o;
            }
            if constexpr (NP == 2) {
                const bool in = c < nv;
                amax = in ? fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w))) : amax;
                ss += in ? (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w) : 0.f;
            }
        }
        if constexpr (NP == 2) {
            amax = wave_max(amax);
            ss = wave_sum(ss);
            const float inv = live_row ? h2::inv_scale_of(amax) : 1.0f;
            if (lane == 0) {
                const float nrm = live_row ? sqrtf(ss) * 1.0001f : 0.f;
                s_scale[rl] = 1.0f / inv; s_norm[rl] = nrm;
                if (live_row) {
                    inv_scale[row] = inv;
                    if (row_norm) row_norm[row] = nrm;
                }
            }
        }
    };
    if constexpr (LN_MAXV > 4) {
#pragma unroll 1
        for (int rr = 0; rr < RPW; ++rr) row_grad(rr);
    } else {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) row_grad(rr);
    }
    // the block's dx rows are re-read by other waves of this workgroup: drain the stores, then make them visible (the lines
    // were never in this CU's L1, but the order store -> load across waves still needs the fence + barrier)
    if constexpr (!STAGE) {   // (staged: phase 2 reads the rows from LDS -- nothing to drain, the fp32 stores leave in the background)
        __threadfence_block();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if constexpr (NP == 2) {
        if (max_slot && wave == 0) {
            const float m = wave_max(lane < ROWS ? s_norm[lane] : 0.f);
            if (lane == 0) atomicMax(max_slot, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(m));
        }
    }
    const int r = lane & (ROWS - 1), h = (lane / ROWS) & 1, ks = lane / (2 * ROWS);   // ks: which of the wave's KPW k-blocks (ROWS = 16: two per pass)
    constexpr int KPW = 32 / ROWS;
    const long row = row0 + r;
    const bool live = row < rows;
    const int KB = cols >> 4;
    const float* dxr = dx + (live ? row : 0) * cols;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NIT = (LN_MAXV * 16 + NW * KPW - 1) / (NW * KPW);
    constexpr int GRP = NIT < 4 ? NIT : 4;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT && (wave + NW * it0) * KPW < KB; it0 += GRP) {
    f32x4 pa[GRP], pb[GRP];
#pragma unroll
    for (int it = 0; it < GRP; ++it) {
        const int kb = (wave + NW * (it0 + it)) * KPW + ks, c0 = (kb < KB ? kb : KB - 1) * 16 + h * 8;
        if constexpr (STAGE) {
            pa[it] = *reinterpret_cast<const f32x4*>(s_stage + r * (cols + 4) + c0);
            pb[it] = *reinterpret_cast<const f32x4*>(s_stage + r * (cols + 4) + c0 + 4);
        } else {
            pa[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dxr + c0));  // past this CU's L1
            pb[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dxr + c0 + 4));
        }
    }
#pragma unroll
    for (int it = 0; it < GRP; ++it) {
        const int kb = (wave + NW * (it0 + it)) * KPW + ks;
        if ((wave + NW * (it0 + it)) * KPW >= KB) break;   // wave-uniform
        if (kb >= KB) continue;
        const f32x4 a = pa[it], b = pb[it];
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (!live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if constexpr (NP == 3) {
            uint4 pl[3];
            tp3::split8(v, pl);
            unsigned char* o = out + (rb * KB + kb) * (long)tp3::BLK + (h * 32 + r_off + r) * 16;
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(o + s * tp3::PIECE) = pl[s];
        } else {
            const float sc = s_scale[r];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= sc;
            uint4 pl[2];
            h2::split8(v, pl);
            unsigned char* o = out + (rb * KB + kb) * (long)h2::BLK + (h * 32 + r_off + r) * 16;
            *reinterpret_cast<uint4*>(o) = pl[0];
            *reinterpret_cast<uint4*>(o + h2::PIECE) = pl[1];
        }
    }
    }
}

}  // namespace

extern "C" int tvl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int64_t rows, int32_t cols, float eps, tvlStream_t stream) {
    TVL_REQUIRE(x && gamma && y, "tvl_layernorm_fwd: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0, "tvl_layernorm_fwd: bad shape rows=%ld cols=%d", (long)rows, cols);
    const bool vec = (cols % 4 == 0) && cols <= 256 * 8 && tvl_aligned16(x) && tvl_aligned16(y) && tvl_aligned16(gamma) &&
                     (!beta || tvl_aligned16(beta));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec && cols <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<true, 4>), dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps);
    else if (vec) hipLaunchKernelGGL((ln_fwd_kernel<true, 8>), dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps);
    else hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps);
    TVL_LAUNCH_CHECK("tvl_layernorm_fwd");
    return 0;
}

extern "C" int tvl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 const float* dres, float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t cols,
                                 tvlStream_t stream) {
    TVL_REQUIRE(dy && x && gamma && mean && rstd && dx, "tvl_layernorm_bwd: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0, "tvl_layernorm_bwd: bad shape rows=%ld cols=%d", (long)rows, cols);
    const bool vec = (cols % 4 == 0) && cols <= 256 * 8 && tvl_aligned16(x) && tvl_aligned16(dy) && tvl_aligned16(dx) &&
                     tvl_aligned16(gamma) && (!dres || tvl_aligned16(dres));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec && cols <= 1024) hipLaunchKernelGGL((ln_bwd_kernel<true, 4>), dim3(grid), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols);
    else if (vec) hipLaunchKernelGGL((ln_bwd_kernel<true, 8>), dim3(grid), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols);
    else hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols);
    TVL_LAUNCH_CHECK("tvl_layernorm_bwd");
    return 0;
}

extern "C" int tvl_layernorm_fwd_tp3(const float* x, const float* gamma, const float* beta, void* y_tp3, float* mean, float* rstd,
                                     int64_t rows, int32_t cols, float eps, tvlStream_t stream) {
    TVL_REQUIRE(x && gamma && y_tp3, "tvl_layernorm_fwd_tp3: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0 && cols % 16 == 0 && cols <= 2048, "tvl_layernorm_fwd_tp3: need cols %% 16 == 0 and cols <= 2048 (rows=%ld cols=%d)", (long)rows, cols);
    TVL_REQUIRE(tvl_aligned16(x) && tvl_aligned16(y_tp3) && tvl_aligned16(gamma) && (!beta || tvl_aligned16(beta)), "tvl_layernorm_fwd_tp3: operands must be 16-byte aligned");
    const unsigned grid = (unsigned)((rows + 31) / 32);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char* out = reinterpret_cast<unsigned char*>(y_tp3);
    if (cols <= 1024) hipLaunchKernelGGL((ln_fwd_tp3_kernel<4, 16>), dim3(grid), dim3(1024), 0, s, x, gamma, beta, out, mean, rstd, (long)rows, cols, eps, (float*)nullptr, (float*)nullptr, (unsigned long long*)nullptr, 0u);
    else hipLaunchKernelGGL((ln_fwd_tp3_kernel<8, 8>), dim3(grid), dim3(512), 0, s, x, gamma, beta, out, mean, rstd, (long)rows, cols, eps, (float*)nullptr, (float*)nullptr, (unsigned long long*)nullptr, 0u);
    TVL_LAUNCH_CHECK("tvl_layernorm_fwd_tp3");
    return 0;
}

extern "C" int tvl_layernorm_bwd_tp3(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     const float* dres, float* dx, void* dx_tp3, int64_t rows, int32_t cols, tvlStream_t stream) {
    TVL_REQUIRE(dy && x && gamma && mean && rstd && dx && dx_tp3, "tvl_layernorm_bwd_tp3: null pointer");
    TVL_REQUIRE(rows > 0 && cols > 0 && cols % 16 == 0 && cols <= 2048, "tvl_layernorm_bwd_tp3: need cols %% 16 == 0 and cols <= 2048 (rows=%ld cols=%d)", (long)rows, cols);
    TVL_REQUIRE(tvl_aligned16(x) && tvl_aligned16(dy) && tvl_aligned16(dx) && tvl_aligned16(dx_tp3) && tvl_aligned16(gamma) && (!dres || tvl_aligned16(dres)),
                "tvl_layernorm_bwd_tp3: operands must be 16-byte aligned");
    const unsigned grid = (unsigned)((rows + 31) / 32);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char* out = reinterpret_cast<unsigned char*>(dx_tp3);
    if (cols <= 1024) hipLaunchKernelGGL((ln_bwd_tp3_kernel<4, 16>), dim3(grid), dim3(1024), 0, s, dy, x, gamma, mean, rstd, dres, dx, out, (long)rows, cols, (float*)nullptr, (float*)nullptr, (unsigned long long*)nullptr, 0u);
    else hipLaunchKernelGGL((ln_bwd_tp3_kernel<8, 8>), dim3(grid), dim3(512), 0, s, dy, x, gamma, mean, rstd, dres, dx, out, (long)rows, cols, (float*)nullptr, (float*)nullptr, (unsigned long long*)nullptr, 0u);
    TVL_LAUNCH_CHECK("tvl_layernorm_bwd_tp3");
    return 0;
}

// The same two kernels writing the h2 operand format (two fp16 pieces of the row scaled by a power of two) + the rows' inverse scales.
extern "C" int tvl_layernorm_fwd_h2(const float* x, const float* gamma, const float* beta, void* y_h2, float* inv_scale, float* row_norm, float* mean, float* rstd,
                                    int64_t rows, int32_t cols, float eps, uint64_t* max_slot, uint32_t tag, tvlStream_t stream) {
    TVL_REQUIRE(x && gamma && y_h2 && inv_scale, "tvl_layernorm_fwd_h2: null pointer");
    TVL_REQUIRE(!max_slot || ((uintptr_t)max_slot % 8 == 0), "tvl_layernorm_fwd_h2: max_slot must be 8-byte aligned");
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(max_slot);
    TVL_REQUIRE(rows > 0 && cols > 0 && cols % 16 == 0 && cols <= 2048, "tvl_layernorm_fwd_h2: need cols %% 16 == 0 and cols <= 2048 (rows=%ld cols=%d)", (long)rows, cols);
    TVL_REQUIRE(tvl_aligned16(x) && tvl_aligned16(y_h2) && tvl_aligned16(gamma) && (!beta || tvl_aligned16(beta)), "tvl_layernorm_fwd_h2: operands must be 16-byte aligned");
    const unsigned grid = (unsigned)((rows + 31) / 32);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char* out = reinterpret_cast<unsigned char*>(y_h2);
    static const bool stage = !(getenv("TVL_LN_STAGE") && atoi(getenv("TVL_LN_STAGE")) == 0);
    if (cols <= 1024 && stage) {
        const size_t lds = 16 * (size_t)(cols + 4) * sizeof(float);   // 16 rows per workgroup of 8 waves
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_fwd_tp3_kernel<4, 8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (1024 + 4) * 4);
        TVL_REQUIRE(attr == hipSuccess, "tvl_layernorm_fwd_h2: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL((ln_fwd_tp3_kernel<4, 8, 2, true>), dim3(2 * grid) /* both halves of the last 32-row image block: rows past the end are written as zeros */, dim3(512), lds, s, x, gamma, beta, out, mean, rstd, (long)rows, cols, eps, inv_scale, row_norm, slot, (unsigned)tag);
    } else if (stage) {   // wide rows: one 131 KB workgroup per CU
        const size_t lds = 16 * (size_t)(cols + 4) * sizeof(float);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_fwd_tp3_kernel<8, 8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (2048 + 4) * 4);
        TVL_REQUIRE(attr == hipSuccess, "tvl_layernorm_fwd_h2: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL((ln_fwd_tp3_kernel<8, 8, 2, true>), dim3(2 * grid), dim3(512), lds, s, x, gamma, beta, out, mean, rstd, (long)rows, cols, eps, inv_scale, row_norm, slot, (unsigned)tag);
    } else if (cols <= 1024) hipLaunchKernelGGL((ln_fwd_tp3_kernel<4, 16, 2>), dim3(grid), dim3(1024), 0, s, x, gamma, beta, out, mean, rstd, (long)rows, cols, eps, inv_scale, row_norm, slot, (unsigned)tag);
    else hipLaunchKernelGGL((ln_fwd_tp3_kernel<8, 8, 2>), dim3(grid), dim3(512), 0, s, x, gamma, beta, out, mean, rstd, (long)rows, cols, eps, inv_scale, row_norm, slot, (unsigned)tag);
    TVL_LAUNCH_CHECK("tvl_layernorm_fwd_h2");
    return 0;
}

extern "C" int tvl_layernorm_bwd_h2(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                    const float* dres, float* dx, void* dx_h2, float* inv_scale, float* row_norm, int64_t rows, int32_t cols,
                                    uint64_t* max_slot, uint32_t tag, tvlStream_t stream) {
    TVL_REQUIRE(dy && x && gamma && mean && rstd && dx && dx_h2 && inv_scale, "tvl_layernorm_bwd_h2: null pointer");
    TVL_REQUIRE(!max_slot || ((uintptr_t)max_slot % 8 == 0), "tvl_layernorm_bwd_h2: max_slot must be 8-byte aligned");
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(max_slot);
    TVL_REQUIRE(rows > 0 && cols > 0 && cols % 16 == 0 && cols <= 2048, "tvl_layernorm_bwd_h2: need cols %% 16 == 0 and cols <= 2048 (rows=%ld cols=%d)", (long)rows, cols);
    TVL_REQUIRE(tvl_aligned16(x) && tvl_aligned16(dy) && tvl_aligned16(dx) && tvl_aligned16(dx_h2) && tvl_aligned16(gamma) && (!dres || tvl_aligned16(dres)),
                "tvl_layernorm_bwd_h2: operands must be 16-byte aligned");
    const unsigned grid = (unsigned)((rows + 31) / 32);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char* out = reinterpret_cast<unsigned char*>(dx_h2);
    static const bool stage = !(getenv("TVL_LN_STAGE") && atoi(getenv("TVL_LN_STAGE")) == 0);
    if (cols <= 1024 && stage) {
        const size_t lds = 16 * (size_t)(cols + 4) * sizeof(float);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_tp3_kernel<4, 8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (1024 + 4) * 4);
        TVL_REQUIRE(attr == hipSuccess, "tvl_layernorm_bwd_h2: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL((ln_bwd_tp3_kernel<4, 8, 2, true>), dim3(2 * grid) /* both halves of the last 32-row image block: rows past the end are written as zeros */, dim3(512), lds, s, dy, x, gamma, mean, rstd, dres, dx, out, (long)rows, cols, inv_scale, row_norm, slot, (unsigned)tag);
    } else if (stage) {
        const size_t lds = 16 * (size_t)(cols + 4) * sizeof(float);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_tp3_kernel<8, 8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (2048 + 4) * 4);
        TVL_REQUIRE(attr == hipSuccess, "tvl_layernorm_bwd_h2: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL((ln_bwd_tp3_kernel<8, 8, 2, true>), dim3(2 * grid), dim3(512), lds, s, dy, x, gamma, mean, rstd, dres, dx, out, (long)rows, cols, inv_scale, row_norm, slot, (unsigned)tag);
    } else if (cols <= 1024) hipLaunchKernelGGL((ln_bwd_tp3_kernel<4, 16, 2>), dim3(grid), dim3(1024), 0, s, dy, x, gamma, mean, rstd, dres, dx, out, (long)rows, cols, inv_scale, row_norm, slot, (unsigned)tag);
    else hipLaunchKernelGGL((ln_bwd_tp3_kernel<8, 8, 2>), dim3(grid), dim3(512), 0, s, dy, x, gamma, mean, rstd, dres, dx, out, (long)rows, cols, inv_scale, row_norm, slot, (unsigned)tag);
    TVL_LAUNCH_CHECK("tvl_layernorm_bwd_h2");
    return 0;
}

