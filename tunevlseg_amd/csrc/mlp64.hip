// The feed-forward half of a CLIPSeg decoder layer (HF CLIPSegDecoderLayer, modeling_clipseg.py:393-410, with the decoder's
// hidden_act = "relu", hidden size reduce_dim = 64):   out = LayerNorm2(x + W2 relu(W1 x + b1) + b2)   -- ONE kernel forward, ONE backward.
//
// Why.  At M = B * 485 rows and 64 channels the op-by-op path is memory traffic and launches: fc1 writes an [M, 2048] fp32
// pre-activation AND activation (2 x 127 MB), fc2 reads it back through a split-K pass, the backward repeats both -- ~160 us forward and
// ~200 us backward per layer for 8 + 12 GFLOP.  Here the 2048-wide intermediate never leaves the registers:
//   stage 1   Z^T[f, r] = W1[f, :] . x[r, :]            (A = weights, B = the 64-row tile of x: the row index r sits on the LANE)
//   relu, split into two fp16 pieces IN the accumulator registers
//   stage 2   out^T[c, r] += W2[c, f] . relu(Z)[f, r]   (the stage-1 accumulators are the B operand as they stand: a 32 x 32 result has
//             its column on the lane and its rows in the 16 registers -- cdna_hip_programming.md "An accumulator tile as the next MFMA's
//             operand" -- so the contraction index f needs no lane movement and no LDS; the frozen W2 is packed once with the matching
//             k permutation)
// and the backward recomputes Z the same way (K = 64: a third of its MFMAs) instead of reading a saved [M, 2048] pre-activation.
//
// Arithmetic: two fp16 pieces per operand scaled by exact powers of two, three MFMAs per product (h0 h1 + h1 h0 + h0 h0), fp32
// accumulation -- the "h2" arithmetic of the vision tower's GEMMs (gemm_h2.hip): x rows and LN-gradient rows by their exact maximum,
// relu(Z) / dZ rows by a Cauchy-Schwarz bound (the row is not known before it is written), weights per tensor.
//
// Work split: a workgroup = 64 rows (two 32-row MFMA column blocks per wave) x 4 waves, wave w owning a quarter of the 2048 hidden
// units; weight fragments go global -> registers in fragment order (1 KB per wave-instruction, L2-resident: 1 MB for all four images),
// the four partial out^T tiles meet in LDS.  ceil(M / 64) = 243 workgroups at the headline batch.
#include "common.h"
#include "tp3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct Mlp64P {
    const float* x;          // [M, 64] the block's input (post-LN1)
    const unsigned char* img;   // the four weight images (tvl_mlp64_pack)
    const float* b1; const float* b2; const float* gamma; const float* beta;
    float* out; float* t2; float* mean; float* rstd;     // forward outputs (t2 / mean / rstd may be null)
    const float* dout; const float* t2_in; const float* mean_in; const float* rstd_in; float* dx;   // backward
    long M; int F;
    float inv_w1, inv_w2;    // inverse tensor scales of the W1 / W2 images
    float w1_rownorm, b1_max;   // max_f ||W1[f, :]||_2 and max |b1|: |z[r, f]| <= ||x_r|| w1_rownorm + b1_max
    float w2_colnorm;        // max_f ||W2[:, f]||_2: |dA[r, f]| <= ||dt2_r|| w2_colnorm
    float eps;
};

__device__ __forceinline__ f32x16 mma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);   // smallest piece products first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
}

__device__ __forceinline__ void split8h(const float (&v)[8], f16x8 (&out)[2]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h0 = (_Float16)v[e];
        out[0][e] = h0;
        out[1][e] = (_Float16)(v[e] - (float)h0);
    }
}

__device__ __forceinline__ void ldfrag2(const unsigned char* p, f16x8 (&f)[2]) {   // two pieces, 1 KB apart
    f[0] = *reinterpret_cast<const f16x8*>(p);
    f[1] = *reinterpret_cast<const f16x8*>(p + 1024);
}

// image geometry (bytes): per 32-unit hidden block `fb`
//   img1 (stage 1, A = W1 rows f, k = d):           frag (fb, s, piece)            at ((fb * 4 + s) * 2 + piece) * 1024
//   img2 (fwd stage 2, A = W2 rows c, k = f perm):  frag (fb, cb, s', piece)       at (((fb * 2 + cb) * 2 + s') * 2 + piece) * 1024
//   img3 (bwd, A = W2^T rows f, k = c):             as img1
//   img4 (bwd, A = W1^T rows d, k = f perm):        as img2
// each image F * 64 * 4 bytes; k permutation of a 16-deep step s' inside a 32-unit block: element j of lane half h is
// unit 16 s' + 8 (j >> 2) + 4 h + (j & 3) -- the order in which a 32 x 32 accumulator hands its rows to the next MFMA.
__device__ __forceinline__ long img_bytes(int F) { return (long)F * 64 * 4; }

// rows of one 32-row column block in B-operand order: lane (r, h) holds x[row, 16 s + 8 h + j]; returns amax and sum of squares of the row
__device__ __forceinline__ void load_rows(const float* __restrict__ base, long row, int h, float (&v)[4][8]) {
    const float* xr = base + row * 64 + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(xr + 16 * s), b = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
        v[s][0] = a.x; v[s][1] = a.y; v[s][2] = a.z; v[s][3] = a.w; v[s][4] = b.x; v[s][5] = b.y; v[s][6] = b.z; v[s][7] = b.w;
    }
}

__device__ __forceinline__ void row_amax_ss(const float (&v)[4][8], float& amax, float& ss) {
    float m = 0.f, q = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { m = fmaxf(m, fabsf(v[s][j])); q += v[s][j] * v[s][j]; }
    amax = xor32_max(m);
    ss = xor32_sum(q);
}

__device__ __forceinline__ void to_frags(const float (&v)[4][8], float scale, f16x8 (&out)[4][2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = v[s][j] * scale;
        split8h(w, out[s]);
    }
}

// cross-wave sum of the four partial 64 x 64 tiles: wave w finalises (column block nb = w >> 1, channel block cb = w & 1); the other
// three waves' partials of that quarter travel through LDS (48 KB), summed in wave order
typedef float4 Red[4][3][4][64];

__device__ __forceinline__ void reduce_quarters(Red& red, const f32x16 (&o)[2][2], int wave, int lane, float (&t)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q == wave) continue;
        const int si = wave < q ? wave : wave - 1;
        const f32x16& src = o[q >> 1][q & 1];
#pragma unroll
        for (int g = 0; g < 4; ++g) red[q][si][g][lane] = make_float4(src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]);
    }
    __syncthreads();
    f32x16 own;   // this wave's own quarter, picked by compares (a run-time index into the register arrays would send them to scratch)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q == wave) own = o[q >> 1][q & 1];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float4 v;
            if (w == wave) v = make_float4(own[4 * g], own[4 * g + 1], own[4 * g + 2], own[4 * g + 3]);
            else v = red[wave][w < wave ? w : w - 1][g][lane];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        t[4 * g] = acc.x; t[4 * g + 1] = acc.y; t[4 * g + 2] = acc.z; t[4 * g + 3] = acc.w;
    }
}

__global__ __launch_bounds__(256) void mlp64_fwd_kernel(Mlp64P p) {
    TVL_KERNEL_ENTRY();
    __shared__ Red red;
    __shared__ float stat[2][2][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * 64;
    const unsigned char* img1 = p.img + lane * 16;
    const unsigned char* img2 = p.img + img_bytes(p.F) + lane * 16;

    f16x8 xh[2][4][2];
    float inv_x[2], inv_a[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        long row = row0 + nb * 32 + r;
        row = row < p.M ? row : p.M - 1;
        float v[4][8], amax, ss;
        load_rows(p.x, row, h, v);
        row_amax_ss(v, amax, ss);
        inv_x[nb] = h2::inv_scale_of(amax);
        to_frags(v, 1.0f / inv_x[nb], xh[nb]);
        inv_a[nb] = h2::inv_scale_of(sqrtf(ss) * 1.0001f * p.w1_rownorm + p.b1_max);
    }

    const int nblk = p.F >> 7;   // 32-unit hidden blocks per wave
    const int fb0 = wave * nblk;
    f16x8 w1[4][2], w2[2][2][2];
    float4 bb[4];
    auto load_w1 = [&](int fb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) ldfrag2(img1 + (long)(fb * 4 + s) * 2048, w1[s]);
#pragma unroll
        for (int g = 0; g < 4; ++g) bb[g] = *reinterpret_cast<const float4*>(p.b1 + fb * 32 + 8 * g + 4 * h);
    };
    auto load_w2 = [&](int fb) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < 2; ++s) ldfrag2(img2 + (long)((fb * 2 + cb) * 2 + s) * 2048, w2[cb][s]);
    };
    load_w1(fb0);
    load_w2(fb0);
    f32x16 o[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[nb][cb][i] = 0.f;

#pragma unroll 1
    for (int it = 0; it < nblk; ++it) {
        const int fb = fb0 + it;
        f32x16 z[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) z[nb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) z[nb] = mma3(w1[s], xh[nb][s], z[nb]);
        }
        float bias[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) { bias[4 * g] = bb[g].x; bias[4 * g + 1] = bb[g].y; bias[4 * g + 2] = bb[g].z; bias[4 * g + 3] = bb[g].w; }
        if (it + 1 < nblk) load_w1(fb + 1);   // the registers are free: the next block's fragments travel under this block's second stage
        f16x8 ah[2][2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float fz = inv_x[nb] * p.inv_w1, sa = 1.0f / inv_a[nb];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(z[nb][8 * s + j] * fz + bias[8 * s + j], 0.f) * sa;
                split8h(v, ah[nb][s]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int s = 0; s < 2; ++s) o[nb][cb] = mma3(w2[cb][s], ah[nb][s], o[nb][cb]);
        if (it + 1 < nblk) load_w2(fb + 1);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float f = inv_a[nb] * p.inv_w2;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[nb][cb][i] *= f;
    }
    float t[16];
    reduce_quarters(red, o, wave, lane, t);
    const int nb = wave >> 1, cb = wave & 1;
    const long row = row0 + nb * 32 + r;
    const bool live = row < p.M;
    const long rl = live ? row : p.M - 1;
    float s1 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = cb * 32 + 8 * g + 4 * h;
        const float4 xr = *reinterpret_cast<const float4*>(p.x + rl * 64 + c), b2 = *reinterpret_cast<const float4*>(p.b2 + c);
        t[4 * g] += b2.x + xr.x; t[4 * g + 1] += b2.y + xr.y; t[4 * g + 2] += b2.z + xr.z; t[4 * g + 3] += b2.w + xr.w;
        if (p.t2 && live) *reinterpret_cast<float4*>(p.t2 + row * 64 + c) = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
        s1 += (t[4 * g] + t[4 * g + 1]) + (t[4 * g + 2] + t[4 * g + 3]);
    }
    s1 = xor32_sum(s1);
    if (h == 0) stat[nb][cb][r] = s1;
    __syncthreads();
    const float mean = (stat[nb][0][r] + stat[nb][1][r]) * (1.0f / 64.0f);
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = t[i] - mean; q += d * d; }
    q = xor32_sum(q);
    if (h == 0) stat[nb][cb][r] = q;
    __syncthreads();
    const float rstd = rsqrtf((stat[nb][0][r] + stat[nb][1][r]) * (1.0f / 64.0f) + p.eps);
    if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cb * 32 + 8 * g + 4 * h;
            const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c), bt = *reinterpret_cast<const float4*>(p.beta + c);
            *reinterpret_cast<float4*>(p.out + row * 64 + c) =
                make_float4((t[4 * g] - mean) * rstd * gm.x + bt.x, (t[4 * g + 1] - mean) * rstd * gm.y + bt.y,
                            (t[4 * g + 2] - mean) * rstd * gm.z + bt.z, (t[4 * g + 3] - mean) * rstd * gm.w + bt.w);
        }
        if (cb == 0 && h == 0) {
            if (p.mean) p.mean[row] = mean;
            if (p.rstd) p.rstd[row] = rstd;
        }
    }
}

// dx = dt2 + W1^T [ (W2^T dt2) * (W1 x + b1 > 0) ]   with   dt2 = LayerNorm2'(dout)   (per row: rstd (g - mean(g) - xhat mean(g xhat)), g = dout gamma)
__global__ __launch_bounds__(256) void mlp64_bwd_kernel(Mlp64P p) {
    TVL_KERNEL_ENTRY();
    __shared__ Red red;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * 64;
    const unsigned char* img1 = p.img + lane * 16;
    const unsigned char* img3 = p.img + 2 * img_bytes(p.F) + lane * 16;
    const unsigned char* img4 = p.img + 3 * img_bytes(p.F) + lane * 16;

    f16x8 xh[2][4][2], dth[2][4][2];
    float inv_x[2], inv_d[2], inv_dz[2], m1[2], m2[2], mu[2], rs[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        long row = row0 + nb * 32 + r;
        row = row < p.M ? row : p.M - 1;
        float v[4][8], amax, ss;
        load_rows(p.x, row, h, v);
        row_amax_ss(v, amax, ss);
        inv_x[nb] = h2::inv_scale_of(amax);
        to_frags(v, 1.0f / inv_x[nb], xh[nb]);
        // LayerNorm2 backward of this row, in the same (lane, k) order
        float dy[4][8], tt[4][8];
        load_rows(p.dout, row, h, dy);
        load_rows(p.t2_in, row, h, tt);
        mu[nb] = p.mean_in[row];
        rs[nb] = p.rstd_in[row];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + 16 * s + 8 * h), g1 = *reinterpret_cast<const float4*>(p.gamma + 16 * s + 8 * h + 4);
            const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                tt[s][j] = (tt[s][j] - mu[nb]) * rs[nb];   // xhat
                dy[s][j] *= gm[j];                          // g
                a1 += dy[s][j];
                a2 += dy[s][j] * tt[s][j];
            }
        }
        a1 = xor32_sum(a1);
        a2 = xor32_sum(a2);
        m1[nb] = a1 * (1.0f / 64.0f);
        m2[nb] = a2 * (1.0f / 64.0f);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) dy[s][j] = rs[nb] * (dy[s][j] - m1[nb] - tt[s][j] * m2[nb]);   // dt2
        row_amax_ss(dy, amax, ss);
        inv_d[nb] = h2::inv_scale_of(amax);
        to_frags(dy, 1.0f / inv_d[nb], dth[nb]);
        inv_dz[nb] = h2::inv_scale_of(sqrtf(ss) * 1.0001f * p.w2_colnorm);
    }

    const int nblk = p.F >> 7;
    const int fb0 = wave * nblk;
    f16x8 w1[4][2], w3[4][2], w4[2][2][2];
    float4 bb[4];
    auto load_w1 = [&](int fb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) ldfrag2(img1 + (long)(fb * 4 + s) * 2048, w1[s]);
#pragma unroll
        for (int g = 0; g < 4; ++g) bb[g] = *reinterpret_cast<const float4*>(p.b1 + fb * 32 + 8 * g + 4 * h);
    };
    auto load_w3 = [&](int fb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) ldfrag2(img3 + (long)(fb * 4 + s) * 2048, w3[s]);
    };
    auto load_w4 = [&](int fb) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s) ldfrag2(img4 + (long)((fb * 2 + db) * 2 + s) * 2048, w4[db][s]);
    };
    load_w1(fb0);
    load_w3(fb0);
    load_w4(fb0);
    f32x16 o[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[nb][db][i] = 0.f;

#pragma unroll 1
    for (int it = 0; it < nblk; ++it) {
        const int fb = fb0 + it;
        f32x16 z[2], da[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) z[nb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) z[nb] = mma3(w1[s], xh[nb][s], z[nb]);
        }
        float bias[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) { bias[4 * g] = bb[g].x; bias[4 * g + 1] = bb[g].y; bias[4 * g + 2] = bb[g].z; bias[4 * g + 3] = bb[g].w; }
        if (it + 1 < nblk) load_w1(fb + 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) da[nb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) da[nb] = mma3(w3[s], dth[nb][s], da[nb]);
        }
        if (it + 1 < nblk) load_w3(fb + 1);
        f16x8 dzh[2][2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float fz = inv_x[nb] * p.inv_w1, fd = inv_d[nb] * p.inv_w2 / inv_dz[nb];   // powers of two: exact
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (z[nb][8 * s + j] * fz + bias[8 * s + j] > 0.f) ? da[nb][8 * s + j] * fd : 0.f;
                split8h(v, dzh[nb][s]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) o[nb][db] = mma3(w4[db][s], dzh[nb][s], o[nb][db]);
        if (it + 1 < nblk) load_w4(fb + 1);
    }
    // per-lane scalars of the finalising quarter (nb = wave >> 1) must not be indexed dynamically: select
    float f_sc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        f_sc[nb] = inv_dz[nb] * p.inv_w1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[nb][db][i] *= f_sc[nb];
    }
    float t[16];
    reduce_quarters(red, o, wave, lane, t);
    const int nb = wave >> 1, db = wave & 1;
    const long row = row0 + nb * 32 + r;
    if (row < p.M) {
        const float mu_ = nb ? mu[1] : mu[0], rs_ = nb ? rs[1] : rs[0], m1_ = nb ? m1[1] : m1[0], m2_ = nb ? m2[1] : m2[0];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = db * 32 + 8 * g + 4 * h;
            const float4 dy = *reinterpret_cast<const float4*>(p.dout + row * 64 + c), tv = *reinterpret_cast<const float4*>(p.t2_in + row * 64 + c);
            const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c);
            float4 o4;
            o4.x = t[4 * g] + rs_ * (dy.x * gm.x - m1_ - (tv.x - mu_) * rs_ * m2_);
            o4.y = t[4 * g + 1] + rs_ * (dy.y * gm.y - m1_ - (tv.y - mu_) * rs_ * m2_);
            o4.z = t[4 * g + 2] + rs_ * (dy.z * gm.z - m1_ - (tv.z - mu_) * rs_ * m2_);
            o4.w = t[4 * g + 3] + rs_ * (dy.w * gm.w - m1_ - (tv.w - mu_) * rs_ * m2_);
            *reinterpret_cast<float4*>(p.dx + row * 64 + c) = o4;
        }
    }
}

// weight images: thread = (image, hidden block fb, sub-fragment, lane); writes both pieces (16 bytes each)
__global__ __launch_bounds__(256) void mlp64_pack_kernel(const float* __restrict__ W1, const float* __restrict__ W2, int F, float s1, float s2,
                                                         unsigned char* __restrict__ img) {
    TVL_KERNEL_ENTRY();
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = gid & 63;
    const long frag = gid >> 6;                 // 0 .. 4 * (F / 32) * 4 - 1
    const int per_img = (F >> 5) * 4;
    if (frag >= 4L * per_img) return;
    const int which = (int)(frag / per_img), fi = (int)(frag % per_img);
    const int fb = fi >> 2, sub = fi & 3;
    const int m = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (which == 0) {            // W1[fb * 32 + m][16 sub + 8 h + j]
            v[j] = W1[(long)(fb * 32 + m) * 64 + 16 * sub + 8 * h + j] * s1;
        } else if (which == 2) {     // W2[c = 16 sub + 8 h + j][fb * 32 + m]
            v[j] = W2[(long)(16 * sub + 8 * h + j) * F + fb * 32 + m] * s2;
        } else {
            const int blk = sub >> 1, s = sub & 1;
            const int f = fb * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            v[j] = which == 1 ? W2[(long)(blk * 32 + m) * F + f] * s2      // W2[c][f]
                              : W1[(long)f * 64 + blk * 32 + m] * s1;      // W1[f][d]
        }
    }
    f16x8 o[2];
    split8h(v, o);
    unsigned char* dst = img + (long)which * F * 64 * 4 + ((long)fi * 2) * 1024 + lane * 16;
    *reinterpret_cast<f16x8*>(dst) = o[0];
    *reinterpret_cast<f16x8*>(dst + 1024) = o[1];
}

}  // namespace

extern "C" int64_t tvl_mlp64_image_bytes(int32_t F) { return (F > 0 && F % 128 == 0) ? (int64_t)4 * F * 64 * 4 : -1; }

extern "C" int tvl_mlp64_pack(const float* W1, const float* W2, int32_t F, float scale1, float scale2, void* img, tvlStream_t stream) {
    TVL_REQUIRE(W1 && W2 && img && F > 0 && F % 128 == 0, "tvl_mlp64_pack: need F %% 128 == 0 (F=%d)", F);
    TVL_REQUIRE(tvl_aligned16(img), "tvl_mlp64_pack: image must be 16-byte aligned");
    const long threads = 4L * (F / 32) * 4 * 64;
    hipLaunchKernelGGL(mlp64_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W1, W2, F, scale1, scale2,
                       reinterpret_cast<unsigned char*>(img));
    TVL_LAUNCH_CHECK("tvl_mlp64_pack");
    return 0;
}

extern "C" int tvl_mlp64_fwd(const float* x, const void* img, const float* b1, const float* b2, const float* gamma, const float* beta, float* out, float* t2,
                             float* mean, float* rstd, int64_t M, int32_t F, float inv_w1, float inv_w2, float w1_rownorm, float b1_max, float eps,
                             tvlStream_t stream) {
    TVL_REQUIRE(x && img && b1 && b2 && gamma && beta && out, "tvl_mlp64_fwd: null pointer");
    TVL_REQUIRE(M > 0 && F > 0 && F % 128 == 0, "tvl_mlp64_fwd: need M > 0, F %% 128 == 0 (M=%ld F=%d)", (long)M, F);
    TVL_REQUIRE(tvl_aligned16(x) && tvl_aligned16(img) && tvl_aligned16(b1) && tvl_aligned16(b2) && tvl_aligned16(gamma) && tvl_aligned16(beta) && tvl_aligned16(out) &&
                (!t2 || tvl_aligned16(t2)), "tvl_mlp64_fwd: operands must be 16-byte aligned");
    Mlp64P p = {};
    p.x = x; p.img = reinterpret_cast<const unsigned char*>(img); p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.beta = beta;
    p.out = out; p.t2 = t2; p.mean = mean; p.rstd = rstd; p.M = M; p.F = F; p.inv_w1 = inv_w1; p.inv_w2 = inv_w2;
    p.w1_rownorm = w1_rownorm; p.b1_max = b1_max; p.eps = eps;
    hipLaunchKernelGGL(mlp64_fwd_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    TVL_LAUNCH_CHECK("tvl_mlp64_fwd");
    return 0;
}

extern "C" int tvl_mlp64_bwd(const float* dout, const float* x, const float* t2, const float* mean, const float* rstd, const void* img, const float* b1,
                             const float* gamma, float* dx, int64_t M, int32_t F, float inv_w1, float inv_w2, float w2_colnorm, tvlStream_t stream) {
    TVL_REQUIRE(dout && x && t2 && mean && rstd && img && b1 && gamma && dx, "tvl_mlp64_bwd: null pointer");
    TVL_REQUIRE(M > 0 && F > 0 && F % 128 == 0, "tvl_mlp64_bwd: need M > 0, F %% 128 == 0 (M=%ld F=%d)", (long)M, F);
    TVL_REQUIRE(tvl_aligned16(dout) && tvl_aligned16(x) && tvl_aligned16(t2) && tvl_aligned16(img) && tvl_aligned16(b1) && tvl_aligned16(gamma) && tvl_aligned16(dx),
                "tvl_mlp64_bwd: operands must be 16-byte aligned");
    Mlp64P p = {};
    p.x = x; p.img = reinterpret_cast<const unsigned char*>(img); p.b1 = b1; p.gamma = gamma; p.dout = dout; p.t2_in = t2; p.mean_in = mean; p.rstd_in = rstd;
    p.dx = dx; p.M = M; p.F = F; p.inv_w1 = inv_w1; p.inv_w2 = inv_w2; p.w2_colnorm = w2_colnorm;
    hipLaunchKernelGGL(mlp64_bwd_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    TVL_LAUNCH_CHECK("tvl_mlp64_bwd");
    return 0;
}
