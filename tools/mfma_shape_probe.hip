// Probe: which fp16 MFMA shape sustains more FLOP/s in the ring GEMM's inner-loop regime on a power-limited MI355X?
//
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe.bin && tools/mfma_shape_probe.bin
//
// Both kernels: 512 threads = 8 waves (two per SIMD), wave tile 128 x 64, operands as two fp16 pieces, three piece products per
// k-step, every fragment re-read from LDS by ds_read_b128 (conflict-free lane-linear images), one raw s_barrier per 16-deep
// k-block, random operands (zero operands would let the chip clock higher: guide rule 25), no global traffic inside the loop.
//   A: v_mfma_f32_32x32x16_f16, 24 MFMAs + 12 reads per 16-deep block (what gemm_tp3_kernel<256, 256, ..., NP = 2> issues)
//   B: v_mfma_f32_16x16x32_f16, 96 MFMAs + 24 reads per 32-deep block (same cycles at the instruction rates of the guide)
// Reported: wall TFLOP/s (algorithmic fp32 FLOPs = MFMA FLOPs / 3) and the in-loop shader clock (s_memtime / s_memrealtime).
// MI355X_MICROARCH.md "DVFS give-back" item 7 measured 1.12-1.15x for B on bare loops; this checks it in OUR loop shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OFF>
__device__ __forceinline__ f16x8 lds_frag(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

constexpr int STAGE = 32 * 1024;   // one 16-deep slab of a 256 x 256 tile: (8 + 8) row blocks x 2 pieces x 1 KiB
constexpr int NSTAGE = 3;

__global__ __launch_bounds__(512) void probe32(const unsigned char* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    for (int i = threadIdx.x; i < NSTAGE * STAGE / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned a_frag = lds0 + lane * 16 + wm * (4 * 2048), b_frag = lds0 + lane * 16 + 16 * 1024 + wn * (2 * 2048);
    f16x8 a[4][2], b[2][2];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    int st = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned aa = a_frag + st * STAGE, bb = b_frag + st * STAGE;
        a[0][0] = lds_frag<0>(aa); a[0][1] = lds_frag<1024>(aa); a[1][0] = lds_frag<2048>(aa); a[1][1] = lds_frag<3072>(aa);
        a[2][0] = lds_frag<4096>(aa); a[2][1] = lds_frag<5120>(aa); a[3][0] = lds_frag<6144>(aa); a[3][1] = lds_frag<7168>(aa);
        b[0][0] = lds_frag<0>(bb); b[0][1] = lds_frag<1024>(bb); b[1][0] = lds_frag<2048>(bb); b[1][1] = lds_frag<3072>(bb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][1], a[i][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][0], a[i][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][0], a[i][0], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_barrier();
        st = st == NSTAGE - 1 ? 0 : st + 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = c1 - c0; }
}

// 16x16x32: lane l holds A[row l & 15][k = 8 (l >> 4) + j], j = 0..7, of a 16-row x 32-k operand.  In the (32-row x 16-k block, piece)
// image a 16 x 32 operand is rows 16 m .. 16 m + 15 of the two consecutive k blocks kb, kb + 1: lane (r = l & 15, q = l >> 4) reads
// 16 bytes at block(kb + (q >> 1)) + (((q & 1) * 32) + 16 m + r) * 16 -- lane-distinct 16-byte slots, conflict-free per 16-lane group.
__global__ __launch_bounds__(512) void probe16(const unsigned char* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    for (int i = threadIdx.x; i < NSTAGE * STAGE / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // here one "stage pair" = two consecutive 16-deep slabs (kb, kb + 1) = stages st, st + 1 of the same LDS image
    const int r = lane & 15, q = lane >> 4;
    const unsigned lane_off = (unsigned)((q >> 1) * STAGE + (((q & 1) * 32) + r) * 16);
    const unsigned a_frag = lds0 + lane_off + wm * (4 * 2048), b_frag = lds0 + lane_off + 16 * 1024 + wn * (2 * 2048);
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
        f16x8 a[8][2], b[4][2];
        // A 16-row operand m of 32-row block i: block offset i * 2048, piece p * 1024, row half (m & 1) * 256
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            a[m][0] = lds_frag<0>(a_frag + (m >> 1) * 2048 + (m & 1) * 256);
            a[m][1] = lds_frag<1024>(a_frag + (m >> 1) * 2048 + (m & 1) * 256);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            b[n][0] = lds_frag<0>(b_frag + (n >> 1) * 2048 + (n & 1) * 256);
            b[n][1] = lds_frag<1024>(b_frag + (n >> 1) * 2048 + (n & 1) * 256);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][1], a[i][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], a[i][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], a[i][0], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) s += acc[i][j][k];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = c1 - c0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4096, nwg = argc > 2 ? atoi(argv[2]) : 256;
    const size_t bytes = (size_t)NSTAGE * STAGE;
    std::vector<_Float16> h(bytes / 2);
    srand(1);
    for (auto& v : h) v = (_Float16)(2.0f * rand() / RAND_MAX - 1.0f);
    unsigned char* d_src; float* d_out; unsigned long long* d_st;
    CK(hipMalloc(&d_src, bytes)); CK(hipMalloc(&d_out, (size_t)nwg * 512 * 4)); CK(hipMalloc(&d_st, (size_t)nwg * 16));
    CK(hipMemcpy(d_src, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = (double)nwg * iters * 2.0 * 256 * 256 * 16;   // algorithmic fp32 FLOPs (3 MFMA products each)
    std::vector<unsigned long long> st(nwg * 2);
    for (int round = 0; round < 4; ++round) {
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < (round ? 1 : 3); ++rep) {   // first round: warm the clock governor
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(probe32, dim3(nwg), dim3(512), bytes, 0, d_src, d_out, d_st, iters);
                else hipLaunchKernelGGL(probe16, dim3(nwg), dim3(512), bytes, 0, d_src, d_out, d_st, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
            double clk = 0; for (int i = 0; i < nwg; ++i) clk += (double)st[2 * i + 1] / (double)st[2 * i] * 0.1; clk /= nwg;
            double cyc = 0; for (int i = 0; i < nwg; ++i) cyc += (double)st[2 * i + 1] / iters; cyc /= nwg;
            if (round) printf("round %d %s: %8.3f ms  %7.1f TFLOP/s (fp32-equivalent; x3 = MFMA rate)  in-loop clock %.3f GHz  cycles per 16-deep block %.0f (MFMA issue floor 1536)\n",
                              round, which == 0 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9, clk, cyc);
        }
    }
    return 0;
}
