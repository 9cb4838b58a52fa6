import sys, torch
sys.path.insert(0, ".")
from tunevlseg_amd import hip
hip.load()
shapes = [(384, 1536, 512), (384, 512, 512), (384, 2048, 512), (384, 512, 2048), (256, 1536, 512), (15840, 64, 768), (346112, 64, 576), (21632, 64, 512)]
for M, N, K in shapes:
    A, B, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
    for _ in range(5): hip.gemm(hip.NT, M, N, K, A, K, B, K, C, N)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.gemm(hip.NT, M, N, K, A, K, B, K, C, N)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = sorted(ts)[3]
    print(f"M={M} N={N} K={K}: {t:7.1f} us {2.0*M*N*K/t/1e6:6.1f} TF/s {hip.gemm_kernel_key(hip.NT, M, N, True, 3, K)[17:]}")
