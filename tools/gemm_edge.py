import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip
hip.load()
torch.manual_seed(0)
M = 495
for N, K in ((768, 3072), (3072, 768), (768, 768), (2304, 768), (768, 2304)):
    A, B = torch.randn(M, K), torch.randn(N, K)
    ref = A.double() @ B.double().T
    scale = A.double().abs() @ B.double().abs().T
    Ad, Bd = A.cuda(), B.cuda()
    outs = {}
    for mode in ("f32", "bf16x6"):
        hip.set_gemm_mode(mode)
        C = torch.empty(M, N, device="cuda"); hip.gemm(hip.NT, M, N, K, Ad, K, Bd, K, C, N); outs[mode + "_NT"] = C.cpu().double()
    hip.set_gemm_mode("f32")
    Bt = Bd.t().contiguous()
    C = torch.empty(M, N, device="cuda"); hip.gemm(hip.NN, M, N, K, Ad, K, Bt, N, C, N); outs["f32_NN"] = C.cpu().double()
    hip.set_gemm_mode("bf16x6")
    outs["tp3"] = hip.gemm_tp3(hip.tp3_pack(Ad), hip.tp3_pack(Bd))[0].cpu().double()
    for k, v in outs.items():
        e = ((v - ref) / scale)
        print(f"N={N} K={K} {k:10s} rms all {e.pow(2).mean().sqrt():.2e}  rows485+ {e[485:].pow(2).mean().sqrt():.2e}  max {e.abs().max():.2e}  mean {e.mean():+.2e}")
