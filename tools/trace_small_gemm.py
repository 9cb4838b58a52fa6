#!/usr/bin/env python
"""Small-M GEMMs at several K under `rocprofv3 --kernel-trace`: GPU-side duration against K gives the per-launch intercept.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <out> -- python <repo>/tools/trace_small_gemm.py
    python <repo>/tools/trace_small_gemm.py --read <out>
"""
import collections
import csv
import glob
import sys
from pathlib import Path

if "--read" in sys.argv:
    f = glob.glob(f"{sys.argv[sys.argv.index('--read') + 1]}/**/*kernel_trace.csv", recursive=True)[0]
    groups = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        groups.setdefault((r["Kernel_Name"][:80], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", "")), []).append(d)
    for (n, g, l), v in groups.items():
        v = sorted(v)
        print(f"{n:82s} grid {g:>8s} lds {l:>6s} n={len(v):3d} median {v[len(v) // 2]:7.1f} us min {v[0]:7.1f}")
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tunevlseg_amd import hip  # noqa: E402

hip.load()
SHAPES = [(384, 512, 32), (384, 512, 64), (384, 512, 512), (384, 512, 2048), (64, 64, 512), (384, 1536, 512), (384, 2048, 512), (260, 1536, 512), (260, 512, 2048)]
COLD = "--cold" in sys.argv   # evict L2 + the 256 MB memory-side cache between launches (a step touches GBs between two uses of a weight)
if "--sweep" in sys.argv:   # duration against the number of workgroups at (almost) no work per workgroup, and at K = 512
    SHAPES = [(384, n, k) for k in (64, 512) for n in (64, 512, 1536, 4096, 8192)] + [(64, 64, 64), (3072, 512, 64)]
for M, N, K in SHAPES:
    A, B, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
    big = torch.empty(160 * 1024 * 1024, device="cuda") if COLD else None   # 640 MB
    for _ in range(30):
        if COLD:
            big.fill_(1.0)
        hip.gemm(hip.NT, M, N, K, A, K, B, K, C, N)
    torch.cuda.synchronize()
x = torch.randn(1024, device="cuda")
for _ in range(30):
    y = x * 2.0
torch.cuda.synchronize()
