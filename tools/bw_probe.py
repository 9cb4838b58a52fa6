import sys, torch
sys.path.insert(0, ".")
from tunevlseg_amd import hip
hip.load()
for mb in (48.7, 146, 194, 780):
    n = int(mb * 1e6 / 4)
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    def t(fn, reps=10):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tf = t(lambda: hip.fill(x, 1.0)); tz = t(lambda: x.zero_()); tc = t(lambda: y.copy_(x))
    print(f"{mb:6.1f} MB: tvl_fill {tf:7.1f} us = {mb/tf*1e6/1e6:.2f} TB/s | torch zero_ {tz:7.1f} us = {mb/tz:.2f} TB/s | copy {tc:7.1f} us = {2*mb/tc:.2f} TB/s (r+w)")
