#!/usr/bin/env python
"""Print a rocprofv3 --kernel-trace --stats summary (kernel_stats.csv): share, calls, average duration per kernel."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms/step calls {int(r['Calls']) / steps:6.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
print(f"kernel time per step {tot / 1e6 / steps:.2f} ms")
