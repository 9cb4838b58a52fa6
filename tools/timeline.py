#!/usr/bin/env python
"""Timeline of one train step inside a rocprofv3 --kernel-trace CSV (steps are delimited by the AdamW launch): per queue busy time,
union busy time, idle time, and one line per kernel (start offset us, duration us, queue, name).

    python tools/timeline.py gpurun_out/<tag>/kernel_trace [out.txt] [--back N]

--back N: the step N before the last one (default 0 = the last).  bench.py ends with two event-profiled steps that run on ONE stream
(text tower on the main stream): --back 2 is the last TIMED step, with the side stream on.
"""
import csv
import glob
import re
import sys

back = 0
if "--back" in sys.argv:
    k = sys.argv.index("--back")
    back = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-2 - back] + 1, idx[-1 - back] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
qs = {}
for r in step:
    qs.setdefault(r["Queue_Id"], []).append(r)
for q, v in qs.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
    print(f"queue {q}: {len(v)} kernels, busy {busy / 1e6:.3f} ms")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cs, ce = iv[0]
tot = 0
for s, e in iv[1:]:
    if s > ce:
        tot += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
tot += ce - cs
print(f"union busy {tot / 1e6:.3f} ms, idle {(t1 - t0 - tot) / 1e6:.3f} ms")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\((?![^<]*>).*", "", n)[:90]


out = [f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])}"
       for r in step]
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
else:
    print("\n".join(out))
