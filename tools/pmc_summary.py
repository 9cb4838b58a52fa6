#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt in name:
            acc[name.split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print(k, "dispatches", n)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}")
