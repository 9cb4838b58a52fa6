#!/usr/bin/env python
"""Per-kernel mean counter values over one or more `rocprofv3 --pmc ... --kernel-trace --output-format csv` directories.

    python tools/pmc_table.py <dir> [<dir> ...] [filter=<substring>] [--json out.json]
Counter semantics on gfx950 (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import csv
import glob
import json
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith(("filter=", "--")) and not a.endswith(".json")]
flt = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("filter=")), "")
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            name = name[5:] if name.startswith("void ") else name
            name = name.rsplit("(", 1)[0] if name.endswith(")") else name
            if flt not in name:
                continue
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (r["Dispatch_Id"], f) not in seen:
                seen.add((r["Dispatch_Id"], f))
                acc[name]["_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {}
for name, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1]["_us"])):
    row = {c: sum(v) / len(v) for c, v in cs.items()}
    row["dispatches"] = len(cs["_us"])
    out[name] = row
    print(f"{name[-100:]}  n={row['dispatches']}  avg {row['_us']:.1f} us")
    wc = row.get("SQ_WAVE_CYCLES")
    for c, v in sorted(row.items()):
        if c in ("_us", "dispatches"):
            continue
        extra = f"  ({100 * v / wc:5.1f} % of wave cycles)" if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
        print(f"     {c:34s} {v:18.1f}{extra}")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
