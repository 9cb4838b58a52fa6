#!/usr/bin/env python
"""Per kernel, from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace` pass
(tools/profile_set.sh, directory pmc_mfma): average duration, effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration
(MI355X_MICROARCH.md "DVFS give-back"), and matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock cycles of the
dispatch).  SQ_VALU_MFMA_BUSY_CYCLES counts, summed over all SIMDs, the cycles in which an MFMA is executing (32 per
v_mfma_f32_32x32x16_*): 100 % = every SIMD's matrix pipe busy for the whole dispatch.

    python tools/mfma_clock.py gpurun_out/<tag>/pmc_mfma [name filter] [--json out.json]
"""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if name.startswith("void "):
            name = name[5:]
        name = name.rsplit("(", 1)[0] if name.endswith(")") else name
        if flt in name:
            k = acc[name]
            k[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                k["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for name, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c or not c["ns"]:
        continue
    n = len(c["ns"])
    ns = sum(c["ns"]) / n
    gui = sum(c["GRBM_GUI_ACTIVE"]) / n
    mfma = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])) / max(len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])), 1)
    cu = sum(c.get("SQ_BUSY_CU_CYCLES", [0.0])) / max(len(c.get("SQ_BUSY_CU_CYCLES", [0.0])), 1)
    clk_ghz = gui / 8.0 / ns
    cycles = gui / 8.0
    rows.append({"kernel": name[-90:], "dispatches": n, "avg_us": round(ns / 1e3, 1), "total_ms": round(ns * n / 1e6, 3), "clock_GHz": round(clk_ghz, 3),
                 "mfma_busy_frac": round(mfma / (1024.0 * cycles), 4) if cycles else None, "mfma_busy_cycles": mfma, "sq_busy_cu_cycles": cu})
rows.sort(key=lambda r: -r["total_ms"])
for r in rows[:40]:
    print(f"{r['total_ms']:9.3f} ms  n={r['dispatches']:4d}  {r['avg_us']:8.1f} us  clk {r['clock_GHz']:.3f} GHz  MFMA busy {100 * (r['mfma_busy_frac'] or 0):5.1f} %  {r['kernel']}")
if "--json" in sys.argv:
    json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
