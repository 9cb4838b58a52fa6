#!/usr/bin/env python
"""HBM bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run).

    python tools/hbm_traffic.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run> [commit] > profiles/<round>_hbm_traffic.json

Output: {"commit": <the build the passes ran>, "csrc_sha": <tunevlseg_amd.hip.csrc_sha() of the sources the passes ran: bench.py reports a
record only while the sources are the same>, "kernels": {name: {...}}}, names without blanks after commas (bench.py's lookup key).

Both counters are reported in KB; FETCH_SIZE is doubled (gfx950 counts 128-byte read requests as 64 B,
MI355X_MICROARCH.md, HBM section).  Kernel names are reduced to `name<template args>` without the namespace."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:  # cut the argument list: first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip().replace(", ", ",")


def load(d: str, counter: str):
    acc = defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
        w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        out[k] = {"dispatches": max(len(fetch.get(k, [])), len(write.get(k, []))), "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                  "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024))}
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    from tunevlseg_amd.hip import csrc_sha

    json.dump({"commit": sys.argv[3] if len(sys.argv) > 3 else None, "csrc_sha": csrc_sha(), "counters": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KB, "
               "mean over dispatches; two separate rocprofv3 --pmc passes", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
