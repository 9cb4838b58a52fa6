#!/bin/bash
# One measurement set on the MI355X box (run through gpurun from the repo root):
#     bash tools/profile_set.sh <tag> [workload]
# writes gpurun_out/<tag>/: bench.json (un-profiled bench line incl. cpu_baseline for vpt), kernel_trace/ (+ bench_under_rocprof.json),
# pmc_fetch/, pmc_write/ (one counter per pass, never together with a trace).  Summaries are copied into profiles/ afterwards with
# tools/kstats.py / tools/hbm_traffic.py (profiles/README.md).
set -eo pipefail
TAG=${1:?tag}; WL=${2:-vpt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
EXTRA=""
[ "$WL" != vpt ] && EXTRA="--workload $WL"
timeout -k 10 900 python bench.py --steps 10 --warmup 3 $EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done: $(head -c 200 "$OUT/bench.json")"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline $EXTRA \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/kernel_trace.err"
echo "kernel trace done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline $EXTRA \
    > "$OUT/bench_under_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "pmc FETCH_SIZE done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline $EXTRA \
    > "$OUT/bench_under_pmc_write.json" 2> "$OUT/pmc_write.err"
echo "pmc WRITE_SIZE done"
# matrix-pipe occupancy and clock: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA is executing, per SIMD summed) and GRBM_GUI_ACTIVE (busy cycles
# summed over the 8 XCDs: / 8 / kernel wall time = effective clock, MI355X_MICROARCH.md "DVFS give-back"); SQ_BUSY_CYCLES for the ratio
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline $EXTRA \
    > "$OUT/bench_under_pmc_mfma.json" 2> "$OUT/pmc_mfma.err" || echo "pmc MFMA pass failed (see pmc_mfma.err)"
echo "pmc MFMA_BUSY done"
# keep what travels back small: the per-dispatch CSVs are what the tools read
find "$OUT" -name "*.csv" -size +20M -delete || true
du -sh "$OUT"
