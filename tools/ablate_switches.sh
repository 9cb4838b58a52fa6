#!/bin/bash
# Same-box ablation of the A/B switches on one workload: the default build, then each switch flipped alone, then all of round 3's off.
#     bash tools/ablate_switches.sh [vpt|maple|cris] [steps]
WL=${1:-vpt}; ST=${2:-15}
run() { env $1 python bench.py --workload $WL --steps $ST --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-78s %8.1f img/s %7.3f ms' % ('$1', d['value'], d['ms_per_step']))"; }
run "TVL_DEFAULT=1"
run "TVL_GEMM_M16=0"
run "TVL_MLP64=0"
run "TVL_TEXT_STREAM=0"
run "TVL_GEMM_F32_DIRECT=0"
run "TVL_GEMM_SPLITK_MIN_K=1024"
run "TVL_GEMM_M16=0 TVL_MLP64=0 TVL_TEXT_STREAM=0 TVL_GEMM_F32_DIRECT=0 TVL_GEMM_SPLITK_MIN_K=1024"
run "TVL_GEMM_H2=0 TVL_ATTN_H2=0 TVL_DQKV_H2=0"
run "TVL_DEFAULT=1"
