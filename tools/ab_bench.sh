# usage: bash tools/ab_bench.sh <workload> "<ENV_A>" "<ENV_B>" [steps]
WL=$1; A=$2; B=$3; ST=${4:-15}
for i in 1 2; do
  for e in "$A" "$B"; do
    env $e python bench.py --workload $WL --steps $ST --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL [$e]:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
