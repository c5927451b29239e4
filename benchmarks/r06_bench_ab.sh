#!/bin/bash
# Round 6: the headline workload (no side workloads, no CPU legs) under tunable settings, A/B on one box.
#   bash benchmarks/r06_bench_ab.sh "search_exp=0" "search_exp=13" ...     [WL=batch|peaky|...]
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
WL=${WL:-batch}
LOG=$OUT/r06_bench_ab_$WL.txt; : > $LOG
for rep in 1 2; do
for t in "$@"; do
  echo "== $WL $t (rep $rep)" >> $LOG
  STT_AMD_TUNING="$t" timeout 900 python bench.py --workload $WL --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-reference-check --no-hybrid-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d.get(k) for k in ('value','ms_per_step','verified')}, (d.get('roofline') or {}).get('avg_launch_ms'))" >> $LOG
done
done
cat $LOG
