#!/bin/bash
# One GPU trip: tests + bench variants; everything lands under gpurun_out/<tag>/.  usage: bash benchmarks/trip.sh <tag> [variants...]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-trip}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
  tail -40 $OUT/tests.log
fi
run() {  # name, env...
  local name=$(echo "$1" | tr -c 'A-Za-z0-9_\n' '_'); shift
  env "$@" timeout 400 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?"
  python - "$OUT/bench_$name.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms/step %.3f  value %.0f  stages %s" % (r["ms_per_step"], r["value"], {k: round(v, 3) for k, v in r["stage_ms_per_step"].items()}))
    print("  phases/stream-step %s" % r.get("decoder_phase_cycles_per_stream_step"))
    print("  counters %s" % r.get("decoder_counters_last_step"))
    st = r.get("decoder_stamp_cycles_per_stream_step") or []
    if any(st):
        print("  stamps[0:16]  %s" % st[:16])
        print("  stamps[16:32] %s" % st[16:32])
        print("  stamps[32:48] %s" % st[32:48])
        print("  stamps[48:64] %s" % st[48:64])
except Exception as e:
    print("  (no result: %s)" % e)
PY
}
for v in "$@"; do
  case $v in
    fast) run fast A=1 ;;
    slow) run slow STT_AMD_FAST=0 ;;
    lm1) run lm1 STT_AMD_LM_WAVES=1 ;;
    lm2) run lm2 STT_AMD_LM_WAVES=2 ;;
    lm4) run lm4 STT_AMD_LM_WAVES=4 ;;
    mband) run mband STT_AMD_DENSE_MBAND=1 ;;
    *) run "$v" $v ;;   # "ENV=1 ENV2=2" -> name sanitised
  esac
done
