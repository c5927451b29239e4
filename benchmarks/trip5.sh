#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_t5; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log; tail -25 $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_batch.json 2> $OUT/bench_batch.err; echo "batch rc=$?"; tail -c 1500 $OUT/bench_batch.json; tail -3 $OUT/bench_batch.err
for W in peaky stream bytes ragged; do
  timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err; echo "$W rc=$?"
  python - $OUT/bench_$W.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value %.0f ms/step %.2f  p50 %s hop %s stages %s" % (r["value"], r["ms_per_step"], r.get("p50_utterance_latency_ms"), r.get("hop_latency_ms"), r.get("stage_ms_per_step")))
except Exception as e:
    print("  no result", e)
PY
done
bash benchmarks/profile_round.sh r02_a > $OUT/profile.log 2>&1; tail -30 $OUT/profile.log
