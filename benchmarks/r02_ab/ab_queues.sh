#!/bin/bash
# A/B inside one box: HW queue count x stage-event profiling (interleaved, twice)
out=gpurun_out/ab_queues.log; : > $out
run_b() { echo "== bench $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 3 $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('p50_utterance_latency_ms'), r.get('stage_ms_per_step'))" >> $out; }
for rep in 1 2; do
EXTRA=--no-profile run_b A=1
EXTRA=--no-profile run_b GPU_MAX_HW_QUEUES=8
EXTRA= run_b A=1
EXTRA= run_b GPU_MAX_HW_QUEUES=8
EXTRA="--no-profile --no-pipeline" run_b A=1
done
cat $out
