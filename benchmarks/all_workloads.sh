#!/bin/bash
# every bench workload once (JSON lines under gpurun_out/<tag>_<workload>.json) + the GPU test suite
TAG=${1:-r02_f}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${TAG}_tests.log
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_batch.json 2> gpurun_out/${TAG}_batch.err
for w in peaky bytes ragged stream; do
  extra=""; [ $w = stream ] && extra="--utterances 128 --steps 2 --warmup 1"; [ $w = ragged ] && extra="--steps 2 --warmup 1"; [ $w = bytes ] && extra="--steps 6 --warmup 2"; [ $w = peaky ] && extra="--steps 10 --warmup 2"
  timeout 400 python bench.py --workload $w --no-cpu-baseline $extra > gpurun_out/${TAG}_$w.json 2> gpurun_out/${TAG}_$w.err
done
cat gpurun_out/${TAG}_tests.log
for w in batch peaky bytes ragged stream; do python - $TAG $w <<'PY'
import json,sys
tag,w=sys.argv[1:3]
try:
    r=json.loads(open('gpurun_out/%s_%s.json'%(tag,w)).read().strip().splitlines()[-1])
    print(w, round(r['ms_per_step'],3), round(r['value']), r.get('p50_utterance_latency_ms'), r.get('hop_latency_ms'), r.get('stage_ms_per_step'), (r.get('cpu_baseline') or {}).get('value'))
except Exception as e: print(w,'FAILED',e)
PY
done
