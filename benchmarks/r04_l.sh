#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for Q in 8 16 24; do
GPU_MAX_HW_QUEUES=$Q timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-check > gpurun_out/r04_l_q$Q.json 2> gpurun_out/r04_l_q$Q.err
python - $Q <<'PY'
import json, sys
r=json.loads(open('gpurun_out/r04_l_q%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
w=r['workloads']
print('Q', sys.argv[1], 'batch ms/step', round(r['ms_per_step'],3), '| stream', round(w['stream']['value']), {k: round(v,2) for k,v in w['stream']['hop_latency_ms'].items() if k in ('p50','p95','p99','max')}, '| ragged', round(w['ragged']['value']), '| bytes ms', round(w['bytes']['ms_per_step'],1))
PY
done
