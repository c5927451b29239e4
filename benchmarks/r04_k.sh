#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --workload stream --steps 3 --warmup 1 --no-extras > gpurun_out/r04_k_stream.json 2> gpurun_out/r04_k_stream.err
tail -c 400 gpurun_out/r04_k_stream.err | grep -v "amdgpu.ids\|TensorFlow\|Coqui"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_k_stream.json').read().strip().splitlines()[-1])
print('stream value', round(r['value']), 'ms/step', round(r['ms_per_step'],1), 'verified', r.get('verified'), r['hop_latency_ms'])
PY
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_k_bench.json 2> gpurun_out/r04_k_bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_k_bench.json').read().strip().splitlines()[-1])
print('ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'verified', r.get('verified'), r.get('verified_against'), r.get('verify_counts'))
for k,v in r.get('workloads',{}).items(): print(k, {kk: v.get(kk) for kk in ('value','ms_per_step','verified','verified_against','hop_latency_ms','error')})
cb=r.get('cpu_baseline',{}); print('cpu', cb.get('value'), cb.get('kind'), cb.get('end_to_end',{}).get('value'))
PY
