#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_d_bench.json 2> gpurun_out/r04_d_bench.err
tail -c 400 gpurun_out/r04_d_bench.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r04_d_bench.json').read().strip().splitlines()[-1])
    print('ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'verified', r.get('verified'), r.get('verified_against'), r.get('verify_counts'))
    print(json.dumps(r.get('verify_mismatches'))[:1500])
    for k,v in r.get('workloads',{}).items(): print(k, {kk: v.get(kk) for kk in ('value','ms_per_step','verified','verified_against','hop_latency_ms','error')}, v.get('verified_what'))
    cb=r.get('cpu_baseline',{}); print('cpu', cb.get('value'), cb.get('kind'), cb.get('end_to_end',{}).get('value'))
except Exception as e:
    print('BENCH FAILED', e)
PY
timeout 1500 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -q -x 2>&1 | tail -15
