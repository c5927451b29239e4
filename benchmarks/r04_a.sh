#!/bin/bash
# round 4, first call: the new bench (a different batch every step, reference-checked), the N = 2 entry point, smoke
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err
tail -c 600 gpurun_out/r04_a_bench.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r04_a_bench.json').read().strip().splitlines()[-1])
    print('ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'verified', r.get('verified'), r.get('verified_against'))
    print(r.get('verified_what'))
    print('roofline', r['roofline']['kernel'], round(r['roofline']['frac'],3), 'search cycles', r['roofline'].get('search_cycles_per_stream_timestep'))
    for k,v in r.get('workloads',{}).items(): print(k, {kk: v.get(kk) for kk in ('value','ms_per_step','verified','verified_against','hop_latency_ms','error')})
    cb=r.get('cpu_baseline',{}); print('cpu', cb.get('value'), cb.get('kind'), cb.get('end_to_end',{}).get('value'))
except Exception as e:
    print('BENCH FAILED', e)
PY
timeout 1500 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_async.py -m gpu -q -x 2>&1 | tail -5
