#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python benchmarks/lstm_cotenant.py 2>&1 | grep -v "^TensorFlow\| Coqui\|^RUN" | grep "rows 128\|\"rows\": 128" | head -24
timeout 600 python -m pytest tests/test_gpu_timedpath.py -q 2>&1 | tail -3
for t in "dense_solo=3"; do
  STT_AMD_TUNING=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err
  python - "$t" <<'PY'
import json,sys
try:
    r=json.loads(open('gpurun_out/r03_e_bench.json').read().strip().splitlines()[-1])
    cp=r['roofline'].get('critical_path',{})
    print(sys.argv[1], '| ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'ver', r.get('verified'), 'p50', round(r.get('p50_utterance_latency_ms'),2), '| stages', {k[:-3]: round(v,2) for k,v in r.get('stage_ms_per_step',{}).items()}, '| lstm us', round(cp.get('us_per_launch'),2), 'rows', cp.get('rows_per_launch'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open('gpurun_out/r03_e_bench.err').read()[-800:])
PY
done
