#!/bin/bash
# round 3, call c: recurrent step with pinned accumulators -- microbenchmark + stamps, the bitwise tests, the bench line
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python benchmarks/lstm_micro.py > gpurun_out/r03_c_lstm_micro.jsonl 2> gpurun_out/r03_c_lstm_micro.err
cat gpurun_out/r03_c_lstm_micro.jsonl; grep LSTM_STAMPS gpurun_out/r03_c_lstm_micro.err
timeout 900 python -m pytest tests/test_gpu_timedpath.py tests/test_gpu_benchshape.py tests/test_gpu_async.py -q 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r03_c_bench.json').read().strip().splitlines()[-1])
    print('bench ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'verified', r.get('verified'), 'p50', r.get('p50_utterance_latency_ms'))
    print('   stages', {k: round(v,3) for k,v in r.get('stage_ms_per_step',{}).items()})
    cp=r['roofline'].get('critical_path',{}); print('   lstm us/launch', cp.get('us_per_launch'), 'rows', cp.get('rows_per_launch'))
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r03_c_bench.err').read()[-1500:])
PY
