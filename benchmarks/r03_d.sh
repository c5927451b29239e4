#!/bin/bash
# round 3, call d: fast activations + pinned 64-row probes (micro), GEMM 128x256 co-tenant form A/B, kernel tests
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python benchmarks/lstm_micro.py > gpurun_out/r03_d_lstm_micro.jsonl 2> gpurun_out/r03_d_lstm_micro.err
cat gpurun_out/r03_d_lstm_micro.jsonl | tr '\n' ' '; echo; grep LSTM_STAMPS gpurun_out/r03_d_lstm_micro.err | awk 'NR%3==1'
timeout 900 python -m pytest tests/test_gpu_timedpath.py tests/test_gpu_benchshape.py tests/test_gpu_kernels.py -q 2>&1 | tail -5
for t in "dense_solo=2" "dense_solo=3" "dense_solo=3,pair=0" "dense_solo=3,active=1" "dense_solo=3,pchunk=32" "dense_solo=3,pchunk=64"; do
  STT_AMD_TUNING=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.err
  python - "$t" <<'PY'
import json,sys
try:
    r=json.loads(open('gpurun_out/r03_d_bench.json').read().strip().splitlines()[-1])
    cp=r['roofline'].get('critical_path',{})
    print(sys.argv[1], '| ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'ver', r.get('verified'), 'p50', round(r.get('p50_utterance_latency_ms'),2), '| stages', {k[:-3]: round(v,2) for k,v in r.get('stage_ms_per_step',{}).items()}, '| lstm us', round(cp.get('us_per_launch'),2), 'rows', cp.get('rows_per_launch'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open('gpurun_out/r03_d_bench.err').read()[-800:])
PY
done
