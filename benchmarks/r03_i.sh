#!/bin/bash
# tunables sweep on the round-3 build (one box): each line = python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for t in "pair=1" "active=1" "pchunk0=48" "pchunk0=32,pchunk=64" "pchunk=96" "lstm_prio=0" "lm_waves=1" "lm_waves=4" "pair=1,pipeline=3,active=2" "pair=1"; do
  STT_AMD_TUNING=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_i_bench.json 2> gpurun_out/r03_i_bench.err
  python - "$t" <<'PY'
import json,sys
try:
    r=json.loads(open('gpurun_out/r03_i_bench.json').read().strip().splitlines()[-1])
    cp=r['roofline'].get('critical_path',{})
    print(sys.argv[1], '| ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'ver', r.get('verified'), 'p50', round(r['p50_utterance_latency_ms'],1), '| stages', {k[:-3]: round(v,2) for k,v in r.get('stage_ms_per_step',{}).items()}, '| lstm us', round(cp.get('us_per_launch'),2), 'cyc', round(r['roofline'].get('search_cycles_per_stream_timestep')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open('gpurun_out/r03_i_bench.err').read()[-500:])
PY
done
