#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for t in "search_prio=0" "search_prio=1" "search_prio=2"; do
  STT_AMD_TUNING=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_k_bench.json 2> gpurun_out/r03_k_bench.err
  python - "$t" <<'PY'
import json,sys
try:
    r=json.loads(open('gpurun_out/r03_k_bench.json').read().strip().splitlines()[-1])
    cp=r['roofline'].get('critical_path',{})
    print(sys.argv[1], '| ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'ver', r.get('verified'), '| stages', {k[:-3]: round(v,2) for k,v in r.get('stage_ms_per_step',{}).items()}, '| lstm us', round(cp.get('us_per_launch'),2), 'cyc', round(r['roofline'].get('search_cycles_per_stream_timestep')))
    print('   ', r.get('decoder_phase_cycles_per_stream_step')); st=r.get('decoder_stamp_cycles_per_stream_step'); print('   arrive', st[:16] if st else None, 'last-first', st[48:50] if st else None)
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open('gpurun_out/r03_k_bench.err').read()[-500:])
PY
done
