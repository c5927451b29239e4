#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_errors.py tests/test_gpu_async.py tests/test_gpu_api.py -q 2>&1 | tail -3
timeout 900 bash benchmarks/profile_round.sh r03_g > gpurun_out/r03_g_profile.log 2>&1; tail -5 gpurun_out/r03_g_profile.log | cut -c1-300
find gpurun_out/prof_r03_g -name "*kernel_trace.csv" -size +20M -delete; find gpurun_out/prof_r03_g -name "*.db" -delete
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_g_bench.json 2> gpurun_out/r03_g_bench.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r03_g_bench.json').read().strip().splitlines()[-1])
    print('bench ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'verified', r.get('verified'), 'p50', r.get('p50_utterance_latency_ms'), 'host', r.get('host_enqueue_ms_per_step'))
    print('   stages', {k: round(v,3) for k,v in r.get('stage_ms_per_step',{}).items()})
    print('   roofline', {k: v for k,v in r['roofline'].items() if k not in ('all','critical_path')})
    for k,v in r['roofline']['all'].items(): print('     ', k[:60], {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk!='note'} if isinstance(v,dict) else v)
    print('   cp', r['roofline'].get('critical_path'))
    for k,v in (r.get('workloads') or {}).items(): print('   wl', k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','ms_per_step','verified','error','hop_latency_ms','p50_utterance_latency_ms')})
    cb=r.get('cpu_baseline',{}); print('   cpu', cb.get('value'), cb.get('cores'), cb.get('acoustic_s_per_utterance'), cb.get('decoder_s_per_utterance'), cb.get('decoder_batch',{}).get('value'), cb.get('acoustic_quiet',{}).get('s_per_utterance'))
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r03_g_bench.err').read()[-1500:])
PY
