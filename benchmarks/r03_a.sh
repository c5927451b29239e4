#!/bin/bash
# round 3, first GPU call: the whole GPU suite (new: tests/test_gpu_timedpath.py), then the bench line with two batches per recurrent
# step (default) against one (pair=0) and against gated searches (active=1)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -40 > gpurun_out/r03_a_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err
STT_AMD_TUNING=pair=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_a_bench_pair0.json 2> gpurun_out/r03_a_bench_pair0.err
STT_AMD_TUNING=active=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_a_bench_active1.json 2> gpurun_out/r03_a_bench_active1.err
STT_AMD_TUNING=pipeline=3 timeout 300 python bench.py --steps 30 --warmup 6 --no-extras --no-cpu-baseline > gpurun_out/r03_a_bench_pipe3.json 2> gpurun_out/r03_a_bench_pipe3.err
cat gpurun_out/r03_a_tests.log | tail -15
for f in bench bench_pair0 bench_active1 bench_pipe3; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    r=json.loads(open('gpurun_out/r03_a_%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'ms/step', round(r['ms_per_step'],3), 'RTF', round(r['value']), 'verified', r.get('verified'), 'p50', r.get('p50_utterance_latency_ms'), 'host', r.get('host_enqueue_ms_per_step'))
    print('   stages', {k: round(v,3) for k,v in r.get('stage_ms_per_step',{}).items()})
    cp=r['roofline'].get('critical_path',{}); print('   lstm us/launch', cp.get('us_per_launch'), 'rows', cp.get('rows_per_launch'), 'frac', cp.get('frac'), 'cyc', r['roofline'].get('search_cycles_per_stream_timestep'))
    for k,v in (r.get('workloads') or {}).items(): print('   wl', k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','ms_per_step','verified','error','hop_latency_ms')})
    if 'cpu_baseline' in r: print('   cpu', {k: v for k,v in r['cpu_baseline'].items() if k in ('value','cores','acoustic_s_per_utterance','decoder_s_per_utterance')}, r['cpu_baseline'].get('decoder_batch',{}).get('value'), r['cpu_baseline'].get('acoustic_quiet',{}).get('s_per_utterance'))
except Exception as e:
    print(f,'FAILED',e); print(open('gpurun_out/r03_a_%s.err'%f).read()[-1500:])
PY
done
