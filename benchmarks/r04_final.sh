#!/bin/bash
# round 4, final build: full GPU suite, the driver's bench line, the profile round (kernel trace + FETCH / WRITE passes), the search alone
# with its SQ counters
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_n_bench_line.json 2> gpurun_out/r04_n_bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_n_bench_line.json').read().strip().splitlines()[-1])
print('ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'verified', r.get('verified'), r.get('verified_against'), r.get('verify_counts'))
print('roofline', r['roofline']['kernel'], round(r['roofline']['frac'],3), 'search cycles', r['roofline'].get('search_cycles_per_stream_timestep'))
for k,v in r.get('workloads',{}).items(): print(k, {kk: v.get(kk) for kk in ('value','ms_per_step','verified','verified_against','hop_latency_ms','error')})
cb=r.get('cpu_baseline',{}); print('cpu', cb.get('value'), cb.get('kind'), cb.get('end_to_end',{}).get('value'))
PY
bash benchmarks/profile_round.sh r04_n > gpurun_out/r04_n_profile.log 2>&1; tail -5 gpurun_out/r04_n_profile.log | cut -c1-300
timeout 600 python benchmarks/search_micro.py --reps 3 2>&1 | grep -v "amdgpu.ids\|TensorFlow\|Coqui" | cut -c1-900 > gpurun_out/r04_n_search_micro.txt; tail -1 gpurun_out/r04_n_search_micro.txt | cut -c1-400
bash benchmarks/pmc_search.sh > gpurun_out/r04_n_pmc_search.log 2>&1; cp gpurun_out/pmc_search.csv gpurun_out/r04_n_pmc_search.csv; grep -E "INSTS_VALU|INSTS_SALU|WAVE_CYCLES|LDS_BANK|LDS_IDX|INSTS_LDS|BRANCH" gpurun_out/r04_n_pmc_search.csv | cut -c1-200
