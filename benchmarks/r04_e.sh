#!/bin/bash
# search kernel after the register diet: decoder parity tests, the search alone, the bench
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fuzz.py tests/test_gpu_lm.py tests/test_gpu_wide.py -m gpu -q -x 2>&1 | tail -5
timeout 600 python benchmarks/search_micro.py --reps 3 2>&1 | grep -v "amdgpu.ids\|TensorFlow\|Coqui" | cut -c1-900 | tee gpurun_out/r04_e_search_micro.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04_e_bench.json 2> gpurun_out/r04_e_bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_e_bench.json').read().strip().splitlines()[-1])
print('ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'verified', r.get('verified'), r.get('verify_counts'))
print('search cycles', r['roofline'].get('search_cycles_per_stream_timestep'), r['stage_ms_per_step'])
PY
