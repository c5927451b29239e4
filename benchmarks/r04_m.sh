#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_decoder.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for SC in synthetic fixture; do
timeout 900 python bench.py --workload bytes --steps 8 --warmup 5 --no-extras --scorer $SC > gpurun_out/r04_m_$SC.json 2> gpurun_out/r04_m_$SC.err
python - $SC <<'PY'
import json, sys
r=json.loads(open('gpurun_out/r04_m_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
c=r['decoder_counters_last_step']
print(sys.argv[1], 'ms/step', round(r['ms_per_step'],2), 'verified', r['verified'], r['verified_against'], 'probes/query', round(c['lm_probes']/max(1,c['lm_queries']),2), 'phases', {k: round(v) for k,v in r['decoder_phase_cycles_per_stream_step'].items()})
PY
done
