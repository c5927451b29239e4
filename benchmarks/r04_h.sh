#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for SC in synthetic fixture; do
timeout 900 python bench.py --workload bytes --steps 8 --warmup 5 --no-extras --scorer $SC > gpurun_out/r04_h_bytes_$SC.json 2> gpurun_out/r04_h_bytes_$SC.err
tail -c 300 gpurun_out/r04_h_bytes_$SC.err
python - $SC <<'PY'
import json, sys
r=json.loads(open('gpurun_out/r04_h_bytes_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'verified', r.get('verified'), r.get('verified_against'), r.get('verify_counts'))
print(' stage', {k: round(v,3) for k,v in r['stage_ms_per_step'].items()})
print(' counters', r['decoder_counters_last_step'])
print(' phases', r['decoder_phase_cycles_per_stream_step'])
print(' ', r['verified_what'][-200:])
PY
done
