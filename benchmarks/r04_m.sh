#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for LG in 18 22 24; do
STT_AMD_TUNING="lm_memo=$LG" timeout 900 python bench.py --workload bytes --steps 8 --warmup 5 --no-extras --no-reference-check > gpurun_out/r04_m_$LG.json 2> gpurun_out/r04_m_$LG.err
python - $LG <<'PY'
import json, sys
r=json.loads(open('gpurun_out/r04_m_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
c=r['decoder_counters_last_step']
print('memo 2^%s' % sys.argv[1], 'ms/step', round(r['ms_per_step'],2), 'probes/query', round(c['lm_probes']/max(1,c['lm_queries']),2), 'lm cycles', round(r['decoder_phase_cycles_per_stream_step']['lm']))
PY
done
