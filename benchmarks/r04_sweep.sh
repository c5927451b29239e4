#!/bin/bash
# tunables around the defaults on the final build, one box: ms per batch of the headline workload (no extras, no CPU baseline, blocking verification only)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/r04_sweep.txt; : > $OUT
for T in "" "pchunk=32" "pchunk=64" "active=2" "lm_waves=4" "lm_waves=1" "pchunk0=32" "search_lds_kb=144" "lstm_prio=0"; do
  STT_AMD_TUNING="$T" timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-reference-check > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - "$T" >> $OUT <<'PY'
import json, sys
try:
    r=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
    print("%-20s ms/step %.3f  value %d  verified %s  lstm %.3f search %.3f dense_in %.3f" % (sys.argv[1] or "defaults", r['ms_per_step'], r['value'], r['verified'], r['stage_ms_per_step']['lstm_ms'], r['stage_ms_per_step']['decoder_next_ms'], r['stage_ms_per_step']['dense_in_ms']))
except Exception as e:
    print("%-20s FAILED %s" % (sys.argv[1], e))
PY
done
cat $OUT
