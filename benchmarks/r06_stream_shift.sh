#!/bin/bash
# does the streaming workload (two cohorts, two model replicas) care which pipes its replicas' streams land on?
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/r06_stream_shift.txt; : > $OUT
for K in 0 1 2 3 5; do
  timeout 600 python bench.py --workload stream --steps 3 --warmup 1 --utterances 1000 --no-cpu-baseline --no-reference-check --idle-streams $K > gpurun_out/ss.json 2> gpurun_out/ss.err
  python - "$K" >> $OUT <<'PY'
import json, sys
try:
    r = json.loads(open('gpurun_out/ss.json').read().strip().splitlines()[-1])
    print("idle streams %s  value %.0f  ms/step %.1f  verified %s" % (sys.argv[1], r['value'], r['ms_per_step'], r['verified']))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e), open('gpurun_out/ss.err').read()[-300:])
PY
done
cat $OUT
