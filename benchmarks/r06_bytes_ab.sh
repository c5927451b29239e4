#!/bin/bash
# Round 6, code-point step A/B: parity tests that use a code-point scorer, the phase probe and the bytes workload under tunable settings.
#   bash benchmarks/r06_bytes_ab.sh "cp_blocks=0" "cp_blocks=1" ...
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_bytes_ab.txt; : > $LOG
for t in "$@"; do
  echo "== $t parity" >> $LOG
  STT_AMD_TUNING="$t" timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_lm.py -m gpu -x -q 2>&1 | tail -2 >> $LOG
done
for t in "$@"; do
  echo "== $t phase probe" >> $LOG
  timeout 900 python benchmarks/bytes_phase_probe.py --set $t 2>&1 | grep ms_profiled | cut -c1-700 >> $LOG
done
for t in "$@"; do
  echo "== $t bench bytes" >> $LOG
  STT_AMD_TUNING="$t" timeout 900 python bench.py --workload bytes --steps 8 --warmup 5 --no-extras --no-cpu-baseline --no-reference-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d.get(k) for k in ('value','ms_per_step','verified')})" >> $LOG
done
cat $LOG
