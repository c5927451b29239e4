#!/bin/bash
# The build without device calls (-DSTT_NO_DEVICE_CALLS): the fault's cuts, the parity tests that touch a scorer, the code-point step's phases and the bytes workload.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_nocalls_${TAG:-x}.txt; : > $OUT
for i in 1 2 3; do TAG=n$i FUZZ_K="bytes-True" bash benchmarks/r06_scribble_fuzz.sh bytes_lm_only_16 "STT_AMD_TUNING=debug_scribble=2,decoder_streams=16" | tail -3 >> $OUT; done
for i in 1 2; do TAG=m$i bash benchmarks/r06_scribble_fuzz.sh whole_fuzz_16 "STT_AMD_TUNING=debug_scribble=2,decoder_streams=16" | tail -3 >> $OUT; done
echo "== parity" >> $OUT
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_lm.py tests/test_gpu_errors.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
echo "== phase probe" >> $OUT
timeout 600 python benchmarks/bytes_phase_probe.py 2>&1 | grep ms_profiled | cut -c1-700 >> $OUT
echo "== bench bytes" >> $OUT
timeout 600 python bench.py --workload bytes --steps 8 --warmup 5 --no-extras --no-cpu-baseline --no-reference-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d.get(k) for k in ('value','ms_per_step','verified')})" >> $OUT
echo "== search micro" >> $OUT
timeout 300 python benchmarks/search_micro.py 2>&1 | tail -3 | cut -c1-600 >> $OUT
cat $OUT
