#!/bin/bash
# A build of ctc.hip with another SGPR-spill code path: the fault's cuts, the parity tests that touch a scorer, the perf of the two search steps.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_spillfix_${TAG:-x}.txt; : > $OUT
for i in 1 2 3; do TAG=s$i FUZZ_K="bytes-True" bash benchmarks/r06_scribble_fuzz.sh B16 "STT_AMD_TUNING=debug_scribble=2,decoder_streams=16" | grep -E "===|rc=|passed|failed|fault" >> $OUT; done
TAG=s4 FUZZ_K="bytes-True" bash benchmarks/r06_scribble_fuzz.sh zero512_default "STT_AMD_TUNING=debug_scribble=129" pattern512_default "STT_AMD_TUNING=debug_scribble=1" | grep -E "===|rc=|passed|failed|fault" >> $OUT
TAG=s5 bash benchmarks/r06_scribble_fuzz.sh whole_fuzz_16 "STT_AMD_TUNING=debug_scribble=2,decoder_streams=16" whole_fuzz_pattern512 "STT_AMD_TUNING=debug_scribble=1" | grep -E "===|rc=|passed|failed|fault" >> $OUT
echo "== parity" >> $OUT
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_lm.py tests/test_gpu_errors.py -m gpu -x -q 2>&1 | tail -2 >> $OUT
echo "== bench bytes" >> $OUT
timeout 600 python bench.py --workload bytes --steps 8 --warmup 5 --no-extras --no-cpu-baseline --no-reference-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d.get(k) for k in ('value','ms_per_step','verified')})" >> $OUT
echo "== search micro" >> $OUT
timeout 300 python benchmarks/search_micro.py 2>&1 | tail -1 | cut -c1-130 >> $OUT
cat $OUT
