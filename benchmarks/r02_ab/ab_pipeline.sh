#!/bin/bash
# A/B inside one box: pipeline depth x chunk schedule of the pipelined batch path x dense tile rule
out=gpurun_out/ab_pipeline.log; : > $out
timeout 300 python -m pytest tests/test_gpu_async.py -x -q 2>&1 | tail -3 >> $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --no-profile --steps 30 --warmup 4 2>/dev/null | tail -1 | cut -c1-200 >> $out; }
run_b STT_AMD_PIPELINE=2 STT_AMD_DENSE_TILE=128
run_b STT_AMD_PIPELINE=3 STT_AMD_DENSE_TILE=128
run_b STT_AMD_PIPELINE=4 STT_AMD_DENSE_TILE=128
run_b STT_AMD_PIPELINE=3 STT_AMD_PCHUNK=128
run_b STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=128 STT_AMD_PCHUNK=128
run_b STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=64 STT_AMD_PCHUNK=96
run_b STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=250 STT_AMD_PCHUNK=250
run_b STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=250 STT_AMD_PCHUNK=250 STT_AMD_DENSE_TILE=256
run_b STT_AMD_PIPELINE=2 STT_AMD_PCHUNK0=128 STT_AMD_PCHUNK=128
run_b STT_AMD_PIPELINE=4 STT_AMD_PCHUNK0=128 STT_AMD_PCHUNK=128
run_b STT_AMD_PIPELINE=2 STT_AMD_DENSE_TILE=128
cat $out
