#!/bin/bash
# A/B inside one box: acoustic model as three engines (STT_AMD_AM_PIPE) x pipeline depth x GEMM LDS floor x chunking
out=gpurun_out/ab_ampipe.log; : > $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --no-profile --steps 30 --warmup 4 2>/dev/null | tail -1 | cut -c1-200 >> $out; }
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=2
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=2
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=4
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_DENSE_LDS_KB=0
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=32 STT_AMD_PCHUNK=32
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=24 STT_AMD_PCHUNK=24
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_PCHUNK0=64 STT_AMD_PCHUNK=64
run_b STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_LSTM_UPW=8
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=2
echo "== profiled default" >> $out
timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('p50_utterance_latency_ms'), r.get('stage_ms_per_step'))" >> $out
cat $out
