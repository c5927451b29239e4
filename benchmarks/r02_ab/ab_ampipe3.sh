#!/bin/bash
out=gpurun_out/ab_ampipe3.log; : > $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), round(r.get('p50_utterance_latency_ms'),2), {k:round(v,2) for k,v in r.get('stage_ms_per_step').items()})" >> $out; }
run_b A=1
run_b STT_AMD_DENSE_LDS_KB=100
run_b STT_AMD_PCHUNK=32
run_b STT_AMD_PCHUNK=96
run_b STT_AMD_PCHUNK0=48
run_b STT_AMD_PCHUNK0=32 STT_AMD_PCHUNK=64
run_b STT_AMD_LSTM_PRIO=0
run_b STT_AMD_LSTM_UPW=8
run_b STT_AMD_PIPELINE=3 STT_AMD_DENSE_LDS_KB=100
run_b STT_AMD_PIPELINE=1
run_b STT_AMD_AM_PIPE=0
run_b A=2
cat $out
