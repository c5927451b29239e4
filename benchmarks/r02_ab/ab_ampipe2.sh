#!/bin/bash
out=gpurun_out/ab_ampipe2.log; : > $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('p50_utterance_latency_ms'), r.get('stage_ms_per_step'))" >> $out; }
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=2
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=2 STT_AMD_LSTM_PRIO=0
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=2 STT_AMD_LSTM_PRIO=1
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_LSTM_PRIO=1
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_LSTM_PRIO=1 STT_AMD_DENSE_LDS_KB=0
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_LSTM_PRIO=1 STT_AMD_LSTM_UPW=8
cat $out
