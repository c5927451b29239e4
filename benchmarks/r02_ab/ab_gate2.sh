#!/bin/bash
out=gpurun_out/ab_gate2.log; : > $out
timeout 300 python -m pytest tests/test_gpu_async.py -x -q 2>&1 | tail -2 >> $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), round(r.get('p50_utterance_latency_ms'),2), round(r.get('host_enqueue_ms_per_step'),2), {k:round(v,2) for k,v in r.get('stage_ms_per_step').items()})" >> $out; }
run_b STT_AMD_PIPELINE=2 STT_AMD_LSTM_PASSES=1
run_b STT_AMD_PIPELINE=3 STT_AMD_LSTM_PASSES=1
run_b STT_AMD_PIPELINE=3 STT_AMD_LSTM_PASSES=2
run_b STT_AMD_PIPELINE=4 STT_AMD_LSTM_PASSES=1
run_b STT_AMD_PIPELINE=3 STT_AMD_ACTIVE=1 STT_AMD_LSTM_PASSES=1
run_b STT_AMD_PIPELINE=3 STT_AMD_ACTIVE=3 STT_AMD_LSTM_PASSES=1
run_b STT_AMD_PIPELINE=3 STT_AMD_LSTM_PASSES=1 STT_AMD_PCHUNK=32
run_b STT_AMD_PIPELINE=2 STT_AMD_LSTM_PASSES=1
run_b STT_AMD_PIPELINE=3 STT_AMD_LSTM_PASSES=1
cat $out
