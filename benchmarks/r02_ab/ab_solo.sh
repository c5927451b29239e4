#!/bin/bash
out=gpurun_out/ab_solo.log; : > $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $out
echo "== am_micro" >> $out; timeout 120 python benchmarks/am_micro.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), round(r.get('p50_utterance_latency_ms'),2), {k:round(v,2) for k,v in r.get('stage_ms_per_step').items()})" >> $out; }
run_b STT_AMD_DENSE_SOLO=1
run_b STT_AMD_DENSE_SOLO=0
run_b STT_AMD_DENSE_SOLO=1 STT_AMD_PCHUNK=32
run_b STT_AMD_DENSE_SOLO=1 STT_AMD_PCHUNK=64
run_b STT_AMD_DENSE_SOLO=1 STT_AMD_PIPELINE=3
run_b STT_AMD_DENSE_SOLO=1 STT_AMD_LSTM_PRIO=0
run_b STT_AMD_DENSE_SOLO=1
cat $out
