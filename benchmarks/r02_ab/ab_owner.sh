#!/bin/bash
out=gpurun_out/ab_owner.log; : > $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $out
for f in 1 3 1 3; do echo "== am_micro FORM=$f" >> $out; STT_AMD_LSTM_FORM=$f timeout 120 python benchmarks/am_micro.py 5 2>/dev/null | tail -1 | cut -c1-130 >> $out; done
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), round(r.get('p50_utterance_latency_ms'),2), round(r.get('host_enqueue_ms_per_step'),2), {k:round(v,2) for k,v in r.get('stage_ms_per_step').items()})" >> $out; }
run_b STT_AMD_LSTM_FORM=3
run_b STT_AMD_LSTM_FORM=2
run_b STT_AMD_LSTM_FORM=3
run_b STT_AMD_LSTM_FORM=2
run_b STT_AMD_LSTM_FORM=3 STT_AMD_PCHUNK=32
cat $out
