#!/bin/bash
# A/B of the batch path's time-chunk schedule and the dense tile side (acoustic stage alone, then the pipelined bench)
out=gpurun_out/sweep_chunks.log; : > $out
run_am() { echo "== am_micro $*" >> $out; env "$@" timeout 120 python benchmarks/am_micro.py 5 2>/dev/null | tail -1 | cut -c1-330 >> $out; }
run_b() { echo "== bench $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['p50_utterance_latency_ms'], r.get('stage_ms_per_step'))" >> $out; }
run_am A=1
run_am STT_AMD_CHUNK0=250 STT_AMD_CHUNK=250
run_am STT_AMD_CHUNK0=250 STT_AMD_CHUNK=250 STT_AMD_DENSE_TILE=256
run_am STT_AMD_CHUNK0=128 STT_AMD_CHUNK=128 STT_AMD_DENSE_TILE=256
run_b GPU_MAX_HW_QUEUES=8
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_CHUNK0=250 STT_AMD_CHUNK=250
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_CHUNK0=250 STT_AMD_CHUNK=250 STT_AMD_DENSE_TILE=256
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_CHUNK0=128 STT_AMD_CHUNK=128 STT_AMD_DENSE_TILE=256
cat $out
