#!/bin/bash
out=gpurun_out/ab_copy.log; : > $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('p50_utterance_latency_ms'), r.get('stage_ms_per_step'))" >> $out; }
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=2 STT_AMD_COPY_KERNEL=0
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=2 STT_AMD_COPY_KERNEL=1
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=3 STT_AMD_COPY_KERNEL=1
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=2 STT_AMD_COPY_KERNEL=1
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_COPY_KERNEL=1
run_b GPU_MAX_HW_QUEUES=8 STT_AMD_AM_PIPE=1 STT_AMD_PIPELINE=3 STT_AMD_COPY_KERNEL=1 STT_AMD_DENSE_LDS_KB=0
run_b STT_AMD_AM_PIPE=0 STT_AMD_PIPELINE=2 STT_AMD_COPY_KERNEL=0
echo "== stream workload copy kernel 0 / 1" >> $out
for c in 0 1; do STT_AMD_COPY_KERNEL=$c timeout 200 python bench.py --workload stream --utterances 128 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r.get('hop_latency_ms'))" >> $out; done
cat $out
