#!/bin/bash
out=gpurun_out/ab_w8.log; : > $out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_async.py tests/test_gpu_benchshape.py -x -q 2>&1 | tail -3 >> $out
echo "== am_micro" >> $out; timeout 120 python benchmarks/am_micro.py 5 2>/dev/null | tail -1 | cut -c1-130 >> $out
run_b() { echo "== $*" >> $out; env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 24 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), round(r.get('p50_utterance_latency_ms'),2), round(r.get('host_enqueue_ms_per_step'),2), {k:round(v,2) for k,v in r.get('stage_ms_per_step').items()})" >> $out; }
run_b STT_AMD_DENSE_SOLO=2
run_b STT_AMD_DENSE_SOLO=1
run_b STT_AMD_DENSE_SOLO=2
run_b STT_AMD_DENSE_SOLO=1
cat $out
