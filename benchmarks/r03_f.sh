#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 ./benchmarks/grid_barrier_probe > gpurun_out/r03_f_grid_barrier.jsonl 2>&1; cat gpurun_out/r03_f_grid_barrier.jsonl
STT_AMD_TUNING=dump_marks=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2> gpurun_out/r03_f_bench.err > gpurun_out/r03_f_bench.json
grep ARENA gpurun_out/r03_f_bench.err | sort | uniq -c | sort -rn | head -30
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
