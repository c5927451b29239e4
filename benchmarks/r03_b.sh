#!/bin/bash
# round 3, call b: recurrent-step microbenchmark (forms, row counts, timing probes) + a kernel trace of the paired bench
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python benchmarks/lstm_micro.py > gpurun_out/r03_b_lstm_micro.jsonl 2> gpurun_out/r03_b_lstm_micro.err
cat gpurun_out/r03_b_lstm_micro.jsonl
rm -rf gpurun_out/prof_r03_b; mkdir -p gpurun_out/prof_r03_b
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_b/trace -o trace --output-format csv -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-extras > gpurun_out/prof_r03_b/trace.log 2>&1
f=$(find gpurun_out/prof_r03_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_b_kernel_stats.csv && head -30 $f | cut -c1-220
find gpurun_out/prof_r03_b -name "*kernel_trace.csv" -size +30M -delete
tail -3 gpurun_out/prof_r03_b/trace.log | cut -c1-400
