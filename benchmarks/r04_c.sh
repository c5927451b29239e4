#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python benchmarks/ref_mismatch_probe.py 2>&1 | grep -v "amdgpu.ids\|TensorFlow\|Coqui" | tail -150
timeout 600 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -q -x 2>&1 | tail -30
