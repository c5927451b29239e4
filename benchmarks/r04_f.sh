#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 600 python benchmarks/search_micro.py --reps 3 2>&1 | grep -v "amdgpu.ids\|TensorFlow\|Coqui" | cut -c1-700 | tee gpurun_out/r04_f_search_micro.txt
