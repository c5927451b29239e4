#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_hybrid.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids\|TensorFlow\|Coqui" | tail -12 | cut -c1-600
