#!/bin/bash
# K idle streams before the model, each point a fresh process: with the placement (round 6 default) and as round 5 ran (no placement, no watch)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp STT_AMD_TEST_HOOKS=0
mkdir -p gpurun_out
OUT=gpurun_out/r06_queue_placement_sweep.jsonl; : > $OUT
for mode in f16 int8; do
  for k in 0 1 2 3 4 5 6 7 8 12 16; do
    python benchmarks/queue_placement.py --idle $k --mode $mode --place 0 --moves 0 2>/dev/null | tail -1 >> $OUT
    python benchmarks/queue_placement.py --idle $k --mode $mode --place 1 --moves 0 2>/dev/null | tail -1 >> $OUT
  done
done
cat $OUT
