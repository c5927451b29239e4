#!/bin/bash
# Round 6: the search alone (search_micro.py, model emissions + peaky) under several tunable settings on ONE box, two passes (A B A B) so that
# drift of the box shows.    bash benchmarks/r06_micro_ab.sh "lm_waves=0" "lm_waves=1" ...
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_micro_ab.txt; : > $LOG
for rep in 1 2; do
for t in "$@"; do
  echo "== $t (pass $rep)" >> $LOG
  timeout 600 python benchmarks/search_micro.py --reps 3 --set $t 2>&1 | grep "rep" >> $LOG
  timeout 600 python benchmarks/search_micro.py --reps 3 --emissions peaky --scorer fixture --set $t 2>&1 | grep "rep" >> $LOG
done
done
