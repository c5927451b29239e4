#!/bin/bash
# Round 6, search step A/B: parity (decoder goldens + fuzz) and the search alone (search_micro.py) for every value of the tunable search_exp
# given on the command line.    bash benchmarks/r06_search_exp.sh "0 1 2 3" [peaky] [extra tunables, e.g. lm_waves=4]
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
VALS=${1:-"0 1 2 3"}
EXTRA=${3:+,$3}
LOG=$OUT/r06_search_exp.txt; : > $LOG
for v in $VALS; do
  echo "== search_exp=$v$EXTRA parity" >> $LOG
  STT_AMD_TUNING="search_exp=$v$EXTRA" timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1 >> $LOG
done
for v in $VALS; do
  echo "== search_exp=$v$EXTRA micro (model emissions)" >> $LOG
  timeout 600 python benchmarks/search_micro.py --reps 3 --set search_exp=$v$EXTRA 2>&1 | grep "rep" >> $LOG
done
if [ "$2" = peaky ]; then
  for v in $VALS; do
    echo "== search_exp=$v$EXTRA micro (peaky, fixture scorer)" >> $LOG
    timeout 600 python benchmarks/search_micro.py --reps 3 --emissions peaky --scorer fixture --set search_exp=$v$EXTRA 2>&1 | grep "rep" >> $LOG
  done
fi
cat $LOG
