#!/bin/bash
# Same-box A / B: the search kernels with everything inlined (the shipped build) against the build with real device calls (-DSTT_DEVICE_CALLS,
# stt_amd/lib/variants/calls/*.so built by hand: DESIGN.md 10.11).  Run on the GPU box's scratch copy: the variant overwrites stt_amd/lib/*.so there.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_calls_ab.txt; : > $OUT
one() {
  for wl in bytes peaky_bytes peaky; do
    timeout 600 python bench.py --workload $wl --steps 8 --warmup 4 --no-extras --no-cpu-baseline --no-reference-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1 $wl', {k:d.get(k) for k in ('value','ms_per_step','verified')})" >> $OUT
  done
  timeout 600 python benchmarks/bytes_phase_probe.py 2>&1 | grep ms_profiled | cut -c1-330 | sed "s/^/$1 /" >> $OUT
  timeout 300 python benchmarks/search_micro.py 2>&1 | tail -1 | cut -c1-130 | sed "s/^/$1 /" >> $OUT
}
mkdir -p /tmp/inl && cp stt_amd/lib/libstt.so stt_amd/lib/libstt_test.so /tmp/inl/
one inlined
cp stt_amd/lib/variants/calls/libstt.so stt_amd/lib/variants/calls/libstt_test.so stt_amd/lib/
one calls
cp /tmp/inl/*.so stt_amd/lib/
one inlined
cat $OUT
