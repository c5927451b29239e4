#!/bin/bash
# Round 6: what sits between two dependent recurrent launches?  The verdict's arithmetic: 16.45 us of kernel per step by rocprofv3, 22.8 us per
# launch as run by HIP events -- 6.4 us x 125 launches per batch that are not kernel time.  Before rebuilding the recurrence, the HIP runtime's
# own switches: kernel arguments in device memory, fence scopes, how graph nodes are turned into AQL packets, the number of hardware queues.
# One box, the headline workload (no extras, no CPU baseline, blocking verification), ms per batch + the engines' busy times.
#   usage: benchmarks/r06_env_knobs.sh [workload] [env|cus]      (batch | batch_i8; the runtime's switches | CU-mask partitions)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
WL=${1:-batch}
OUT=gpurun_out/r06_${2:-env}_knobs_$WL.txt; : > $OUT
run() {   # $1 = label, rest = env assignments
  local label="$1"; shift
  env "$@" timeout 300 python bench.py --workload $WL --steps 24 --warmup 8 --no-extras --no-cpu-baseline --no-reference-check > gpurun_out/ek.json 2> gpurun_out/ek.err
  python - "$label" >> $OUT <<'PY'
import json, sys
try:
    r = json.loads(open('gpurun_out/ek.json').read().strip().splitlines()[-1])
    s = r.get('stage_ms_per_step', {})
    print("%-44s ms/step %.3f  verified %s  lstm %.3f  dense_in %.3f  dense_out %.3f  search %.3f  moves %s  us/launch %.2f" % (
        sys.argv[1], r['ms_per_step'], r['verified'], s.get('lstm_ms', 0), s.get('dense_in_ms', 0), s.get('dense_out_ms', 0), s.get('decoder_next_ms', 0),
        r['config'].get('queue_moves'), 1e3 * r['roofline']['avg_launch_ms']))
except Exception as e:
    print("%-44s FAILED %r %s" % (sys.argv[1], e, open('gpurun_out/ek.err').read()[-300:].replace("\n", " | ")))
PY
}
if [ "${2:-env}" = env ]; then
run "defaults" X=1
run "defaults (again)" X=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "AMD_OPT_FLUSH=0 (system-scope fences)" AMD_OPT_FLUSH=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "ROC_SYSTEM_SCOPE_SIGNAL=0" ROC_SYSTEM_SCOPE_SIGNAL=0
run "GPU_MAX_HW_QUEUES=4" GPU_MAX_HW_QUEUES=4
run "eager recurrence (lstm_graph=0)" STT_AMD_TUNING=lstm_graph=0
elif [ "$2" = chunks ]; then
# with the engines' streams placed (no more 3.0-or-6.1 lottery between runs): chunk lengths and row groups, both paths
run "defaults" X=1
run "defaults (again)" X=1
run "pchunk=32" STT_AMD_TUNING=pchunk=32
run "pchunk=64" STT_AMD_TUNING=pchunk=64
run "pchunk=96" STT_AMD_TUNING=pchunk=96
run "pchunk0=32" STT_AMD_TUNING=pchunk0=32
run "pchunk0=48,pchunk=64" STT_AMD_TUNING=pchunk0=48,pchunk=64
run "lstm_i8_rows=128" STT_AMD_TUNING=lstm_i8_rows=128
run "active=2" STT_AMD_TUNING=active=2
elif [ "$2" = i8 ]; then
# the int8 path's recurrence: row groups, chunk lengths, searches side by side (run with workload batch_i8)
run "defaults" X=1
run "lstm_i8_rows=128 (one row group)" STT_AMD_TUNING=lstm_i8_rows=128
run "lstm_i8_rows=32" STT_AMD_TUNING=lstm_i8_rows=32
run "search on 128 CUs" STT_AMD_TUNING=search_cus=128
run "pchunk=32" STT_AMD_TUNING=pchunk=32
run "pchunk=64" STT_AMD_TUNING=pchunk=64
run "active=2" STT_AMD_TUNING=active=2
run "pipeline=3" STT_AMD_TUNING=pipeline=3
else
# a fixed partition of the chip (hipExtStreamCreateWithCUMask; benchmarks/cumask_probe.hip says where the bits land)
run "defaults" X=1
run "search on 128 CUs, engines anywhere" STT_AMD_TUNING=search_cus=128
run "search 128 | all three engines on the rest" STT_AMD_TUNING=search_cus=128,am_cus=7
run "search 128 | recurrence on the rest" STT_AMD_TUNING=search_cus=128,am_cus=2
run "search 128 | GEMM + output engines on the rest" STT_AMD_TUNING=search_cus=128,am_cus=5
run "search 112 | engines on the rest" STT_AMD_TUNING=search_cus=112,am_cus=7
run "search 144 | engines on the rest" STT_AMD_TUNING=search_cus=144,am_cus=7
run "search 128, two searches side by side" STT_AMD_TUNING=search_cus=128,active=2
fi
cat $OUT
