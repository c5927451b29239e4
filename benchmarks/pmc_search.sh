#!/bin/bash
# SQ / instruction-cache counter passes over the search kernel alone (benchmarks/search_micro.py); counters only, no trace domains.
# usage (on the GPU box): bash benchmarks/pmc_search.sh      -> gpurun_out/pmc_search.csv (per kernel: sum and average per launch)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/pmc_search; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|SQC)_[A-Z0-9_]+" | sort -u > $OUT/avail.txt
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
 "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_ACTIVE_INST_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAVES"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  Q=""; for c in $P; do grep -qx "$c" $OUT/avail.txt && Q="$Q $c"; done
  [ -z "$Q" ] && continue
  rocprofv3 --pmc $Q -d $OUT/p$i -o p$i --output-format csv -- python benchmarks/search_micro.py --reps 2 "$@" > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_search/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "ctc_next" not in k: continue
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
with open("gpurun_out/pmc_search.csv", "w") as o:
    o.write("Kernel,Counter,Launches,Sum,PerStreamTimestep(64x250 per launch)\n")
    for (k, c), v in sorted(acc.items()):
        o.write('"%s",%s,%d,%.0f,%.1f\n' % (k, c, calls[(k, c)], v, v / calls[(k, c)] / 16000.0))
print(open("gpurun_out/pmc_search.csv").read())
PY
