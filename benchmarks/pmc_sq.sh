#!/bin/bash
# SQ counter passes over bench.py (counters only, no trace domains); per-kernel sums -> gpurun_out/pmc_sq.csv
# usage (on the GPU box): bash benchmarks/pmc_sq.sh [extra bench args]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P -d $OUT/p$i -o p$i --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
with open("gpurun_out/pmc_sq.csv", "w") as o:
    o.write("Kernel,Counter,Calls,Sum,AvgPerCall\n")
    for (k, c), v in sorted(acc.items()):
        o.write('"%s",%s,%d,%.0f,%.1f\n' % (k, c, calls[(k, c)], v, v / calls[(k, c)]))
print(open("gpurun_out/pmc_sq.csv").read())
PY
