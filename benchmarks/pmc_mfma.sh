#!/bin/bash
# MFMA utilisation per kernel: one rocprofv3 --pmc pass (counters only) over bench.py; per-kernel sums -> gpurun_out/<tag>_pmc_mfma.csv
# (counter mode runs every dispatch by itself: these are each kernel's figures ALONE, not beside its co-tenants)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r03_x}
OUT=gpurun_out/pmc_mfma_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p -o p --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/p.log 2>&1
python - "$TAG" "$OUT" <<'PY'
import csv, glob, collections, sys
tag, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
kern = sorted({k for k, _ in acc})
with open("gpurun_out/%s_pmc_mfma.csv" % tag, "w") as o:
    o.write("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras\n")
    o.write("# mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) * 100 (rocprofv3's MfmaUtil expression); every dispatch runs alone in counter mode\n")
    o.write("Kernel,Calls,GRBM_GUI_ACTIVE_avg,MFMA_BUSY_avg,MFMA_F16_insts_avg,mfma_util_pct,wait_any_pct,wait_inst_pct,active_inst_pct\n")
    for k in kern:
        n = calls[(k, "GRBM_GUI_ACTIVE")] or 1
        g = acc[(k, "GRBM_GUI_ACTIVE")] / n; mb = acc[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] / n; mi = acc[(k, "SQ_INSTS_VALU_MFMA_F16")] / n
        wc = acc[(k, "SQ_WAVE_CYCLES")] or 1.0
        o.write('"%s",%d,%.0f,%.0f,%.0f,%.2f,%.1f,%.1f,%.1f\n' % (k, n, g, mb, mi, 100.0 * mb / (g * 1024.0) if g else 0.0, 100 * acc[(k, "SQ_WAIT_ANY")] / wc, 100 * acc[(k, "SQ_WAIT_INST_ANY")] / wc, 100 * acc[(k, "SQ_ACTIVE_INST_ANY")] / wc))
print(open("gpurun_out/%s_pmc_mfma.csv" % tag).read())
PY
