#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/prof_r04_stream; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python benchmarks/stream_rolling.py --passes 2 --scorer fixture > $OUT/trace.log 2>&1
grep -v "amdgpu.ids\|TensorFlow\|Coqui" $OUT/trace.log | tail -4 | cut -c1-600
python - <<'PY'
import glob, csv
st = glob.glob("gpurun_out/prof_r04_stream/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), ("%.1f" % (float(r["TotalDurationNs"])/1e6)).rjust(9), "ms", ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(8), "us", r["Percentage"])
import shutil; shutil.copy(st[0], "gpurun_out/r04_stream_kernel_stats.csv")
PY
