#!/bin/bash
# kernel trace of the rolling streaming benchmark (2 cohorts x 128 live streams, fixture scorer): where a hop's GPU time goes
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_stream; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python benchmarks/stream_rolling.py --passes 2 --scorer fixture --cohorts 2 > $OUT/trace.log 2>&1
grep "^{" $OUT/trace.log | tail -1 | cut -c1-500
python - <<'PY'
import glob, csv, shutil
st = glob.glob("gpurun_out/prof_stream/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot/1e6, 1))
for r in rows[:14]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(7), ("%.1f" % (float(r["TotalDurationNs"])/1e6)).rjust(8), "ms", ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(8), "us", r["Percentage"])
shutil.copy(st[0], "gpurun_out/r04_stream_kernel_stats.csv")
PY
