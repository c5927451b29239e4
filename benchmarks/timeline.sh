#!/bin/bash
# Kernel timeline of one batch (which kernel runs when, on which stream): rocprofv3 --kernel-trace of a short bench run,
# the trace CSV reduced to {name, start, end, stream/queue} rows of the last step under gpurun_out/<tag>_timeline.csv
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r02_tl}
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -o trace --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
python - "$TAG" "$OUT" <<'PY'
import csv, glob, sys
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("gpurun_out/%s_timeline.csv" % tag, "w") as o:
    o.write("kernel,queue,start_us,end_us\n")
    for r in rows:
        o.write('"%s",%s,%.1f,%.1f\n' % (r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
print(len(rows))
PY
tail -3 $OUT/trace.log
