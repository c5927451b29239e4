#!/bin/bash
# configs[4] (bytes-output mode, code-point scorer, beam 1024) under rocprofv3: kernel trace, then FETCH_SIZE / WRITE_SIZE / L2 hit-miss
# counters in passes of their own -- what the search kernel of that workload moves per launch, against how long a launch takes.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_bytes; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --workload bytes --steps 4 --warmup 4 --no-cpu-baseline --no-extras --no-reference-check"
timeout 170 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_')
  timeout 170 rocprofv3 --pmc $C -d $D -o pmc --output-format csv -- $CMD > $D.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, collections, sys
out = sys.argv[1]
res = {"command": "python bench.py --workload bytes --steps 4 --warmup 4 --no-cpu-baseline --no-extras --no-reference-check", "kernel": "ctc_next_kernel<2, 1024, false>"}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ctc_next_kernel<2, 1024" in r["Name"]:
            res["launches"] = int(r["Calls"]); res["avg_launch_ms"] = float(r["AverageNs"]) / 1e6; res["share_of_kernel_time_pct"] = float(r["Percentage"])
acc = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ctc_next_kernel<2, 1024" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
res["counters_per_launch"] = {k: v / calls[k] for k, v in acc.items()}
res["counter_launches"] = dict(calls)
for line in open(out + "/trace.log"):
    if line.startswith("{"):
        d = json.loads(line); res["traced_ms_per_batch"] = d["ms_per_step"]
json.dump(res, open("gpurun_out/r04_q_bytes_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
tail -3 $OUT/pmc_TCC_HIT_sum_TCC_MISS_sum.log $OUT/pmc_TCC_EA0_RDREQ_sum_TCC_EA0_RDREQ_32B_sum.log | cut -c1-300
