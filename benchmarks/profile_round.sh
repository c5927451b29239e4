#!/bin/bash
# Per-round profile of `python bench.py` on the GPU box (scratch output under gpurun_out/prof_<tag>/, summaries copied
# to profiles/ by hand):  1. rocprofv3 --kernel-trace --stats  2. --pmc FETCH_SIZE  3. --pmc WRITE_SIZE (separate passes:
# the two cannot share the TCC slots, and counters never ride on a trace/timed run).
# usage: bash benchmarks/profile_round.sh <tag> [workload]      (workload: batch (default) or batch_i8 -- the int8 path's sub-line)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r03_x}
WL=${2:-batch}
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python bench.py --workload $WL --steps 6 --warmup 4 --no-cpu-baseline --no-extras --no-reference-check > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-reference-check > $OUT/pmc_$C.log 2>&1
done
python - "$TAG" "$OUT" <<'PY'
import csv, glob, json, collections, sys, shutil
tag, out = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if st:
    shutil.copy(st[0], "gpurun_out/%s_kernel_stats.csv" % tag)
    print(open(st[0]).read()[:3000])
acc = collections.defaultdict(float); calls = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
steps = 11.0   # --steps 4 --warmup 2 + the untimed phase-counter step + one blocking verification call per distinct timed batch (4): batches of 64
kern = {}
with open("gpurun_out/%s_pmc_per_kernel.csv" % tag, "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE (pass 1) and --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-reference-check; values in KB as reported by rocprofv3\n")
    o.write("Kernel,Counter,Calls,SumKB,AvgKBPerCall\n")
    for (k, c), v in sorted(acc.items()):
        n = calls[(k, c)]
        o.write('"%s",%s,%d,%.1f,%.2f\n' % (k, c, n, v, v / n))
        e = kern.setdefault(k.replace("void ", ""), {"launches_per_batch": round(n / steps, 3)})
        e["fetch_kb_per_launch" if c == "FETCH_SIZE" else "write_kb_per_launch"] = round(v / n, 1)
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes), python bench.py --steps 4 --warmup 2 --no-extras, MI355X; KB per launch as reported "
                   "(FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950, MI355X_MICROARCH.md HBM section)", "kernels": kern},
          open("gpurun_out/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(kern, indent=1)[:2500])
PY
