#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r04_b_bench.json 2> gpurun_out/r04_b_bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_b_bench.json').read().strip().splitlines()[-1])
print(r['verified'], json.dumps(r['verify_mismatches'], indent=1)[:3000])
PY
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-extras --no-cpu-baseline --scorer fixture > gpurun_out/r04_b_n2.json 2> gpurun_out/r04_b_n2.err
tail -c 2500 gpurun_out/r04_b_n2.err; tail -c 600 gpurun_out/r04_b_n2.json
