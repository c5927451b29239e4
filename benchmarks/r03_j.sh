#!/bin/bash
# end-of-round dress rehearsal: smoke, the N = 2 plumbing of bench.py on one GPU (gloo), the full GPU suite
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
STT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/r03_j_n2.json 2> gpurun_out/r03_j_n2.err
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r03_j_n2.json').read().strip().splitlines()[-1])
    print('N=2 (gloo, one GPU shared): ms/step', round(r['ms_per_step'],3), 'value', round(r['value']), 'n_gpus', r['n_gpus'], 'verified', r.get('verified'), 'global_batch', r['config']['global_batch'])
except Exception as e:
    print('N=2 FAILED', e); print(open('gpurun_out/r03_j_n2.err').read()[-1500:])
PY
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4
