#!/bin/bash
# dense_wide_kernel in two stages of K = 64 (dense_solo = 3, rounds 3-5) against four stages of K = 32 (dense_solo = 4, round 6): alone and beside the
# recurrent step (lstm_cotenant harness), then the whole pipeline, both arithmetic paths.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/r06_gemm_form.txt; : > $OUT
python - >> $OUT 2>&1 <<'PY'
import os, sys, json, tempfile
os.environ["STT_AMD_TEST_HOOKS"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np
from stt_amd import Model, modelfile, native, synth
H = 2048
w = synth.synth_weights(0, n_hidden=H)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.sttw"); modelfile.write_model(p, w, synth.ENGLISH_LABELS, beam_width=500); m = Model(p)
x = (np.random.default_rng(0).standard_normal((4 * 128, 4 * H)) * 1.5).astype(np.float32)
for solo in (3, 4, 3, 4):
    native.set_tuning("dense_solo", solo)
    native.set_tuning("lstm_cotenant", 12); ms = m.lstmSteps(x, 128, 2, graph=True, timing=True)[3]        # the GEMMs (nearly) alone: LSTM_COTENANT line on stderr
    native.set_tuning("lstm_cotenant", 24); native.set_tuning("lstm_stamps", 1)
    ms = m.lstmSteps(x, 128, 500, graph=True, timing=True)[3]
    print(json.dumps({"dense_solo": solo, "step_us_beside_the_gemms": round(1e3 * ms / 500, 2)}), flush=True)
    native.set_tuning("lstm_cotenant", 0); native.set_tuning("lstm_stamps", 0)
PY
for WL in batch batch_i8; do
  for T in dense_solo=3 dense_solo=4 dense_solo=3 dense_solo=4; do
    STT_AMD_TUNING=$T timeout 300 python bench.py --workload $WL --steps 24 --warmup 8 --no-extras --no-cpu-baseline --no-reference-check > gpurun_out/gf.json 2> gpurun_out/gf.err
    python - "$WL $T" >> $OUT <<'PY'
import json, sys
try:
    r = json.loads(open('gpurun_out/gf.json').read().strip().splitlines()[-1]); s = r['stage_ms_per_step']
    print("%-24s ms/step %.3f verified %s lstm %.3f dense_in %.3f dense_out %.3f search %.3f us/launch %.2f" % (sys.argv[1], r['ms_per_step'], r['verified'], s['lstm_ms'], s['dense_in_ms'], s['dense_out_ms'], s['decoder_next_ms'], 1e3 * r['roofline']['avg_launch_ms']))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e), open('gpurun_out/gf.err').read()[-300:])
PY
  done
done
cat $OUT
