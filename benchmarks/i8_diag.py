"""Round-5 diagnostics of the int8 path (needs a MI355X):
  1. MFCC: the feature kernel against oracle/am_ref.py -- how many outputs differ, by how much (the hybrid path quantises these rows:
     a last-bit difference can flip an int8 and the recurrence amplifies it);
  2. the acoustic chain fed with the ORACLE's windows (no feature kernel involved) against oracle/am_hybrid.py on 4 utterances;
  3. the 128-row recurrent step alone: microseconds per step, and with parts switched off (tunable lstm_probe)."""
import json
import os

os.environ.setdefault("STT_AMD_TEST_HOOKS", "1")   # a probe of single kernels: needs libstt_test.so (include/stt_amd_test.h)
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import am_hybrid, am_ref          # noqa: E402
from stt_amd import Model, native, synth, tflitefile      # noqa: E402

out = {}
w = synth.synth_weights(0, n_hidden=2048)
d = tempfile.mkdtemp()
path = os.path.join(d, "q.tflite")
tflitefile.write_tflite(path, w, synth.ENGLISH_LABELS, quantize=True, beam_width=500)
m = Model(path)
assert m.acousticMode() == 1
audio = list(synth.synth_audio_batch(4, 80000, seed=4242))
spec = am_ref.MfccSpec()
# 1
mf_ref = [spec.frames_fast(np.asarray(a, dtype=np.int16)) for a in audio]
mf_gpu = [m.computeMfcc(a) for a in audio]
dd = np.concatenate([(g - r).ravel() for g, r in zip(mf_gpu, mf_ref)])
out["mfcc"] = {"values": int(dd.size), "differ": int((dd != 0).sum()), "max_abs": float(np.abs(dd).max()), "rms": float(np.sqrt((dd ** 2).mean())),
               "differ_by_more_than_2e-6": int((np.abs(dd) > 2e-6).sum())}
# 2
win = np.stack([am_ref.context_windows(f) for f in mf_ref])            # [B][T][494]
hm = am_hybrid.HybridModel(w)
want = hm.forward_batch(win)
got = m.hybridChain(win.transpose(1, 0, 2))["probs"]
dl = np.log(got) - np.log(want)
out["chain_on_oracle_windows"] = {"rms_dlnp": float(np.sqrt((dl ** 2).mean())), "max_abs_dlnp": float(np.abs(dl).max())}
win_g = np.stack([am_ref.context_windows(f) for f in mf_gpu])
got_g = m.hybridChain(win_g.transpose(1, 0, 2))["probs"]
dl = np.log(got_g) - np.log(want)
out["chain_on_gpu_mfcc_windows"] = {"rms_dlnp": float(np.sqrt((dl ** 2).mean())), "max_abs_dlnp": float(np.abs(dl).max())}
e2e = np.stack(m.acousticProbs(audio))
out["end_to_end_equals_chain_on_gpu_mfcc"] = bool(np.array_equal(e2e, got_g))
# 3
rng = np.random.default_rng(1)
T, B = 48, 128
wb = (rng.standard_normal((T, B, 494)) * 4).astype(np.float32)
times = {}
native.set_tuning("lstm_i8_rows", 128)          # (the probe kernels are forms of the one-group 128-row step)
for probe in (0, 1, 2, 4, 8, 15):
    native.set_tuning("lstm_probe", probe)
    best = 1e9
    for _ in range(4):
        best = min(best, m.hybridChain(wb)["lstm_ms"])
    times[str(probe)] = 1e3 * best / T
native.set_tuning("lstm_probe", 0)
out["i8_step_us_at_128_rows"] = {"as_shipped": times["0"], "cheap_activations": times["1"], "no_reduction": times["2"], "no_cell_operand_loads": times["4"],
                                  "operands_from_L1": times["8"], "all_four": times["15"]}
native.set_tuning("lstm_i8_rows", 128)
for Bx in (64, 16):
    best = 1e9
    for _ in range(4):
        best = min(best, m.hybridChain(wb[:, :Bx])["lstm_ms"])
    out["i8_step_us_at_%d_rows" % Bx] = 1e3 * best / T
# 4. row groups: the launch's batch tiles dealt to 2 / 4 / 8 workgroups per 16-unit slice (tunable lstm_i8_rows = rows per workgroup)
out["i8_step_us_by_rows_per_workgroup"] = {}
for Bx in (128, 64):
    for rows in (128, 64, 32, 16):
        if rows > Bx:
            continue
        native.set_tuning("lstm_i8_rows", rows)
        best = 1e9
        for _ in range(4):
            best = min(best, m.hybridChain(wb[:, :Bx])["lstm_ms"])
        out["i8_step_us_by_rows_per_workgroup"]["%d_rows_in_groups_of_%d" % (Bx, rows)] = 1e3 * best / T
native.set_tuning("lstm_i8_rows", 64)
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "i8_diag.json"), "w"), indent=1)
