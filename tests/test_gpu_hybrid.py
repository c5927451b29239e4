"""The stated tolerance against the reference's *TFLite CPU path*: released models are dynamic-range quantised and that path quantises
the activations of every FULLY_CONNECTED to int8 per call (oracle/am_hybrid.py restates the published kernel).  Two engine paths:
  * "int8" (round 5; what a quantised `.tflite` takes by default): the same arithmetic -- int8 activations per row, int32 sums, f32 rescale --
    end to end; what is left is the last bit of expf / tanhf in the cell, which now and then flips one int8 of a later quantised row;
  * "f16" (tunable am_i8 = 0; north_star's f16 MFMA / f32 accumulate): the same int8 weights de-quantised to f16, activations f16 / f32.
At the bench's shape (n_hidden 2048, 64 x 5 s, beam 500) this measures how far each is from the hybrid path on the softmax outputs and
what that does to transcripts.  Needs a MI355X."""
import json
import os

import numpy as np
import pytest

from conftest import OUT
from stt_amd import synth, tflitefile

pytestmark = pytest.mark.gpu

# The stated tolerance of the HIP path against the hybrid-int8 restatement, over every class of every frame of 64 x 5 s: with the
# reference's initialisation scale |d p| <= 1e-3 and |d ln p| <= 1.25e-2 (measured 3.0e-4 / 7.2e-3, rms 1.6e-3); the error is made
# in the output layer's input and therefore grows with that layer's weights -- `head` times the weights, `head` times the bound
# (measured at x 8: 8.8e-3 / 5.8e-2).  It consists of the activation quantisation noise of the reference's OWN CPU path: against
# the float graph of the same de-quantised weights the engine stays within 1e-4 / 2e-3 (tests/test_gpu_benchshape.py).
# int8 path: what is left is float rounding in the softmax (the engine multiplies by 1 / sum, the restatement divides) -- measured
# |d p| 1.5e-8 ... 3e-7, |d ln p| 4.8e-7 ... 9.5e-7 at every head scale (profiles/r05_hybrid_tolerance.json): the bound does not grow with the head.
ABS_TOL = {"f16": 1e-3, "int8": 1e-6}
LOG_TOL = {"f16": 1.25e-2, "int8": 4e-6}
# transcripts equal to the reference decoder's on the hybrid path's probabilities, of 64: a random-init head is a coin toss per frame
# (mean top probability 0.04) and any perturbation re-routes the beam; the more a model commits, the fewer transcripts move
MIN_EQUAL = {"f16": {1.0: 0, 8.0: 48, 32.0: 62},       # measured: 4, 57, 64 (profiles/r04_hybrid_tolerance.json)
             "int8": {1.0: 60, 8.0: 63, 32.0: 64}}     # measured: 63, 64, 64 -- and every transcript that differs must be a boundary tie of the SEARCH (below)


_WANT = {}


def _model(tmp, w, name, mode, beam=500):
    from stt_amd import Model, native
    path = str(tmp / (name + ".tflite"))
    tflitefile.write_tflite(path, w, synth.ENGLISH_LABELS, quantize=True, beam_width=beam)
    native.set_tuning("am_i8", -1 if mode == "int8" else 0)
    try:
        m = Model(path)
    finally:
        native.set_tuning("am_i8", -1)
    assert m.acousticMode() == (1 if mode == "int8" else 0)
    return m


@pytest.mark.parametrize("mode", ["int8", "f16"])
@pytest.mark.parametrize("head", [1.0, 8.0, 32.0], ids=["random-init head", "head x 8", "head x 32 (peaky outputs)"])
def test_engine_against_the_hybrid_int8_path(tmp_path, ref, port, english, fix, head, mode):
    from oracle import am_hybrid
    B = 64
    w = synth.synth_weights(0, n_hidden=2048)
    w["layer_6/weights"] = (w["layer_6/weights"] * head).astype(np.float32)
    model = _model(tmp_path, w, "q%d" % int(head), mode)
    model.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    audio = list(synth.synth_audio_batch(B, 80000, seed=4242))
    got = np.stack(model.acousticProbs(audio))                                       # [B][250][29], the HIP path on the de-quantised weights
    if head not in _WANT:                                                            # the hybrid int8 kernels (shared by the two engine paths)
        _WANT[head] = am_hybrid.utterance_probs_batch(audio, w)
    want = _WANT[head]
    assert got.shape == want.shape == (B, 250, 29)
    a_err = float(np.abs(got - want).max())
    l_err = float(np.abs(np.log(got) - np.log(want)).max())
    rms = float(np.sqrt(np.mean((np.log(got) - np.log(want)) ** 2)))
    # transcripts: the engine's own against the REAL reference decoder run on the hybrid path's probabilities (what the reference's CPU
    # path would print), same scorer, same beam
    texts = model.sttBatch(audio)
    A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
    S = ref.Scorer(os.path.join(fix, "pruned_lm.scorer"), A)
    res = ref.decode_batch(want.astype(np.float64), [250] * B, A, 500, os.cpu_count() or 1, S)
    ref_texts = [A.decode(tok).decode("utf-8", "replace") for _, tok in res]
    same = sum(1 for x, y in zip(texts, ref_texts) if x == y)
    n_tie = n_bits = 0
    if mode == "int8":
        # the acoustic halves agree to the last bits (|d ln p| <= 4e-6, asserted below), so a transcript may only differ (a) where the reference
        # DECODER's own choice is implementation-defined: a (score, character) tie across the beam boundary -- the restatement must
        # show one and agree with the engine; or (b) where the last-bit difference of the two emission matrices itself decides a near-tie: then
        # the SEARCHES agree with each other on either input (restatement on the engine's emissions == the engine, restatement on the hybrid
        # path's emissions == the reference) and the whole difference is those last bits.  Anything else is a search or an acoustic error.
        labels, space = english
        P = port.Scorer(os.path.join(fix, "pruned_lm.scorer"))

        def port_text(p_):
            d = port.Decoder(labels, space, 500, P)
            d.next(p_)
            return d.boundary_ties(), port.decode_text(labels, d.decode(1)[0][1]).decode()
        for b in range(B):
            if texts[b] != ref_texts[b]:
                ties_g, text_g = port_text(got[b])
                assert texts[b] == text_g, (b, texts[b], text_g)                      # the engine's search == the restatement on the same emissions, always
                if ties_g > 0:
                    n_tie += 1
                    continue
                ties_w, text_w = port_text(want[b].astype(np.float32))
                assert text_w == ref_texts[b], (b, text_w, ref_texts[b], ties_w)      # no tie anywhere: the two searches agree on the hybrid path's emissions too
                n_bits += 1
    # ... and how sure the model is: mean probability of the best class (a random-init head is near-uniform: 1 / 29 = 0.034)
    top = float(got.max(axis=2).mean())
    line = {"engine_path": mode, "head_scale": head, "max_abs_dp": a_err, "max_abs_dlnp": l_err, "rms_dlnp": rms, "transcripts_equal": same, "differ_at_a_boundary_tie_of_the_search": n_tie, "differ_by_the_last_bits_of_the_emissions": n_bits, "of": B, "mean_top_probability": top}
    print("hybrid-int8 tolerance:", json.dumps(line))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "hybrid_tolerance_%s_head%d.json" % (mode, int(head))), "w") as f:
        json.dump(line, f)
    assert l_err <= LOG_TOL[mode] * (head if mode == "f16" else 1.0), line      # (f16: a bound on logits, it scales with the output layer)
    if head == 1.0:
        assert a_err <= ABS_TOL[mode], line
    assert same >= MIN_EQUAL[mode][head], line


# The SECOND checker (round-5 advisor): the same restatement with LOGISTIC / TANH evaluated in float32 as a TFLite build would (numpy's float32
# exp / tanh: 1-3 ulp from the correctly rounded value the engine and the first checker compute).  TFLite itself is not in the tree, so which
# float exp the reference's CPU path links is unknowable here; this is the SIZE of that unknown at the bench's shape: the quantised recurrence
# turns a last-bit difference of one activation into a different int8 somewhere in the next row now and then.  Stated: rms |d ln p| <=
# 1e-3 x head, max |d ln p| <= 2.5e-2 x head -- the same order as the f16 path's tolerance against the first checker, i.e. "which exp" costs
# about what "f16 instead of int8 activations" costs; transcripts at a random-init head are a coin toss per frame either way.
F32_RMS_TOL, F32_MAX_TOL = 1e-3, 2.5e-2
F32_MIN_EQUAL = {1.0: 24, 32.0: 60}


@pytest.mark.parametrize("head", [1.0, 32.0], ids=["random-init head", "head x 32 (peaky outputs)"])
def test_int8_path_against_the_float32_activation_variant(tmp_path, ref, english, fix, head):
    from oracle import am_hybrid
    B = 64
    w = synth.synth_weights(0, n_hidden=2048)
    w["layer_6/weights"] = (w["layer_6/weights"] * head).astype(np.float32)
    model = _model(tmp_path, w, "qf%d" % int(head), "int8")
    model.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    audio = list(synth.synth_audio_batch(B, 80000, seed=4242))
    got = np.stack(model.acousticProbs(audio))
    want = am_hybrid.utterance_probs_batch(audio, w, activations="f32")
    d = np.log(got) - np.log(want)
    l_err, rms, a_err = float(np.abs(d).max()), float(np.sqrt(np.mean(d ** 2))), float(np.abs(got - want).max())
    texts = model.sttBatch(audio)
    A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
    S = ref.Scorer(os.path.join(fix, "pruned_lm.scorer"), A)
    res = ref.decode_batch(want.astype(np.float64), [250] * B, A, 500, os.cpu_count() or 1, S)
    same = sum(1 for x, (_, tok) in zip(texts, res) if x == A.decode(tok).decode("utf-8", "replace"))
    line = {"engine_path": "int8", "checker": "hybrid restatement with float32 LOGISTIC / TANH (numpy)", "head_scale": head, "max_abs_dp": a_err, "max_abs_dlnp": l_err,
            "rms_dlnp": rms, "transcripts_equal": same, "of": B}
    print("hybrid-int8 tolerance, second checker:", json.dumps(line))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "hybrid_tolerance_int8_f32act_head%d.json" % int(head)), "w") as f:
        json.dump(line, f)
    assert rms <= F32_RMS_TOL * head and l_err <= F32_MAX_TOL * head, line
    assert same >= F32_MIN_EQUAL[head], line
