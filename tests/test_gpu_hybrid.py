"""The stated tolerance against the reference's *TFLite CPU path*: released models are dynamic-range quantised and that path quantises
the activations of every FULLY_CONNECTED to int8 per call (oracle/am_hybrid.py restates the published kernel); the engine de-quantises
the same int8 weights to f16 and keeps activations in f16 / f32.  At the bench's shape (n_hidden 2048, 64 x 5 s, beam 500) this
measures how far the two are apart on the softmax outputs and what that does to transcripts.  Needs a MI355X."""
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
ABS_TOL, LOG_TOL = 1e-3, 1.25e-2
# transcripts equal to the reference decoder's on the hybrid path's probabilities, of 64: a random-init head is a coin toss per frame
# (mean top probability 0.04) and any perturbation re-routes the beam; the more a model commits, the fewer transcripts move
MIN_EQUAL = {1.0: 0, 8.0: 48, 32.0: 62}      # measured: 4, 57, 64 (profiles/r04_hybrid_tolerance.json)


def _model(tmp, w, name, beam=500):
    from stt_amd import Model
    path = str(tmp / (name + ".tflite"))
    tflitefile.write_tflite(path, w, synth.ENGLISH_LABELS, quantize=True, beam_width=beam)
    return Model(path)


@pytest.mark.parametrize("head", [1.0, 8.0, 32.0], ids=["random-init head", "head x 8", "head x 32 (peaky outputs)"])
def test_engine_against_the_hybrid_int8_path(tmp_path, ref, fix, head):
    from oracle import am_hybrid
    B = 64
    w = synth.synth_weights(0, n_hidden=2048)
    w["layer_6/weights"] = (w["layer_6/weights"] * head).astype(np.float32)
    model = _model(tmp_path, w, "q%d" % int(head))
    model.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    audio = list(synth.synth_audio_batch(B, 80000, seed=4242))
    got = np.stack(model.acousticProbs(audio))                                       # [B][250][29], the HIP path on the de-quantised weights
    want = am_hybrid.utterance_probs_batch(audio, w)                                 # the hybrid int8 kernels
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
    # ... and how sure the model is: mean probability of the best class (a random-init head is near-uniform: 1 / 29 = 0.034)
    top = float(got.max(axis=2).mean())
    line = {"head_scale": head, "max_abs_dp": a_err, "max_abs_dlnp": l_err, "rms_dlnp": rms, "transcripts_equal": same, "of": B, "mean_top_probability": top}
    print("hybrid-int8 tolerance:", json.dumps(line))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "hybrid_tolerance_head%d.json" % int(head)), "w") as f:
        json.dump(line, f)
    assert l_err <= LOG_TOL * head, line                          # the bound is one on logits: it scales with the output layer
    if head == 1.0:
        assert a_err <= ABS_TOL, line
    assert same >= MIN_EQUAL[head], line
