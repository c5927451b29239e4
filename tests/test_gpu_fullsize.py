"""BASELINE.json sizes (English geometry n_hidden=2048, beam 500, scorer, 5 s utterances): size-independent properties."""
import os

import numpy as np
import pytest

from stt_amd import modelfile, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(tmp_path_factory, fix):
    from stt_amd import Model
    w = synth.synth_weights(0, n_hidden=2048)
    path = str(tmp_path_factory.mktemp("big") / "english.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
    m = Model(path)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    return m, w


def test_fullsize_batch_properties(big, port, english, fix):
    model, w = big
    B = 64
    audio = [synth.synth_audio(80000, seed=1000 + i) for i in range(B)]
    r1 = model.sttBatch(audio)
    r2 = model.sttBatch(audio)
    assert r1 == r2                                                          # deterministic
    perm = np.random.RandomState(0).permutation(B)
    r3 = model.sttBatch([audio[i] for i in perm])
    assert [r3[k] for k in np.argsort(perm)] == r1                           # batch position does not matter
    for i in (0, 17, 63):
        assert model.stt(audio[i]) == r1[i]                                  # batch == one-shot
    # GPU probabilities -> oracle decoder == GPU transcript, on three utterances
    labels, space = english
    P = port.Scorer(os.path.join(fix, "pruned_lm.scorer"))
    probs = model.acousticProbs([audio[i] for i in (0, 17, 63)])
    for k, i in enumerate((0, 17, 63)):
        assert probs[k].shape == (250, 29)
        d = port.Decoder(labels, space, 500, P); d.next(probs[k])
        tok = d.decode(1)[0][1]
        assert b"".join(labels[t] for t in tok).decode() == r1[i], i


def test_fullsize_acoustic_tolerance(big):
    """One second of audio through the 2048-wide model vs the f64 oracle with f16-rounded weights/activations."""
    from oracle import am_ref
    model, w = big
    a = synth.synth_audio(16000, seed=9)
    got = model.acousticProbs([a])[0]
    want = am_ref.utterance_probs(a, w, weight_round=np.float16)
    assert np.abs(got - want).max() < 3e-3, np.abs(got - want).max()
