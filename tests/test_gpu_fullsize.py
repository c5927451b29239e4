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
    a_err, l_err = np.abs(got - want).max(), np.abs(np.log(got) - np.log(want)).max()
    print("fullsize acoustic: max |dp| %.3e  max |dlnp| %.3e" % (a_err, l_err))
    assert a_err < 1e-4 and l_err < 2e-3, (a_err, l_err)          # the stated tolerance (tests/test_gpu_benchshape.py)


def test_fullsize_huge_vocab_scorer_against_the_real_reference_decoder(big, ref, port, english, fix, tmp_path):
    """configs[1] with a scorer of the bench's kind (synthetic, 50 k words here to keep the test short, order 5, quantised array
    trie written by stt_amd/tools): GPU transcripts == the REAL reference decoder (oracle/_ref) on the GPU's own emissions."""
    from stt_amd import scorertools
    model, w = big
    lm, vocab, pkg = str(tmp_path / "lm.binary"), str(tmp_path / "vocab.txt"), str(tmp_path / "synth50k.scorer")
    scorertools.synth_lm(lm, vocab, words=50000, order=5, seed=11, avg={2: 12, 3: 1.0, 4: 0.6, 5: 0.5})
    scorertools.generate_scorer_package(lm, vocab, pkg, alphabet=os.path.join(fix, "alphabet.txt"),
                                        default_alpha=0.931289039105002, default_beta=1.1834137581510284)
    model.enableExternalScorer(pkg)
    try:
        audio = [synth.synth_audio(80000, seed=2000 + i) for i in range(64)]
        got = model.sttBatch(audio)
        pick = (0, 31, 63)
        probs = model.acousticProbs([audio[i] for i in pick])
        A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
        S = ref.Scorer(pkg, A)
        for k, i in enumerate(pick):
            d = ref.Decoder(A, 500, S)
            d.next(probs[k].astype(np.float64))
            conf, tok, ts = d.decode(1)[0]
            assert A.decode(tok).decode() == got[i], i
            md = model.sttWithMetadata(audio[i], 1)["transcripts"][0]
            assert md["confidence"] == conf and [t[1] for t in md["tokens"]] == [int(x) for x in ts], i
    finally:
        model.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
