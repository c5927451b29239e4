"""KenLM PROBING binaries on the device (ctc.hip: kenlm_full_score_probing): FullScore against the real KenLM's answers, and the whole
decoder with a scorer package around a probing binary against the REAL reference decoder (oracle/_ref loads it through KenLM itself).
Needs a MI355X."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, canon
from stt_amd import modelfile, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flavour", ["probing", "probing20"])
def test_device_fullscore_on_a_probing_binary_equals_kenlm(fix, flavour):
    from stt_amd import native
    with open(os.path.join(GOLD, "kenlm_probing_golden.json")) as f:
        rows = json.load(f)[flavour]
    lm = open(os.path.join(fix, "kenlm_test_%s.bin" % flavour), "rb").read()
    for row in rows:
        pr, ln = native.lm_score(lm, row["words"], row["bos"], mode=1)
        assert [float(x) for x in pr] == row["probs"], (flavour, row["words"])
        assert [int(x) for x in ln] == row["lens"], (flavour, row["words"])


@pytest.mark.parametrize("beam", [16, 100, 500])
def test_decoder_with_a_probing_scorer_equals_the_reference_decoder(tmp_path, ref, fix, beam):
    from stt_amd import Model
    pkg = os.path.join(fix, "probing_lm.scorer")
    vocab = open(os.path.join(fix, "probing_lm.vocab.txt")).read().split()
    w = synth.synth_weights(3, n_hidden=128)
    path = str(tmp_path / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=beam)
    m = Model(path)
    m.enableExternalScorer(pkg)
    A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
    S = ref.Scorer(pkg, A)
    rng = np.random.RandomState(17 + beam)
    for it in range(6):
        sent = " ".join(rng.choice(vocab, size=rng.randint(2, 8)))
        lab = [0 if ch == " " else ord(ch) - ord("a") + 1 for ch in sent]
        p = synth.peaky_emissions(lab, 24 + 5 * len(lab), 29, 28, seed=90 + it, noise=[0.02, 0.05, 0.1, 0.3, 0.6, 1.0][it])
        hot = {vocab[2]: 3.0} if it == 3 else None
        dr = ref.Decoder(A, beam, S, hot_words=hot)
        dr.next(p.astype(np.float64))
        if hot:
            for wd, b in hot.items():
                m.addHotWord(wd, b)
        d = m.createDecoder(1, beam)
        for k in range(0, len(p), 16):            # fed in chunks, as a stream is
            d.next(p[k:k + 16])
        if hot:
            m.clearHotWords()
        n = min(beam, 10)
        got, want = d.decode(n)[0], dr.decode(n)
        assert d.stats()["error"] == 0
        assert canon(got) == canon(want), (it, sent)
        if it == 0:
            text = A.decode(got[0][1]).decode()
            assert text == sent, (text, sent)
