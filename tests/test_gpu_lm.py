"""KenLM FullScore on the device -- the trie walk (ctc.hip: kenlm_full_score) and the hashed n-gram index as the search step
reads it (ctc.hip: lm_full_score_indexed) -- against the answers of the real KenLM (tests/golden/kenlm_golden.json,
written by the reference library) on all four trie flavours: plain (model type 2), quantised (3), array-compressed (4)
and quantised + array (5); and against the C port on random word sequences over the shipped scorer's model."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flavour", ["trie", "array", "quant", "qarray"])
@pytest.mark.parametrize("mode", [1, 2], ids=["trie-walk", "index"])
def test_device_lm_equals_kenlm_known_answers(fix, flavour, mode):
    from stt_amd import native
    with open(os.path.join(GOLD, "kenlm_golden.json")) as f:
        rows = json.load(f)["kenlm"][flavour]
    lm = open(os.path.join(fix, "kenlm_test_%s.bin" % flavour), "rb").read()
    for row in rows:
        pr, ln = native.lm_score(lm, row["words"], row["bos"], mode=mode)
        assert [float(x) for x in pr] == row["probs"], (flavour, mode, row["words"])
        assert [int(x) for x in ln] == row["lens"], (flavour, mode, row["words"])


def test_device_lm_random_sequences_equal_port(port, fix):
    from stt_amd import native
    data = open(os.path.join(fix, "pruned_lm.scorer"), "rb").read()
    lm = data[:port.lib().port_scorer_lm_end(port.Scorer(data=data).h)]
    Pl = port.Scorer(data=lm, lm_only=True)
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(11)
    for it in range(60):
        n = rng.randint(1, 9)
        words = [vocab[i] for i in rng.randint(0, len(vocab), n)]
        if it % 5 == 0:
            words[rng.randint(0, n)] = "zzzzqq"
        if it % 3 == 0:
            j = rng.randint(0, len(vocab) - 8); words = vocab[j:j + n]
        bos = bool(it & 1)
        want_p, want_l = Pl.score(words, bos)
        for mode in (1, 2):
            got_p, got_l = native.lm_score(lm, words, bos, mode=mode)
            assert np.array_equal(got_p, want_p) and np.array_equal(got_l, want_l), (mode, words, bos)
