"""The hashed n-gram index (stt_amd/csrc/lmindex.h, built by parse_scorer() at scorer load) on the host: FullScore through
the index must give the floats and matched lengths of the real KenLM -- the committed answers of
tests/golden/kenlm_golden.json (written by the reference library, make_golden.py) on all four trie flavours, and the C
port's trie walk on random word sequences over the shipped scorer's model.  No GPU: STTX_TestLm mode 0."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD


@pytest.fixture(scope="module")
def native():
    from stt_amd import native as n
    if not os.path.exists(n.LIB_PATH):
        from stt_amd import build
        build.build(verbose=False)
    return n


@pytest.mark.parametrize("flavour", ["trie", "array", "quant", "qarray"])
def test_index_answers_equal_kenlm_known_answers(native, fix, flavour):
    with open(os.path.join(GOLD, "kenlm_golden.json")) as f:
        rows = json.load(f)["kenlm"][flavour]
    lm = open(os.path.join(fix, "kenlm_test_%s.bin" % flavour), "rb").read()
    for row in rows:
        pr, ln = native.lm_score(lm, row["words"], row["bos"], mode=0)
        assert [float(x) for x in pr] == row["probs"], (flavour, row["words"])
        assert [int(x) for x in ln] == row["lens"], (flavour, row["words"])


def test_index_equals_port_trie_walk_on_the_shipped_scorer(native, port, fix):
    data = open(os.path.join(fix, "pruned_lm.scorer"), "rb").read()
    P = port.Scorer(data=data)                      # order 4, quant-array-trie, 3451 words
    lm_end = port.lib().port_scorer_lm_end(P.h)
    lm = data[:lm_end]
    Pl = port.Scorer(data=lm, lm_only=True)
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(5)
    for it in range(300):
        n = rng.randint(1, 9)
        words = [vocab[i] for i in rng.randint(0, len(vocab), n)]
        if it % 7 == 0:
            words[rng.randint(0, n)] = "zzzzqq"      # out of vocabulary: <unk> as the new word and then in the history
        if it % 3 == 0:                              # a real corpus continuation: deeper matches
            j = rng.randint(0, max(1, len(vocab) - 8)); words = vocab[j:j + n]
        bos = bool(it & 1)
        want_p, want_l = Pl.score(words, bos)
        got_p, got_l = native.lm_score(lm, words, bos, mode=0)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_l, want_l), (words, bos, got_p, want_p, got_l, want_l)


def test_lm_only_parse_errors(native):
    with pytest.raises(RuntimeError, match="0x2006"):
        native.lm_score(b"not a kenlm file" * 20, ["a"], True, mode=0)
