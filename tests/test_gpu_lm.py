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


def test_device_bigram_blocks_equal_the_index_and_the_port(port, tmp_path):
    """Round 6: FullScore of a code point through the bigram blocks ON THE DEVICE (ctc.hip: lm_full_score_blocks, what the code-point search step
    runs; STTX_TestLm mode 4) against the hashed index on the device (mode 2) and the C port's trie walk, on a synthetic code-point model (1500
    three-byte units, order 5): random sequences, chains grown along the model so that orders 3 - 5 are reached (the hand-over from a block's
    record to the index), and sweeps of one context over the 64 code points of a block -- floats and matched lengths equal."""
    from stt_amd import native, scorertools
    lm_path, vocab_path = str(tmp_path / "cp.binary"), str(tmp_path / "cp.vocab")
    scorertools.synth_lm(lm_path, vocab_path, words=1500, order=5, seed=9, avg={2: 40, 3: 2.0, 4: 1.0, 5: 0.7}, codepoints=True)
    lm, units = open(lm_path, "rb").read(), open(vocab_path, encoding="utf-8").read().split()
    Pl = port.Scorer(data=lm, lm_only=True)
    rng = np.random.RandomState(21)

    def check(words, bos):
        want_p, want_l = Pl.score(words, bos)
        for mode in (2, 4):
            got_p, got_l = native.lm_score(lm, words, bos, mode=mode)
            assert np.array_equal(got_p, want_p) and np.array_equal(got_l, want_l), (mode, words, bos)
        return want_l

    for it in range(40):
        words = [str(units[i]) for i in rng.randint(0, len(units), int(rng.randint(1, 10)))]
        check(words, bool(it & 1))
    deep = 0
    for it in range(8):
        words = [str(units[int(rng.randint(len(units)))])]
        for _ in range(6):
            cand = [str(units[i]) for i in rng.randint(0, len(units), 300)]
            words.append(max(cand, key=lambda u: int(Pl.score(words + [u], True)[1][-1])))
        deep += int((check(words, True) >= 3).sum())
    assert deep > 6
    found2 = 0
    for it in range(3):
        ctx = [str(units[i]) for i in rng.randint(0, len(units), int(rng.randint(1, 4)))]
        nxt = max((str(units[i]) for i in rng.randint(0, len(units), 300)), key=lambda u: int(Pl.score(ctx + [u], True)[1][-1]))
        base = ord(nxt) & ~63
        for cp in range(base, base + 64):
            if chr(cp) in units:                                          # (mode 4 answers for code points of the vocabulary; the step sends the others through the index)
                found2 += int(check(ctx + [chr(cp)], True)[-1] >= 2)
    assert found2 >= 3
