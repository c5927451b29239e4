"""stt_amd/csrc/scorer_dev.cpp: the code-point bigram blocks (context x 64 consecutive code points -> one table entry + one slice; the
layout DESIGN.md 9.3 / 7.1 argues the code-point search step needs) on the host: FullScore through them (STTX_TestLm mode 3) must
give the floats and matched lengths of the hashed index (mode 0) and of the C port's trie walk, on a synthetic code-point LM written by
stt_amd/tools (three-byte units, order 5) -- sequences of model units, sibling sweeps over a 64-unit block, unknown units.  No GPU."""
import os

import numpy as np
import pytest

from stt_amd import native, scorertools


@pytest.fixture(scope="module")
def cp_lm(tmp_path_factory):
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libstt.so not built")
    d = tmp_path_factory.mktemp("cpb")
    lm, vocab = str(d / "cp.binary"), str(d / "cp.vocab")
    scorertools.synth_lm(lm, vocab, words=1500, order=5, seed=9, avg={2: 40, 3: 2.0, 4: 1.0, 5: 0.7}, codepoints=True)
    return open(lm, "rb").read(), open(vocab, encoding="utf-8").read().split()


def _both(lm, words, bos):
    native.set_tuning("cp_blocks", 1)
    try:
        return native.lm_score(lm, words, bos, mode=3), native.lm_score(lm, words, bos, mode=0)
    finally:
        native.set_tuning("cp_blocks", 1)     # (the default since round 6)


def test_blocks_equal_the_index_and_the_port_trie_walk(cp_lm, port):
    lm, units = cp_lm
    Pl = port.Scorer(data=lm, lm_only=True)
    rng = np.random.RandomState(7)
    deep = 0
    for it in range(100):
        n = int(rng.randint(1, 10))
        words = [str(units[i]) for i in rng.randint(0, len(units), n)]
        if it % 5 == 0:
            words[int(rng.randint(0, n))] = chr(0x3042 + it)          # a code point the model does not know: <unk>, then <unk> in the history
        bos = bool(it & 1)
        (bp, bl), (ip, il) = _both(lm, words, bos)
        assert np.array_equal(bp, ip) and np.array_equal(bl, il), (words, bos)
        want_p, want_l = Pl.score(words, bos)
        assert np.array_equal(bp, want_p) and np.array_equal(bl, want_l), (words, bos)
    # chains that FOLLOW the model (random units almost never continue a stored bigram): grown greedily with the port -- the next unit is
    # one that gives the longest match -- so that orders 3 .. 5 are reached and the hand-over from the block's record to the index is exercised
    for it in range(12):
        words = [str(units[int(rng.randint(len(units)))])]
        for _ in range(7):
            cand = [str(units[i]) for i in rng.randint(0, len(units), 400)]
            best = max(cand, key=lambda u: int(Pl.score(words + [u], True)[1][-1]))
            words.append(best)
        (bp, bl), (ip, il) = _both(lm, words, True)
        want_p, want_l = Pl.score(words, True)
        assert np.array_equal(bp, ip) and np.array_equal(bl, il), (words,)
        assert np.array_equal(bp, want_p) and np.array_equal(bl, want_l), (words,)
        deep += int((bl >= 3).sum())
    assert deep > 10                                                   # (matches beyond the bigram)


def test_sibling_sweep_over_a_block(cp_lm, port):
    """What the search step asks: ONE context, the 64 code points of a block (first two bytes fixed) -- present and absent bigrams."""
    lm, units = cp_lm
    Pl = port.Scorer(data=lm, lm_only=True)
    rng = np.random.RandomState(8)
    found2 = 0
    for it in range(8):                                                  # (every call parses the model and builds its tables: ~20 ms)
        ctx = [str(units[i]) for i in rng.randint(0, len(units), int(rng.randint(1, 4)))]
        nxt = max((str(units[i]) for i in rng.randint(0, len(units), 400)), key=lambda u: int(Pl.score(ctx + [u], True)[1][-1]))
        base = ord(nxt) & ~63                                             # a block that holds at least one stored continuation of the context
        for cp in range(base, base + 64):
            words = ctx + [chr(cp)]
            (bp, bl), (ip, il) = _both(lm, words, True)
            assert np.array_equal(bp, ip) and np.array_equal(bl, il), (words,)
            want_p, want_l = Pl.score(words, True)
            assert np.array_equal(bp, want_p) and np.array_equal(bl, want_l), (words,)
            found2 += int(bl[-1] >= 2)
    assert found2 >= 8


def test_blocks_need_the_tunable(cp_lm):
    lm, units = cp_lm
    native.set_tuning("cp_blocks", 0)
    try:
        with pytest.raises(RuntimeError):
            native.lm_score(lm, [str(units[0])], True, mode=3)           # cp_blocks = 0: not built
    finally:
        native.set_tuning("cp_blocks", 1)
