"""stt_amd/csrc/scorer_dev.cpp: the dictionary of a word-mode scorer package unfolded into the tree of its word prefixes (what the
search kernel follows without arc reads) accepts the same label sequences, and ends words at the same places, as the minimised
automaton the package holds (path_trie.cpp:54-90 only asks those two questions).  Host only."""
import ctypes as C
import os

import numpy as np
import pytest

from stt_amd import native


def _walk(blob, space, seqs, tree_mb):
    L = native.lib()
    native.set_tuning("dict_tree_mb", tree_mb)
    try:
        n, ln = seqs.shape
        out = np.zeros((n, ln), dtype=np.int32)
        rc = L.STTX_TestDictionaryWalk(blob, len(blob), space, seqs.ctypes.data, n, ln, out.ctypes.data)
        assert rc == 0, rc
        return out
    finally:
        native.set_tuning("dict_tree_mb", 2048)


def _labels(word):
    return [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in word]


def test_tree_unfolding_accepts_the_same_language(fix):
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libstt.so not built")
    blob = open(os.path.join(fix, "pruned_lm.scorer"), "rb").read()
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(5)
    ln = 24
    seqs = []
    for w in vocab[:400]:                                           # words, then a second word behind the space
        seqs.append(_labels(w + " " + str(rng.choice(vocab)))[:ln])
    for _ in range(400):                                            # corrupted words and random strings
        w = list(str(rng.choice(vocab)))
        w[int(rng.randint(len(w)))] = chr(ord("a") + int(rng.randint(26)))
        seqs.append(_labels("".join(w) + " ")[:ln])
        seqs.append([int(x) for x in rng.randint(0, 28, size=int(rng.randint(1, ln)))])
    arr = np.full((len(seqs), ln), -1, dtype=np.int32)
    for i, sq in enumerate(seqs):
        arr[i, :len(sq)] = sq
    space = 0
    a = _walk(blob, space, arr, 0)                                  # the package's automaton, repacked
    t = _walk(blob, space, arr, 2048)                               # unfolded
    assert (a[a >= 0] & 2).sum() == 0 and (t[t >= 0] & 2).all(), "the tunable selects the form"
    assert np.array_equal(a >= 0, t >= 0), "same label sequences accepted"
    assert np.array_equal(a[a >= 0] & 1, t[t >= 0] & 1), "words end at the same places"
    assert (a[:400] >= 0).sum() > 400 * 3                            # (the vocabulary's own words are walked to their ends)
    assert (a[400:] < 0).any()


def test_tree_cap_falls_back_to_the_automaton(fix):
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libstt.so not built")
    blob = open(os.path.join(fix, "pruned_lm.scorer"), "rb").read()
    arr = np.array([_labels("the ") + [-1] * 4], dtype=np.int32)
    native.set_tuning("dict_tree_mb", 2048)
    big = _walk(blob, 0, arr, 2048)
    tiny = _walk(blob, 0, arr, 1)                                    # 1 MiB / 21 B = ~50 k nodes
    assert np.array_equal(big >= 0, tiny >= 0)
