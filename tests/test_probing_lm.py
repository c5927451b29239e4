"""KenLM PROBING binaries (model type 0, lm/search_hashed.hh): the reference loads any KenLM binary (scorer.cpp:119-126, RecognizeBinary +
LoadVirtual), round 5 refused everything but tries.  Host side, no GPU: the layout parsed by parse_scorer() and FullScore through the file's
own hash tables (STTX_TestLm mode 0) against the REAL KenLM's answers (tests/golden/kenlm_probing_golden.json, written by oracle/_ref on
binaries made by the vendored build_binary: tests/golden/make_probing_golden.py), on two probing multipliers; and a scorer PACKAGE around
a probing binary (the reference's own packaging) parsed down to its dictionary."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD


@pytest.fixture(scope="module")
def native():
    from stt_amd import native as n
    return n


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLD, "kenlm_probing_golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("flavour", ["probing", "probing20"])
def test_host_fullscore_on_a_probing_binary_equals_kenlm(native, fix, golden, flavour):
    lm = open(os.path.join(fix, "kenlm_test_%s.bin" % flavour), "rb").read()
    for row in golden[flavour]:
        pr, ln = native.lm_score(lm, row["words"], row["bos"], mode=0)
        assert [float(x) for x in pr] == row["probs"], (flavour, row["words"])
        assert [int(x) for x in ln] == row["lens"], (flavour, row["words"])


def test_a_package_around_a_probing_binary_parses_and_scores_like_the_reference(native, fix, golden):
    """get_log_cond_prob (scorer.cpp:308-338): OOV -> -1000, else FullScore's log10 probability of the last word / log10(e) in float."""
    pkg = open(os.path.join(fix, "probing_lm.scorer"), "rb").read()
    for row in golden["scorer"]:
        pr, ln = native.lm_score(pkg, row["words"], row["bos"], mode=0)      # (lm_only parse: the package's trailer is simply behind the model)
        if row["value"] == -1000.0:
            assert int(ln[-1]) >= 1 and row["words"][-1] not in golden["vocabulary"]
            continue
        want = np.float32(row["value"])
        got = np.float32(np.float32(pr[-1]) / np.float32(0.4342944819))
        assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)), (row, float(got))
    # the whole package: header, dictionary automaton, label bitmaps (STTX_TestDictionaryWalk parses it as STT_EnableExternalScorer does)
    import ctypes as C
    labs = np.full((2, 8), -1, dtype=np.int32)
    for i, w in enumerate(["looking", "zq"]):
        for k, ch in enumerate(w):
            labs[i, k] = ord(ch) - ord("a") + 1
    out = np.zeros((2, 8), dtype=np.int32)
    rc = native.lib().STTX_TestDictionaryWalk(pkg, len(pkg), 0, labs.ctypes.data, 2, 8, out.ctypes.data)
    assert rc == 0
    assert out[0, 6] >= 1 and (out[0, 6] & 1) == 1        # "looking" is a word: a space arc follows its last letter
    assert out[1, 1] == -1                                # "zq" leaves the dictionary


def test_rest_probing_and_garbage_are_refused(native, fix):
    lm = bytearray(open(os.path.join(fix, "kenlm_test_probing.bin"), "rb").read())
    lm[88 + 8] = 1                                        # FixedWidthParameters.model_type = REST_PROBING: three floats per record -- other offsets, nothing lines up
    with pytest.raises(RuntimeError):
        native.lm_score(bytes(lm), ["a"], True, mode=0)
    lm[88 + 8] = 0
    lm[88 + 4:88 + 8] = np.float32(0.5).tobytes()         # a probing multiplier below 1 cannot have been written
    with pytest.raises(RuntimeError):
        native.lm_score(bytes(lm), ["a"], True, mode=0)
