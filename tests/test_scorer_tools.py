"""Scorer packaging tool (stt_amd/tools/scorer_tools.cpp): `package` against the reference's own package of the same LM
and vocabulary, `synth-lm` against the real KenLM reader (oracle/_ref) and the C port."""
import os

import numpy as np
import pytest

from conftest import canon
from stt_amd import scorertools, synth


def _emissions(vocab, rng, seed, noise, n_words=(2, 6)):
    sent = " ".join(rng.choice(vocab, size=rng.randint(*n_words)))
    lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
    return synth.peaky_emissions(lab, 30 + 5 * len(lab), 29, 28, seed=seed, noise=noise)


def test_package_matches_reference_package(port, english, fix, tmp_path):
    """Same KenLM blob + same vocabulary -> a package that decodes exactly like the one the reference tool wrote."""
    labels, space = english
    orig = os.path.join(fix, "pruned_lm.scorer")
    data = open(orig, "rb").read()
    P0 = port.Scorer(orig)
    lm_end = port.lib().port_scorer_lm_end(P0.h)
    lm = tmp_path / "lm.binary"
    lm.write_bytes(data[:lm_end])
    out = str(tmp_path / "re.scorer")
    scorertools.generate_scorer_package(str(lm), os.path.join(fix, "vocab.pruned.txt"), out, alphabet=os.path.join(fix, "alphabet.txt"),
                                        default_alpha=0.75, default_beta=1.85)
    P1 = port.Scorer(out)
    assert (P1.utf8, P1.order, P1.model_type) == (False, 4, 5)
    assert abs(P1.alpha - 0.75) < 1e-9 and abs(P1.beta - 1.85) < 1e-6
    s0, a0, f0 = P0.fst(); s1, a1, f1 = P1.fst()
    assert len(f1) == len(f0) and a1.shape == a0.shape      # both minimal: same number of states and arcs
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(11)
    for it in range(4):
        p = _emissions(vocab, rng, 40 + it, [0.02, 0.2, 0.6, 1.0][it])
        for beam in (30, 200):
            d0 = port.Decoder(labels, space, beam, P0); d1 = port.Decoder(labels, space, beam, P1)
            d0.next(p); d1.next(p)
            assert canon(d0.decode(beam)) == canon(d1.decode(beam)), (it, beam)


def test_package_bytes_mode(port, fix, tmp_path):
    orig = os.path.join(fix, "pruned_lm.bytes.scorer")
    data = open(orig, "rb").read()
    P0 = port.Scorer(orig)
    lm_end = port.lib().port_scorer_lm_end(P0.h)
    lm = tmp_path / "lm.binary"; lm.write_bytes(data[:lm_end])
    _, a0, _ = P0.fst()
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(chr(int(l)) for l in sorted(set(a0[:, 1]))) + "\n")   # the single-byte words of the fixture
    out = str(tmp_path / "re.scorer")
    scorertools.generate_scorer_package(str(lm), str(vocab), out, force_bytes_output_mode=True, default_alpha=P0.alpha, default_beta=P0.beta)
    P1 = port.Scorer(out)
    assert P1.utf8 and P1.order == P0.order
    s1, a1, f1 = P1.fst()
    assert sorted(map(tuple, a1.tolist())) == sorted(map(tuple, a0.tolist())) and len(f1) == 2


@pytest.fixture(scope="module")
def synth_scorer(tmp_path_factory, fix):
    d = tmp_path_factory.mktemp("synthlm")
    lm, vocab, pkg = str(d / "lm.binary"), str(d / "vocab.txt"), str(d / "synth.scorer")
    scorertools.synth_lm(lm, vocab, words=3000, order=5, seed=5)
    scorertools.generate_scorer_package(lm, vocab, pkg, alphabet=os.path.join(fix, "alphabet.txt"), default_alpha=0.9, default_beta=1.2)
    return pkg, open(vocab).read().split()


def test_synth_lm_is_a_valid_kenlm_trie(port, synth_scorer):
    pkg, vocab = synth_scorer
    P = port.Scorer(pkg)
    assert (P.utf8, P.order, P.model_type) == (False, 5, 5)          # QUANT_ARRAY_TRIE, like `build_binary -a 255 -q 8 trie`
    assert P.index("<s>") > 0 and P.index(vocab[0]) > 0 and P.index("definitelynotaword") == 0
    v = P.log_cond_prob([vocab[0]], True)
    assert np.isfinite(v) and v < 0
    assert P.log_cond_prob(["definitelynotaword"]) == -1000.0


def test_synth_lm_against_real_kenlm_reader(port, ref, english, synth_scorer, fix):
    """The writer is only trusted because the *reference* (real KenLM + OpenFst) reads its files and agrees with the port."""
    pkg, vocab = synth_scorer
    labels, space = english
    A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
    S = ref.Scorer(pkg, A)
    P = port.Scorer(pkg)
    rng = np.random.RandomState(0)
    for it in range(1500):
        n = rng.randint(1, 7)
        ws = [vocab[min(int(rng.zipf(1.3)) - 1, len(vocab) - 1)] if rng.rand() < 0.7 else vocab[rng.randint(len(vocab))] for _ in range(n)]
        if rng.rand() < 0.05:
            ws[rng.randint(n)] = "zzzzqq"
        bos = bool(rng.rand() < 0.5)
        assert S.log_cond_prob(ws, bos) == P.log_cond_prob(ws, bos), (ws, bos)
    for it in range(3):
        p = _emissions(vocab, rng, 700 + it, [0.05, 0.4, 1.0][it])
        for beam in (50, 300):
            dr = ref.Decoder(A, beam, S); dp = port.Decoder(labels, space, beam, P)
            dr.next(p.astype(np.float64)); dp.next(p)
            assert canon(dr.decode(beam)) == canon(dp.decode(beam)), (it, beam)


def test_synth_code_point_lm_packaged_in_bytes_mode(port, ref, tmp_path):
    """`synth-lm --codepoints`: the units are three-byte code points (the scorer SURVEY.md 8d Config 5 names: a code-point level LM in
    bytes-output mode).  The REAL KenLM + OpenFst read the package, agree with the port on random n-grams, and the two decoders agree
    on byte-alphabet emissions that spell units of the vocabulary."""
    lm, vocab, pkg = str(tmp_path / "cp.binary"), str(tmp_path / "cp.vocab"), str(tmp_path / "cp.scorer")
    scorertools.synth_lm(lm, vocab, words=500, order=4, seed=3, avg={2: 40, 3: 2.0, 4: 1.0}, codepoints=True)
    scorertools.generate_scorer_package(lm, vocab, pkg, force_bytes_output_mode=True, default_alpha=0.9, default_beta=1.1)
    units = open(vocab, encoding="utf-8").read().split()
    assert len(units) == 500 and all(len(u) == 1 and len(u.encode("utf-8")) == 3 for u in units)
    A = ref.Alphabet(None)
    S = ref.Scorer(pkg, A)
    P = port.Scorer(pkg)
    assert S.utf8 and S.order == 4 and (P.utf8, P.order) == (True, 4)
    rng = np.random.RandomState(1)
    for it in range(600):
        ws = [units[min(len(units) - 1, int(rng.zipf(1.4)) - 1)] for _ in range(rng.randint(1, 6))]
        if rng.rand() < 0.05:
            ws[0] = "Ж"                                   # a two-byte code point the model does not know
        bos = bool(it & 1)
        assert S.log_cond_prob(ws, bos) == P.log_cond_prob(ws, bos), (ws, bos)
    ulabels, uspace = port.utf8_alphabet()
    for it, noise in enumerate((0.02, 0.3)):
        text = "".join(rng.choice(units, size=6)).encode("utf-8")
        lab = [b - 1 for b in text]                            # UTF8Alphabet: label = byte - 1
        p = synth.peaky_emissions(lab, 30 + 5 * len(lab), 256, 255, seed=60 + it, noise=noise)
        for beam in (40, 200):
            dr = ref.Decoder(A, beam, S); dp = port.Decoder(ulabels, uspace, beam, P)
            dr.next(p.astype(np.float64)); dp.next(p)
            assert canon(dr.decode(min(beam, 20))) == canon(dp.decode(min(beam, 20))), (it, beam)
