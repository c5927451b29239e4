"""Pins the C port (oracle/stt_port.c) against the REAL reference decoder (oracle/_ref, built from /root/reference) on
seeded random cases beyond the committed goldens: uniform-ish and peaky emissions, both cut-off mechanisms, scorer on/off,
word and byte mode, hot words.  Emissions are continuous (no exact probability ties), so the orders the reference leaves
to libstdc++ do not come into play, and only the top results are compared (a prefix that never
received a finite probability makes the reference dereference a null timestep node).  CPU only; skipped where oracle/_ref
is not built."""
import os

import numpy as np

from conftest import canon
from stt_amd import synth


def test_port_equals_reference_decoder_on_random_cases(port, ref, fix):
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rigs = {}
    for mode in ("word", "bytes"):
        if mode == "word":
            labels, space = port.parse_alphabet_file(os.path.join(fix, "alphabet.txt"))
            A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
            sp = os.path.join(fix, "pruned_lm.scorer")
        else:
            labels, space = port.utf8_alphabet()
            A = ref.Alphabet(None)
            sp = os.path.join(fix, "pruned_lm.bytes.scorer")
        rigs[mode] = (labels, space, A, port.Scorer(sp), ref.Scorer(sp, A))
    rng = np.random.RandomState(31337)
    for case in range(200):
        mode = "word" if case % 4 else "bytes"
        labels, space, A, P, S = rigs[mode]
        C = len(labels) + 1
        lm = bool(rng.randint(2))
        beam = int(rng.choice([4, 16, 50, 100] if mode == "word" else [8, 32]))
        T = int(rng.randint(6, 40 if mode == "word" else 20))
        cp, ctn = [(1.0, 40), (0.999, 40), (0.95, 40), (1.0, 8)][int(rng.randint(4))]
        if mode == "bytes" and not lm and cp == 1.0 and ctn >= 40:
            cp = 0.999
        hot = {}
        if lm and mode == "word" and rng.rand() < 0.3:
            hot = {str(rng.choice(vocab)): float(rng.choice([-2.0, 4.0]))}
        if rng.rand() < 0.5:
            x = rng.randn(T, C) * rng.choice([0.5, 1.5])
            p = np.exp(x - x.max(1, keepdims=True)); p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        else:
            sent = " ".join(rng.choice(vocab, size=rng.randint(1, 4)))
            lab = [b - 1 for b in sent.encode()] if mode == "bytes" else [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
            p = synth.peaky_emissions(lab, T, C, C - 1, seed=int(rng.randint(1 << 30)), noise=float(rng.choice([0.05, 0.5])), lead=2)
        o = port.Decoder(labels, space, beam, P if lm else None, cp, ctn, hot or None)
        r = ref.Decoder(A, beam, S if lm else None, cp, ctn, hot or None)
        for k in range(0, T, 16):
            o.next(p[k:k + 16]); r.next(p[k:k + 16])
        n = 2
        assert canon(o.decode(n)) == canon(r.decode(n)), (case, mode, lm, beam, T, cp, ctn, hot)


def test_reference_order_restatement_equals_the_reference_on_tie_cases(port, ref, fix):
    """stt_port.c Part D: the pointer trie + libstdc++'s nth_element / partial_sort restated.  Emissions built to PRODUCE ties -- rows drawn from
    a handful of values, so that different prefixes collect exactly equal float scores -- where the flat restatement (and the kernels,
    which share its rule) may keep another member of a tied group than the reference: here every case must equal the compiled reference,
    the whole vector of prefixes after every step (score, blank / non-blank probability, last label, length, IN ORDER) and the results."""
    labels, space = port.parse_alphabet_file(os.path.join(fix, "alphabet.txt"))
    A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
    sp = os.path.join(fix, "pruned_lm.scorer")
    P, S = port.Scorer(sp), ref.Scorer(sp, A)
    rng = np.random.RandomState(4242)
    C = len(labels) + 1
    n_tie_cases = n_flat_differs = 0
    for case in range(60):
        lm = bool(case % 2)
        beam = int(rng.choice([3, 8, 20, 64]))
        T = int(rng.randint(8, 40))
        levels = rng.choice([0.02, 0.05, 0.05, 0.2, 0.6], size=(T, C)).astype(np.float32)       # few distinct values: exact ties
        p = (levels / levels.sum(1, keepdims=True)).astype(np.float32)
        o = port.Decoder(labels, space, beam, P if lm else None, reference_order=True)
        f = port.Decoder(labels, space, beam, P if lm else None)
        r = ref.Decoder(A, beam, S if lm else None)
        for t in range(T):
            o.next(p[t:t + 1]); r.next(p[t:t + 1]); f.next(p[t:t + 1])
            a, b = o.raw_beam(), r.raw_beam()
            assert len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a, b)), (case, t)
        assert canon(o.decode(2)) == canon(r.decode(2)), case
        n_tie_cases += 1 if f.boundary_ties() else 0
        n_flat_differs += 1 if canon(f.decode(2)) != canon(r.decode(2)) else 0
    assert n_tie_cases >= 10, n_tie_cases          # (the point of the construction)
    print("tie cases %d of 60, flat restatement differs from the reference in %d" % (n_tie_cases, n_flat_differs))
