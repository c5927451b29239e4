#!/usr/bin/env python
"""How often does the reference's own choice depend on libstdc++ (a (score, character) tie across the beam boundary)?  CPU only.

Three kinds of emissions at configs[1]'s shape (250 frames, 29 classes, beam 500, the pruned_lm.scorer fixture's vocabulary):
  peaky     what a trained model emits (blank ~0.9, labels held two frames, noise 0.02) -- sentences from vocab.pruned.txt
  noisy     the same with noise 0.2 (a poorly trained / out-of-domain model)
  uniform   softmax of Gaussian logits, sigma 0.5 (a random-init head: bench.py's headline workload is of this kind)
For every utterance: the flat restatement (the kernels' tie rule) counts the steps with a boundary tie; its transcript is compared with the
REAL reference decoder's (oracle/_ref) and with the reference-order restatement's (stt_port.c Part D), which must equal the reference always.

    python benchmarks/tie_rate.py [--utterances 2000] > profiles/r05_tie_rate.json
"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import port, ref      # noqa: E402
from stt_amd import synth         # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "fixtures")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=2000)
    ap.add_argument("--beam", type=int, default=500)
    a = ap.parse_args()
    labels, space = port.parse_alphabet_file(os.path.join(FIX, "alphabet.txt"))
    A = ref.Alphabet(os.path.join(FIX, "alphabet.txt"))
    sp = os.path.join(FIX, "pruned_lm.scorer")
    P, S = port.Scorer(sp), ref.Scorer(sp, A)
    vocab = open(os.path.join(FIX, "vocab.pruned.txt")).read().split()
    T, C = 250, 29

    def emissions(kind, i):
        rng = np.random.RandomState(1000003 * (1 + ["peaky", "noisy", "uniform"].index(kind)) + i)
        if kind == "uniform":
            x = rng.randn(T, C) * 0.5
            p = np.exp(x - x.max(1, keepdims=True))
            return (p / p.sum(1, keepdims=True)).astype(np.float32)
        sent = ""
        while len(sent) < 48:
            sent += (" " if sent else "") + str(rng.choice(vocab))
        lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent[:56]]
        return synth.peaky_emissions(lab, T, C, C - 1, seed=int(rng.randint(1 << 30)), noise=0.02 if kind == "peaky" else 0.2)

    def one(job):
        kind, i = job
        p = emissions(kind, i)
        f = port.Decoder(labels, space, a.beam, P)
        f.next(p)
        rf = f.decode(1)
        o = port.Decoder(labels, space, a.beam, P, reference_order=True)
        o.next(p)
        ro = o.decode(1)
        return kind, i, f.boundary_ties(), port.decode_text(labels, rf[0][1]) if rf else b"", float(rf[0][0]) if rf else 0.0, port.decode_text(labels, ro[0][1]) if ro else b"", float(ro[0][0]) if ro else 0.0, p

    out = {}
    cores = os.cpu_count() or 1
    for kind in ("peaky", "noisy", "uniform"):
        n = a.utterances if kind != "uniform" else max(200, a.utterances // 4)
        with ThreadPoolExecutor(max_workers=cores) as ex:
            res = list(ex.map(one, [(kind, i) for i in range(n)]))
        ems = np.stack([r[7] for r in res]).astype(np.float64)
        rr = ref.decode_batch(ems, [T] * n, A, a.beam, cores, S)
        ref_t = [port.decode_text(labels, tok) for _, tok in rr]
        ref_c = [float(c) for c, _ in rr]
        tie_utts = sum(1 for r in res if r[2] > 0)
        tie_steps = sum(r[2] for r in res)
        flat_differs = sum(1 for r, t, c in zip(res, ref_t, ref_c) if r[3] != t or r[4] != c)
        flat_differs_without_tie = sum(1 for r, t, c in zip(res, ref_t, ref_c) if (r[3] != t or r[4] != c) and r[2] == 0)
        order_differs = sum(1 for r, t, c in zip(res, ref_t, ref_c) if r[5] != t or r[6] != c)
        out[kind] = {"utterances": n, "frames": T, "beam": a.beam, "utterances_with_a_boundary_tie": tie_utts, "boundary_tie_steps": int(tie_steps),
                     "steps": n * T, "flat_restatement_differs_from_the_reference": flat_differs, "of_those_without_a_tie": flat_differs_without_tie,
                     "reference_order_restatement_differs_from_the_reference": order_differs}
        sys.stderr.write("%s %s\n" % (kind, json.dumps(out[kind])))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
