#!/usr/bin/env python
"""A long run of tests/test_oracle_fuzz.py's comparison: the C restatement (oracle/stt_port.c) against the REAL reference decoder
(oracle/_ref) on seeded random cases -- more seeds, longer emissions, wider beams, several results.  CPU only (test infrastructure
checking test infrastructure).  A mismatch is classified by the restatement's own boundary-tie counter.

    python benchmarks/oracle_fuzz_long.py [--seeds 20] [--cases 300] > profiles/rNN_oracle_fuzz_long.txt
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import canon  # noqa: E402
from oracle import port, ref  # noqa: E402
from stt_amd import synth  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "fixtures")


def equal_score_order(ca, cb, port_more):
    """The results differ only where prefix_compare leaves the order open (ctc_beam_search_decoder.cpp / path_trie.h: equal score AND equal
    last character -> `false` both ways; std::sort / partial_sort decide): the same confidences, and every result the reference returns is
    among the restatement's results when it is asked for more of them -- with the same tokens, timesteps and confidence: the reference
    picked another member of a group the restatement ranks as equals."""
    if sorted(x[0] for x in ca) != sorted(x[0] for x in cb):
        return False
    return all(y in port_more for y in cb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--reference-order", action="store_true", help="the pointer-trie restatement with libstdc++'s nth_element / partial_sort restated (stt_port.c, Part D): every case must EQUAL the reference, tie cases included")
    ap.add_argument("--one-seed", type=int, default=-1, help="(child) run this seed only")
    ap.add_argument("--show-case", type=int, default=-1, help="(child) with --one-seed: print both decoders' results of this case")
    ap.add_argument("--only-case", type=int, default=-1, help="(child) with --one-seed: decode this case with the restatement only (does IT survive?)")
    a = ap.parse_args()
    if a.one_seed < 0:
        return parent(a)
    vocab = open(os.path.join(FIX, "vocab.pruned.txt")).read().split()
    rigs = {}
    for mode in ("word", "bytes"):
        if mode == "word":
            labels, space = port.parse_alphabet_file(os.path.join(FIX, "alphabet.txt"))
            A = ref.Alphabet(os.path.join(FIX, "alphabet.txt"))
            sp = os.path.join(FIX, "pruned_lm.scorer")
        else:
            labels, space = port.utf8_alphabet()
            A = ref.Alphabet(None)
            sp = os.path.join(FIX, "pruned_lm.bytes.scorer")
        rigs[mode] = (labels, space, A, port.Scorer(sp), ref.Scorer(sp, A))
    # configs[4]'s kind of scorer: a code-point level LM (three-byte units, order 5) in bytes-output mode, written by stt_amd/tools
    import tempfile
    from stt_amd import scorertools
    tmp = tempfile.TemporaryDirectory()
    lmf, vf, pkg = (os.path.join(tmp.name, x) for x in ("cp.binary", "cp.vocab", "cp.scorer"))
    scorertools.synth_lm(lmf, vf, words=1500, order=5, seed=9, avg={2: 40, 3: 2.0, 4: 1.0, 5: 0.7}, codepoints=True)
    scorertools.generate_scorer_package(lmf, vf, pkg, force_bytes_output_mode=True, default_alpha=0.93, default_beta=1.18)
    units = open(vf, encoding="utf-8").read().split()
    labels, space = port.utf8_alphabet()
    A = ref.Alphabet(None)
    rigs["cp"] = (labels, space, A, port.Scorer(pkg), ref.Scorer(pkg, A))
    t0 = time.time()
    tot = dict(cases=0, equal=0, differ_with_boundary_tie=0, differ_in_the_order_of_equal_scores=0, unexplained=0)
    if a.reference_order:
        tot = dict(cases=0, equal=0, equal_and_the_flat_restatement_differs=0, equal_with_a_boundary_tie=0, unexplained=0)
    bad = []
    for seed in [a.one_seed]:
        rng = np.random.RandomState(9000 + seed)
        for case in range(a.cases):
            mode = "word" if case % 4 else ("bytes" if case % 8 else "cp")
            labels, space, A, P, S = rigs[mode]
            C = len(labels) + 1
            lm = bool(rng.randint(2)) or mode == "cp"
            beam = int(rng.choice([4, 16, 50, 100, 500] if mode == "word" else [8, 32, 128]))
            T = int(rng.randint(6, 120 if mode == "word" else 40))
            cp, ctn = [(1.0, 40), (0.999, 40), (0.95, 40), (1.0, 8), (0.97, 12)][int(rng.randint(5))]
            if mode != "word" and not lm and cp == 1.0 and ctn >= 40:
                cp = 0.999
            hot = {}
            if lm and mode == "word" and rng.rand() < 0.3:
                hot = {str(rng.choice(vocab)): float(rng.choice([-2.0, 4.0, 10.0]))}
            if rng.rand() < 0.5:
                x = rng.randn(T, C) * rng.choice([0.5, 1.5, 3.0])
                p = np.exp(x - x.max(1, keepdims=True)); p = (p / p.sum(1, keepdims=True)).astype(np.float32)
            else:
                sent = "".join(rng.choice(units, size=rng.randint(1, 8))) if mode == "cp" else " ".join(rng.choice(vocab, size=rng.randint(1, 8)))
                lab = [b - 1 for b in sent.encode()] if mode != "word" else [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
                p = synth.peaky_emissions(lab, T, C, C - 1, seed=int(rng.randint(1 << 30)), noise=float(rng.choice([0.02, 0.05, 0.5])), lead=2)
            chunk = int(rng.choice([1, 16, 48]))
            n = int(rng.choice([1, 2]))   # (more results than prefixes with a finite probability: the REFERENCE dereferences a null timestep node -- seen with n = 5)
            if a.only_case >= 0 and case != a.only_case:
                continue
            print("case %d" % case, file=sys.stderr, flush=True)
            o = port.Decoder(labels, space, beam, P if lm else None, cp, ctn, hot or None, reference_order=a.reference_order)
            flat = port.Decoder(labels, space, beam, P if lm else None, cp, ctn, hot or None) if a.reference_order else None
            if a.only_case >= 0:
                for k in range(0, T, chunk):
                    o.next(p[k:k + chunk])
                o.decode(n)
                return 0
            r = ref.Decoder(A, beam, S if lm else None, cp, ctn, hot or None)
            for k in range(0, T, chunk):
                o.next(p[k:k + chunk]); r.next(p[k:k + chunk])
                if flat is not None:
                    flat.next(p[k:k + chunk])
            tot["cases"] += 1
            if a.show_case == case:
                ra, rb = o.decode(n), r.decode(n)
                print(mode, "lm", lm, "beam", beam, "T", T, "cutoff", cp, ctn, "chunk", chunk, "n", n, "boundary ties", (flat or o).boundary_ties())
                for i in range(max(len(ra), len(rb))):
                    for nm, rr in (("port", ra), ("ref ", rb)):
                        print(i, nm, (rr[i][0], [int(x) for x in rr[i][1]], [int(x) for x in rr[i][2]]) if i < len(rr) else None)
                np.save("/tmp/fuzz_case_emissions.npy", p)
                return 0
            ca, cb = canon(o.decode(n)), canon(r.decode(n))
            if a.reference_order:
                if ca == cb:
                    tot["equal"] += 1
                    tot["equal_and_the_flat_restatement_differs"] += 1 if canon(flat.decode(n)) != cb else 0
                    tot["equal_with_a_boundary_tie"] += 1 if flat.boundary_ties() > 0 else 0
                else:
                    tot["unexplained"] += 1
                    bad.append((seed, case, mode, lm, beam, T, cp, ctn, hot, chunk, n))
            elif ca == cb:
                tot["equal"] += 1
            elif o.boundary_ties() > 0:
                tot["differ_with_boundary_tie"] += 1
            elif equal_score_order(ca, cb, canon(o.decode(n + 64))):
                tot["differ_in_the_order_of_equal_scores"] += 1
            else:
                tot["unexplained"] += 1
                bad.append((seed, case, mode, lm, beam, T, cp, ctn, hot, chunk, n))
    print(json.dumps({"seed": 9000 + a.one_seed, **tot, "unexplained_cases": bad[:20]}), flush=True)
    return 0


def parent(a):
    """one child per seed: the reference decoder can take the process down (it dereferences a null timestep node for some dead beams);
    the case it died in is then decoded by the restatement alone, in a child of its own"""
    import subprocess
    t0 = time.time()
    tot, bad, ref_crashes = {}, [], []
    for seed in range(a.seeds):
        extra = ["--reference-order"] if a.reference_order else []
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-seed", str(seed), "--cases", str(a.cases)] + extra, capture_output=True, text=True)
        last = [l for l in r.stderr.splitlines() if l.startswith("case ")]
        if r.returncode != 0:
            case = int(last[-1].split()[1]) if last else -1
            alone = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-seed", str(seed), "--cases", str(a.cases), "--only-case", str(case)] + extra, capture_output=True, text=True)
            ref_crashes.append({"seed": 9000 + seed, "case": case, "child_rc": r.returncode, "restatement_alone_rc": alone.returncode})
            print(json.dumps({"seed": 9000 + seed, "died_in_case": case, "restatement_alone_rc": alone.returncode, "elapsed_s": round(time.time() - t0, 1)}), flush=True)
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        bad += d.pop("unexplained_cases"); d.pop("seed")
        for k, v in d.items():
            tot[k] = tot.get(k, 0) + v
        print(json.dumps({"seed": 9000 + seed, **tot, "elapsed_s": round(time.time() - t0, 1)}), flush=True)
    print(json.dumps({"total": tot, "unexplained_cases": bad[:20], "seeds_the_reference_decoder_died_in": ref_crashes}))
    return 1 if (tot.get("unexplained") or any(c["restatement_alone_rc"] != 0 for c in ref_crashes)) else 0


if __name__ == "__main__":
    sys.exit(main())
