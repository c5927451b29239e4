"""tests/golden/make_golden.py -- regenerates the committed golden vectors FROM THE REAL REFERENCE.

Run in the build container (needs /root/reference and oracle/_ref built by `make -C oracle ref`):
    python tests/golden/make_golden.py
Outputs (committed):
    tests/golden/decoder_golden.npz   reference DecoderState outputs on seeded synthetic emissions
    tests/golden/kenlm_golden.json    reference KenLM / Scorer query results
    tests/golden/fixtures/kenlm_test_*.bin  vendored build_binary run on kenlm/lm/test.arpa (4 trie flavours)
Fixture data copied verbatim from the reference tree (data, not source): data/alphabet.txt,
data/smoke_test/{pruned_lm.scorer,pruned_lm.bytes.scorer,vocab.pruned.txt,LDC93S1_pcms16le_1_16000.wav},
tests/test_data/alphabet_{unix,macos,windows}.txt.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from stt_amd import synth  # noqa: E402

FIX = os.path.join(HERE, "fixtures")
REF = "/root/reference"


def decoder_cases():
    """(name, mode, beam, use_scorer, emissions kwargs, extra) -- emissions are regenerated from the seed by the tests."""
    vocab = open(os.path.join(FIX, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(1234)
    cases = []
    for i, (noise, beam, lm) in enumerate([(0.02, 1, False), (0.02, 64, False), (0.02, 64, True), (0.1, 200, True), (0.3, 500, True),
                                            (0.3, 500, False), (1.0, 128, True), (0.05, 1024, True)]):
        words = list(rng.choice(vocab, size=rng.randint(3, 8)))
        cases.append(dict(name="word%d" % i, mode="word", beam=beam, lm=lm, sentence=" ".join(words), noise=noise, seed=100 + i, T=0))
    for i, (noise, beam, lm) in enumerate([(0.002, 32, False), (0.01, 128, True), (0.02, 512, True)]):
        words = list(rng.choice(vocab, size=rng.randint(2, 4)))
        cases.append(dict(name="bytes%d" % i, mode="bytes", beam=beam, lm=lm, sentence=" ".join(words), noise=noise, seed=200 + i, T=0))
    cases.append(dict(name="hot0", mode="word", beam=100, lm=True, sentence="she had your dark suit", noise=0.3, seed=300, T=0,
                      hot={"dark": 5.0, "suit": -3.0}))
    cases.append(dict(name="cut0", mode="word", beam=100, lm=True, sentence="she had your dark suit", noise=0.3, seed=301, T=0,
                      cutoff_prob=0.95, cutoff_top_n=10))
    cases.append(dict(name="stream0", mode="word", beam=64, lm=True, sentence="in greasy wash water all year", noise=0.2, seed=302, T=0, chunk=16))
    return cases


def labels_for(case):
    s = case["sentence"]
    if case["mode"] == "bytes":
        return [b - 1 for b in s.encode()], 256, 255
    return [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in s], 29, 28


def emissions_for(case):
    lab, C, blank = labels_for(case)
    T = case["T"] or (30 + 5 * len(lab))
    return synth.peaky_emissions(lab, T, C, blank, seed=case["seed"], noise=case["noise"])


def main():
    A = ref.Alphabet(os.path.join(FIX, "alphabet.txt"))
    AU = ref.Alphabet(None)
    S = ref.Scorer(os.path.join(FIX, "pruned_lm.scorer"), A)
    SU = ref.Scorer(os.path.join(FIX, "pruned_lm.bytes.scorer"), AU)
    out = {}
    cases = decoder_cases()
    for c in cases:
        p = emissions_for(c).astype(np.float64)  # float32-representable doubles, as the reference gets from stt.cc:327
        alpha, sc = (AU, SU) if c["mode"] == "bytes" else (A, S)
        d = ref.Decoder(alpha, c["beam"], sc if c["lm"] else None, cutoff_prob=c.get("cutoff_prob", 1.0),
                        cutoff_top_n=c.get("cutoff_top_n", 40), hot_words=c.get("hot"))
        if c.get("chunk"):
            for i in range(0, len(p), c["chunk"]):
                d.next(p[i:i + c["chunk"]])
        else:
            d.next(p)
        nres = min(c["beam"], 50)
        res = d.decode(nres)
        out[c["name"] + "/conf"] = np.array([r[0] for r in res])
        out[c["name"] + "/lens"] = np.array([len(r[1]) for r in res], dtype=np.int32)
        out[c["name"] + "/tokens"] = np.concatenate([r[1] for r in res]).astype(np.uint32) if res else np.zeros(0, np.uint32)
        out[c["name"] + "/timesteps"] = np.concatenate([r[2] for r in res]).astype(np.uint32) if res else np.zeros(0, np.uint32)
        print(c["name"], len(res), alpha.decode(res[0][1]))
    np.savez_compressed(os.path.join(HERE, "decoder_golden.npz"), **out)
    with open(os.path.join(HERE, "decoder_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)

    # ---- KenLM / scorer goldens from the reference library
    kg = {"scorer": [], "kenlm": {}}
    for words, bos in [(["she", "had", "your"], True), (["she", "had", "your"], False), (["zzzz"], False), (["she"], True),
                       (["dark", "suit", "in", "greasy"], False), (["had", "your", "dark", "suit"], True), (["water"], False)]:
        kg["scorer"].append(dict(words=words, bos=bos, value=S.log_cond_prob(words, bos)))
    kg["scorer_bytes"] = [dict(words=w, bos=b, value=SU.log_cond_prob(w, b)) for w, b in [(["s"], True), (["s", "h"], False), (["q", "z"], False)]]
    sentences = [["looking", "on", "a", "little", "more", "loin"], ["looking", "on", "a", "little", "the", "biarritz", "not_found", "more", ".", "</s>"],
                 ["also", "would", "consider", "higher", "looking"], ["higher", "looking", "not_found"]]
    build_binary = os.path.join(ROOT, "oracle", "_ref", "build_binary")
    arpa = os.path.join(REF, "native_client/kenlm/lm/test.arpa")
    for name, args in [("trie", ["trie"]), ("array", ["-a", "22", "trie"]), ("quant", ["-q", "8", "-b", "8", "trie"]),
                       ("qarray", ["-a", "22", "-q", "8", "-b", "8", "trie"])]:
        path = os.path.join(FIX, "kenlm_test_%s.bin" % name)
        if os.path.exists(arpa):
            subprocess.run([build_binary] + args + [arpa, path], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        m = ref.KenLM(path)
        rows = []
        for s in sentences:
            for bos in (True, False):
                pr, ln = m.score(s, bos)
                rows.append(dict(words=s, bos=bos, probs=[float(x) for x in pr], lens=[int(x) for x in ln]))
        kg["kenlm"][name] = rows
    with open(os.path.join(HERE, "kenlm_golden.json"), "w") as f:
        json.dump(kg, f, indent=1)
    print("wrote goldens")


if __name__ == "__main__":
    main()
