"""Fixtures for KenLM PROBING binaries (model type 0: lm/search_hashed.hh) -- generated HERE from the reference, committed as data:
    tests/golden/fixtures/kenlm_test_probing.bin      vendored build_binary ("probing", default multiplier 1.5) on kenlm/lm/test.arpa
    tests/golden/fixtures/kenlm_test_probing20.bin    ... with -p 2.0 (another bucket count)
    tests/golden/fixtures/probing_lm.scorer           the probing binary packaged with the words of test.arpa that data/alphabet.txt can spell
                                                     (the reference's own packaging code through oracle/_ref: ref_make_scorer)
    tests/golden/kenlm_probing_golden.json            FullScore answers of the REAL KenLM (oracle/_ref) on both binaries, model_test.cc's sequences
Needs /root/reference and oracle/_ref (make -C oracle ref).  Run once; tests only read the outputs."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
FIX = os.path.join(HERE, "fixtures")
REF = "/root/reference"


def main():
    from oracle import ref
    assert ref.available()
    build_binary = os.path.join(ROOT, "oracle", "_ref", "build_binary")
    arpa = os.path.join(REF, "native_client/kenlm/lm/test.arpa")
    sentences = [["looking", "on", "a", "little", "more", "loin"], ["looking", "on", "a", "little", "the", "biarritz", "not_found", "more", ".", "</s>"],
                 ["also", "would", "consider", "higher", "looking"], ["higher", "looking", "not_found"], ["a", "little", "more", "loin", "also", "would", "consider"]]
    out = {}
    for name, args in [("probing", ["probing"]), ("probing20", ["-p", "2.0", "probing"])]:
        path = os.path.join(FIX, "kenlm_test_%s.bin" % name)
        subprocess.run([build_binary] + args + [arpa, path], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        m = ref.KenLM(path)
        rows = []
        for s in sentences:
            for bos in (True, False):
                pr, ln = m.score(s, bos)
                rows.append(dict(words=s, bos=bos, probs=[float(x) for x in pr], lens=[int(x) for x in ln]))
        out[name] = rows
    # a scorer package around the probing binary: the reference's own packaging (scorer.cpp fill_dictionary + save_dictionary through the shim)
    words = []
    for line in open(arpa):
        parts = line.rstrip("\n").split("\t")
        if len(parts) >= 2 and parts[1] and " " not in parts[1] and parts[1].isalpha() and parts[1].islower():
            words.append(parts[1])
    words = sorted(set(words))
    vocab = os.path.join(FIX, "probing_lm.vocab.txt")
    open(vocab, "w").write(" ".join(words) + "\n")
    pkg = os.path.join(FIX, "probing_lm.scorer")
    import tempfile
    with tempfile.TemporaryDirectory() as d:     # a package holds the binary WITHOUT its vocabulary strings (build_binary -v: data/lm/generate_lm.py:118-126)
        lm_v = os.path.join(d, "probing_v.bin")
        subprocess.run([build_binary, "-v", "probing", arpa, lm_v], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ref.make_scorer(lm_v, vocab, os.path.join(FIX, "alphabet.txt"), 0.93, 1.18, pkg)
    A = ref.Alphabet(os.path.join(FIX, "alphabet.txt"))
    S = ref.Scorer(pkg, A)
    out["scorer"] = [dict(words=w, bos=b, value=S.log_cond_prob(w, b)) for w, b in
                     [(["looking", "on", "a"], True), (["looking", "on", "a"], False), (["zzzz"], False), (["a", "little", "more", "loin"], True), (["higher"], True)]]
    out["vocabulary"] = words
    json.dump(out, open(os.path.join(HERE, "kenlm_probing_golden.json"), "w"), indent=1)
    print("wrote", len(words), "words;", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
