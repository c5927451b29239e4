"""Wide alphabets (BASELINE.json configs[4] names a 6000-character output layer): the per-class arrays of a timestep do
not fit in LDS next to a beam of 1024, so the search reads per-row records prepared by a row-parallel kernel
(stt_amd/csrc/ctc.hip, "wide alphabets").  Same bar as everywhere: tokens, timesteps and f32 scores identical to the
oracle (the real reference decoder when oracle/_ref is built, else the C port, which tests/test_oracle_port.py pins)."""
import os

import numpy as np
import pytest

from conftest import canon
from stt_amd import modelfile, synth

pytestmark = pytest.mark.gpu
N_LABELS = 6000


@pytest.fixture(scope="module")
def wide(tmp_path_factory, port, fix):
    """English labels first (so the shipped English scorer's dictionary still applies), then CJK ideographs up to 6000."""
    from stt_amd import Model
    tmp = tmp_path_factory.mktemp("wide")
    eng = open(os.path.join(fix, "alphabet.txt"), encoding="utf-8").read()
    path = str(tmp / "alphabet_wide.txt")
    with open(path, "w", encoding="utf-8") as f:
        f.write(eng if eng.endswith("\n") else eng + "\n")
        for i in range(N_LABELS - 28):
            f.write(chr(0x4E00 + i) + "\n")
    labels, space = port.parse_alphabet_file(path)
    assert len(labels) == N_LABELS and space == 0
    w = synth.synth_weights(11, n_hidden=128, n_classes=N_LABELS + 1)
    w["layer_6/weights"] = (w["layer_6/weights"] * 8.0).astype(np.float32)
    mpath = str(tmp / "wide.sttw")
    modelfile.write_model(mpath, w, labels, beam_width=32)
    m = Model(mpath)
    return {"alphabet": path, "labels": labels, "space": space, "model": m, "weights": w}


def _oracle(wide, fix, port, beam, lm, cutoff_prob, cutoff_top_n):
    from oracle import ref
    scorer = os.path.join(fix, "pruned_lm.scorer")
    if ref.available():
        A = ref.Alphabet(wide["alphabet"])
        return ref.Decoder(A, beam, ref.Scorer(scorer, A) if lm else None, cutoff_prob, cutoff_top_n), "ref"
    return port.Decoder(wide["labels"], wide["space"], beam, port.Scorer(scorer) if lm else None, cutoff_prob, cutoff_top_n), "port"


def test_wide_decoder_matches_oracle(wide, port, fix):
    from oracle import ref
    m = wide["model"]
    C = N_LABELS + 1
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(5)
    sent = " ".join(rng.choice(vocab, size=4))
    lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
    p = synth.peaky_emissions(lab, 30 + 5 * len(lab), C, C - 1, seed=3, noise=0.05)
    # (beam, scorer, cutoff_prob, cutoff_top_n): with the API's fixed 1.0 / 40 every class is kept, only re-ordered (the
    # reference sorts but does not cut when cutoff_prob == 1, ctc_beam_search_decoder.cpp:337-352)
    cases = [(100, True, 1.0, 40), (64, False, 0.99, 40), (200, False, 0.9999, 60)]
    if ref.available():
        cases.append((1024, True, 1.0, 40))     # (the single-threaded C port needs minutes for this one)
    for beam, lm, cp, ctn in cases:
        if lm:
            m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
        else:
            m.disableExternalScorer()
        o, kind = _oracle(wide, fix, port, beam, lm, cp, ctn)
        o.next(p)
        d = m.createDecoder(1, beam, cp, ctn)
        for k in range(0, len(p), 16):            # chunked like the streaming path
            d.next(p[k:k + 16])
        n = min(beam, 20)
        got, want = d.decode(n)[0], o.decode(n)
        assert canon(got) == canon(want), (beam, lm, cp, ctn, kind)
        assert d.stats()["error"] == 0
    m.disableExternalScorer()


def test_wide_model_end_to_end(wide, port, fix):
    """MFCC -> dense -> LSTM -> 6001-way output layer + softmax -> wide search: probabilities within the stated tolerance
    of the numpy restatement; transcript == oracle decoder on the GPU's emissions; one-shot == streaming == batch."""
    from oracle import am_ref
    m = wide["model"]
    a = synth.synth_audio(16000, seed=21)
    probs = m.acousticProbs([a])[0]
    want = am_ref.utterance_probs(a, wide["weights"], weight_round=np.float16)
    assert probs.shape == want.shape and probs.shape[1] == N_LABELS + 1
    a_err, l_err = np.abs(probs - want).max(), np.abs(np.log(probs) - np.log(want)).max()
    print("wide model: max |dp| %.3e  max |dlnp| %.3e" % (a_err, l_err))
    assert a_err < 1e-4 and l_err < 2e-3, (a_err, l_err)          # the stated tolerance (tests/test_gpu_benchshape.py)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    o, kind = _oracle(wide, fix, port, 32, True, 1.0, 40)
    o.next(probs)
    want_text = b"".join(wide["labels"][t] for t in o.decode(1)[0][1]).decode("utf-8")
    text = m.stt(a)
    assert text == want_text, (text, want_text, kind)
    s = m.createStream()
    for k in range(0, len(a), 5120):
        s.feedAudioContent(a[k:k + 5120])
    assert s.finishStream() == text
    b = synth.synth_audio(9000, seed=22)
    assert m.sttBatch([a, b]) == [text, m.stt(b)]
    m.disableExternalScorer()


def test_wide_random_cases(wide, port, fix):
    """Seeded sweep in wide mode: beams across the capacity buckets, cut-offs that make the kept-class count vary per row,
    several streams per launch, random chunking, peaky and flat emissions."""
    m = wide["model"]
    C = N_LABELS + 1
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(77)
    for case in range(10):
        lm = case % 3 == 0
        beam = int(rng.choice([3, 64, 100]) if lm else rng.choice([1, 17, 64, 130, 600]))
        cp, ctn = (1.0, 40) if lm else [(0.99, 40), (0.9, 40), (0.999, 12), (0.9999, 200)][int(rng.randint(4))]
        T = int(rng.randint(3, 40))
        n_streams = int(rng.choice([1, 2]))
        probs = []
        for _ in range(n_streams):
            if rng.rand() < 0.7:
                sent = " ".join(rng.choice(vocab, size=rng.randint(1, 4)))
                lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
                lab = [int(rng.randint(28, N_LABELS)) if (not lm and rng.rand() < 0.3) else l for l in lab]   # some ideographs
                probs.append(synth.peaky_emissions(lab, T, C, C - 1, seed=int(rng.randint(1 << 30)), noise=float(rng.choice([0.02, 0.3]))))
            else:
                x = rng.randn(T, C) * 2.0
                p = np.exp(x - x.max(1, keepdims=True)); probs.append((p / p.sum(1, keepdims=True)).astype(np.float32))
        if lm:
            m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
        else:
            m.disableExternalScorer()
        d = m.createDecoder(n_streams, beam, cp, ctn)
        cuts = sorted(set(int(x) for x in rng.randint(1, T + 1, size=int(rng.randint(0, 3))))) + [T]
        k0 = 0
        for k1 in cuts:
            if k1 > k0:
                d.next(np.stack([p[k0:k1] for p in probs]))
            k0 = k1
        got = d.decode(3)
        assert d.stats()["error"] == 0
        for s in range(n_streams):
            o, kind = _oracle(wide, fix, port, beam, lm, cp, ctn)
            o.next(probs[s])
            assert canon(got[s]) == canon(o.decode(3)), (case, lm, beam, cp, ctn, T, n_streams, s, kind)
    m.disableExternalScorer()
