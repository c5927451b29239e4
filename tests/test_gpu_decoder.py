"""The HIP CTC beam-search decoder against the oracle: bit-exact scores, identical tokens and timesteps."""
import os

import numpy as np
import pytest

from conftest import canon, case_emissions, dump, golden_results
from stt_amd import modelfile, synth

pytestmark = pytest.mark.gpu


def _mk(tmp, labels, beam, name):
    from stt_amd import Model
    C = len(labels) + 1
    w = synth.synth_weights(3, n_hidden=128, n_classes=C)
    path = str(tmp / (name + ".sttw"))
    modelfile.write_model(path, w, labels, beam_width=beam)
    return Model(path)


@pytest.fixture(scope="module")
def models(tmp_path_factory, port, fix):
    tmp = tmp_path_factory.mktemp("dec")
    word = _mk(tmp, synth.ENGLISH_LABELS, 500, "word")
    word_lm = _mk(tmp, synth.ENGLISH_LABELS, 500, "word_lm")
    word_lm.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    ulabels, _ = port.utf8_alphabet()
    byte = _mk(tmp, ulabels, 500, "bytes")
    byte_lm = _mk(tmp, ulabels, 500, "bytes_lm")
    byte_lm.enableExternalScorer(os.path.join(fix, "pruned_lm.bytes.scorer"))
    return {("word", False): word, ("word", True): word_lm, ("bytes", False): byte, ("bytes", True): byte_lm}


def _gpu_decode(models, case, probs):
    m = models[(case["mode"], case["lm"])]
    m.clearHotWords() if case["lm"] else None
    for wd, boost in (case.get("hot") or {}).items():
        m.addHotWord(wd, boost)
    d = m.createDecoder(1, case["beam"], case.get("cutoff_prob", 1.0), case.get("cutoff_top_n", 40))
    if case.get("chunk"):
        for i in range(0, len(probs), case["chunk"]):
            d.next(probs[i:i + case["chunk"]])
    else:
        d.next(probs)
    res = d.decode(min(case["beam"], 50))[0]
    st = d.stats()
    if case["lm"]:
        m.clearHotWords()
    return res, st, d


@pytest.mark.parametrize("step", [0, 2], ids=["generic-step", "bitmap-step"])
def test_decoder_matches_reference_goldens(models, decoder_cases, step):
    """Tunable search_step = 2: the word-mode cases with a scorer run the step with dictionary label bitmaps, two language-model
    waves and FullScore through the hashed n-gram index; everything else takes the generic step either way."""
    from stt_amd import native
    native.set_tuning("search_step", step)
    try:
        _goldens(models, decoder_cases)
    finally:
        native.set_tuning("search_step", 2)


def _goldens(models, decoder_cases):
    cases, gold = decoder_cases
    bad = []
    for case in cases:
        res, st, _ = _gpu_decode(models, case, case_emissions(case))
        want = golden_results(gold, case["name"])
        ok = canon(res) == sorted(want)
        print(case["name"], "OK" if ok else "MISMATCH", st)
        if not ok:
            bad.append(case["name"])
            dump("dec_gold_" + case["name"], got_conf=np.array([r[0] for r in res]), want_conf=np.array([r[0] for r in want]),
                 got_len=np.array([len(r[1]) for r in res]), want_len=np.array([len(r[1]) for r in want]))
        assert st["error"] == 0, (case["name"], st)
    assert not bad, bad


def test_decoder_whole_beam_vs_port(models, port, english, fix):
    """Fresh seeds; the complete beam (every prefix's score / blank / non-blank log-probs and last character) bit for bit."""
    labels, space = english
    P = port.Scorer(os.path.join(fix, "pruned_lm.scorer"))
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(4242)
    for it in range(6):
        sent = " ".join(rng.choice(vocab, size=rng.randint(2, 8)))
        lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
        T = 30 + 5 * len(lab)
        noise = [0.02, 0.1, 0.3, 1.0, 0.05, 0.5][it]
        p = synth.peaky_emissions(lab, T, 29, 28, seed=900 + it, noise=noise)
        for beam, lm in [(1, False), (37, True), (500, True), (500, False), (1024, True)]:
            dp = port.Decoder(labels, space, beam, P if lm else None)
            dp.next(p)
            d = models[("word", lm)].createDecoder(1, beam)
            d.next(p)
            gs, gb, gnb, gch = d.raw_beam(0)
            ps, pb, pnb, pch, _ = dp.raw_beam()
            tag = "it%d beam%d lm%d" % (it, beam, lm)
            if not (len(gs) == len(ps) and np.array_equal(gs.view(np.uint32), ps.view(np.uint32))):
                dump("beam_mismatch_%d_%d_%d" % (it, beam, lm), gs=gs, ps=ps, gch=gch, pch=pch, gb=gb, pb=pb, gnb=gnb, pnb=pnb)
            assert len(gs) == len(ps), tag
            assert np.array_equal(gs.view(np.uint32), ps.view(np.uint32)), tag
            assert np.array_equal(gb.view(np.uint32), pb.view(np.uint32)) and np.array_equal(gnb.view(np.uint32), pnb.view(np.uint32)), tag
            assert np.array_equal(gch, pch), tag
            n = min(beam, 30)
            assert canon(d.decode(n)[0]) == canon(dp.decode(n)), tag
            assert d.stats()["error"] == 0


def test_decoder_random_softmax_and_edge_cases(models, port, english, fix):
    labels, space = english
    P = port.Scorer(os.path.join(fix, "pruned_lm.scorer"))
    rng = np.random.RandomState(5)
    logits = rng.randn(90, 29) * 0.7
    p = np.exp(logits); p = (p / p.sum(1, keepdims=True)).astype(np.float32)   # near-uniform: worst case for the beam
    for beam, lm in [(64, False), (200, True)]:
        dp = port.Decoder(labels, space, beam, P if lm else None); dp.next(p)
        d = models[("word", lm)].createDecoder(1, beam); d.next(p)
        assert canon(d.decode(beam)[0]) == canon(dp.decode(beam)), (beam, lm)
    # nothing fed / only blanks (start_expanding never set, ctc_beam_search_decoder.cpp:125-132) / single frame
    for frames in (None, np.tile(np.eye(29, dtype=np.float32)[28], (7, 1)), p[:1]):
        dp = port.Decoder(labels, space, 16, P); d = models[("word", True)].createDecoder(1, 16)
        if frames is not None:
            dp.next(frames); d.next(frames)
        assert canon(d.decode(4)[0]) == canon(dp.decode(4))


def test_decoder_streams_are_independent_and_chunking_is_exact(models, port, english, fix):
    """A batch of streams decoded together == each decoded alone; feeding 16-frame chunks == feeding all at once."""
    labels, space = english
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(77)
    probs, nfr = [], []
    for i in range(5):
        sent = " ".join(rng.choice(vocab, size=rng.randint(2, 6)))
        lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
        probs.append(synth.peaky_emissions(lab, 30 + 5 * len(lab), 29, 28, seed=i, noise=0.2)); nfr.append(len(probs[-1]))
    tmax = max(nfr)
    batch = np.zeros((5, tmax, 29), np.float32)
    for i, p in enumerate(probs):
        batch[i, :len(p)] = p
    m = models[("word", True)]
    d = m.createDecoder(5, 100)
    d.next(batch, nfr)
    together = d.decode(10)
    for i, p in enumerate(probs):
        d1 = m.createDecoder(1, 100); d1.next(p)
        d2 = m.createDecoder(1, 100)
        for k in range(0, len(p), 16):
            d2.next(p[k:k + 16])
        a, b, c = canon(together[i]), canon(d1.decode(10)[0]), canon(d2.decode(10)[0])
        assert a == b == c, i


def test_decoders_of_one_model_on_several_host_threads(models, english, fix):
    """Round 6: a STTX_Decoder runs on a stream and result blocks of its own, so several decoders of ONE model may be driven side by side from
    several host threads (bench.py's decoder-stage workloads keep four in flight).  Eight decoders on four threads, fed in chunks with a
    decode after every chunk: every result equals the same work done one decoder after the other."""
    import threading
    vocab = open(os.path.join(fix, "vocab.pruned.txt")).read().split()
    rng = np.random.RandomState(91)
    m = models[("word", True)]
    jobs = []
    for j in range(8):
        rows = []
        for i in range(6):
            sent = " ".join(rng.choice(vocab, size=rng.randint(2, 6)))
            lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
            rows.append(synth.peaky_emissions(lab, 120, 29, 28, seed=100 * j + i, noise=0.2)[:120])
        jobs.append(np.stack(rows).astype(np.float32))

    def run(batch):
        d = m.createDecoder(len(batch), 100)
        out = []
        for k in range(0, batch.shape[1], 40):
            d.next(batch[:, k:k + 40])
            out.append([canon(r) for r in d.decode(3)])
        return out

    want = [run(b) for b in jobs]
    got, errs = [None] * len(jobs), []
    from stt_amd import native
    native.set_tuning("decoder_streams", 4)      # (a pool of four streams for this model's decoders; default 1: the model's own)

    def worker(t):
        try:
            for rep in range(3):
                for j in range(t, len(jobs), 4):
                    got[j] = run(jobs[j])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    native.set_tuning("decoder_streams", 1)
    assert not errs, errs
    assert got == want


def test_decoder_with_synthetic_order5_scorer(models, port, english, fix, tmp_path):
    """A scorer written by stt_amd/tools (synthetic order-5 quantised array trie, 3000 pseudo-words, words up to 15 letters):
    exercises every trie level, the Bhiksha hint table, the interpolation search and the 16-byte word registers."""
    from stt_amd import scorertools
    labels, space = english
    lm, vocab_p, pkg = str(tmp_path / "lm.binary"), str(tmp_path / "vocab.txt"), str(tmp_path / "synth.scorer")
    scorertools.synth_lm(lm, vocab_p, words=3000, order=5, seed=5)
    scorertools.generate_scorer_package(lm, vocab_p, pkg, alphabet=os.path.join(fix, "alphabet.txt"), default_alpha=0.9, default_beta=1.2)
    vocab = open(vocab_p).read().split()
    P = port.Scorer(pkg)
    m = _mk(tmp_path, synth.ENGLISH_LABELS, 500, "synth_lm")
    m.enableExternalScorer(pkg)
    rng = np.random.RandomState(321)
    for it in range(5):
        sent = " ".join(rng.choice(vocab, size=rng.randint(2, 7)))
        lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent]
        p = synth.peaky_emissions(lab, 30 + 5 * len(lab), 29, 28, seed=40 + it, noise=[0.02, 0.1, 0.3, 0.6, 1.0][it])
        for beam in (64, 500, 1024):
            hot = {vocab[3]: 2.5, vocab[10]: -1.0} if it == 2 else None
            dp = port.Decoder(labels, space, beam, P, hot_words=hot); dp.next(p)
            if hot:
                for wd, b in hot.items():
                    m.addHotWord(wd, b)
            d = m.createDecoder(1, beam); d.next(p)
            if hot:
                m.clearHotWords()
            gs, gb, gnb, gch = d.raw_beam(0)
            ps, pb, pnb, pch, _ = dp.raw_beam()
            tag = "it%d beam%d" % (it, beam)
            assert len(gs) == len(ps) and np.array_equal(gs.view(np.uint32), ps.view(np.uint32)), tag
            assert np.array_equal(gch, pch), tag
            n = min(beam, 20)
            assert canon(d.decode(n)[0]) == canon(dp.decode(n)), tag
            assert d.stats()["error"] == 0


def test_decoder_profile_counters_do_not_change_results(models, port, english, fix):
    """STTX_DecoderSetProfiling / STTX_DecoderGetProfile (include/stt_amd.h; benchmarks/search_micro.py): the kernel's phase cycle
    counters and stamps are a measurement, not a mode -- the same beam comes out with them on, the phases add up to a positive number of
    cycles and the bitmap step's own stamps (item table complete, chunks taken) are there."""
    labels, space = english
    m = models[("word", True)]
    P = port.Scorer(os.path.join(fix, "pruned_lm.scorer"))
    rng = np.random.RandomState(77)
    x = rng.randn(3, 60, 29).astype(np.float32)
    probs = np.exp(x - x.max(2, keepdims=True)); probs = (probs / probs.sum(2, keepdims=True)).astype(np.float32)
    plain = m.createDecoder(3, 500); plain.next(probs)
    want = [canon(r) for r in plain.decode(10)]
    prof = m.createDecoder(3, 500); prof.setProfiling(2); prof.next(probs)
    assert [canon(r) for r in prof.decode(10)] == want
    phases, stamps, ms = prof.profile()
    steps = prof.stats()["steps"]
    assert steps == 3 * 60 and ms > 0.0
    serial = sum(v for k, v in phases.items() if not k.startswith("lm_wave"))
    assert 1000 * steps < serial < 400000 * steps, serial / steps          # thousands to tens of thousands of cycles per stream-timestep
    assert stamps[50] > 0 and stamps[52] > 0                                # bitmap step: table-complete time, chunks taken by wave 9
    o = port.Decoder(labels, space, 500, P); o.next(probs[1])
    assert want[1] == canon(o.decode(10))


def test_a_counter_wait_that_times_out_ends_as_an_error_not_as_a_transcript(models, port, english, fix):
    """ctc.hip: wait_count -- the bitmap step's waits on LDS counters are bounded (a logic error must not hang the GPU); a wait that
    gives up sets error bit 0x10 and the beam may be anything.  Every decode path reports the bit instead of a result
    (check_decoder_errors: STT_* return NULL, STTX_DecoderDecode fails).  Tunable wait_spins = 1 forces the time-out."""
    from stt_amd import native
    p = synth.peaky_emissions([8, 5, 12, 12, 15, 0, 23, 15, 18, 12, 4], 120, 29, 28, seed=1, noise=0.8)
    m = models[("word", True)]
    native.set_tuning("wait_spins", 1)
    try:
        d = m.createDecoder(4, 500)
        d.next(np.stack([p] * 4))
        st = d.stats()
        assert st["error"] != 0, st                      # (STT_ERR_FAIL_RUN_SESS: some stream carries an error bit -- the waves did overtake a counter)
        with pytest.raises(RuntimeError):
            d.decode(1)
    finally:
        native.set_tuning("wait_spins", 0)
    d = m.createDecoder(1, 500); d.next(p)              # ... and with the default bound the same input decodes
    assert d.stats()["error"] == 0 and d.decode(1)[0]


def test_load_time_tunables_are_frozen_while_a_model_is_alive(models):
    """lstm_upw decides how the recurrent matrix is packed at load: changing it under a live model is refused (STT_ERR_INVALID_SHAPE)."""
    from stt_amd import native
    cur = native.get_tuning("lstm_upw")
    with pytest.raises(Exception):
        native.set_tuning("lstm_upw", 8 if cur != 8 else 16)
    native.set_tuning("lstm_upw", cur)                   # the same value is fine
    assert native.get_tuning("lstm_upw") == cur
