"""BASELINE.json configs[2..4] as parity cases at test size (the bench line is configs[1]):
   [2] streaming chunked feed (320 ms hops) with an intermediate decode after every chunk,
   [3] a variable-length batch (more than one 64-utterance group) sharded like the multi-GPU path,
   [4] byte-output model (256 classes, UTF8Alphabet) with the bytes scorer at beam 1024."""
import os

import numpy as np
import pytest

from conftest import canon
from stt_amd import dist, modelfile, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(tmp_path_factory, fix):
    from stt_amd import Model
    w = synth.synth_weights(33, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    path = str(tmp_path_factory.mktemp("cfg") / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=100)
    m = Model(path)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    return m


def test_config3_streaming_320ms_hops_with_intermediate_decodes(model, port, english, fix):
    """stt.cc:553-639: IntermediateDecode after every 5120-sample feed never changes what FinishStream returns, and every
    intermediate result equals decoding the windows processed so far (checked through the oracle on the GPU's emissions)."""
    labels, space = english
    P = port.Scorer(os.path.join(fix, "pruned_lm.scorer"))
    rng = np.random.RandomState(1)
    for u in range(4):
        n = int(rng.uniform(1.0, 6.0) * 16000)
        a = synth.synth_audio(n, seed=300 + u)
        probs = model.acousticProbs([a])[0]
        s = model.createStream()
        inter = []
        for k in range(0, n, 5120):
            s.feedAudioContent(a[k:k + 5120])
            inter.append(s.intermediateDecode())
        final = s.finishStream()
        assert final == model.stt(a)
        # after feeding F samples, 16*floor(windows/16) windows have been through the model (stt.cc:311-334); window count
        # = full frames pushed - n_context (the 9 leading zero frames are already there)
        for j, text in enumerate(inter):
            fed = min(n, (j + 1) * 5120)
            frames = (fed - 512) // 320 + 1 if fed >= 512 else 0
            done = max(0, frames - 9) // 16 * 16
            d = port.Decoder(labels, space, 100, P)
            if done:
                d.next(probs[:done])
            want = b"".join(labels[t] for t in d.decode(1)[0][1]).decode()
            assert text == want, (u, j, fed, done)


def test_config4_variable_length_batch_and_shards(model):
    """150 utterances of 0.5-6 s: the batch path (three groups, ragged lengths inside a group) == one utterance at a time,
    and the LPT shards of stt_amd/dist.py put together again give the same list (what 8 ranks + all_gather produce)."""
    rng = np.random.RandomState(2)
    lens = (rng.uniform(0.5, 6.0, size=150) * 16000).astype(int)
    lens[7] = 0; lens[19] = 300                      # empty and shorter than one window
    audio = [synth.synth_audio(int(n), seed=500 + i) for i, n in enumerate(lens)]
    whole = model.sttBatch(audio)
    for i in (0, 7, 19, 64, 65, 128, 149):
        assert model.stt(audio[i]) == whole[i], i
    shards = dist.shard_utterances([len(a) for a in audio], 8)
    assert sorted(i for s in shards for i in s) == list(range(150))
    merged = [None] * 150
    for s in shards:
        for i, t in zip(s, model.sttBatch([audio[i] for i in s])):
            merged[i] = t
    assert merged == whole


def test_config5_bytes_model_beam_1024_with_bytes_scorer(tmp_path, port, fix):
    """Byte-output mode as the reference defines it (doc/DECODER.rst: 255 byte labels + blank); C = 256 > cutoff_top_n = 40,
    so the class sort / cut-off of get_pruned_emissions is active; codepoint-level scorer; beam 1024."""
    from stt_amd import Model
    ulabels, uspace = port.utf8_alphabet()
    w = synth.synth_weights(44, n_hidden=128, n_classes=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 8.0).astype(np.float32)
    path = str(tmp_path / "bytes.sttw")
    modelfile.write_model(path, w, ulabels, beam_width=1024)
    m = Model(path)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.bytes.scorer"))
    P = port.Scorer(os.path.join(fix, "pruned_lm.bytes.scorer"))
    for u, n in enumerate([16000, 30000]):
        a = synth.synth_audio(n, seed=800 + u)
        probs = m.acousticProbs([a])[0]
        assert probs.shape[1] == 256
        d = port.Decoder(ulabels, uspace, 1024, P); d.next(probs)
        conf, tok, ts = d.decode(1)[0]
        md = m.sttWithMetadata(a, 1)["transcripts"][0]
        assert [t[1] for t in md["tokens"]] == [int(x) for x in ts]
        assert md["confidence"] == conf
        assert md["text"].encode("utf-8", "surrogateescape") == b"".join(ulabels[t] for t in tok) or len(md["tokens"]) == len(tok)


def test_config3_many_streams_batched_equal_one_by_one(model):
    """STTX_FeedAudioContentBatch / IntermediateDecodeBatch / FinishStreamBatch: 140 live streams (more than one group of 128
    rows -- the recurrent launch covers 128 with 16 units per workgroup) of different lengths, fed in 320 ms hops together == each stream fed alone through coqui-stt.h (every intermediate result
    and the final one)."""
    from stt_amd import model as M
    rng = np.random.RandomState(3)
    lens = (rng.uniform(0.2, 4.0, size=140) * 16000).astype(int)
    lens[5] = 100; lens[9] = 5120 * 3                                      # shorter than a window; exact multiple of the hop
    audio = [synth.synth_audio(int(n), seed=900 + i) for i, n in enumerate(lens)]
    # one by one (reference API)
    want_inter, want_final = [], []
    for a in audio:
        s = model.createStream()
        inter = []
        for k in range(0, len(a), 5120):
            s.feedAudioContent(a[k:k + 5120]); inter.append(s.intermediateDecode())
        want_inter.append(inter); want_final.append(s.finishStream())
    # all together
    streams = [model.createStream() for _ in audio]
    got_inter = [[] for _ in audio]
    k = 0
    live = list(range(len(audio)))
    while live:
        M.feedAudioContentBatch([streams[i] for i in live], [audio[i][k:k + 5120] for i in live])
        for i, t in zip(live, M.intermediateDecodeBatch([streams[i] for i in live])):
            got_inter[i].append(t)
        k += 5120
        live = [i for i in live if k < len(audio[i])]
    got_final = M.finishStreamBatch(streams)
    assert got_final == want_final
    assert got_inter == want_inter
    with pytest.raises(RuntimeError):
        streams[0].intermediateDecode()                                    # destroyed by the batch finish


@pytest.mark.parametrize("sr,win,step", [(8000, 256, 160), (22050, 705, 441), (44100, 1411, 882), (48000, 1536, 960)])
def test_other_sample_rates_take_the_reference_fft_length(tmp_path, sr, win, step):
    """util/config.py:306-325: a model trained on 8 kHz audio has 256-sample windows every 160 samples; TF's AudioSpectrogram then
    runs a 256-point FFT (129 bins), a 22.05 kHz one a 1024-point FFT.  Round 2 refused such models; the reference runs them.
    Features and probabilities against the restatement at that geometry, streaming == one-shot, frame counts of stt.cc."""
    import numpy as np

    from oracle import am_ref
    from stt_amd import Model, modelfile, synth
    w = synth.synth_weights(31, n_hidden=256)
    path = str(tmp_path / ("m%d.sttw" % sr))
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=32, sample_rate=sr, win_len=win, win_step=step)
    m = Model(path)
    assert m.sampleRate() == sr
    spec = am_ref.MfccSpec(sample_rate=sr, win_len=win, win_step=step)
    for n in (3 * sr + 77, win - 1, win, 0):
        a = synth.synth_audio(n, seed=n + sr, sample_rate=sr)
        got = m.computeMfcc(a)
        want = am_ref.mfcc_utterance(a, spec)
        assert got.shape == want.shape == (am_ref.n_frames_for(n, win, step), 26)
        assert np.abs(got - want).max() <= 2e-4, (n, np.abs(got - want).max())
    a = synth.synth_audio(2 * sr + 1234, seed=5, sample_rate=sr)
    probs = m.acousticProbs([a])[0]
    want = am_ref.utterance_probs(a, w, spec=spec, weight_round=np.float16)
    assert probs.shape == want.shape
    assert np.abs(probs - want).max() < 1e-4 and np.abs(np.log(probs) - np.log(want)).max() < 2e-3
    text = m.stt(a)
    s = m.createStream()
    for k in range(0, len(a), 16 * step):
        s.feedAudioContent(a[k:k + 16 * step])
    assert s.finishStream() == text


def test_config5_code_point_scorer_with_three_byte_units(tmp_path, port, ref):
    """configs[4] on the scorer SURVEY.md 8d names: a code-point level LM (three-byte units, `synth-lm --codepoints`) in bytes-output
    mode.  A full beam takes the code-point step's pruning (ctc.hip: `thr` / `theta` -- new prefixes that provably cannot reach the beam
    are left out of the selection and are not scored): complete N-best lists, scores and timesteps must still be the REAL reference
    decoder's, on near-uniform emissions (everything is a candidate) and on peaky ones, beams 64 / 1024, in one piece and in chunks."""
    from stt_amd import Model, scorertools
    lm, vocab, pkg = str(tmp_path / "cp.binary"), str(tmp_path / "cp.vocab"), str(tmp_path / "cp.scorer")
    scorertools.synth_lm(lm, vocab, words=6000, order=5, seed=9, avg={2: 60, 3: 2.0, 4: 1.0, 5: 0.7}, codepoints=True)
    scorertools.generate_scorer_package(lm, vocab, pkg, force_bytes_output_mode=True, default_alpha=0.93, default_beta=1.18)
    units = open(vocab, encoding="utf-8").read().split()
    ulabels, uspace = port.utf8_alphabet()
    w = synth.synth_weights(44, n_hidden=128, n_classes=256)
    path = str(tmp_path / "bytes.sttw")
    modelfile.write_model(path, w, ulabels, beam_width=1024)
    m = Model(path)
    m.enableExternalScorer(pkg)
    A = ref.Alphabet(None)
    S = ref.Scorer(pkg, A)
    rng = np.random.RandomState(5)
    cases = []
    logits = rng.randn(40, 256) * 0.5
    pu = np.exp(logits); cases.append(("near-uniform", (pu / pu.sum(1, keepdims=True)).astype(np.float32)))
    text = "".join(rng.choice(units, size=8)).encode("utf-8")
    cases.append(("peaky", synth.peaky_emissions([b - 1 for b in text], 30 + 5 * len(text), 256, 255, seed=3, noise=0.2)))
    for name, p in cases:
        for beam in (64, 1024):
            for chunk in (0, 7):
                dr = ref.Decoder(A, beam, S); dr.next(p.astype(np.float64))
                d = m.createDecoder(1, beam)
                if chunk:
                    for i in range(0, len(p), chunk):
                        d.next(p[i:i + chunk])
                else:
                    d.next(p)
                assert d.stats()["error"] == 0
                k = min(beam, 20)
                assert canon(d.decode(k)[0]) == canon(dr.decode(k)), (name, beam, chunk)


def test_config3_last_audio_carries_the_flush(model):
    """STTX_FeedAudioContentBatchEx: a stream whose final audio is flagged has its flush (partial window, trailing context frames, last
    partial batch: stt.cc:236-254) done in the same pass as the other streams' hop; its finish only decodes.  A rolling live set --
    streams end in different hops, new ones take their place -- gives the transcripts of STT_SpeechToText on every utterance, also
    for utterances shorter than a window / an exact multiple of the hop, and a flagged stream finished through plain coqui-stt.h."""
    from stt_amd import model as M
    rng = np.random.RandomState(8)
    lens = (rng.uniform(0.1, 3.0, size=40) * 16000).astype(int)
    lens[3] = 100; lens[7] = 5120 * 2; lens[11] = 5120 * 2 + 1; lens[13] = 512
    audio = [synth.synth_audio(int(n), seed=1500 + i) for i, n in enumerate(lens)]
    want = [model.stt(a) for a in audio]
    got = [None] * len(audio)
    S, nxt, live = 9, 0, []
    while nxt < len(audio) or live:
        while len(live) < S and nxt < len(audio):
            live.append([nxt, model.createStream(), 0]); nxt += 1
        M.feedAudioContentBatch([s for _, s, _ in live], [audio[u][k:k + 5120] for u, _, k in live], last=[k + 5120 >= len(audio[u]) for u, _, k in live])
        inter = M.intermediateDecodeBatch([s for _, s, _ in live])
        for e in live:
            e[2] += 5120
        done = [e for e in live if e[2] >= len(audio[e[0]])]
        for (u, s, _), t_inter in zip(live, inter):
            if any(u == d[0] for d in done):
                assert t_inter == want[u], u                 # everything has gone through the model: the intermediate result is the final one
        if done:
            if len(done) == 1:
                got[done[0][0]] = done[0][1].finishStream()  # plain STT_FinishStream on a flagged stream: no second flush
            else:
                for e, t in zip(done, M.finishStreamBatch([e[1] for e in done])):
                    got[e[0]] = t
            live = [e for e in live if e[2] < len(audio[e[0]])]
    assert got == want


def test_config3_deferred_flush_tail(model):
    """aLast == 2: the flagged stream's flush leaves a handful of windows behind the call's first pass; they ride in the stream's next
    batched call (empty buffer, beside the live streams' hop) or -- if none comes -- are processed by its finish, batched or plain.
    Same transcripts as STT_SpeechToText either way."""
    from stt_amd import model as M
    rng = np.random.RandomState(9)
    lens = (rng.uniform(0.1, 4.0, size=48) * 16000).astype(int)
    lens[2] = 100; lens[5] = 5120 * 3; lens[6] = 5120 * 3 + 1; lens[9] = 512; lens[10] = 5120 * 3 - 1
    audio = [synth.synth_audio(int(n), seed=1700 + i) for i, n in enumerate(lens)]
    want = [model.stt(a) for a in audio]
    got = [None] * len(audio)
    S, nxt, live, drain, hop = 9, 0, [], [], 0
    empty = np.zeros(0, dtype=np.int16)
    while nxt < len(audio) or live or drain:
        while len(live) + len(drain) < S and nxt < len(audio):
            live.append([nxt, model.createStream(), 0]); nxt += 1
        ride = drain if hop % 3 else []              # every third hop the drained streams go straight to their finish: it does the tail
        M.feedAudioContentBatch([s for _, s, _ in live] + [s for _, s, _ in ride], [audio[u][k:k + 5120] for u, _, k in live] + [empty] * len(ride),
                                last=[2 if k + 5120 >= len(audio[u]) else 0 for u, _, k in live] + [0] * len(ride))
        if hop % 2:                                  # STTX_DecodeStreamsBatch: the hop's intermediate results and the finishes in one launch
            inter = M.intermediateDecodeBatch([s for _, s, _ in live]) if hop % 4 == 1 else None
            out = M.decodeStreamsBatch([s for _, s, _ in live] + [s for _, s, _ in drain], [False] * len(live) + [True] * len(drain))
            assert inter is None or out[:len(live)] == inter
            for e, t in zip(drain, out[len(live):]):
                got[e[0]] = t
                with pytest.raises(RuntimeError):
                    e[1].intermediateDecode()        # destroyed
        else:
            M.intermediateDecodeBatch([s for _, s, _ in live])
            if len(drain) == 1:
                got[drain[0][0]] = drain[0][1].finishStream()
            elif drain:
                for e, t in zip(drain, M.finishStreamBatch([e[1] for e in drain])):
                    got[e[0]] = t
        for e in live:
            e[2] += 5120
        drain = [e for e in live if e[2] >= len(audio[e[0]])]
        live = [e for e in live if e[2] < len(audio[e[0]])]
        hop += 1
    assert got == want



def test_config3_hop_replayed_as_one_graph_gives_the_same_results(model):
    """The acoustic + search pass of a batched hop is captured into one hipGraph per live-set shape the second time the shape comes up and
    replayed from then on (engine.cpp: streams_process; tunable stream_graph).  A fixed live set of 12 streams over 30 hops: every
    intermediate result and every final transcript equals the launch-by-launch run, and the graph path really ran (counter hop_replays)."""
    from stt_amd import model as M, native
    audio = [synth.synth_audio(5120 * 30, seed=1900 + i) for i in range(12)]

    def run(graph):
        native.set_tuning("stream_graph", graph)
        streams = [model.createStream() for _ in audio]
        inter = []
        for k in range(0, 5120 * 30, 5120):
            M.feedAudioContentBatch(streams, [a[k:k + 5120] for a in audio])
            inter.append(M.intermediateDecodeBatch(streams))
        return inter, M.finishStreamBatch(streams)
    try:
        before = native.get_tuning("hop_replays")
        with_graph = run(1)
        replays = native.get_tuning("hop_replays") - before
        without = run(0)
    finally:
        native.set_tuning("stream_graph", 1)
    assert with_graph == without
    assert replays >= 20, replays
    assert with_graph[1] == [model.stt(a) for a in audio]
