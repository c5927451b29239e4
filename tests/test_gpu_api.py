"""End-to-end through coqui-stt.h on the GPU: the reference's behavioural contract and
end-to-end parity (GPU acoustic probabilities -> oracle decoder == GPU transcript)."""
import os

import numpy as np
import pytest

from conftest import canon, dump
from stt_amd import modelfile, native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(tmp_path_factory, fix):
    from stt_amd import Model
    w = synth.synth_weights(21, n_hidden=256)
    # make the random-init model a bit "peaky" so transcripts are non-trivial: scale the output layer
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    path = str(tmp_path_factory.mktemp("api") / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=64)
    m = Model(path)
    return m


def test_model_properties_and_scorer_contract(model, fix):
    assert model.sampleRate() == 16000 and model.beamWidth() == 64
    model.setBeamWidth(32); assert model.beamWidth() == 32; model.setBeamWidth(64)
    L = native.lib()
    # without a scorer the hot-word calls fail with STT_ERR_SCORER_NOT_ENABLED (stt.cc:451-506; BasicTest.java of the reference)
    assert L.STT_AddHotWord(model._impl, b"x", 1.0) == 0x2004
    assert L.STT_ClearHotWords(model._impl) == 0x2004 and L.STT_DisableExternalScorer(model._impl) == 0x2004
    assert L.STT_SetScorerAlphaBeta(model._impl, 1.0, 1.0) == 0x2004
    assert L.STT_EnableExternalScorer(model._impl, b"/no/such/file") == 0x2002          # collapses to INVALID_SCORER, stt.cc:428-430
    model.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    model.addHotWord("dark", 2.0)
    assert L.STT_AddHotWord(model._impl, b"dark", 3.0) == 0x3008                        # duplicate insert
    assert L.STT_EraseHotWord(model._impl, b"nope") == 0x3010
    model.eraseHotWord("dark"); model.clearHotWords()
    assert model.disableExternalScorer() == 0


def _oracle_transcript(port, english, fix, probs, beam, lm):
    labels, space = english
    P = port.Scorer(os.path.join(fix, "pruned_lm.scorer")) if lm else None
    d = port.Decoder(labels, space, beam, P)
    d.next(probs)
    return d.decode(1)[0]


@pytest.mark.parametrize("lm", [False, True])
def test_one_shot_streaming_and_batch_agree_with_oracle_decoder(model, port, english, fix, lm):
    if lm:
        model.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    audio = [synth.synth_audio(n, seed=40 + i) for i, n in enumerate([24000, 9000, 46797 // 2, 700, 0])]
    one_shot = [model.stt(a) for a in audio]
    batch = model.sttBatch(audio)
    assert batch == one_shot
    probs = model.acousticProbs(audio)
    for i, a in enumerate(audio):
        conf, tok, ts = _oracle_transcript(port, english, fix, probs[i], 64, lm)
        want = b"".join(english[0][t] for t in tok).decode()
        assert one_shot[i] == want, (i, one_shot[i], want)
        # streaming in 320 ms hops (5120 samples = 16 frames) and in ragged hops
        for hop in (5120, 777):
            s = model.createStream()
            for k in range(0, len(a), hop):
                s.feedAudioContent(a[k:k + hop])
            assert s.finishStream() == want, (i, hop)
        md = model.sttWithMetadata(a, 3)
        assert md["transcripts"][0]["text"] == want
        assert [t[1] for t in md["transcripts"][0]["tokens"]] == [int(x) for x in ts]
        assert abs(md["transcripts"][0]["confidence"] - conf) == 0.0
        for t in md["transcripts"][0]["tokens"]:
            assert abs(t[2] - t[1] * (320 / 16000)) < 1e-6                 # start_time, modelstate.cc:57
    if lm:
        model.disableExternalScorer()


def test_intermediate_decode_semantics(model):
    a = synth.synth_audio(32000, seed=77)
    s = model.createStream()
    s.feedAudioContent(a[:20000])
    x1 = s.intermediateDecode(); x2 = s.intermediateDecode()
    assert x1 == x2                                                        # const, stt.cc:596-600
    s.feedAudioContent(a[20000:])
    final = s.finishStream()
    assert final == model.stt(a)
    with pytest.raises(RuntimeError):
        s.feedAudioContent(a[:10])                                          # stream is gone after finish
    # flush variant processes the partial window/batch early (coqui-stt.h:393-399 of the reference) and may differ afterwards
    s = model.createStream(); s.feedAudioContent(a[:20000])
    assert isinstance(s.intermediateDecodeFlushBuffers(), str)
    s.feedAudioContent(a[20000:]); assert isinstance(s.finishStream(), str)


def test_two_streams_interleaved_on_one_model(model):
    """native_client/test/concurrent_streams.py:43-54"""
    a = synth.synth_audio(30000, seed=1); b = synth.synth_audio(26000, seed=2)
    s1, s2 = model.createStream(), model.createStream()
    for k in range(0, 30000, 2048):
        s1.feedAudioContent(a[k:k + 2048]); s2.feedAudioContent(b[k:k + 2048])
    assert s1.finishStream() == model.stt(a) and s2.finishStream() == model.stt(b)


def test_with_emissions(model):
    a = synth.synth_audio(12000, seed=5)
    md = model.sttWithEmissions(a, 1)
    e = md["emissions"]
    assert e.shape[1] == 29 and md["symbols"][-1] == "\t" and len(md["symbols"]) == 29
    assert np.allclose(e.sum(1), 1.0, atol=1e-4)
    probs = model.acousticProbs([a])[0]
    T = len(probs)
    last = T % 16 or 16                                                     # stt.cc:326-329 keeps only the last batch
    assert e.shape[0] == last and np.array_equal(e.astype(np.float32), probs[T - last:])
