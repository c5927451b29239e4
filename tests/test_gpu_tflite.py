"""STT_CreateModel on a `.tflite` export: the model behaves exactly like the raw container holding
the tensors a TFLite interpreter would compute with (de-quantised int8 / f16), through coqui-stt.h."""
import os

import numpy as np
import pytest

from stt_amd import modelfile, native, synth, tflitefile

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [{}, {"quantize": True}, {"quantize": True, "per_channel": True}, {"f16_weights": True}],
                         ids=["f32", "int8", "int8-per-channel", "f16"])
def test_tflite_model_equals_container_with_the_same_tensors(tmp_path, fix, kw):
    from stt_amd import Model
    w = synth.synth_weights(21, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    p_tfl, p_raw = str(tmp_path / "m.tflite"), str(tmp_path / "m.sttw")
    eff = tflitefile.write_tflite(p_tfl, w, synth.ENGLISH_LABELS, beam_width=64, **kw)
    modelfile.write_model(p_raw, {n: eff[n] for n in modelfile.TENSOR_ORDER}, synth.ENGLISH_LABELS, beam_width=64)
    # (am_i8 = 0: the reader's de-quantised tensors on the f16 path; by default a quantised file takes TFLite's hybrid int8 arithmetic instead --
    # tests/test_gpu_i8_path.py)
    native.set_tuning("am_i8", 0)
    try:
        a, b = Model(p_tfl), Model(p_raw)
    finally:
        native.set_tuning("am_i8", -1)
    assert a.acousticMode() == 0 and b.acousticMode() == 0
    assert a.sampleRate() == 16000 and a.beamWidth() == 64
    for m in (a, b):
        m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    audio = [synth.synth_audio(int(16000 * s), seed=40 + i) for i, s in enumerate((0.4, 1.0, 2.3))]
    pa, pb = a.acousticProbs(audio), b.acousticProbs(audio)
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)
    assert a.sttBatch(audio) == b.sttBatch(audio)
    assert a.sttWithMetadata(audio[2], 3) == b.sttWithMetadata(audio[2], 3)
    s1, s2 = a.createStream(), b.createStream()
    for k in range(0, len(audio[2]), 4000):
        s1.feedAudioContent(audio[2][k:k + 4000]); s2.feedAudioContent(audio[2][k:k + 4000])
        assert s1.intermediateDecode() == s2.intermediateDecode()
    assert s1.finishStream() == s2.finishStream()


def test_tflite_error_paths(tmp_path):
    L = native.lib()
    import ctypes as C
    w = synth.synth_weights(1, n_hidden=128)
    data, _ = tflitefile.tflite_bytes(w, synth.ENGLISH_LABELS, graph_version=5)
    ctx = C.c_void_p()
    assert L.STT_CreateModelFromBuffer(data, len(data), C.byref(ctx)) == 0x2003 and not ctx.value      # tflitemodelstate.cc:256-264
    data, _ = tflitefile.tflite_bytes(w, synth.ENGLISH_LABELS[:-1])                                       # alphabet one label short
    assert L.STT_CreateModelFromBuffer(data, len(data), C.byref(ctx)) == 0x2000 and not ctx.value      # :319-329 STT_ERR_INVALID_ALPHABET
