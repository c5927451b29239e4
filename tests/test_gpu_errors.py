"""A stream whose search arenas overflow has a damaged beam.  The reference cannot get there (it allocates nodes on the
heap); here the arenas are sized up front and grown between chunks, and if that ever fails every decode path must say so
-- NULL from the STT_* calls (coqui-stt.h: "NULL on error"), STT_ERR_FAIL_RUN_SESS from the STTX_* ones -- instead of
returning a transcript.  STTX_DebugLimitArena makes the arenas too small on purpose."""
import os

import numpy as np
import pytest

from stt_amd import modelfile, synth

pytestmark = pytest.mark.gpu


def test_arena_overflow_is_reported_by_every_decode_path(tmp_path, fix):
    from stt_amd import Model, native
    w = synth.synth_weights(5, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    path = str(tmp_path / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=64)
    m = Model(path)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    a = synth.synth_audio(32000, seed=3)
    good = m.stt(a)
    L = native.lib()
    L.STTX_DebugLimitArena(2)            # room for 4 timesteps of a 64-wide beam; the utterance has 100
    try:
        with pytest.raises(RuntimeError):
            m.stt(a)
        assert m.sttWithMetadata(a, 2) is None
        with pytest.raises(RuntimeError):
            m.sttBatch([a, a])
        s = m.createStream()
        s.feedAudioContent(a)
        assert s.intermediateDecode() is None
        assert s.finishStream() is None
        d = m.createDecoder(1, 64)
        x = np.random.RandomState(0).rand(40, 29).astype(np.float32)
        d.next(x / x.sum(1, keepdims=True))
        with pytest.raises(RuntimeError):
            d.decode(1)
        assert d.stats()["error"] != 0
    finally:
        L.STTX_DebugLimitArena(0)
    assert m.stt(a) == good              # streams created afterwards are healthy again
    assert L.STT_SetModelBeamWidth(m._impl, 5000) != 0 and m.beamWidth() == 64   # beyond the LDS beam: refused when set
