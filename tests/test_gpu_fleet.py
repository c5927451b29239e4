"""Several GPUs behind the C ABI (include/stt_amd.h: STTX_Fleet*, stt_amd/csrc/fleet.cpp): replicas, LPT dealing, one host thread
per device, RCCL all-gather of the transcripts.  On a one-GPU box the fleet has one member (the RCCL exchange is then a
one-rank all-gather: the plumbing, not the bandwidth); with more GPUs visible every one of them takes part."""
import os

import numpy as np
import pytest

from stt_amd import modelfile, synth

pytestmark = pytest.mark.gpu


def test_fleet_transcripts_equal_the_single_model_batch(tmp_path, fix):
    from stt_amd import Model, native
    from stt_amd.model import Fleet
    w = synth.synth_weights(5, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    path = str(tmp_path / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=100)
    scorer = os.path.join(fix, "pruned_lm.scorer")
    rng = np.random.RandomState(2)
    audio = [synth.synth_audio(int(n), seed=80 + i) for i, n in enumerate(rng.randint(0, 60000, size=23))]
    m = Model(path)
    m.enableExternalScorer(scorer)
    want = m.sttBatch(audio)
    ndev = native.lib().STTX_GetDeviceCount()
    assert ndev >= 1
    f = Fleet(path, list(range(ndev)))
    assert f.size() == ndev
    f.enableExternalScorer(scorer)
    assert f.setBeamWidth(100) == 0
    assert f.sttBatch(audio) == want
    assert f.sttBatch(audio[:1]) == want[:1]                 # fewer utterances than devices is fine
    assert f.sttBatch([]) == []
    native.lib().STTX_SetDevice(0)


def test_a_failing_shard_returns_null_instead_of_hanging(tmp_path, fix):
    """A shard that fails before it decodes still enters the first all-gather and announces -1; every rank sees it and none enters
    the second one (round 2: the other ranks would have waited inside the collective forever).  The fleet stays usable."""
    from stt_amd import native
    from stt_amd.model import Fleet
    w = synth.synth_weights(6, n_hidden=256)
    path = str(tmp_path / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=50)
    ndev = native.lib().STTX_GetDeviceCount()
    f = Fleet(path, list(range(ndev)))
    audio = [synth.synth_audio(9000 + 1000 * i, seed=30 + i) for i in range(5)]
    want = f.sttBatch(audio)
    for shard in range(ndev):
        assert native.lib().STTX_DebugFleetFailShard(f._impl, shard) == 0
        with pytest.raises(RuntimeError):
            f.sttBatch(audio)
    native.lib().STTX_DebugFleetFailShard(f._impl, -1)
    assert f.sttBatch(audio) == want
    native.lib().STTX_SetDevice(0)
