"""STTX_BatchSubmitDevice / STTX_BatchCollect: several batches in flight give the transcripts of the blocking call (needs a MI355X)."""
import numpy as np
import pytest

from stt_amd import modelfile, native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    from stt_amd import Model
    w = synth.synth_weights(11, n_hidden=256)
    path = str(tmp_path_factory.mktemp("m") / "small.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=64)
    return Model(path)


class _DeviceArray:
    """int16 [B][stride] in HBM through the HIP runtime libstt.so is bound to in this process -- whichever copy of
    libamdhip64 was mapped first (the ROCm one, or the one bundled with torch when an earlier test imported torch)."""
    _hip = None
    ptr = 0

    def __init__(self, host):
        import ctypes
        if _DeviceArray._hip is None:
            path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln][0]
            _DeviceArray._hip = ctypes.CDLL(path)
            _DeviceArray._hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            _DeviceArray._hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            _DeviceArray._hip.hipFree.argtypes = [ctypes.c_void_p]
        p = ctypes.c_void_p()
        assert _DeviceArray._hip.hipMalloc(ctypes.byref(p), host.nbytes) == 0
        assert _DeviceArray._hip.hipMemcpy(p, host.ctypes.data, host.nbytes, 1) == 0   # hipMemcpyHostToDevice
        self.ptr = p.value

    def data_ptr(self):
        return self.ptr

    def __del__(self):
        if self.ptr:
            _DeviceArray._hip.hipFree(self.ptr)


def _device_batches(n_batches, B, seed):
    out = []
    for k in range(n_batches):
        lens = [8000 + 1777 * ((i * 7 + k) % 9) for i in range(B)]
        stride = max(lens)
        host = np.zeros((B, stride), dtype=np.int16)
        for i, n in enumerate(lens):
            host[i, :n] = synth.synth_audio(n, seed=seed + 100 * k + i)
        out.append((_DeviceArray(host), stride, lens))
    return out


def test_pipelined_batches_equal_blocking_calls(model):
    batches = _device_batches(7, 9, seed=40)
    want = [model.sttBatchDevice(d.data_ptr(), stride, lens) for d, stride, lens in batches]
    depth = model.pipelineDepth()
    assert 1 <= depth <= 8
    got, inflight = [], []
    for d, stride, lens in batches:                     # keep the pipeline full: collect the oldest only when there is no room
        if len(inflight) == depth:
            got.append(model.collectBatch(inflight.pop(0)))
        inflight.append(model.submitBatchDevice(d.data_ptr(), stride, lens))
    while inflight:
        got.append(model.collectBatch(inflight.pop(0)))
    assert got == want
    assert any(any(s for s in b) for b in want), "empty transcripts prove nothing"
    # the blocking entry point still works afterwards (no batch left in flight)
    assert model.sttBatchDevice(batches[0][0].data_ptr(), batches[0][1], batches[0][2]) == want[0]


def test_blocking_call_with_more_groups_than_slots(model):
    """> 64 * depth utterances in one blocking call: the groups go round the slots (and take the pipelined chunk schedule)."""
    depth = model.pipelineDepth()
    n = 64 * depth + 70
    lens = [4000 + 531 * (i % 13) for i in range(n)]
    stride = max(lens)
    host = np.zeros((n, stride), dtype=np.int16)
    for i, ln in enumerate(lens):
        host[i, :ln] = synth.synth_audio(ln, seed=300 + (i % 23))
    d = _DeviceArray(host)
    got = model.sttBatchDevice(d.data_ptr(), stride, lens)
    ref = {}
    for i in range(23 * 13):                            # utterance i only depends on (i % 23, i % 13)
        if i < n and (i % 23, i % 13) not in ref:
            ref[(i % 23, i % 13)] = model.stt(host[i, :lens[i]])
    assert all(got[i] == ref[(i % 23, i % 13)] for i in range(n) if (i % 23, i % 13) in ref)
    assert any(got)


def test_pipeline_misuse_is_an_error(model):
    d, stride, lens = _device_batches(1, 4, seed=90)[0]
    depth = model.pipelineDepth()
    tickets = [model.submitBatchDevice(d.data_ptr(), stride, lens) for _ in range(depth)]
    with pytest.raises(RuntimeError):
        model.submitBatchDevice(d.data_ptr(), stride, lens)          # one more: every slot is taken
    with pytest.raises(RuntimeError):
        model.sttBatchDevice(d.data_ptr(), stride, lens)             # the blocking call needs the slots
    a = model.collectBatch(tickets[0])
    with pytest.raises(RuntimeError):
        model.collectBatch(tickets[0])                               # collected already
    rest = [model.collectBatch(t) for t in tickets[1:]]
    assert all(b == a for b in rest)
    with pytest.raises(RuntimeError):
        model.submitBatchDevice(d.data_ptr(), stride, [100] * 65)    # more than one group


def test_search_bound_setup_takes_four_slots(tmp_path):
    """A beam beyond 512 (or a code-point scorer) makes the search the long pole: STTX_BatchPipelineDepthFor says 4 and the
    four batches in flight (all searches side by side) still give the blocking call's transcripts."""
    from stt_amd import Model
    w = synth.synth_weights(12, n_hidden=256)
    path = str(tmp_path / "wide_beam.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=600)
    m = Model(path)
    assert m.pipelineDepth() == 4
    batches = _device_batches(6, 5, seed=70)
    want = [m.sttBatchDevice(d.data_ptr(), stride, lens) for d, stride, lens in batches]
    got, inflight = [], []
    for d, stride, lens in batches:
        if len(inflight) == 4:
            got.append(m.collectBatch(inflight.pop(0)))
        inflight.append(m.submitBatchDevice(d.data_ptr(), stride, lens))
    while inflight:
        got.append(m.collectBatch(inflight.pop(0)))
    assert got == want
    m.setBeamWidth(100)                                  # nothing in flight: the depth follows the configuration
    assert m.pipelineDepth() == (4 if native.get_tuning("pair") else 2)   # two slots, two batches per slot when batches pair up



def test_collect_with_confidence_and_with_metadata(model):
    """STTX_BatchCollectScored / STTX_BatchCollectWithMetadata: a submitted batch comes back with the confidence (and tokens, timesteps)
    STT_SpeechToTextWithMetadata reports for the same audio (stt.cc:349-365), whichever way it is collected."""
    batches = _device_batches(4, 6, seed=77)
    hosts = []
    for k in range(4):
        lens = [8000 + 1777 * ((i * 7 + k) % 9) for i in range(6)]
        hosts.append([synth.synth_audio(n, seed=77 + 100 * k + i) for i, n in enumerate(lens)])
    want = [[model.sttWithMetadata(a, 1)["transcripts"][0] for a in utts] for utts in hosts]
    tickets = [model.submitBatchDevice(d.data_ptr(), stride, lens) for d, stride, lens in batches]
    for k, t in enumerate(tickets):
        if k % 2 == 0:
            texts, conf = model.collectBatchScored(t)
            assert texts == [w["text"] for w in want[k]]
            assert conf == [w["confidence"] for w in want[k]]
        else:
            md = model.collectBatchWithMetadata(t)
            got = [m["transcripts"][0] for m in md]
            assert [g["text"] for g in got] == [w["text"] for w in want[k]]
            assert [g["confidence"] for g in got] == [w["confidence"] for w in want[k]]
            assert [[tk[1] for tk in g["tokens"]] for g in got] == [[tk[1] for tk in w["tokens"]] for w in want[k]]


@pytest.mark.parametrize("pair", [1, 0])
def test_host_audio_submit_equals_the_device_path(model, pair):
    """STTX_BatchSubmit (the ABI's own contract: host buffers, coqui-stt.h:294-297): staged through page-locked memory, copied on its own
    queue, the group gated by the copy's event -- 25 batches (the staging ring wraps twice), ragged lengths incl. an empty utterance,
    the caller's buffers overwritten right after the call; transcripts and confidences of the device-resident path."""
    native.set_tuning("pair", pair)
    try:
        rng = np.random.RandomState(5)
        n_batches, B = 25, 9
        host = []
        for k in range(n_batches):
            lens = [0 if (i == 4 and k % 3 == 0) else 6000 + int(rng.randint(0, 20000)) for i in range(B)]
            host.append([synth.synth_audio(n, seed=700 + 50 * k + i) for i, n in enumerate(lens)])
        want = [model.sttBatch(h) for h in host]
        depth = model.pipelineDepth()
        got, inflight = [], []
        for h in host:
            if len(inflight) == depth:
                got.append(model.collectBatchScored(inflight.pop(0)))
            mine = [a.copy() for a in h]
            inflight.append(model.submitBatch(mine))
            for a in mine:
                a[:] = 12345                                  # the buffers belong to the caller again
        while inflight:
            got.append(model.collectBatchScored(inflight.pop(0)))
        assert [g[0] for g in got] == want
        assert any(any(s for s in b) for b in want)
        # the same batches through the device entry: equal confidences, bit for bit
        for k in (0, 11, 24):
            stride = max(8, max(len(a) for a in host[k]))
            arr = np.zeros((B, stride), dtype=np.int16)
            for i, a in enumerate(host[k]):
                arr[i, :len(a)] = a
            d = _DeviceArray(arr)
            t = model.submitBatchDevice(d.data_ptr(), stride, [len(a) for a in host[k]])
            texts, conf = model.collectBatchScored(t)
            assert texts == got[k][0] and list(conf) == list(got[k][1])
    finally:
        native.set_tuning("pair", 1)


def test_host_audio_submit_refuses_null_buffers_and_serves_large_batches(model, tmp_path):
    """STTX_BatchSubmit: a NULL buffer with a non-zero sample count is an error code (round 5 dereferenced it), a NULL buffer of length 0 is an
    empty utterance; batches above 1 MB go through the persistent gather pool (eight parts on the caller's thread + three workers) --
    several in a row, from two threads at once (the second gathers on its own thread), all equal to a blocking call."""
    import ctypes
    import threading
    L = native.lib()
    a = synth.synth_audio(8000, seed=1)
    bufs = (ctypes.c_void_p * 2)(a.ctypes.data, None)
    sizes = (ctypes.c_uint * 2)(len(a), 100)
    assert L.STTX_BatchSubmit(model._impl, ctypes.cast(bufs, ctypes.POINTER(ctypes.c_void_p)), sizes, 2) < 0
    assert L.STTX_BatchSubmit(model._impl, None, sizes, 2) < 0
    sizes[1] = 0
    t = L.STTX_BatchSubmit(model._impl, ctypes.cast(bufs, ctypes.POINTER(ctypes.c_void_p)), sizes, 2)
    assert t > 0
    texts, _ = model.collectBatchScored(t)
    assert texts == model.sttBatch([a, a[:0]])
    big = [synth.synth_audio(40000 + 997 * i, seed=900 + i) for i in range(16)]        # 1.4 MB: the pooled gather
    want = model.sttBatch(big)
    for _ in range(3):
        assert model.collectBatchScored(model.submitBatch([x.copy() for x in big]))[0] == want
    # the pool under two callers: a second model's submit while the first gathers
    from stt_amd import Model
    path = str(tmp_path / "small2.sttw")
    modelfile.write_model(path, synth.synth_weights(11, n_hidden=256), synth.ENGLISH_LABELS, beam_width=64)
    other = Model(path)
    if other is not None:
        out = {}

        def run(m, key):
            out[key] = [m.collectBatchScored(m.submitBatch([x.copy() for x in big]))[0] for _ in range(4)]
        th = [threading.Thread(target=run, args=(model, "a")), threading.Thread(target=run, args=(other, "b"))]
        [t_.start() for t_ in th]
        [t_.join() for t_ in th]
        assert all(r == want for r in out["a"])
        assert all(r == want for r in out["b"])
