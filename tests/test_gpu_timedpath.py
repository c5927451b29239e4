"""The kernel instances and the schedule bench.py TIMES, tied to the oracle numerically (needs a MI355X).

bench.py submits 64 x 5 s batches to STTX_BatchSubmitDevice: three acoustic engines on three streams, x-projections and recurrent
outputs handed over through ring slots, the recurrence of a chunk replayed as one hipGraph, the eight-wave one-per-CU GEMM form,
the owner-form recurrent step -- 64 rows per step, or 128 when two submitted batches share one recurrence (tunable `pair`).
None of that is what STTX_AcousticProbs runs (one stream, blocking forms), which is what the other numeric tests go through.
Here:
  * STTX_DebugBatchProbs returns the probabilities of a batch in flight exactly as that path computed them: bitwise equal to the
    blocking path (same k order in every GEMM form, same reduction order in every recurrent form) for the 1st batch (eager
    launches), the 2nd (first graph capture) and the 5th (graph replay, ring slots wrapped), and within the stated tolerance of
    the f64 restatement (|dp| <= 1e-4, |d ln p| <= 2e-3; deepspeech_model.py:144-168, native_client/stt.cc:311-334).
  * the recurrent step kernel alone: every form x prefetch depth x {eager, graph} over 250 steps gives the same bits, at 17, 64,
    100 and 128 batch rows, and a row's result does not depend on how many rows share the launch.
  * streams fed / finished and STT_SpeechToText called WHILE batches are in flight (the memory fault of round 2, build "g").
"""
import numpy as np
import pytest

from stt_amd import modelfile, native, synth
from test_gpu_async import _DeviceArray

pytestmark = pytest.mark.gpu

H, B, N = 2048, 64, 80000


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    from stt_amd import Model
    w = synth.synth_weights(0, n_hidden=H)             # the bench's weights
    path = str(tmp_path_factory.mktemp("timed") / "english.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
    m = Model(path)
    m.enableExternalScorer(__import__("os").path.join(__import__("conftest").FIX, "pruned_lm.scorer"))
    return m, w


def _batch_audio(k):
    return [synth.synth_audio(N, seed=1000 * k + i) for i in range(B)]


@pytest.mark.parametrize("pair", [1, 0])
def test_probabilities_of_batches_in_flight_equal_the_blocking_path_and_the_oracle(big, pair):
    from oracle import am_ref
    model, w = big
    native.set_tuning("pair", pair)
    try:
        depth = model.pipelineDepth()
        assert depth == (4 if pair else 2)
        n_batches = 6
        audio = {k: _batch_audio(k) for k in (0, 1, 4)}
        filler = _batch_audio(9)
        dev = {k: _DeviceArray(np.stack(audio.get(k, filler))) for k in range(n_batches)}
        sizes = [N] * B
        inflight, probs, texts = [], {}, {}

        def retire():
            k, t = inflight.pop(0)
            if k in audio:
                probs[k] = model.batchProbs(t, B)        # before the collect: the slot still holds the block its search reads
            texts[k] = model.collectBatch(t)

        for k in range(n_batches):
            if len(inflight) == depth:
                retire()
            inflight.append((k, model.submitBatchDevice(dev[k].data_ptr(), N, sizes)))
        while inflight:
            retire()
        for k in (0, 1, 4):
            want = model.acousticProbs(audio[k])         # one stream, blocking kernel forms, whole utterance as one chunk
            assert all(p.shape == (250, 29) for p in probs[k])
            for i in range(B):
                assert np.array_equal(probs[k][i], want[i]), (pair, k, i, float(np.abs(probs[k][i] - want[i]).max()))
            for i in (5, 58):                            # and against the f64 restatement, at the stated tolerance
                ref = am_ref.utterance_probs(audio[k][i], w, weight_round=np.float16)
                a = float(np.abs(probs[k][i] - ref).max())
                l = float(np.abs(np.log(probs[k][i]) - np.log(ref)).max())
                assert a < 1e-4 and l < 2e-3, (pair, k, i, a, l)
            assert texts[k] == model.sttBatchDevice(dev[k].data_ptr(), N, sizes), (pair, k)
        assert texts[1] != texts[4] or any(texts[1])     # (different audio per batch: nothing can be a cached answer)
    finally:
        native.set_tuning("pair", 1)


def _xproj(period, rows, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((period, rows, 4 * H)) * 1.5).astype(np.float32)


def test_recurrent_step_forms_graph_and_eager_give_the_same_bits(big):
    """lstm_step_kernel<NT, G, MT, PHS> / lstm_step8_kernel<G>: PHS 1 (one pass), 2 (two passes), 3 (owner form) x G 1, 2, 4,
    launched one by one or replayed from one hipGraph, 250 dependent steps."""
    model, _ = big
    T, P = 250, 8
    x128 = _xproj(P, 128, 3)
    try:
        outs = {}
        for rows in (17, 64, 100, 128):
            x = np.ascontiguousarray(x128[:, :rows]).reshape(P * rows, 4 * H)
            base = None
            forms = (1, 2, 3) if rows == 64 else (0,)            # other row counts have one reduction form each
            for form in forms:
                for pf in ((1, 2, 4) if rows == 64 else (2,)):
                    for graph in (False, True):
                        native.set_tuning("lstm_form", form); native.set_tuning("lstm_prefetch", pf)
                        c, h, hall = model.lstmSteps(x, rows, T, graph=graph)
                        assert np.isfinite(c).all() and np.abs(h).max() > 1e-3
                        if base is None:
                            base = (c, h, hall)
                        else:
                            assert np.array_equal(c, base[0]) and np.array_equal(h, base[1]) and np.array_equal(hall, base[2]), (rows, form, pf, graph)
            outs[rows] = base
        # a row's 250-step trajectory does not depend on how many rows share the launch (1-, 2-, 4- and 8-tile instances)
        for rows in (17, 64, 100):
            c, h, hall = outs[rows]
            c8, h8, hall8 = outs[128]
            assert np.array_equal(c, c8[:rows]) and np.array_equal(h, h8[:rows]), rows
            assert np.array_equal(hall.reshape(P, rows, H), hall8.reshape(P, 128, H)[:, :rows]), rows
        # and it is the LSTM cell (deepspeech_model.py:144-168): one step from zero state against f64
        x1 = np.ascontiguousarray(x128[:1, :64]).reshape(64, 4 * H)
        c, h, _ = model.lstmSteps(x1, 64, 1)
        i, j, f, o = np.split(x1.astype(np.float64), 4, axis=1)
        sig = lambda v: 1 / (1 + np.exp(-v))
        c_ref = sig(i) * np.tanh(j)
        assert np.abs(c - c_ref).max() < 1e-5 and np.abs(h - sig(o) * np.tanh(c_ref)).max() < 1e-5
    finally:
        native.set_tuning("lstm_form", 0); native.set_tuning("lstm_prefetch", 2)


def test_streams_and_blocking_calls_while_batches_are_in_flight(big):
    """Round 2, build "g": the streaming calls shared state buffers with the batch path's engines and a growing DevBuf was freed
    under a running kernel -- "Memory access fault by GPU node".  Feed / decode / finish streams and call STT_SpeechToText while
    the pipeline is full; everything must equal what the same calls give on an idle model."""
    model, _ = big
    audio = _batch_audio(2)
    dev = _DeviceArray(np.stack(audio))
    sizes = [N] * B
    long_a = synth.synth_audio(11 * 16000, seed=77)          # longer than anything seen so far: the streaming buffers grow
    short_a = synth.synth_audio(24000, seed=78)

    def side_calls():
        out = []
        s = model.createStream()
        for k in range(0, len(long_a), 5120):
            s.feedAudioContent(long_a[k:k + 5120])
            if k % (5120 * 8) == 0:
                out.append(s.intermediateDecode())
        out.append(s.finishStream())
        out.append(model.stt(short_a))
        s2 = model.createStream()
        s2.feedAudioContent(short_a)
        out.append(s2.finishStreamWithMetadata(2)["transcripts"][0]["text"])
        return out

    want_batch = model.sttBatchDevice(dev.data_ptr(), N, sizes)
    want_side = side_calls()
    depth = model.pipelineDepth()
    for round_ in range(2):
        tickets = [model.submitBatchDevice(dev.data_ptr(), N, sizes) for _ in range(depth)]
        got_side = side_calls()                              # the GPU is busy with `depth` batches the whole time
        got = [model.collectBatch(t) for t in tickets]
        assert got_side == want_side, round_
        assert all(g == want_batch for g in got), round_
    assert any(want_batch) or any(want_side)
