"""The released models' own arithmetic as an ENGINE PATH (needs a MI355X): a dynamic-range quantised `.tflite` (export.py:145-146) runs every
FULLY_CONNECTED as TensorFlow Lite's hybrid kernel -- what the reference's CPU path does (native_client/tflitemodelstate.cc:200,369-405) --
restated in oracle/am_hybrid.py.  Layer by layer:
  * layers 1-3 (row quantisation, int8 GEMM, rescale + bias + clipped ReLU) and the x half of the cell's int32 sums: BIT EQUAL;
  * layers 5-6 on the engine's own recurrent outputs: BIT EQUAL;
  * one cell step from a given state: |d c|, |d h| <= 4e-6 (the integer sums and the rescale are exact; TFLite's LOGISTIC / TANH are float kernels
    and the engine's expf / tanhf differ from numpy's in the last bits) -- including rows whose max |h| exceeds max |x_t| (the joint
    scale is then the h half's: the step's slow path);
  * a chunk of 48 steps: |d h| <= 2e-4 (a last-bit difference in an activation can flip one int8 of the next step's quantised row);
  * the forms the batch path times (three engines, 128 rows per step, hipGraph) give the bits of the one-stream path; streams == one-shot.
"""
import os

import numpy as np
import pytest

from conftest import FIX
from stt_amd import native, synth, tflitefile

pytestmark = pytest.mark.gpu


def _model(tmp_path, w, name, per_channel=False, beam=100):
    from stt_amd import Model
    path = str(tmp_path / (name + ".tflite"))
    tflitefile.write_tflite(path, w, synth.ENGLISH_LABELS, quantize=True, per_channel=per_channel, beam_width=beam)
    m = Model(path)
    assert m.acousticMode() == 1
    return m


def _windows(rng, T, B, scale=1.0):
    w = (rng.standard_normal((T, B, 494)) * rng.uniform(0.5, 12.0, size=(T, B, 1)) * scale).astype(np.float32)
    w[0, 0] = 0.0                              # an all-zero row: range 0 -> zeros, scale 1
    return w


def _oracle_front(hm, win):
    T, B, _ = win.shape
    x = win.reshape(T * B, -1)
    return hm._dense(hm._dense(hm._dense(x, "layer_1"), "layer_2"), "layer_3")


def _oracle_cell(hm, x, c, h):
    from oracle import am_hybrid
    z = hm._fc(np.concatenate([x, h], axis=1), "lstm/kernel", hm.b["lstm/bias"])
    i, j, f, o = np.split(z, 4, axis=1)
    c = (am_hybrid._sigmoid(f) * c + am_hybrid._sigmoid(i) * am_hybrid._tanh(j)).astype(np.float32)
    h = (am_hybrid._sigmoid(o) * am_hybrid._tanh(c)).astype(np.float32)
    return c, h


@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("B", [1, 5, 20, 70])
def test_dense_layers_bit_equal_and_the_cell_within_its_bound(tmp_path, per_channel, B):
    from oracle import am_hybrid
    rng = np.random.default_rng(100 + B)
    H, T = 256, 9
    w = synth.synth_weights(3, n_hidden=H)
    m = _model(tmp_path, w, "c%d_%d" % (B, per_channel), per_channel)
    hm = am_hybrid.HybridModel(w, per_channel)
    win = _windows(rng, T, B)
    c0 = (rng.standard_normal((B, H)) * 0.5).astype(np.float32)
    h0 = np.tanh(rng.standard_normal((B, H))).astype(np.float32) * np.float32(0.7)
    out = m.hybridChain(win, c0, h0)
    # layers 1-3
    l3 = _oracle_front(hm, win)
    assert np.array_equal(out["l3"].reshape(T * B, H), l3), float(np.abs(out["l3"].reshape(T * B, H) - l3).max())
    # x half of the cell's sums: layer 3's rows quantised at their own scale times the kernel's x columns, exactly
    q3, _ = am_hybrid.symmetric_quantize_rows(l3)
    kq = hm.q["lstm/kernel"][0].astype(np.float64)               # [4H][2H]
    assert np.array_equal(out["accx"].reshape(T * B, 4 * H).astype(np.float64), q3 @ kq[:, :H].T)
    # the recurrence, step by step FROM THE ENGINE'S OWN previous state: one step's error, not an accumulated one
    l3 = l3.reshape(T, B, H)
    c, h = c0, h0
    for t in range(T):
        cw, hw = _oracle_cell(hm, l3[t], c, h)
        he = out["h_all"][t]
        assert float(np.abs(he - hw).max()) <= 4e-6, (t, float(np.abs(he - hw).max()))
        # continue from the engine's h (c is not exposed per step: the restatement's own, equal to within the same bound)
        c, h = cw, he
    assert np.array_equal(out["h"], out["h_all"][T - 1])
    assert float(np.abs(out["c"] - c).max()) <= 2e-5
    # layers 5-6 on the engine's own h: bit equal
    l5 = hm._dense(out["h_all"].reshape(T * B, H), "layer_5")
    logits = hm._fc(l5, "layer_6/weights", hm.b["layer_6/bias"])
    assert np.array_equal(out["logits"].reshape(T * B, -1), logits), float(np.abs(out["logits"].reshape(T * B, -1) - logits).max())
    p = hm.softmax(logits).reshape(T, B, -1).transpose(1, 0, 2)
    assert np.array_equal(out["probs"], p), float(np.abs(out["probs"] - p).max())      # the correctly rounded softmax, on both sides: bit equal
    if B > 1:
        assert out["slow_rows"] >= 0                # (rows with small windows may take the joint-scale path; the next test counts them against the restatement)


def test_rows_whose_h_outgrows_x_take_the_joint_scale(tmp_path):
    """max |h_(t-1)| > max |x_t|: TFLite quantises concat([x_t, h]) with the h half's range; the hoisted x half (quantised at max |x_t|) is
    not valid for that row and the step computes it again.  Layer 3 scaled down until nearly every row is such a row."""
    from oracle import am_hybrid
    rng = np.random.default_rng(7)
    H, T, B = 256, 6, 19
    w = synth.synth_weights(4, n_hidden=H)
    w["layer_3/weights"] = (w["layer_3/weights"] * 0.02).astype(np.float32)
    w["layer_3/bias"] = (w["layer_3/bias"] * 0.02).astype(np.float32)
    m = _model(tmp_path, w, "slow")
    hm = am_hybrid.HybridModel(w)
    win = _windows(rng, T, B)
    win[2, 3] *= 400.0                                # one row whose x is large again: the hoisted form, between slow ones
    c0 = (rng.standard_normal((B, H)) * 0.5).astype(np.float32)
    h0 = (np.tanh(rng.standard_normal((B, H))) * 0.9).astype(np.float32)
    h0[4] = 0.0                                       # and one that starts from a zero state
    out = m.hybridChain(win, c0, h0)
    l3 = _oracle_front(hm, win).reshape(T, B, H)
    assert np.array_equal(out["l3"], l3)
    c, h = c0, h0
    n_slow = 0
    for t in range(T):
        n_slow += int((np.abs(h).max(axis=1) > np.abs(l3[t]).max(axis=1)).sum())
        cw, hw = _oracle_cell(hm, l3[t], c, h)
        he = out["h_all"][t]
        assert float(np.abs(he - hw).max()) <= 4e-6, (t, float(np.abs(he - hw).max()))
        c, h = cw, he
    assert n_slow > T * B // 2 and out["slow_rows"] == n_slow, (n_slow, out["slow_rows"])
    assert m.slowRows() == n_slow                    # (STTX_DebugSlowRows: the model's running count, every engine form)


def test_a_chunk_of_steps_and_the_state_carried_between_calls(tmp_path):
    from oracle import am_hybrid
    rng = np.random.default_rng(8)
    H, T, B = 256, 48, 33
    w = synth.synth_weights(5, n_hidden=H)
    m = _model(tmp_path, w, "chunk")
    hm = am_hybrid.HybridModel(w)
    win = _windows(rng, T, B)
    want = hm.forward_batch(win.transpose(1, 0, 2))                     # [B][T][C], zero state
    whole = m.hybridChain(win)
    assert float(np.abs(whole["probs"] - want).max()) <= 2e-4
    # the same 48 steps as 16 + 32 with the state handed back and in again: the bits of the single call
    a = m.hybridChain(win[:16])
    b = m.hybridChain(win[16:], a["c"], a["h"])
    assert np.array_equal(np.concatenate([a["probs"], b["probs"]], axis=1), whole["probs"])
    assert np.array_equal(b["h"], whole["h"]) and np.array_equal(b["c"], whole["c"])
    # a row's result does not depend on what shares the launch (17 rows: two batch tiles; 1 row: one)
    one = m.hybridChain(win[:, 7:8])
    assert np.array_equal(one["probs"][0], whole["probs"][7])


@pytest.mark.parametrize("B", [128, 100, 64, 40])
def test_row_groups_of_the_step_give_the_same_bits(tmp_path, B):
    """The recurrent step deals its batch tiles to 1 / 2 / 4 / 8 workgroups per 16-unit slice (tunable lstm_i8_rows: 128 / 64 / 32 / 16 rows per
    workgroup): integer sums, per-row flags and maxima do not depend on which workgroup holds a row -- every form the bits of the one-group
    form, slow-path rows (layer 3 scaled down: about half the rows) included."""
    rng = np.random.default_rng(70 + B)
    H, T = 256, 12
    w = synth.synth_weights(6, n_hidden=H)
    w["layer_3/weights"] = (w["layer_3/weights"] * 0.05).astype(np.float32)
    w["layer_3/bias"] = (w["layer_3/bias"] * 0.05).astype(np.float32)
    m = _model(tmp_path, w, "rg%d" % B)
    win = _windows(rng, T, B)
    win[:, ::3] *= 60.0                                # a third of the rows on the hoisted form throughout
    c0 = (rng.standard_normal((B, H)) * 0.5).astype(np.float32)
    h0 = (np.tanh(rng.standard_normal((B, H))) * 0.9).astype(np.float32)
    try:
        native.set_tuning("lstm_i8_rows", 128)
        want = m.hybridChain(win, c0, h0)
        assert 0 < want["slow_rows"] < T * B
        for rows in (64, 32, 16):
            native.set_tuning("lstm_i8_rows", rows)
            got = m.hybridChain(win, c0, h0)
            for k in ("h_all", "c", "h", "probs"):
                assert np.array_equal(got[k], want[k]), (rows, k)
            assert got["slow_rows"] == want["slow_rows"], rows
    finally:
        native.set_tuning("lstm_i8_rows", 64)


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    w = synth.synth_weights(0, n_hidden=2048)
    m = _model(tmp_path_factory.mktemp("i8big"), w, "english_q", beam=500)
    m.enableExternalScorer(os.path.join(FIX, "pruned_lm.scorer"))
    return m, w


def test_bench_shape_one_step_and_a_chunk(big):
    """n_hidden 2048 (the 128 x 256 tile on every layer, the pinned 128-row step): one step at 128 rows against the restatement, layers bit equal."""
    from oracle import am_hybrid
    m, w = big
    hm = am_hybrid.HybridModel(w)
    rng = np.random.default_rng(9)
    H, T, B = 2048, 3, 128
    win = _windows(rng, T, B)
    c0 = (rng.standard_normal((B, H)) * 0.5).astype(np.float32)
    h0 = (np.tanh(rng.standard_normal((B, H))) * 0.6).astype(np.float32)
    out = m.hybridChain(win, c0, h0)
    l3 = _oracle_front(hm, win)
    assert np.array_equal(out["l3"].reshape(T * B, H), l3)
    l3 = l3.reshape(T, B, H)
    c, h = c0, h0
    for t in range(T):
        cw, hw = _oracle_cell(hm, l3[t], c, h)
        assert float(np.abs(out["h_all"][t] - hw).max()) <= 4e-6, t
        c, h = cw, out["h_all"][t]
    l5 = hm._dense(out["h_all"].reshape(T * B, H), "layer_5")
    assert np.array_equal(out["logits"].reshape(T * B, -1), hm._fc(l5, "layer_6/weights", hm.b["layer_6/bias"]))


def test_batches_in_flight_and_streams_give_the_bits_of_the_one_stream_path(big):
    from test_gpu_async import _DeviceArray
    m, w = big
    B, N = 64, 80000
    native.set_tuning("pair", 1)
    depth = m.pipelineDepth()
    audio = {k: [synth.synth_audio(N, seed=2000 * k + i) for i in range(B)] for k in range(3)}
    dev = {k: _DeviceArray(np.stack(audio[k % 3])) for k in range(6)}
    inflight, probs, texts = [], {}, {}

    def retire():
        k, t = inflight.pop(0)
        probs[k] = m.batchProbs(t, B)
        texts[k] = m.collectBatch(t)
    for k in range(6):
        if len(inflight) == depth:
            retire()
        inflight.append((k, m.submitBatchDevice(dev[k].data_ptr(), N, [N] * B)))
    while inflight:
        retire()
    for k in (0, 1, 5):
        want = m.acousticProbs(audio[k % 3])
        for i in range(B):
            assert np.array_equal(probs[k][i], want[i]), (k, i, float(np.abs(probs[k][i] - want[i]).max()))
        assert texts[k] == m.sttBatchDevice(dev[k].data_ptr(), N, [N] * B)
    # a stream fed in 320 ms hops == the one-shot call (stt.cc:641-688), on the int8 path
    a = audio[0][3]
    st = m.createStream()
    for off in range(0, N, 5120):
        st.feedAudioContent(a[off:off + 5120])
    assert st.finishStream() == m.stt(a) == texts[0][3]


def test_a_float_container_quantised_at_load_equals_the_quantised_file(tmp_path):
    """am_i8 = 1 quantises a float model's matrices the way the converter does (oracle/am_hybrid.py: quantize_weights): the same int8, the same bits."""
    from stt_amd import Model, modelfile
    w = synth.synth_weights(11, n_hidden=256)
    a = _model(tmp_path, w, "file")
    p_raw = str(tmp_path / "float.sttw")
    modelfile.write_model(p_raw, w, synth.ENGLISH_LABELS, beam_width=100)
    native.set_tuning("am_i8", 1)
    try:
        b = Model(p_raw)
    finally:
        native.set_tuning("am_i8", -1)
    assert b.acousticMode() == 1 and Model(p_raw).acousticMode() == 0
    audio = [synth.synth_audio(int(16000 * s), seed=60 + i) for i, s in enumerate((0.3, 1.1, 2.0))]
    for x, y in zip(a.acousticProbs(audio), b.acousticProbs(audio)):
        assert np.array_equal(x, y)
    for m in (a, b):
        m.enableExternalScorer(os.path.join(FIX, "pruned_lm.scorer"))
    assert a.sttBatch(audio) == b.sttBatch(audio)


def test_ragged_batches_on_the_int8_path(tmp_path):
    """Utterances of different lengths in one group: the rows of a short one run on zero windows behind its end (never read, and never
    worth the step's slow path: LstmI8Args::row_frames) -- every utterance's probabilities and transcript are those it gets alone."""
    w = synth.synth_weights(12, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    m = _model(tmp_path, w, "ragged")
    m.enableExternalScorer(os.path.join(FIX, "pruned_lm.scorer"))
    rng = np.random.RandomState(21)
    lens = (rng.uniform(0.2, 2.6, size=14) * 16000).astype(int)
    lens[3] = 300; lens[9] = 0
    audio = [synth.synth_audio(int(n), seed=2100 + i) for i, n in enumerate(lens)]
    together = m.acousticProbs(audio)
    texts = m.sttBatch(audio)
    for i, a in enumerate(audio):
        alone = m.acousticProbs([a])[0]
        assert np.array_equal(together[i], alone), i
        assert texts[i] == m.stt(a), i
    assert any(texts)
