"""oracle/am_hybrid.py -- TEST INFRASTRUCTURE ONLY: the acoustic model as TensorFlow Lite's *hybrid* kernels compute it.

Released models are dynamic-range quantised (training/coqui_stt_training/export.py:145-146: `converter.optimizations =
[tf.lite.Optimize.DEFAULT]` without a representative dataset): every FULLY_CONNECTED weight matrix -- layers 1-3, 5, 6 and the
unrolled LSTM cell's [x_t, h] kernel -- is stored int8 (symmetric, zero point 0, one scale per tensor; newer converters one per
output row), and the reference's CPU path (native_client/tflitemodelstate.cc:200,369-405) runs them through TFLite's hybrid
FULLY_CONNECTED: the float input is quantised ON EVERY CALL, per batch row, to int8 with scale max|x| / 127, the dot products are
accumulated in int32, rescaled by (row scale x weight scale) and added to the float bias; everything between the matrix products
(ReLU/minimum, LOGISTIC, TANH, MUL, ADD, SOFTMAX) is float32.

TensorFlow Lite is an un-vendored submodule (`.gitmodules:5-7`), so this file restates its PUBLISHED algorithm:
  tensorflow/lite/kernels/fully_connected.cc            EvalHybrid: all-zero input -> bias only; quantise each batch row;
                                                        scaling_factor[b] *= filter scale; MatrixBatchVectorMultiplyAccumulate
  tensorflow/lite/kernels/internal/reference/portable_tensor_utils.cc
      PortableSymmetricQuantizeFloats                   range = max(|min|, |max|); range == 0 -> zeros, scale 1;
                                                        q = clamp(TfLiteRound(x * (127 / range)), -127, 127), scale = range / 127
      PortableMatrixBatchVectorMultiplyAccumulate (int8) result[b][r] += (int32 dot product) * scaling_factor[b]   (float32)
"Parity unpinned" like oracle/am_ref.py: nothing reference-held (no TFLite build, no released model offline) pins it.  It exists to
put a NUMBER on the difference between the engine (int8 weights de-quantised to f16, f16 activations, f32 accumulation -- no
activation quantisation) and the reference's CPU path: tests/test_gpu_hybrid.py.

Integer dot products are evaluated with float64 BLAS: |q_x q_w| <= 127^2 and K <= 4096 keep every partial sum below 2^53, so the
result is the exact int32 value.
"""
import numpy as np

from . import am_ref

F32 = np.float32


def quantize_weights(w_in_out, per_channel=False):
    """What the converter stores for a [in][out] float matrix: int8 [out][in], f32 scale(s) (stt_amd/tflitefile.py does the same)."""
    w = np.ascontiguousarray(np.asarray(w_in_out, dtype=F32).T)
    amax = np.abs(w).max(axis=1) if per_channel else np.array([np.abs(w).max()])
    scale = (np.maximum(amax, 1e-30) / 127.0).astype(F32)
    # the converter's quantize_weights pass goes through the same tensor_utils::SymmetricQuantizeFloats as the activations: round(w * (127 / range))
    inv = (F32(127.0) / np.maximum(amax, 1e-30).astype(F32)).astype(F32)
    t = (w * (inv[:, None] if per_channel else inv[0])).astype(F32)
    q = np.clip(np.sign(t) * np.floor(np.abs(t).astype(np.float64) + 0.5), -127, 127).astype(np.int8)
    return q, scale


def symmetric_quantize_rows(x):
    """PortableSymmetricQuantizeFloats per batch row: -> (int8-valued float64 [B][K], f32 scaling_factor [B])."""
    x = np.asarray(x, dtype=F32)
    rng = np.abs(x).max(axis=1)
    inv = np.where(rng > 0, F32(127.0) / np.where(rng > 0, rng, F32(1)), F32(0)).astype(F32)
    t = (x * inv[:, None]).astype(F32)
    q = np.sign(t) * np.floor(np.abs(t).astype(np.float64) + 0.5)            # TfLiteRound = std::round: half away from zero
    q = np.clip(q, -127, 127)
    sf = np.where(rng > 0, rng / F32(127.0), F32(1)).astype(F32)
    return q, sf


def fully_connected_hybrid(x, wq, wscale, bias, wq_f64_t=None):
    """EvalHybrid: x f32 [B][K], wq int8 [N][K], wscale f32 [1] or [N], bias f32 [N] -> f32 [B][N] (no activation).
    `wq_f64_t`: wq.T as float64, if the caller keeps one (the conversion of a 4096 x 8192 matrix per call is the cost of a step)."""
    x = np.asarray(x, dtype=F32)
    out = np.broadcast_to(np.asarray(bias, dtype=F32), (x.shape[0], wq.shape[0])).copy()
    if not np.any(x):                                            # IsZeroVector: the output is the bias
        return out
    q, sf = symmetric_quantize_rows(x)
    acc = q @ (wq.astype(np.float64).T if wq_f64_t is None else wq_f64_t)      # exact int32 dot products
    scale = (sf[:, None] * np.asarray(wscale, dtype=F32)[None, :]).astype(F32)      # scaling_factors[b] *= filter->params.scale
    return (out + (acc.astype(F32) * scale).astype(F32)).astype(F32)               # (int32 -> float conversion, then float multiply-add)


# LOGISTIC and TANH are float kernels in TFLite, and their last bits depend on the build: the reference kernels call std::exp / std::tanh
# (reference/logistic.h, reference/tanh.h), the optimised ones Eigen's rational approximations -- each within an ulp or two of the correctly
# rounded float of the real function.  That value is what this restatement takes: float64 evaluation, one rounding to float32.  (numpy's own
# float32 exp / tanh are SIMD approximations 1-3 ulp off: not a better stand-in for TFLite than the correctly rounded value, and the
# quantised recurrence amplifies last-bit differences -- tests/test_gpu_hybrid.py.)  oracle/checks/cr_activations_check.c pins the float64
# algorithm the engine uses for them against this definition.
def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))).astype(F32)


def _tanh(x):
    return np.tanh(np.asarray(x, dtype=np.float64)).astype(F32)


def _sigmoid_f32(x):
    """The float form of TFLite's reference LOGISTIC, `1.f / (1.f + std::exp(-x))` (reference/logistic.h), with whatever float32 exp the host's
    numpy has (a SIMD approximation, 1-3 ulp off the correctly rounded value -- like any libm's or Eigen's): the SECOND checker."""
    x = np.asarray(x, dtype=F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def _tanh_f32(x):
    return np.tanh(np.asarray(x, dtype=F32), dtype=F32)


class HybridModel:
    """The exported graph (stt_amd/tflitefile.py mirrors export.py's) with every FULLY_CONNECTED as the hybrid kernel.
    activations: "cr" = LOGISTIC / TANH correctly rounded (float64 evaluation, one rounding: what the engine's int8 path computes, so the
    two agree to the last bits); "f32" = evaluated in float32 as a TFLite build would (1-3 ulp from "cr"; which ulp is build-specific).
    The engine is stated against BOTH: bit-level against "cr", and within the tolerance tests/test_gpu_hybrid.py writes against "f32" --
    that second figure is the honest size of "which float exp did your TFLite link" (round-5 advisor)."""

    def __init__(self, weights, per_channel=False, relu_clip=am_ref.RELU_CLIP, activations="cr"):
        assert activations in ("cr", "f32")
        self.sig, self.tanh = (_sigmoid, _tanh) if activations == "cr" else (_sigmoid_f32, _tanh_f32)
        self.q = {}
        for k in ("layer_1/weights", "layer_2/weights", "layer_3/weights", "lstm/kernel", "layer_5/weights", "layer_6/weights"):
            w = np.asarray(weights[k], dtype=F32)
            self.q[k] = quantize_weights(w, per_channel) if w.size >= 1024 else None        # the converter leaves small tensors in float
            if self.q[k] is None:
                self.q[k] = (w.T.copy(), None)
        self.b = {k: np.asarray(v, dtype=F32) for k, v in weights.items() if k.endswith("bias")}
        self.H = self.b["layer_1/bias"].shape[0]
        self.clip = F32(relu_clip)
        self._f64 = {k: (None if s_ is None else np.ascontiguousarray(q_.T.astype(np.float64))) for k, (q_, s_) in self.q.items()}

    def effective_weights(self):
        """The f32 matrices a de-quantising loader computes with (checkpoint orientation [in][out])."""
        eff = dict(self.b)
        for k, (q, s) in self.q.items():
            eff[k] = (q.T.astype(F32) if s is None else (q.astype(F32) * (s[:, None] if len(s) > 1 else s[0])).T.astype(F32))
        return eff

    def _fc(self, x, name, bias):
        q, s = self.q[name]
        if s is None:
            return (np.asarray(x, F32) @ q.T.astype(F32) + bias).astype(F32)
        return fully_connected_hybrid(x, q, s, bias, self._f64[name])

    def _dense(self, x, name):
        y = self._fc(x, name + "/weights", self.b[name + "/bias"])
        return np.minimum(np.maximum(y, F32(0)), self.clip).astype(F32)

    def forward_batch(self, windows):
        """windows f32 [B][T][494] (zero state at t = 0) -> probs f32 [B][T][C].  Rows are quantised one by one, so processing B
        utterances together changes nothing (the reference runs batch 1, n_steps 16: tflitemodelstate.cc:369-405)."""
        B, T, D = windows.shape
        x = windows.reshape(B * T, D).astype(F32)
        l3 = self._dense(self._dense(self._dense(x, "layer_1"), "layer_2"), "layer_3").reshape(B, T, self.H)
        c = np.zeros((B, self.H), F32)
        h = np.zeros((B, self.H), F32)
        hs = np.zeros((B, T, self.H), F32)
        for t in range(T):
            z = self._fc(np.concatenate([l3[:, t], h], axis=1), "lstm/kernel", self.b["lstm/bias"])     # concat([x_t, h]) . kernel + bias
            i, j, f, o = np.split(z, 4, axis=1)                                                          # gate order i, j, f, o (deepspeech_model.py:144-168)
            c = (self.sig(f) * c + self.sig(i) * self.tanh(j)).astype(F32)
            h = (self.sig(o) * self.tanh(c)).astype(F32)
            hs[:, t] = h
        l5 = self._dense(hs.reshape(B * T, self.H), "layer_5")
        logits = self._fc(l5, "layer_6/weights", self.b["layer_6/bias"])
        return self.softmax(logits).reshape(B, T, -1)

    def softmax(self, logits):
        """SOFTMAX of the float logits.  "cr": like LOGISTIC / TANH, the correctly rounded float of the real function at each of its two steps --
        e = float(exp(l - max)) evaluated in float64, p = float(e / sum(e)) with the sum and the quotient in float64 (one rounding); the engine's
        int8 path computes exactly this (kernels_am.hip: softmax_kernel, `exact`).  "f32": TFLite's reference float kernel as written --
        float exp, a float sum, one float division per class -- with numpy's float32 exp."""
        logits = np.asarray(logits, dtype=F32)
        z = (logits - logits.max(axis=1, keepdims=True)).astype(F32)
        if self.sig is _sigmoid:
            e = np.exp(z.astype(np.float64)).astype(F32).astype(np.float64)
            return (e / e.sum(axis=1, keepdims=True)).astype(F32)
        e = np.exp(z, dtype=F32)
        return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def utterance_probs_batch(audios, weights, per_channel=False, spec=None, activations="cr"):
    """Equal-length int16 utterances -> probs [B][T][C] through the hybrid kernels (features as oracle/am_ref.py)."""
    spec = spec or am_ref.MfccSpec()
    win = np.stack([am_ref.context_windows(spec.frames_fast(np.asarray(a, dtype=np.int16))) for a in audios])
    return HybridModel(weights, per_channel, activations=activations).forward_batch(win)
