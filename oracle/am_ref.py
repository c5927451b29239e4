"""oracle/am_ref.py -- TEST INFRASTRUCTURE ONLY ("port" oracle for the acoustic half).

CPU restatement (numpy, float64 internally) of SURVEY.md section 8a rows a2-a6:

  feed/frame bookkeeping   native_client/stt.cc:105-128 (feedAudioContent), :226-261
                           (processAudioWindow / flushBuffers / addZeroMfccWindow), :519-551
  AudioSpectrogram + Mfcc  training/coqui_stt_training/util/feeding.py:51-73 (op attributes);
                           the op *kernels* live in TensorFlow (un-vendored submodule
                           coqui-ai/tensorflow, .gitmodules:5-7; training side pins
                           tensorflow==1.15.4, setup.py:41).  Their algorithm is restated from
                           upstream tensorflow/core/kernels/{spectrogram,mfcc,
                           mfcc_mel_filterbank,mfcc_dct}.cc as summarised in SURVEY.md 8c.
  context windows          native_client/stt.cc:272-309, deepspeech_model.py:42-63
  dense / clipped ReLU     training/coqui_stt_training/deepspeech_model.py:66-89
  LSTMCell                 deepspeech_model.py:144-168 (tf LSTMCell, forget_bias=0, gate order i,j,f,o)
  layer order + softmax    deepspeech_model.py:171-263, :357

PARITY UNPINNED for this half: no TensorFlow/TFLite runtime and no model file exist in
/root/reference or in this image, and the reference has no test that pins these stages with
in-tree data.  The HIP kernels are compared against this restatement with the
tolerances written in tests/test_gpu_kernels.py, test_gpu_benchshape.py and test_gpu_timedpath.py (and, for the reference's
hybrid-int8 arithmetic, oracle/am_hybrid.py + tests/test_gpu_hybrid.py); the restatement itself cannot be checked against
the reference binary.
"""
import numpy as np

SAMPLE_RATE = 16000
WIN_LEN = 512
WIN_STEP = 320
N_INPUT = 26
N_CONTEXT = 9
N_MEL = 40
RELU_CLIP = 20.0


def n_frames_for(n_samples, win_len=WIN_LEN, win_step=WIN_STEP):
    """Frames stt.cc produces for one utterance: every full window + the flushed partial one."""
    full = (n_samples - win_len) // win_step + 1 if n_samples >= win_len else 0
    return full + 1


# ----------------------------------------------------------------------------- features
class MfccSpec:
    """Constant tables of the TF Mfcc op (40 channels, 20 Hz .. sr/2, 26 DCT coefficients)."""

    def __init__(self, sample_rate=SAMPLE_RATE, win_len=WIN_LEN, n_mel=N_MEL, n_coef=N_INPUT, lower=20.0, upper=None, win_step=None):
        upper = sample_rate / 2.0 if upper is None else upper
        self.win_len = win_len
        self.win_step = WIN_STEP if win_step is None else win_step     # (util/config.py:306-325: 20 ms of the sample rate)
        self.fft_len = 1 << int(np.ceil(np.log2(win_len)))
        self.n_bins = self.fft_len // 2 + 1
        i = np.arange(win_len, dtype=np.float64)
        self.window = 0.5 - 0.5 * np.cos(2.0 * np.pi * i / win_len)  # periodic Hann (spectrogram.cc)
        mel = lambda f: 1127.0 * np.log1p(f / 700.0)
        mel_low, mel_hi = mel(lower), mel(upper)
        spacing = (mel_hi - mel_low) / (n_mel + 1)
        center = mel_low + spacing * (np.arange(n_mel + 1) + 1)  # center_frequencies_[0..n_mel]
        hz_per_sbin = 0.5 * sample_rate / (self.n_bins - 1)
        self.start_index = int(1.5 + lower / hz_per_sbin)
        self.end_index = int(upper / hz_per_sbin)
        band_mapper = np.full(self.n_bins, -2, dtype=np.int32)
        weights = np.zeros(self.n_bins, dtype=np.float64)
        channel = 0
        for b in range(self.n_bins):
            melf = mel(b * hz_per_sbin)
            if b < self.start_index or b > self.end_index:
                band_mapper[b] = -2
            else:
                while channel < n_mel and center[channel] < melf:
                    channel += 1
                band_mapper[b] = channel - 1
        for b in range(self.n_bins):
            ch = band_mapper[b]
            if b < self.start_index or b > self.end_index:
                weights[b] = 0.0
            elif ch >= 0:
                weights[b] = (center[ch + 1] - mel(b * hz_per_sbin)) / (center[ch + 1] - center[ch])
            else:
                weights[b] = (center[0] - mel(b * hz_per_sbin)) / (center[0] - mel_low)
        self.band_mapper, self.weights = band_mapper, weights
        self.n_mel, self.n_coef = n_mel, n_coef
        fnorm = np.sqrt(2.0 / n_mel)
        arg = np.pi / n_mel
        ii, jj = np.meshgrid(np.arange(n_coef), np.arange(n_mel), indexing="ij")
        self.dct = fnorm * np.cos(ii * arg * (jj + 0.5))  # cosines_[i][j] (mfcc_dct.cc)

    def frame(self, samples_f32):
        """One 512-sample window (float32 in [-1,1)) -> 26 float32 coefficients."""
        x = np.zeros(self.fft_len, dtype=np.float64)
        x[:len(samples_f32)] = samples_f32.astype(np.float64)
        x[:len(self.window)] *= self.window
        spec = np.fft.rfft(x)
        power = (spec.real * spec.real + spec.imag * spec.imag).astype(np.float32)  # spectrogram output tensor is float
        return self.from_power(power)

    def from_power(self, power_f32):
        """The Mfcc op proper (mfcc.cc: sqrt of the squared-magnitude spectrogram, mel filterbank, log, DCT): n_bins float32 -> n_coef float32.
        This is the entry upstream TensorFlow's own unit test drives (core/kernels/mfcc_test.cc; tests/test_oracle_am.py holds its vector)."""
        amp = np.sqrt(np.asarray(power_f32, dtype=np.float32).astype(np.float64))
        mel = np.zeros(self.n_mel, dtype=np.float64)
        for b in range(self.start_index, self.end_index + 1):  # mfcc_mel_filterbank.cc Compute()
            w = amp[b] * self.weights[b]
            ch = self.band_mapper[b]
            if ch >= 0:
                mel[ch] += w
            ch += 1
            if ch < self.n_mel:
                mel[ch] += amp[b] - w
        mel = np.log(np.maximum(mel, 1e-12))
        out = np.zeros(self.n_coef, dtype=np.float64)
        for i in range(self.n_coef):  # sequential j, as the op does
            acc = 0.0
            for j in range(self.n_mel):
                acc += self.dct[i, j] * mel[j]
            out[i] = acc
        return out.astype(np.float32)

    def frames_fast(self, audio_i16):
        """Vectorised version of stt.cc framing + frame() for a whole utterance -> [F, 26] float32.
        Mel/DCT accumulation order differs from frame() only in float64 summation order."""
        n = len(audio_i16)
        WL, WS = self.win_len, self.win_step
        F = n_frames_for(n, WL, WS)
        x = np.zeros((F - 1) * WS + WL, dtype=np.float32)
        x[:n] = audio_i16.astype(np.float32) * np.float32(1.0 / 32768.0)
        idx = np.arange(F)[:, None] * WS + np.arange(WL)[None, :]
        fr = x[idx].astype(np.float64) * self.window[None, :]
        spec = np.fft.rfft(fr, n=self.fft_len, axis=1)
        power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
        amp = np.sqrt(power.astype(np.float64))
        mel = np.zeros((F, self.n_mel))
        for b in range(self.start_index, self.end_index + 1):
            w = amp[:, b] * self.weights[b]
            ch = self.band_mapper[b]
            if ch >= 0:
                mel[:, ch] += w
            if ch + 1 < self.n_mel:
                mel[:, ch + 1] += amp[:, b] - w
        mel = np.log(np.maximum(mel, 1e-12))
        return (mel @ self.dct.T).astype(np.float32)


def mfcc_utterance(audio_i16, spec=None):
    """int16 mono 16 kHz -> [F, 26] float32 following stt.cc's streaming framing exactly."""
    spec = spec or MfccSpec()
    n = len(audio_i16)
    WL, WS = spec.win_len, spec.win_step
    F = n_frames_for(n, WL, WS)
    x = audio_i16.astype(np.float32) * np.float32(1.0 / 32768.0)  # stt.cc:113-114
    out = np.zeros((F, spec.n_coef), dtype=np.float32)
    for f in range(F):
        out[f] = spec.frame(x[f * WS: f * WS + WL])  # short tail => zero padded (tflitemodelstate.cc:341-355)
    return out


def context_windows(mfcc, n_context=N_CONTEXT):
    """[F, 26] -> [F, 19*26]: stt.cc:272-309 with n_context zero frames on both sides (:533, :242-247)."""
    F, D = mfcc.shape
    pad = np.zeros((n_context, D), dtype=mfcc.dtype)
    p = np.concatenate([pad, mfcc, pad], axis=0)
    return np.stack([p[t:t + 2 * n_context + 1].reshape(-1) for t in range(F)], axis=0)


# ----------------------------------------------------------------------------- acoustic model
def synth_weights(seed=0, n_input=N_INPUT, n_context=N_CONTEXT, n_hidden=2048, n_classes=29, dtype=np.float32):
    """Seeded random-init weights of the reference architecture.
    dense: VarianceScaling(fan_avg, uniform) as deepspeech_model.py:69-75; LSTM kernel the same scheme;
    biases small normal.  Names follow the checkpoint variables."""
    rng = np.random.default_rng(seed)

    def vs(fan_in, fan_out):
        lim = np.sqrt(3.0 * 1.0 / ((fan_in + fan_out) / 2.0))
        return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)

    n_in1 = n_input * (2 * n_context + 1)
    w = {}
    w["layer_1/weights"], w["layer_1/bias"] = vs(n_in1, n_hidden), (0.05 * rng.standard_normal(n_hidden)).astype(dtype)
    w["layer_2/weights"], w["layer_2/bias"] = vs(n_hidden, n_hidden), (0.05 * rng.standard_normal(n_hidden)).astype(dtype)
    w["layer_3/weights"], w["layer_3/bias"] = vs(n_hidden, n_hidden), (0.05 * rng.standard_normal(n_hidden)).astype(dtype)
    w["lstm/kernel"], w["lstm/bias"] = vs(2 * n_hidden, 4 * n_hidden), (0.05 * rng.standard_normal(4 * n_hidden)).astype(dtype)
    w["layer_5/weights"], w["layer_5/bias"] = vs(n_hidden, n_hidden), (0.05 * rng.standard_normal(n_hidden)).astype(dtype)
    w["layer_6/weights"], w["layer_6/bias"] = vs(n_hidden, n_classes), (0.05 * rng.standard_normal(n_classes)).astype(dtype)
    return w


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def am_forward(windows, w, c0=None, h0=None, dtype=np.float64, relu_clip=RELU_CLIP, weight_round=None):
    """windows [T, 494] -> (probs [T, C], c, h).  `weight_round` (e.g. np.float16) rounds weights and the
    inter-layer activations the way the HIP path stores them, to separate algorithmic error from storage error."""
    rnd = (lambda a: a.astype(weight_round).astype(dtype)) if weight_round is not None else (lambda a: a.astype(dtype))
    W = {k: (rnd(v) if k.endswith("weights") or k.endswith("kernel") else v.astype(dtype)) for k, v in w.items()}
    act = lambda a: rnd(a)
    n_hidden = W["layer_1/bias"].shape[0]
    clip = lambda a: np.minimum(np.maximum(a, 0.0), relu_clip)
    x = act(windows.astype(dtype))
    l1 = act(clip(x @ W["layer_1/weights"] + W["layer_1/bias"]))
    l2 = act(clip(l1 @ W["layer_2/weights"] + W["layer_2/bias"]))
    l3 = act(clip(l2 @ W["layer_3/weights"] + W["layer_3/bias"]))
    T = x.shape[0]
    c = np.zeros(n_hidden, dtype=dtype) if c0 is None else c0.astype(dtype)
    h = np.zeros(n_hidden, dtype=dtype) if h0 is None else h0.astype(dtype)
    Kx, Kh = W["lstm/kernel"][:n_hidden], W["lstm/kernel"][n_hidden:]
    xproj = l3 @ Kx + W["lstm/bias"]
    hs = np.zeros((T, n_hidden), dtype=dtype)
    for t in range(T):
        z = xproj[t] + act(h) @ Kh
        i, j, f, o = np.split(z, 4)
        c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(j)
        h = _sigmoid(o) * np.tanh(c)
        hs[t] = h
    l5 = act(clip(act(hs) @ W["layer_5/weights"] + W["layer_5/bias"]))
    logits = l5 @ W["layer_6/weights"] + W["layer_6/bias"]
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    probs = e / e.sum(axis=1, keepdims=True)
    return probs.astype(np.float32), c, h


def utterance_probs(audio_i16, w, spec=None, **kw):
    """One-shot STT_SpeechToText acoustic path: audio -> probs [T, C] (float32).  `spec`: another feature geometry (8 kHz ...)."""
    spec = spec or MfccSpec()
    feats = spec.frames_fast(np.asarray(audio_i16, dtype=np.int16))
    return am_forward(context_windows(feats), w, **kw)[0]
