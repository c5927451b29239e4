"""stt_amd/synth.py -- seeded synthetic inputs (no datasets or checkpoints exist offline).

Recipes follow SURVEY.md section 8d: audio = band-limited noise + sinusoid bursts; weights = the reference's
initialisers (VarianceScaling fan_avg uniform, deepspeech_model.py:69-75); "peaky" emissions = the probe recipe of
SURVEY.md Appendix B.
"""
import numpy as np

ENGLISH_LABELS = [b" "] + [bytes([c]) for c in range(ord("a"), ord("z") + 1)] + [b"'"]  # data/alphabet.txt of the reference


def synth_weights(seed=0, n_input=26, n_context=9, n_hidden=2048, n_classes=29, dtype=np.float32):
    rng = np.random.default_rng(seed)

    def vs(fan_in, fan_out):
        lim = np.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
        return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)

    b = lambda n: (0.05 * rng.standard_normal(n)).astype(dtype)
    n_in1 = n_input * (2 * n_context + 1)
    w = {}
    w["layer_1/weights"], w["layer_1/bias"] = vs(n_in1, n_hidden), b(n_hidden)
    w["layer_2/weights"], w["layer_2/bias"] = vs(n_hidden, n_hidden), b(n_hidden)
    w["layer_3/weights"], w["layer_3/bias"] = vs(n_hidden, n_hidden), b(n_hidden)
    w["lstm/kernel"], w["lstm/bias"] = vs(2 * n_hidden, 4 * n_hidden), b(4 * n_hidden)
    w["layer_5/weights"], w["layer_5/bias"] = vs(n_hidden, n_hidden), b(n_hidden)
    w["layer_6/weights"], w["layer_6/bias"] = vs(n_hidden, n_classes), b(n_classes)
    return w


def synth_audio(n_samples, seed=0, sample_rate=16000):
    """Band-limited Gaussian noise at about -26 dBFS plus 3-5 random sinusoid bursts per second, int16."""
    rng = np.random.default_rng(seed)
    n = int(n_samples)
    if n == 0:
        return np.zeros(0, dtype=np.int16)
    x = rng.standard_normal(n)
    k = np.ones(8) / 8.0
    x = np.convolve(x, k, mode="same")
    x *= 0.05 / (x.std() + 1e-9)
    t = np.arange(n) / sample_rate
    secs = max(1, int(np.ceil(n / sample_rate)))
    for _ in range(int(rng.integers(3, 6)) * secs):
        f0 = rng.uniform(150.0, 3500.0)
        start = rng.uniform(0, n / sample_rate)
        dur = rng.uniform(0.03, 0.25)
        env = np.exp(-0.5 * ((t - start - dur / 2) / (dur / 4 + 1e-6)) ** 2)
        x += rng.uniform(0.02, 0.2) * env * np.sin(2 * np.pi * f0 * t + rng.uniform(0, 6.28))
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def synth_audio_batch(count, n_samples, seed=0, sample_rate=16000):
    """int16 [count][n_samples]: the recipe of synth_audio (band-limited noise at about -26 dBFS + 3-5 sinusoid bursts per second),
    vectorised -- every burst is evaluated only where its envelope is above 1e-7 of its peak -- so that a bench can afford a
    different batch for every timed step.  Rows are independent draws of ONE generator seeded with `seed` (not the rows
    synth_audio(seed=...) would give)."""
    rng = np.random.default_rng(seed)
    n, count = int(n_samples), int(count)
    out = np.zeros((count, n), dtype=np.int16)
    if n == 0 or count == 0:
        return out
    x = rng.standard_normal((count, n + 7), dtype=np.float32).astype(np.float64)
    cs = np.cumsum(x, axis=1)
    x = (cs[:, 7:] - np.concatenate([np.zeros((count, 1)), cs[:, :-8]], axis=1)) / 8.0       # 8-tap moving average
    x *= 0.05 / (x.std(axis=1, keepdims=True) + 1e-9)
    secs = max(1, int(np.ceil(n / sample_rate)))
    for r in range(count):
        for _ in range(int(rng.integers(3, 6)) * secs):
            f0, start, dur = rng.uniform(150.0, 3500.0), rng.uniform(0, n / sample_rate), rng.uniform(0.03, 0.25)
            amp, ph = rng.uniform(0.02, 0.2), rng.uniform(0, 6.28)
            mid, sig = start + dur / 2, dur / 4 + 1e-6
            a, b = max(0, int((mid - 5.7 * sig) * sample_rate)), min(n, int((mid + 5.7 * sig) * sample_rate) + 1)
            if a >= b:
                continue
            t = np.arange(a, b) / sample_rate
            x[r, a:b] += amp * np.exp(-0.5 * ((t - mid) / sig) ** 2) * np.sin(2 * np.pi * f0 * t + ph)
    np.clip(np.round(x * 32768.0), -32768, 32767, out=x)
    out[:] = x
    return out


def peaky_emissions(label_seq, T, n_classes, blank, seed=0, noise=0.02, hold=2, lead=20):
    """float32 [T, C]: blank ~0.9 between labels, each label held `hold` frames; noise * U(0,1) everywhere, renormalised."""
    rng = np.random.RandomState(seed)
    plan = [blank] * lead
    for l in label_seq:
        plan += [l] * hold + [blank] * 2
    plan = (plan + [blank] * T)[:T]
    p = noise * rng.rand(T, n_classes)
    for t, c in enumerate(plan):
        p[t, c] += 0.6 + 0.35 * rng.rand()
    p /= p.sum(1, keepdims=True)
    return p.astype(np.float32)
