#!/usr/bin/env python
"""benchmarks/stream_latency.py -- BASELINE.json configs[2] as a measurement (bench.py is configs[1]):
synthetic utterances of 1-15 s fed in 320 ms hops (5120 samples = exactly n_steps = 16 frames) through
STT_FeedAudioContent, with STT_IntermediateDecode after every hop; reports p50/p95 of (feed + intermediate decode) per hop.

    python benchmarks/stream_latency.py [--utterances N] [--streams S]

With --streams S > 1 the S streams are interleaved on one model (the reference's concurrent_streams.py pattern)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stt_amd import Model, modelfile, synth  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "fixtures")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=20)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--hidden", type=int, default=2048)
    args = ap.parse_args()
    w = synth.synth_weights(0, n_hidden=args.hidden, n_classes=29)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.sttw")
        modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
        m = Model(path)
    m.enableExternalScorer(os.path.join(FIX, "pruned_lm.scorer"))
    rng = np.random.RandomState(1)
    lat, fin = [], []
    total_audio = 0.0
    t_all = time.perf_counter()
    for u0 in range(0, args.utterances, args.streams):
        group = []
        for u in range(u0, min(args.utterances, u0 + args.streams)):
            n = int(rng.uniform(1, 15) * 16000)
            group.append((synth.synth_audio(n, seed=u), m.createStream()))
            total_audio += n / 16000
        k = 0
        while any(k < len(a) for a, _ in group):
            for a, s in group:
                if k < len(a):
                    t0 = time.perf_counter()
                    s.feedAudioContent(a[k:k + 5120])
                    s.intermediateDecode()
                    lat.append(time.perf_counter() - t0)
            k += 5120
        for a, s in group:
            t0 = time.perf_counter()
            s.finishStream()
            fin.append(time.perf_counter() - t0)
    el = time.perf_counter() - t_all
    lat = np.array(lat) * 1e3
    print("hops %d  feed+intermediate-decode per 320 ms hop: p50 %.2f ms  p95 %.2f ms  max %.2f ms;  finish p50 %.2f ms;  "
          "aggregate RTF %.1f with %d interleaved stream(s)"
          % (len(lat), np.percentile(lat, 50), np.percentile(lat, 95), lat.max(), np.percentile(np.array(fin) * 1e3, 50),
             total_audio / el, args.streams))


if __name__ == "__main__":
    main()
