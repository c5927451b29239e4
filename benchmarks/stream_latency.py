#!/usr/bin/env python
"""benchmarks/stream_latency.py -- BASELINE.json configs[2] as a measurement (bench.py is configs[1]):
synthetic utterances of 1-15 s fed in 320 ms hops (5120 samples = exactly n_steps = 16 frames) through
STT_FeedAudioContent, with STT_IntermediateDecode after every hop; reports p50/p95 of (feed + intermediate decode) per hop.

    python benchmarks/stream_latency.py [--utterances N] [--streams S]

With --streams S > 1, S streams are live at a time.  --batched (default) advances them with STTX_FeedAudioContentBatch /
STTX_IntermediateDecodeBatch (one acoustic + one search launch per hop for all of them); --no-batched interleaves
STT_FeedAudioContent calls on one model (the reference's concurrent_streams.py pattern)."""
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
    ap.add_argument("--batched", dest="batched", action="store_true", default=True)
    ap.add_argument("--no-batched", dest="batched", action="store_false")
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
    base = synth.synth_audio(15 * 16000, seed=3)
    audios = []
    for u in range(args.utterances):                       # synthetic variety: rotated copies of one noise/tone mixture
        n = int(rng.uniform(1, 15) * 16000)
        audios.append(np.roll(base, 977 * u)[:n].copy())
        total_audio += n / 16000
    t_all = time.perf_counter()
    for u0 in range(0, args.utterances, args.streams):
        group = []
        for u in range(u0, min(args.utterances, u0 + args.streams)):
            group.append((audios[u], m.createStream()))
        k = 0
        if args.batched and args.streams > 1:
            from stt_amd import model as M
            live = list(group)
            while live:
                t0 = time.perf_counter()
                M.feedAudioContentBatch([s for _, s in live], [a[k:k + 5120] for a, _ in live])
                M.intermediateDecodeBatch([s for _, s in live])
                lat.append(time.perf_counter() - t0)          # one hop of ALL live streams
                k += 5120
                live = [(a, s) for a, s in live if k < len(a)]
            t0 = time.perf_counter()
            M.finishStreamBatch([s for _, s in group])
            fin.append(time.perf_counter() - t0)
            continue
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
    print("hops %d  feed+intermediate-decode per 320 ms hop%s: p50 %.2f ms  p95 %.2f ms  max %.2f ms;  finish p50 %.2f ms;  "
          "aggregate RTF %.1f with %d live stream(s)%s"
          % (len(lat), " (all live streams together)" if args.batched and args.streams > 1 else "", np.percentile(lat, 50), np.percentile(lat, 95),
             lat.max(), np.percentile(np.array(fin) * 1e3, 50), total_audio / el, args.streams,
             " [batched]" if args.batched and args.streams > 1 else ""))


if __name__ == "__main__":
    main()
