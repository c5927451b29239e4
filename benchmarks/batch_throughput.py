#!/usr/bin/env python
"""benchmarks/batch_throughput.py -- BASELINE.json configs[3] on one GPU: a LibriSpeech-shaped batch (lengths U(1, 15) s,
seed 2) through STTX_SpeechToTextBatchDevice, English geometry, beam 500, the bench's synthetic 500 k-word scorer.
One rank's share of the 10 k-utterance job is 1250 utterances (the 8 ranks run the same code on their LPT shards).

    python benchmarks/batch_throughput.py [--utterances 1250] [--scorer synthetic|fixture]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stt_amd import Model, modelfile, scorertools, synth  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "fixtures")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=1250)
    ap.add_argument("--scorer", default="synthetic", choices=["synthetic", "fixture"])
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    w = synth.synth_weights(0, n_hidden=2048, n_classes=29)
    d = tempfile.TemporaryDirectory()
    path = os.path.join(d.name, "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
    m = Model(path)
    scorer = os.path.join(FIX, "pruned_lm.scorer")
    if args.scorer == "synthetic":
        lm, vocab, scorer = os.path.join(d.name, "lm.binary"), os.path.join(d.name, "vocab.txt"), os.path.join(d.name, "s.scorer")
        scorertools.synth_lm(lm, vocab, words=500000, order=5, seed=7, avg={2: 24, 3: 1.2, 4: 0.7, 5: 0.5})
        scorertools.generate_scorer_package(lm, vocab, scorer, alphabet=os.path.join(FIX, "alphabet.txt"),
                                            default_alpha=0.931289039105002, default_beta=1.1834137581510284)
    m.enableExternalScorer(scorer)
    rng = np.random.RandomState(2)
    lens = (rng.uniform(1.0, 15.0, size=args.utterances) * 16000).astype(np.int64)
    stride = int(lens.max())
    base = synth.synth_audio(stride, seed=5)
    host = np.zeros((args.utterances, stride), dtype=np.int16)
    for i, n in enumerate(lens):                      # cheap synthetic variety: rotated copies of one noise/tone mixture
        host[i, :n] = np.roll(base, 977 * i)[:n]
    d_audio = torch.from_numpy(host).to(dev)
    sizes = [int(x) for x in lens]
    m.sttBatchDevice(d_audio.data_ptr(), stride, sizes[:64])        # warm-up
    best = None
    for _ in range(args.repeat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.sttBatchDevice(d_audio.data_ptr(), stride, sizes)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    audio_s = float(lens.sum()) / 16000.0
    print("utterances %d  audio %.0f s  wall %.3f s  RTF %.0f  (%.2f ms per utterance, %d non-empty transcripts)"
          % (args.utterances, audio_s, best, audio_s / best, 1e3 * best / args.utterances, sum(1 for t in out if t)))


if __name__ == "__main__":
    main()
