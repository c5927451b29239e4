#!/usr/bin/env python
"""benchmarks/bytes_mode.py -- BASELINE.json configs[4] on one GPU: byte-output model (UTF8Alphabet, 256 classes,
doc/DECODER.rst:193), codepoint-level scorer (the reference's data/smoke_test/pruned_lm.bytes.scorer), beam_width 1024,
64 synthetic 5 s utterances, full-size acoustic model.  Prints the step time and the stage times.

    python benchmarks/bytes_mode.py [--beam 1024] [--steps 3] [--no-scorer]"""
import argparse
import json
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
    ap.add_argument("--beam", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-scorer", action="store_true")
    ap.add_argument("--gain", type=float, default=1.0, help="scale of the output layer (1 = near-uniform softmax, the worst case)")
    args = ap.parse_args()
    labels = [bytes([i + 1]) for i in range(255)]        # UTF8Alphabet: label i = byte i+1 (alphabet.h:83-91)
    w = synth.synth_weights(0, n_hidden=2048, n_classes=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * args.gain).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "bytes.sttw")
        modelfile.write_model(path, w, labels, beam_width=args.beam)
        m = Model(path)
    if not args.no_scorer:
        m.enableExternalScorer(os.path.join(FIX, "pruned_lm.bytes.scorer"))
    audio = [synth.synth_audio(80000, seed=i) for i in range(64)]
    m.sttBatch(audio)
    m.setProfiling(1)
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        m.sttBatch(audio)
        for k, v in m.stageTimes().items():
            acc[k] = acc.get(k, 0.0) + v
    el = (time.perf_counter() - t0) / args.steps
    st = m.decoderStats()
    m.setProfiling(2)
    m.sttBatch(audio)
    ph = m.decoderPhaseCycles()
    print(json.dumps({"beam": args.beam, "scorer": not args.no_scorer, "ms_per_step": round(1e3 * el, 3), "rtf": round(320.0 / el),
                      "stage_ms": {k: round(v / args.steps, 3) for k, v in acc.items() if k.endswith("_ms")},
                      "decoder_counters": st,
                      "phase_cycles_per_stream_step": {k: round(v / max(1, st["steps"]), 1) for k, v in ph.items()}}))


if __name__ == "__main__":
    main()
