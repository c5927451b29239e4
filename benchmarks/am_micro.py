#!/usr/bin/env python
"""Acoustic stage alone (beam 1, no scorer): MFCC -> dense x3 -> LSTM -> dense x2 -> softmax over 64 x 5 s utterances at the
English geometry; prints the HIP-event stage times per pass.  Used to A/B kernel variants (STT_AMD_* environment switches)."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stt_amd import Model, modelfile, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    w = synth.synth_weights(0, n_hidden=2048, n_classes=29)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.sttw")
        modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
        m = Model(path)
    audio = [synth.synth_audio(80000, seed=i) for i in range(64)]
    m.setBeamWidth(1)                      # no scorer, beam 1: the search is negligible and the stages run unperturbed
    m.sttBatch(audio)
    m.setProfiling(True)
    acc = {}
    for _ in range(reps):
        m.sttBatch(audio)
        for k, v in m.stageTimes().items():
            acc[k] = acc.get(k, 0.0) + v
    m.setProfiling(2)
    m.sttBatch(audio)
    ph = m.decoderPhaseCycles(); st = m.decoderStats()
    p = m.acousticProbs(audio[:2])
    out = {k: round(v / reps, 4) for k, v in acc.items()}
    out["checksum"] = float(np.asarray(p[0], dtype=np.float64).sum())
    out["decoder_phase_cycles_per_stream_step"] = {k: round(v / max(1, st["steps"]), 1) for k, v in ph.items()}
    out["env"] = {k: v for k, v in os.environ.items() if k.startswith("STT_AMD_")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
