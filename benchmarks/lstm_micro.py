#!/usr/bin/env python
"""The recurrent step kernel alone (STTX_TestLstmSteps, HIP-event timed): 250 dependent steps at the English geometry (n_hidden 2048),
64 and 128 batch rows, every shipped form, plus the timing probes of kernels_am.hip (one operand stream removed) that say where a
step's time goes.  Prints one JSON line per variant: microseconds per step."""
import json
import os

os.environ.setdefault("STT_AMD_TEST_HOOKS", "1")   # a probe of single kernels: needs libstt_test.so (include/stt_amd_test.h)
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stt_amd import Model, modelfile, native, synth  # noqa: E402

H, T, P = 2048, 250, 4


def main():
    w = synth.synth_weights(0, n_hidden=H, n_classes=29)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.sttw")
        modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
        m = Model(path)
    rng = np.random.default_rng(0)
    x128 = (rng.standard_normal((P, 128, 4 * H)) * 1.5).astype(np.float32)

    def run(rows, graph, **tun):
        for k, v in tun.items():
            native.set_tuning(k, v)
        x = np.ascontiguousarray(x128[:, :rows]).reshape(P * rows, 4 * H)
        best = 1e9
        for _ in range(3):
            best = min(best, m.lstmSteps(x, rows, T, graph=graph, timing=True)[3])
        for k in tun:
            native.set_tuning(k, {"lstm_prefetch": 2}.get(k, 0))
        print(json.dumps({"rows": rows, "graph": graph, **tun, "us_per_step": round(1e3 * best / T, 3)}), flush=True)

    for graph in (False, True):
        for rows in (64, 128):
            run(rows, graph)
    for form in (1, 2, 3):
        for pf in (1, 2, 4):
            run(64, True, lstm_form=form, lstm_prefetch=pf)
    for rows in (64, 128):
        for probe in (1, 2, 3):
            run(rows, True, lstm_form=3, lstm_probe=probe)
    for probe in (10, 11, 12, 13):
        run(128, True, lstm_probe=probe)
    for probe in (20, 21, 22):                 # the 64-row step with pinned accumulators, G = 1, 2, 4
        run(64, True, lstm_probe=probe)
    for rows in (16, 32, 96):
        run(rows, True)
    # where the time of a step goes: in-kernel REFCLK stamps (summary lines on stderr)
    for rows in (64, 128):
        for probe in (0, 3):
            run(rows, False, lstm_form=3, lstm_probe=probe, lstm_stamps=1)
    run(128, False, lstm_probe=10, lstm_stamps=1)
    run(128, False, lstm_probe=13, lstm_stamps=1)


if __name__ == "__main__":
    main()
