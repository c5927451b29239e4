#!/usr/bin/env python
"""The recurrent step beside its co-tenant: STTX_TestLstmSteps with x-projection GEMMs of a 128-row chunk running on a second stream
(tunable lstm_cotenant), in-kernel REFCLK stamps on.  What does a GEMM workgroup on the same CU cost a step, and where?"""
import json
import os

os.environ.setdefault("STT_AMD_TEST_HOOKS", "1")   # a probe of single kernels: needs libstt_test.so (include/stt_amd_test.h)
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stt_amd import Model, modelfile, native, synth  # noqa: E402

H, P = 2048, 4


def main():
    w = synth.synth_weights(0, n_hidden=H, n_classes=29)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.sttw")
        modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
        m = Model(path)
    rng = np.random.default_rng(0)
    x128 = (rng.standard_normal((P, 128, 4 * H)) * 1.5).astype(np.float32)
    defaults = {}

    def run(rows, steps, **tun):
        for k, v in tun.items():
            defaults.setdefault(k, native.get_tuning(k))
            native.set_tuning(k, v)
        x = np.ascontiguousarray(x128[:, :rows]).reshape(P * rows, 4 * H)
        sys.stderr.write("RUN rows %d steps %d %s\n" % (rows, steps, json.dumps(tun)))
        sys.stderr.flush()
        ms = m.lstmSteps(x, rows, steps, graph=True, timing=True)[3]
        for k in tun:
            native.set_tuning(k, defaults[k])
        print(json.dumps({"rows": rows, "steps": steps, **tun, "us_per_step": round(1e3 * ms / steps, 3)}), flush=True)

    for rows in (128, 64):
        run(rows, 500, lstm_stamps=1)
        for solo in (3, 2):
            run(rows, 500, lstm_stamps=1, lstm_cotenant=24, dense_solo=solo)
        run(rows, 500, lstm_stamps=1, lstm_cotenant=24, dense_solo=3, lstm_prio=0)
    run(128, 2, lstm_cotenant=12, dense_solo=3)      # the GEMMs (nearly) alone
    run(128, 2, lstm_cotenant=12, dense_solo=2)
    run(128, 2, lstm_cotenant=12, dense_solo=0, dense_lds_kb=0)
    run(128, 500, lstm_stamps=1, lstm_cotenant=24, dense_solo=3, lstm_probe=3)     # no operand loads in the step: pure issue / pipe contention
    run(128, 500, lstm_stamps=1, lstm_cotenant=24, dense_solo=3, lstm_probe=1)
    run(128, 500, lstm_stamps=1, lstm_cotenant=24, dense_solo=3, lstm_probe=2)


if __name__ == "__main__":
    main()
