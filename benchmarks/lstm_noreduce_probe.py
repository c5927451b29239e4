#!/usr/bin/env python
"""What could a recurrent step without a cross-wave reduction save?  The 128-row step alone and beside x-projection GEMMs, as shipped
and with timing probe 4 (every wave settles its tile from its own partial sums: no LDS parking, no barrier; wrong results)."""
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
w = synth.synth_weights(0, n_hidden=H, n_classes=29)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=500)
    m = Model(path)
x = (np.random.default_rng(0).standard_normal((P * 128, 4 * H)) * 1.5).astype(np.float32)
for tun in ({}, {"lstm_probe": 4}, {"lstm_cotenant": 24, "dense_solo": 3}, {"lstm_cotenant": 24, "dense_solo": 3, "lstm_probe": 4},
            {"lstm_probe": 7}, {"lstm_cotenant": 24, "dense_solo": 3, "lstm_probe": 7}):
    old = {k: native.get_tuning(k) for k in tun}
    for k, v in tun.items():
        native.set_tuning(k, v)
    ms = m.lstmSteps(x, 128, 500, graph=True, timing=True)[3]
    for k, v in old.items():
        native.set_tuning(k, v)
    print(json.dumps({**tun, "us_per_step": round(1e3 * ms / 500, 2)}), flush=True)
