#!/usr/bin/env python
"""benchmarks/deep_pipeline_probe.py -- how much does a deeper batch pipeline buy on one GPU?

Experiment, not the bench: N model replicas in one process, each driving the library's own batches-in-flight pipeline
(STTX_BatchSubmitDevice / STTX_BatchCollect), submitted round-robin from one host thread.  With N = 2 there are two
acoustic streams and up to four search streams on the device at once.  Prints ms per 64 x 5 s batch for every variant.
"""
import ctypes
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from stt_amd import Model, modelfile, native, synth
    native.lib().STTX_SetDevice(0)
    dev = torch.device("cuda", 0)
    K = int(os.environ.get("PROBE_STEPS", "24"))
    n_models = int(os.environ.get("PROBE_MODELS", "2"))
    weights = synth.synth_weights(0, n_hidden=2048, n_classes=29)
    sd = tempfile.TemporaryDirectory()
    scorer, _ = bench.synth_scorer(sd.name)
    models = []
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synth.sttw")
        modelfile.write_model(path, weights, synth.ENGLISH_LABELS, beam_width=500)
        for _ in range(n_models):
            m = Model(path)
            m.enableExternalScorer(scorer)
            models.append(m)
    n = 5 * 16000
    audio = np.stack([synth.synth_audio(n, seed=i) for i in range(64)])
    d_audio = torch.from_numpy(audio).to(dev)
    csz = (ctypes.c_uint * 64)(*([n] * 64))

    def run(depth_per_model, use_models):
        """K batches; every model keeps `depth_per_model` batches in flight; models take turns."""
        ms = models[:use_models]
        inflight = []  # (model, ticket) oldest first
        cap = depth_per_model * use_models
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = None
        for k in range(K):
            if len(inflight) == cap:
                m, t = inflight.pop(0)
                out = m.collectBatch(t)
                ref = ref or out
                assert out == ref
            m = ms[k % use_models]
            inflight.append((m, m.submitBatchDevice(d_audio.data_ptr(), n, csz)))
        while inflight:
            m, t = inflight.pop(0)
            out = m.collectBatch(t)
            assert out == ref
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / K

    for _ in range(2):
        for m in models:
            m.sttBatchDevice(d_audio.data_ptr(), n, [n] * 64)
    for depth, use in ((1, 1), (2, 1), (1, 2), (2, 2)) + (((1, 3), (2, 3)) if n_models >= 3 else ()):
        if use > n_models:
            continue
        r = [run(depth, use) for _ in range(3)]
        print("models %d x depth %d (in flight %d): ms per batch %s" % (use, depth, use * depth, " ".join("%.3f" % x for x in r)), flush=True)


if __name__ == "__main__":
    main()
