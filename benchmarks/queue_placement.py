"""One point of the queue-placement sweep (needs a MI355X): K idle HIP streams created BEFORE the model -- what a binding user's own copy
streams, a second library or torch do to a process -- then the batch pipeline (three acoustic engines, two groups in flight) on 64 x 5 s
batches; prints one JSON line: ms per batch in steady state, placements made, watch moves.  Round 5: 3.0 or 6.1 ms depending on K
(profiles/r05_queue_placement.json); since round 6 the engine PLACES the recurrence's and the output engine's streams by probing which
dispatch pipe each candidate shares (engine.cpp: place_engine_streams, benchmarks/pipe_probe.hip).

    python benchmarks/queue_placement.py --idle 3 --mode int8 [--place 0] [--moves 0]
tests/test_gpu_placement.py runs K = 0 .. 8 x {f16, int8} in fresh processes and asserts the spread."""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--idle", type=int, default=0)
    ap.add_argument("--mode", default="f16", choices=["f16", "int8"])
    ap.add_argument("--place", type=int, default=1)
    ap.add_argument("--moves", type=int, default=6)
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--bytes", action="store_true", help="a search-bound setup instead: byte-output model (256 classes), pruned_lm.bytes.scorer, beam 1024, four searches side by side")
    a = ap.parse_args()
    from stt_amd import Model, modelfile, native, synth
    from test_gpu_async import _DeviceArray
    native.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    scratch = _DeviceArray(np.zeros(1024, np.int16))
    keep = []
    for _ in range(a.idle):      # a stream takes its hardware queue when it is first used
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        assert hip.hipMemsetAsync(ctypes.c_void_p(scratch.data_ptr()), 0, 64, st) == 0
        assert hip.hipStreamSynchronize(st) == 0
        keep.append(st)
    native.set_tuning("am_place", a.place)
    native.set_tuning("am_moves", a.moves)
    native.set_tuning("am_i8", 1 if a.mode == "int8" else 0)
    w = synth.synth_weights(0, n_hidden=a.hidden, n_classes=256 if a.bytes else 29)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.sttw")
        modelfile.write_model(path, w, [bytes([i + 1]) for i in range(255)] if a.bytes else synth.ENGLISH_LABELS, beam_width=1024 if a.bytes else 500)
        m = Model(path)
    assert m.acousticMode() == (1 if a.mode == "int8" else 0)
    m.enableExternalScorer(os.path.join(ROOT, "tests", "golden", "fixtures", "pruned_lm.bytes.scorer" if a.bytes else "pruned_lm.scorer"))
    B, N = 64, 80000
    dev = [_DeviceArray(synth.synth_audio_batch(B, N, seed=100003 + v)) for v in range(4)]
    depth = m.pipelineDepth()

    def run(n):
        inflight, texts = [], []
        for k in range(n):
            if len(inflight) == depth:
                texts.append(m.collectBatch(inflight.pop(0)))
            inflight.append(m.submitBatchDevice(dev[k % 4].data_ptr(), N, [N] * B))
        while inflight:
            texts.append(m.collectBatch(inflight.pop(0)))
        return texts
    first = run(12)             # warm-up: rings, graphs, the placement (and whatever moves the watch still makes)
    t0 = time.perf_counter()
    again = run(a.batches)
    dt = time.perf_counter() - t0
    ok = all(again[k] == first[k % 4] for k in range(min(len(again), 8)))
    print(json.dumps({"idle_streams_before_the_model": a.idle, "mode": a.mode, "am_place": a.place, "am_moves": a.moves, "ms_per_batch": round(1e3 * dt / a.batches, 3),
                      "placements": native.get_tuning("am_placed") & 0xff, "candidates_behind_the_gemm_engine": native.get_tuning("am_placed") >> 8,
                      "watch_moves": native.get_tuning("am_moved"), "watched_step_us": native.get_tuning("am_step_us_x10") / 10.0, "transcripts_repeat": bool(ok)}), flush=True)


if __name__ == "__main__":
    main()
