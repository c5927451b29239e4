"""Round-5 diagnostic (needs a MI355X): the int8 path's batch pipeline as one acoustic queue and as three engines (tunable am_i8_pipe) on the
same batches -- milliseconds per batch and how many rows took the recurrent step's slow path -- under the conditions in which bench.py runs
it (a float container quantised at load, torch in the process, a second model alive), one at a time."""
import json
import os

os.environ.setdefault("STT_AMD_TEST_HOOKS", "1")   # a probe of single kernels: needs libstt_test.so (include/stt_amd_test.h)
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from stt_amd import Model, modelfile, native, synth, tflitefile      # noqa: E402
from test_gpu_async import _DeviceArray                               # noqa: E402

SCORER = os.path.join(ROOT, "tests", "golden", "fixtures", "pruned_lm.scorer")
w = synth.synth_weights(0, n_hidden=2048)
d = tempfile.mkdtemp()
qpath, fpath = os.path.join(d, "q.tflite"), os.path.join(d, "f.sttw")
tflitefile.write_tflite(qpath, w, synth.ENGLISH_LABELS, quantize=True, beam_width=500)
modelfile.write_model(fpath, w, synth.ENGLISH_LABELS, beam_width=500)
B, N, K = 64, 80000, 12
audio = [synth.synth_audio_batch(B, N, seed=100003 + v) for v in range(4)]
out = {}
keep_alive = []


def load(kind):
    if kind == "quantised .tflite":
        return Model(qpath)
    native.set_tuning("am_i8", 1)
    try:
        return Model(fpath)
    finally:
        native.set_tuning("am_i8", -1)


def measure(label, kind, pipe, use_torch=False):
    native.set_tuning("am_i8_pipe", pipe)
    m = load(kind)
    assert m.acousticMode() == 1
    m.enableExternalScorer(SCORER)
    if use_torch:
        import torch
        dev = [torch.from_numpy(a).to("cuda:0") for a in audio]
    else:
        dev = [_DeviceArray(a) for a in audio]
    depth = m.pipelineDepth()

    def run(n):
        inflight = []
        for k in range(n):
            if len(inflight) == depth:
                m.collectBatch(inflight.pop(0))
            inflight.append(m.submitBatchDevice(dev[k % 4].data_ptr(), N, [N] * B))
        while inflight:
            m.collectBatch(inflight.pop(0))
    run(4)
    s0 = m.slowRows()
    t0 = time.perf_counter()
    run(K)
    dt = time.perf_counter() - t0
    out["%s, am_i8_pipe=%d" % (label, pipe)] = {"ms_per_batch": round(1e3 * dt / K, 3), "slow_rows_per_batch": (m.slowRows() - s0) / K}
    print(label, pipe, out["%s, am_i8_pipe=%d" % (label, pipe)], flush=True)


import ctypes                                          # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
scratch = _DeviceArray(np.zeros(1024, np.int16))


def shift_queues(n):
    """n more HIP streams, each used once (a stream takes its hardware queue when it is created or first used): the next streams the engine
    creates land n hardware queues further on."""
    for _ in range(n):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        assert hip.hipMemsetAsync(ctypes.c_void_p(scratch.data_ptr()), 0, 64, st) == 0
        assert hip.hipStreamSynchronize(st) == 0
        keep_alive.append(st)


which = sys.argv[1] if len(sys.argv) > 1 else "models"
if which == "shift":
    for n in range(0, 9):
        if n:
            shift_queues(1)
        measure("%d streams created before the model" % n, "float", 1)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "i8_pipe_diag_shift.json"), "w"), indent=1)
    sys.exit(0)
for pipe in (0, 1):
    measure("quantised .tflite", "quantised .tflite", pipe)
for pipe in (0, 1):
    measure("float container quantised at load", "float", pipe)
keep_alive.append(Model(fpath))                     # an f16 model of the same size alive in the process (bench.py holds one)
keep_alive[0].enableExternalScorer(SCORER)
for pipe in (0, 1):
    measure("... + an f16 model alive", "float", pipe)
keep_alive[0].sttBatch([audio[0][i] for i in range(4)])
for pipe in (0, 1):
    measure("... + that model has run a batch", "float", pipe)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "i8_pipe_diag.json"), "w"), indent=1)
