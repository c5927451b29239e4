"""Two models on two threads (the reference's safe pattern is one thread per model; a GPU engine "may be more permissive but
must not be less").  Found in round 6 by bench.py's streaming workload, one run in eight: while one thread CAPTURED a hop of its live set
into a hipGraph, the other thread grew a device buffer (hipDeviceSynchronize + hipFree) -- HIP answered "operation not permitted when
stream is capturing", invalidated the capture, and the call failed.  A process-wide lock now keeps captures and device-wide
synchronisations apart (engine.h: hip_capture_mutex).  This test drives exactly that overlap: thread A changes the shape of its live set
all the time (every new shape is captured at its second sighting), thread B keeps growing its buffers (longer and longer utterances).
Needs a MI355X."""
import threading

import numpy as np
import pytest

from stt_amd import model as M
from stt_amd import modelfile, native, synth

pytestmark = pytest.mark.gpu


def _model(tmp, name, seed):
    from stt_amd import Model
    w = synth.synth_weights(seed, n_hidden=256)
    path = str(tmp / (name + ".sttw"))
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=32)
    return Model(path)


def test_a_capture_on_one_thread_and_buffer_growth_on_another(tmp_path):
    ma, mb = _model(tmp_path, "a", 5), _model(tmp_path, "b", 6)
    audio = [synth.synth_audio(16000, seed=300 + i) for i in range(24)]
    hops = [5120] * 3
    want_a = [ma.stt(a[:sum(hops)]) for a in audio]
    errs, done = [], threading.Event()
    replays0 = native.get_tuning("hop_replays")

    def streamer():
        try:
            for rep in range(30):
                n = 2 + (rep * 5) % 19                       # a different number of live streams nearly every time: new graph keys
                for twice in range(2):                       # ... each seen twice: the second sighting is the capture
                    ss = [ma.createStream() for _ in range(n)]
                    k = 0
                    for h in hops:
                        M.feedAudioContentBatch(ss, [audio[i][k:k + h] for i in range(n)])
                        M.intermediateDecodeBatch(ss)
                        k += h
                    got = M.finishStreamBatch(ss)
                    assert got == want_a[:n], (rep, n)
        except Exception as ex:                              # noqa: BLE001
            errs.append(ex)
        finally:
            done.set()

    def grower():
        try:
            n = 4000
            while not done.is_set():
                n = n + 4099 if n < 400000 else 4000         # longer every time: features, windows, activations, probabilities all grow
                a = synth.synth_audio(n, seed=n)
                assert mb.acousticProbs([a, a[: n // 2]])[0].shape[0] > 0
        except Exception as ex:                              # noqa: BLE001
            errs.append(ex)

    ta, tb = threading.Thread(target=streamer), threading.Thread(target=grower)
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    assert native.get_tuning("hop_replays") > replays0       # (graphs were captured and replayed meanwhile)
