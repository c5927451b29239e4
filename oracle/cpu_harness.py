"""oracle/cpu_harness.py -- BASELINE INFRASTRUCTURE ONLY: the reference's CPU evaluation pattern for bench.py's cpu_baseline.

training/coqui_stt_training/evaluate_export.py:65-80 runs the exported model on the host with one worker PROCESS per
core group, each pulling utterances from a queue and running the whole native_client path on them (batch 1, interpreter
with 4 threads, tflitemodelstate.cc:200).  Here a worker runs: MFCC + acoustic model = oracle/am_torch.py (torch-CPU f32,
4 threads; restatement, TFLite itself is not in the tree) and the beam search = the REAL reference decoder
(oracle/_ref/libctcdecode_ref.so: DecoderState with the same scorer, beam 500), one utterance at a time.
Workers are spawned (not forked: the parent holds a HIP context), load the weights from memory-mapped .npy files and the
scorer by path, report ready, and are timed from a common start."""
import os
import sys
import time


def _worker(rank, n_workers, wdir, audio_path, scorer, alphabet, beam, threads, ready, go, out):
    import warnings

    import numpy as np
    warnings.filterwarnings("ignore")     # (torch warns about read-only memory-mapped weights; they are never written)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import torch
    torch.set_num_threads(threads)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import am_torch, ref
    W = am_torch.to_torch({k: np.load(os.path.join(wdir, k.replace("/", "__") + ".npy"), mmap_mode="r") for k in
                           ["layer_1/weights", "layer_1/bias", "layer_2/weights", "layer_2/bias", "layer_3/weights", "layer_3/bias",
                            "lstm/kernel", "lstm/bias", "layer_5/weights", "layer_5/bias", "layer_6/weights", "layer_6/bias"]})
    audio = np.load(audio_path, mmap_mode="r")
    A = ref.Alphabet(alphabet)
    S = ref.Scorer(scorer, A) if scorer else None
    am_torch.utterance_probs(audio[0][:8000], W)          # warm-up: thread pool, page cache
    ready.put(rank)
    go.wait()
    t_am = t_dec = 0.0
    texts = []
    for i in range(rank, audio.shape[0], n_workers):
        t0 = time.perf_counter()
        probs = am_torch.utterance_probs(audio[i], W)
        t1 = time.perf_counter()
        d = ref.Decoder(A, beam, S)
        d.next(probs.astype(np.float64))
        res = d.decode(1)
        t2 = time.perf_counter()
        t_am += t1 - t0; t_dec += t2 - t1
        texts.append((i, len(res[0][1]) if res else 0))
    out.put((rank, t_am, t_dec, texts))


def run(weights, audio, scorer, alphabet, beam, workers, threads=4, tmpdir=None):
    """weights: dict of f32 arrays; audio: list of equal-length int16 arrays.  Returns a dict with wall seconds etc."""
    import multiprocessing as mp
    import tempfile

    import numpy as np
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        for k, v in weights.items():
            np.save(os.path.join(d, k.replace("/", "__") + ".npy"), np.ascontiguousarray(v, dtype=np.float32))
        apath = os.path.join(d, "audio.npy")
        np.save(apath, np.stack(audio))
        n = min(workers, len(audio))
        ready, out, go = ctx.Queue(), ctx.Queue(), ctx.Event()
        procs = [ctx.Process(target=_worker, args=(r, n, d, apath, scorer, alphabet, beam, threads, ready, go, out)) for r in range(n)]
        for p in procs:
            p.start()
        for _ in range(n):
            ready.get(timeout=600)
        t0 = time.perf_counter()
        go.set()
        res = [out.get(timeout=1200) for _ in range(n)]
        wall = time.perf_counter() - t0
        for p in procs:
            p.join(timeout=60)
    return {"wall_s": wall, "workers": n, "threads_per_worker": threads,
            "am_s_per_utt": sum(r[1] for r in res) / len(audio), "dec_s_per_utt": sum(r[2] for r in res) / len(audio)}
