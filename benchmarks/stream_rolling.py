#!/usr/bin/env python
"""configs[2] with a ROLLING live set: S streams are live at all times -- a stream that has consumed its utterance is finished and the
next utterance takes its place -- fed in 320 ms hops with an intermediate decode of every live stream after every hop.  Per hop: feed
+ decode latency of ALL live streams, what else happened in that hop (streams finished / created), and the outliers with their context.

    python benchmarks/stream_rolling.py [--utterances 1000] [--streams 128] [--scorer synthetic|fixture] [--set name=value,...]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


ONE_DECODE = True   # STTX_DecodeStreamsBatch; --two-decodes: STTX_IntermediateDecodeBatch + STTX_FinishStreamBatch
DEFER = True  # ... and what the flush leaves behind the hop's pass rides in the next hop (aLast = 2); --no-defer: a pass of its own
FOLD = True   # a stream's last hop carries its flush (STTX_FeedAudioContentBatchEx); --no-fold: the flush is finishStreamBatch's own pass


def run(model, utts, S, M):
    """-> (texts by utterance, hop records [(feed_s, decode_s, finish_s, create_s, n_live, n_finished, n_created)])"""
    texts = [None] * len(utts)
    nxt = 0
    live = []      # [utterance index, stream, samples consumed]
    drain = []     # DEFER: streams whose last audio went in with the previous hop (aLast = 2); the rest of their flush rides in this hop
    recs = []
    empty = np.zeros(0, dtype=np.int16)
    while nxt < len(utts) or live or drain:
        t0 = time.perf_counter()
        n_created = 0
        while len(live) + len(drain) < S and nxt < len(utts):
            live.append([nxt, model.createStream(), 0]); nxt += 1; n_created += 1
        t1 = time.perf_counter()
        code = 2 if DEFER else 1
        M.feedAudioContentBatch([s for _, s, _ in live] + [s for _, s, _ in drain], [utts[u][k:k + 5120] for u, _, k in live] + [empty] * len(drain),
                                last=([code if k + 5120 >= len(utts[u]) else 0 for u, _, k in live] + [0] * len(drain) if FOLD else None))
        t2 = time.perf_counter()
        if DEFER and ONE_DECODE:   # the hop's intermediate results and the finishes of the drained streams in one launch
            out = M.decodeStreamsBatch([s for _, s, _ in live] + [s for _, s, _ in drain], [False] * len(live) + [True] * len(drain))
            for e, t in zip(drain, out[len(live):]):
                texts[e[0]] = t
        else:
            M.intermediateDecodeBatch([s for _, s, _ in live])
        t3 = time.perf_counter()
        for e in live:
            e[2] += 5120
        done = drain if DEFER else [e for e in live if e[2] >= len(utts[e[0]])]
        if done and not (DEFER and ONE_DECODE):
            for e, t in zip(done, M.finishStreamBatch([e[1] for e in done])):
                texts[e[0]] = t
        drain = [e for e in live if e[2] >= len(utts[e[0]])] if DEFER else []
        live = [e for e in live if e[2] < len(utts[e[0]])]
        t4 = time.perf_counter()
        recs.append((t2 - t1, t3 - t2, t4 - t3, t1 - t0, len(live) + len(done), len(done), n_created))
    return texts, recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=1000)
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--passes", type=int, default=2, help="the first pass warms the stream pool; the last one is reported")
    ap.add_argument("--scorer", default="synthetic", choices=["synthetic", "fixture"])
    ap.add_argument("--set", default="")
    ap.add_argument("--no-fold", action="store_true")
    ap.add_argument("--no-defer", action="store_true")
    ap.add_argument("--two-decodes", action="store_true")
    ap.add_argument("--cohorts", type=int, default=1, help="independent live sets of --streams streams, each on its own Model replica and host thread (their passes overlap on the GPU)")
    a = ap.parse_args()
    global FOLD, DEFER, ONE_DECODE
    ONE_DECODE = not a.two_decodes
    FOLD = not a.no_fold
    DEFER = FOLD and not a.no_defer
    from stt_amd import model as M
    from stt_amd import native, synth
    native.lib()
    for kv in filter(None, a.set.split(",")):
        k, v = kv.split("=")
        native.set_tuning(k, int(v))
    models = [bench.make_model(29, bench.BEAM, synth.ENGLISH_LABELS)[0] for _ in range(a.cohorts)]
    model = models[0]
    with tempfile.TemporaryDirectory() as d:
        sp = bench.synth_scorer(d)[0] if a.scorer == "synthetic" else bench.FIXTURE_SCORER
        for m_ in models:
            m_.enableExternalScorer(sp)
        rng = np.random.RandomState(1)
        base = synth.synth_audio(15 * 16000, seed=3)
        utts = [np.roll(base, 977 * u)[:int(rng.uniform(1, 15) * 16000)].copy() for u in range(a.utterances)]
        audio_s = sum(len(x) for x in utts) / 16000.0
        for ps in range(a.passes):
            t0 = time.perf_counter()
            if a.cohorts == 1:
                texts, recs = run(model, utts, a.streams, M)
            else:
                import threading
                out = [None] * a.cohorts
                parts = [list(range(c, len(utts), a.cohorts)) for c in range(a.cohorts)]

                def work(c):
                    out[c] = run(models[c], [utts[i] for i in parts[c]], a.streams, M)
                th = [threading.Thread(target=work, args=(c,)) for c in range(a.cohorts)]
                [t.start() for t in th]; [t.join() for t in th]
                texts = [None] * len(utts)
                recs = []
                for c in range(a.cohorts):
                    for i, t in zip(parts[c], out[c][0]):
                        texts[i] = t
                    recs += out[c][1]
            wall = time.perf_counter() - t0
            r = np.array(recs)
            hop = (r[:, 0] + r[:, 1]) * 1e3
            full = r[:, 4] == a.streams
            line = {"pass": ps, "utterances": a.utterances, "streams": a.streams, "audio_s": round(audio_s, 1), "wall_s": round(wall, 3), "rtf_x": round(audio_s / wall, 1),
                    "hops": int(len(hop)), "hops_with_a_full_live_set": int(full.sum()),
                    "hop_ms": {k: round(float(np.percentile(hop, q)), 3) for k, q in (("p50", 50), ("p90", 90), ("p95", 95), ("p99", 99), ("max", 100))},
                    "feed_ms_p50": round(float(np.median(r[:, 0])) * 1e3, 3), "decode_ms_p50": round(float(np.median(r[:, 1])) * 1e3, 3),
                    "finish_ms_per_hop_mean": round(float(r[:, 2].mean()) * 1e3, 3), "create_ms_per_hop_mean": round(float(r[:, 3].mean()) * 1e3, 3),
                    "non_empty": sum(1 for t in texts if t)}
            print(json.dumps(line), flush=True)
            worst = np.argsort(-hop)[:8]
            print("  outliers (hop index: feed, decode, finish, create ms | live, finished, created):",
                  [(int(i), round(r[i, 0] * 1e3, 2), round(r[i, 1] * 1e3, 2), round(r[i, 2] * 1e3, 2), round(r[i, 3] * 1e3, 2), int(r[i, 4]), int(r[i, 5]), int(r[i, 6])) for i in worst], flush=True)


if __name__ == "__main__":
    main()
