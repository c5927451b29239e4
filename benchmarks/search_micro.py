"""The search kernel alone on configs[1]'s own emissions: the acoustic model's probabilities for 64 x 5 s of synthetic audio (near-uniform
softmax of a random-init model: the beam search's worst case, what bench.py times), 64 decoder streams, beam 500, the synthetic 500 k-word
scorer.  Prints ms per 64 x 250 frames (HIP events around the search launch), shader cycles per stream-timestep by phase and the
fine-grained stamps (ctc.hip: DecParams::stamps).  `--set name=value,...` applies tunables first.

    python benchmarks/search_micro.py [--reps 3] [--set search_step=0] [--scorer fixture]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--set", default="")
    ap.add_argument("--scorer", default="synthetic", choices=["synthetic", "fixture"])
    ap.add_argument("--emissions", default="model", choices=["model", "peaky"], help="peaky: bench.py's `peaky` workload (trained-like emissions of vocab.pruned.txt sentences; use --scorer fixture)")
    a = ap.parse_args()
    from stt_amd import native, synth
    native.lib()
    for kv in filter(None, a.set.split(",")):
        k, v = kv.split("=")
        native.set_tuning(k, int(v))
    model, _ = bench.make_model(29, bench.BEAM, synth.ENGLISH_LABELS)
    with tempfile.TemporaryDirectory() as d:
        path = bench.synth_scorer(d)[0] if a.scorer == "synthetic" else bench.FIXTURE_SCORER
        model.enableExternalScorer(path)
        n = int(bench.SECONDS * 16000)
        base = synth.synth_audio(n + 977 * a.streams, seed=11)
        audio = [base[977 * u:977 * u + n] for u in range(a.streams)]
        probs = np.stack(model.acousticProbs(audio))
        if a.emissions == "peaky":      # (the recipe of bench.py's peaky workload)
            vocab = open(os.path.join(bench.FIX, "vocab.pruned.txt")).read().split()
            rng = np.random.RandomState(7)
            em = []
            for i in range(a.streams):
                sent = ""
                while len(sent) < 48:
                    sent += (" " if sent else "") + str(rng.choice(vocab))
                lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent[:56]]
                em.append(synth.peaky_emissions(lab, 250, 29, 28, seed=int(rng.randint(1 << 30)), noise=0.02))
            probs = np.stack(em).astype(np.float32)
        T = probs.shape[1]
        for rep in range(a.reps):
            dec = model.createDecoder(a.streams, bench.BEAM)
            dec.setProfiling(2 if rep == a.reps - 1 else 1)
            dec.next(probs)
            ph, st, ms = dec.profile()
            stats = dec.stats()
            steps = max(1, stats["steps"])
            line = {"rep": rep, "search_ms": round(ms, 3), "us_per_stream_step": round(1e3 * ms / T, 3), "steps": stats["steps"],
                    "cand_per_step": round(stats["candidates"] / steps, 1), "lmq_per_step": round(stats["lm_queries"] / steps, 2)}
            if rep == a.reps - 1:
                line["phase_cycles_per_stream_step"] = {k: round(v / steps, 1) for k, v in ph.items()}
                line["cycles_per_stream_step"] = round(sum(v for k, v in ph.items() if not k.startswith("lm_wave")) / steps, 1)
                line["stamps_per_stream_step"] = [round(v / steps, 1) for v in st]
            print(line, flush=True)
            dec.close()


if __name__ == "__main__":
    main()
