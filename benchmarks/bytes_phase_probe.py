#!/usr/bin/env python
"""Where a bytes-mode search step spends its cycles: the decoder stage alone (STTX_Decoder*), code-point scorer, beam 1024, on
(a) peaky byte emissions of code-point sentences (bench.py --workload peaky_bytes) and (b) near-uniform emissions (what the random-init
model of --workload bytes produces).  Per stream-timestep: phase cycles (in-kernel counters), candidates, LM queries, memo probes.

    python benchmarks/bytes_phase_probe.py [--set name=value,...]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="")
    ap.add_argument("--frames", type=int, default=250)
    a = ap.parse_args()
    from stt_amd import native, synth
    native.lib()
    for kv in filter(None, a.set.split(",")):
        k, v = kv.split("=")
        native.set_tuning(k, int(v))
    model, _ = bench.make_model(256, 1024, [bytes([i + 1]) for i in range(255)])
    with tempfile.TemporaryDirectory() as d:
        sp, desc = bench.synth_codepoint_scorer(d)
        model.enableExternalScorer(sp)
        rng = np.random.RandomState(11)
        T = a.frames
        peaky = []
        for i in range(64):
            lab = []
            for cp in 0x4E00 + rng.randint(0, 6000, size=19):
                lab += [(0xE0 | (cp >> 12)) - 1, (0x80 | ((cp >> 6) & 0x3F)) - 1, (0x80 | (cp & 0x3F)) - 1]
            peaky.append(synth.peaky_emissions(lab, T, 256, 255, seed=int(rng.randint(1 << 30)), noise=0.02 * 29 / 256, lead=10))
        flat = rng.dirichlet(np.full(256, 40.0), size=(64, T)).astype(np.float32)     # near-uniform: every class within a factor ~1.5
        for name, em in (("peaky", np.stack(peaky).astype(np.float32)), ("near-uniform", flat)):
            for level in (0, 2):
                dec = model.createDecoder(64, 1024)
                dec.setProfiling(level)
                t0 = time.perf_counter()
                dec.next(em)
                res = dec.decode(1, 256)
                ms = 1e3 * (time.perf_counter() - t0)
                if level:
                    ph, st, kms = dec.profile()
                    s = dec.stats()
                    n = max(1, s["steps"])
                    print(json.dumps({"emissions": name, "ms_profiled": round(ms, 2), "search_ms": round(kms, 2), "per_stream_timestep": {
                        "phase_cycles": {k: int(v / n) for k, v in ph.items()}, "candidates": round(s["candidates"] / n, 1), "lm_queries": round(s["lm_queries"] / n, 1),
                        "memo_probes": round(s["lm_probes"] / n, 1),
                        # wave 0's share of the expand loop (cycles per stream-timestep): locating the item + issuing its arc read | waiting for the arc,
                        # class position, log-probability, key | filter + path hash | event or candidate record; [54] = items wave 0 took
                        "expand_wave0": {"locate": int(st[50] / n), "arc_to_key": int(st[51] / n), "hash": int(st[52] / n), "record": int(st[53] / n), "items": round(st[54] / n, 1)}},
                        "error": s["error"], "non_empty": sum(1 for r in res if r and len(r[0][1]))}), flush=True)
                else:
                    print(json.dumps({"emissions": name, "ms": round(ms, 2)}), flush=True)
                dec.close()


if __name__ == "__main__":
    main()
