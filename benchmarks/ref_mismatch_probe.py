"""Where does the GPU beam search leave the real reference decoder?  Feeds the emissions of chosen utterances of bench.py's timed batches
frame by frame to oracle/_ref (the reference), oracle's C port and the GPU decoder (both word-mode steps) and reports the first frame at
which their beams differ.  Test infrastructure (imports oracle/); writes the emissions to gpurun_out/ for analysis without a GPU."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import port, ref  # noqa: E402
from stt_amd import native, synth  # noqa: E402

CASES = [(5, 45), (6, 1), (6, 18), (6, 60)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]


def canon(sc, pb, pnb, ch):
    a = np.stack([sc.view(np.uint32).astype(np.int64), pb.view(np.uint32).astype(np.int64), pnb.view(np.uint32).astype(np.int64), ch.astype(np.int64)], 1)
    return a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]


def main():
    model, _ = bench.make_model(29, 500, synth.ENGLISH_LABELS)
    d = tempfile.TemporaryDirectory()
    scorer, _ = bench.synth_scorer(d.name)
    model.enableExternalScorer(scorer)
    A = ref.Alphabet(os.path.join(bench.FIX, "alphabet.txt"))
    S = ref.Scorer(scorer, A)
    labels, space = port.parse_alphabet_file(os.path.join(bench.FIX, "alphabet.txt"))
    P = port.Scorer(scorer)
    out = {}
    for kk, b in CASES:
        audio = synth.synth_audio_batch(64, 80000, seed=100003 + kk)
        probs = model.acousticProbs([audio[b]])[0]
        np.save(os.path.join(ROOT, "gpurun_out", "probe_probs_%d_%d.npy" % (kk, b)), probs)
        dr, dp = ref.Decoder(A, 500, S), port.Decoder(labels, space, 500, P)
        gpus = {}
        for step in (2, 0):
            native.set_tuning("search_step", step)
            gpus[step] = model.createDecoder(1, 500)
        first = {}
        for t in range(probs.shape[0]):
            fr = probs[t:t + 1]
            dr.next(fr.astype(np.float64)); dp.next(fr)
            rs = canon(*dr.raw_beam()[:4]); ps = canon(*dp.raw_beam()[:4])
            beams = {"port": ps}
            for step, g in gpus.items():
                native.set_tuning("search_step", step)
                g.next(fr)
                beams["gpu%d" % step] = canon(*g.raw_beam(0))
            for name, bm in beams.items():
                if name not in first and not (bm.shape == rs.shape and np.array_equal(bm, rs)):
                    only_r = sorted(set(map(tuple, rs)) - set(map(tuple, bm)))
                    only_b = sorted(set(map(tuple, bm)) - set(map(tuple, rs)))
                    first[name] = {"frame": t, "n_ref": int(len(rs)), "n_other": int(len(bm)), "only_ref": len(only_r), "only_other": len(only_b),
                                   "ref_worst_score": float(np.sort(dr.raw_beam()[0])[0]),
                                   "only_ref_scores": [float(np.array([x[0]], dtype=np.uint32).view(np.float32)[0]) for x in only_r[:6]],
                                   "only_other_scores": [float(np.array([x[0]], dtype=np.uint32).view(np.float32)[0]) for x in only_b[:6]],
                                   "only_ref_ch": [int(x[3]) for x in only_r[:6]], "only_other_ch": [int(x[3]) for x in only_b[:6]]}
        native.set_tuning("search_step", 2)
        rr = dr.decode(1)[0]; pr = dp.decode(1)[0]
        res = {"first_divergence_from_reference": first, "ref": (float(rr[0]), A.decode(rr[1]).decode()), "port": (float(pr[0]), A.decode(pr[1]).decode())}
        for step, g in gpus.items():
            native.set_tuning("search_step", step)
            gr = g.decode(1)[0][0]
            res["gpu%d" % step] = (float(gr[0]), A.decode(gr[1]).decode(), g.stats() if hasattr(g, "stats") else None)
        native.set_tuning("search_step", 2)
        out["%d:%d" % (kk, b)] = res
        print(kk, b, json.dumps(res, indent=1), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ref_mismatch_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
