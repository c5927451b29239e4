#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

metric   audio-seconds per wall-second (RTF x), whole job, audio already resident in HBM when the clock starts
step     one pass of the hot path (MFCC -> dense x3 -> LSTM-2048 -> dense x2 -> softmax -> CTC beam search + KenLM/FST scorer)
         over one batch per GPU; configs[1]: 64 synthetic 5 s 16 kHz utterances, English geometry, beam 500, scorer
weights  seeded random init of the reference architecture (no checkpoint exists offline); scorer = a synthetic
         huge-vocabulary package written at start-up by stt_amd/tools (500 k pseudo-words, order 5, 30 M n-grams, KenLM
         `-a 255 -q 8 trie` layout = the release recipe of doc/LANGUAGE_MODEL.rst:52-62; no corpus or lmplz offline).
         --scorer fixture switches to the reference's small data/smoke_test/pruned_lm.scorer.
scaling  weak: every rank decodes its own 64 utterances; one RCCL gather of the transcripts per step

One JSON line on rank 0, including `roofline` (dominant kernel, algorithmic bytes / measured HIP-event time on the
engine's own stream) and `cpu_baseline` (oracle on the host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FIX = os.path.join(ROOT, "tests", "golden", "fixtures")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
H, C, BEAM, BATCH, SECONDS = 2048, 29, 500, 64, 5.0
SCORER_PATH = os.path.join(FIX, "pruned_lm.scorer")
SCORER_DESC = "pruned_lm.scorer fixture (quant-array-trie order 4)"


def cpu_baseline(model_weights, audio, probs_gpu, n_utts):
    """Oracle on the host cores: numpy restatement of the acoustic stage (port) + the compiled reference decoder
    (ctc_beam_search_decoder_batch, one thread per core) on the same emissions.  Bounded sample of the same workload."""
    from oracle import am_ref, ref
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    for a in audio[:n_utts]:
        am_ref.utterance_probs(a, model_weights, dtype=np.float32)
    t_am = time.perf_counter() - t0
    kind = "port"
    t1 = time.perf_counter()
    if ref.available():
        A = ref.Alphabet(os.path.join(FIX, "alphabet.txt"))
        S = ref.Scorer(SCORER_PATH, A)
        p = np.stack([probs_gpu[i] for i in range(n_utts)]).astype(np.float64)
        ref.decode_batch(p, [p.shape[1]] * n_utts, A, BEAM, cores, S)
        dec = "reference ctc_beam_search_decoder_batch (oracle/_ref), %d threads" % cores
    else:
        from oracle import port
        labels, space = port.parse_alphabet_file(os.path.join(FIX, "alphabet.txt"))
        P = port.Scorer(SCORER_PATH)
        for i in range(n_utts):
            d = port.Decoder(labels, space, BEAM, P); d.next(probs_gpu[i]); d.decode(1)
        dec = "C port decoder, 1 thread"
    t_dec = time.perf_counter() - t1
    secs = n_utts * SECONDS
    return {"value": secs / (t_am + t_dec), "unit": "audio-seconds/sec", "cores": cores, "kind": kind,
            "sample": "%d of the %d utterances (%.0f audio-s): numpy f32 restatement of MFCC+acoustic model (BLAS threads, one utterance at a time like "
                      "the reference's batch-1 interpreter) %.2f s + %s (one utterance per thread: %d busy) %.2f s"
                      % (n_utts, BATCH, secs, t_am, dec, min(n_utts, cores), t_dec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scorer", default="synthetic", choices=["synthetic", "fixture"])
    ap.add_argument("--no-profile", action="store_true", help="experiment: no HIP-event stage timing inside the timed region")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the engine has no CPU path")
    # STT_BENCH_BACKEND=gloo: plumbing check of the N>1 path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives on host tensors); the measured configuration is always nccl (= RCCL), one rank per GPU
    backend = os.environ.get("STT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else None      # where the collectives' tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from stt_amd import dist as _sd
        _sd.assume_equal_batches()      # weak scaling: every rank decodes BATCH utterances -> the gather is one collective

    from stt_amd import Model, modelfile, native, synth
    from stt_amd import dist as sdist
    native.lib().STTX_SetDevice(local_rank)

    weights = synth.synth_weights(0, n_hidden=H, n_classes=C)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "english_synth.sttw")
        modelfile.write_model(path, weights, synth.ENGLISH_LABELS, beam_width=BEAM)
        model = Model(path)
    global SCORER_PATH, SCORER_DESC
    scorer_dir = None
    if args.scorer == "synthetic":
        from stt_amd import scorertools
        scorer_dir = tempfile.TemporaryDirectory()
        lm, vocab = os.path.join(scorer_dir.name, "lm.binary"), os.path.join(scorer_dir.name, "vocab.txt")
        SCORER_PATH = os.path.join(scorer_dir.name, "synthetic_500k.scorer")
        t_s = time.perf_counter()
        scorertools.synth_lm(lm, vocab, words=500000, order=5, seed=7, avg={2: 24, 3: 1.2, 4: 0.7, 5: 0.5})
        scorertools.generate_scorer_package(lm, vocab, SCORER_PATH, alphabet=os.path.join(FIX, "alphabet.txt"),
                                            default_alpha=0.931289039105002, default_beta=1.1834137581510284)   # doc/LANGUAGE_MODEL.rst:80-81
        SCORER_DESC = ("synthetic huge-vocabulary scorer (500 k words, order 5, quant-array-trie `-a 255 -q 8`, %.0f MB, built in %.0f s)"
                       % (os.path.getsize(SCORER_PATH) / 1e6, time.perf_counter() - t_s))
    model.enableExternalScorer(SCORER_PATH)

    n = int(SECONDS * 16000)
    audio = [synth.synth_audio(n, seed=1000 * rank + i) for i in range(BATCH)]
    stride = n
    d_audio = torch.from_numpy(np.stack(audio)).to(dev)     # int16 [B][stride], resident in HBM before the clock starts
    sizes = [n] * BATCH
    ptr = d_audio.data_ptr()

    def step():
        texts = model.sttBatchDevice(ptr, stride, sizes)
        return sdist.gather_transcripts(texts, device=cdev) if world > 1 else [texts]

    for _ in range(args.warmup):
        step()
    model.setProfiling(not args.no_profile)
    stage = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        st = model.stageTimes()
        for k, v in st.items():
            stage[k] = stage.get(k, 0.0) + v
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dstats = model.decoderStats()
    model.setProfiling(2)            # one extra, untimed step with the search kernel's phase cycle counters on
    step()
    dphase = model.decoderPhaseCycles()
    dstamps = model.decoderStamps()
    model.setProfiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0 and args.no_profile:
        print(json.dumps({"experiment": "no-profile", "ms_per_step": 1e3 * elapsed / args.steps, "value": world * BATCH * SECONDS * args.steps / elapsed}))
    elif rank == 0:
        audio_s = world * BATCH * SECONDS * args.steps
        K = args.steps
        T = 250
        # ---- roofline of the dominant kernel (by HIP-event time on the engine stream)
        lstm_launches = stage["lstm_launches"]
        lstm_avg_ms = stage["lstm_ms"] / max(1.0, lstm_launches)
        # recurrent matrix H x 4H f16 once per batch-timestep + per-row x-projection (f32 4H) in and h (f16 H) in/out, c (f32 H) in/out
        lstm_bytes = H * 4 * H * 2 + BATCH * (4 * H * 4 + 2 * H * 2 + 2 * H * 4)
        dec_ms = stage["decoder_next_ms"] / K
        # per utterance-timestep C*4 B of probabilities in, beam state ~ beam*40 B read + written, 8 B per counted LM probe
        dec_bytes = BATCH * T * (C * 4 + 2 * BEAM * 40) + 8.0 * dstats["lm_probes"]
        kernels = {
            "lstm_step_kernel<4, 2>": {"avg_ms": lstm_avg_ms, "launches_per_step": lstm_launches / K, "bytes": lstm_bytes,
                                    "share_ms": stage["lstm_ms"] / K},
            "ctc_next_kernel": {"avg_ms": dec_ms, "launches_per_step": 1, "bytes": dec_bytes, "share_ms": dec_ms},
        }
        dom = max(kernels, key=lambda k: kernels[k]["share_ms"])
        ach = kernels[dom]["bytes"] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json: FETCH_SIZE and
        # WRITE_SIZE cannot share a pass, and counters are never collected inside a timed run); bytes per batch, like `achieved`
        traffic, traffic_note = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            key = [k for k in pmc if k.startswith(dom.split("<")[0])][0]
            e = pmc[key]
            wide = dom.startswith("lstm")   # 16 B/lane coalesced streams: FETCH_SIZE reads 1/2 on gfx950 (MI355X_MICROARCH.md, HBM)
            traffic = (e["fetch_kb_per_launch"] * (2.0 if wide else 1.0) + e["write_kb_per_launch"]) * 1024.0 * (e["launches_per_batch"] if not wide else 1.0)
            traffic_note = "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (separate passes, profiles/r01_pmc_per_kernel.csv), bytes per %s" % ("launch" if wide else "batch (%d chunk launches)" % round(e["launches_per_batch"]))
        except Exception:
            pass
        roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_note": traffic_note,
                    "all": {k: {"GB/s": v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9, "avg_ms": v["avg_ms"], "ms_per_step": v["share_ms"]}
                            for k, v in kernels.items()}}
        res = {
            "metric": "audio-seconds/sec (RTF)", "value": audio_s / elapsed, "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (MFMA operands, f32 accumulate/state; decoder f32+f64)", "data": "synthetic",
            "config": {"workload": "configs[1]: batch=64 synthetic 5 s 16 kHz utterances per GPU, English geometry (n_hidden 2048, 29 classes), "
                                   "beam_width=500, KenLM scorer = " + SCORER_DESC,
                       "global_batch": world * BATCH, "parallelism": "dp%d (utterance shards, RCCL transcript gather)" % world},
            "p50_utterance_latency_ms": 1e3 * elapsed / args.steps,   # a batch completes together: submit -> transcripts on host
            "stage_ms_per_step": {k: v / K for k, v in stage.items() if k.endswith("_ms")},
            "decoder_counters_last_step": dstats,
            # (the LM wave runs beside the expand phases: not part of the serial sum)
            "decoder_phase_cycle_share": {k: round(v / max(1, sum(x for n, x in dphase.items() if not n.startswith("lm_wave"))), 4)
                                          for k, v in dphase.items() if not k.startswith("lm_wave")},
            "decoder_phase_cycles_per_stream_step": {k: round(v / max(1, dstats["steps"]), 1) for k, v in dphase.items()},
            "decoder_stamp_cycles_per_stream_step": [round(v / max(1, dstats["steps"]), 1) for v in dstamps],
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            probs = model.acousticProbs(audio[:8])
            res["cpu_baseline"] = cpu_baseline(weights, audio, probs, 8)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
