#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W [--workload batch|stream|ragged|bytes|peaky]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

metric   audio-seconds per wall-second (RTF x), whole job, audio already resident in HBM when the clock starts
step     one pass of the hot path (MFCC -> dense x3 -> LSTM-2048 -> dense x2 -> softmax -> CTC beam search + KenLM/FST scorer)
         over one batch per GPU
workload batch  (default, the driver's line) configs[1]: 64 synthetic 5 s 16 kHz utterances, English geometry, beam 500, scorer
         stream configs[2]: synthetic utterances of 1-15 s fed in 320 ms hops with an intermediate decode after every hop,
                --streams live streams advanced together (STTX_*Batch); a step = one pass over --utterances utterances
         ragged configs[3]: this rank's LPT shard of a LibriSpeech-shaped job (--utterances per rank, lengths U(1,15) s)
         bytes  configs[4]: byte-output model (256 classes), pruned_lm.bytes.scorer, beam 1024, 64 x 5 s (different audio every step)
         peaky  configs[1]'s decoder stage alone on peaky synthetic emissions (SURVEY.md 8d Config 2: blank ~0.9, labels held two
                frames) of sentences drawn from vocab.pruned.txt, 64 streams x 250 frames: DecoderState::next + decode, state
                slabs allocated before the clock starts
         The default run (batch, one GPU) appends the other four as `workloads` sub-lines, measured in the same process on the
         same build (--no-extras skips them).
weights  seeded random init of the reference architecture (no checkpoint exists offline); scorer = a synthetic
         huge-vocabulary package written at start-up by stt_amd/tools (500 k pseudo-words, order 5, 30 M n-grams, KenLM
         `-a 255 -q 8 trie` layout = the release recipe of doc/LANGUAGE_MODEL.rst:52-62; no corpus or lmplz offline).
         --scorer fixture switches to the reference's small data/smoke_test/pruned_lm.scorer.
scaling  weak: every rank decodes its own utterances; one RCCL gather of the transcripts per step
verified after the clock stops the transcripts of EVERY timed batch are compared with one blocking call on the same audio

One JSON line on rank 0, including `roofline` (dominant kernel + every engine's kernels: algorithmic bytes or flops / HIP-event
time on the engine's own stream) and `cpu_baseline` (the reference's CPU paths on the host cores, rank 0 at N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch initialises HIP: the engine's streams each get a hardware queue (STTX_ConfigureRuntime)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FIX = os.path.join(ROOT, "tests", "golden", "fixtures")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: f16 / bf16 MFMA dense
H, BEAM, BATCH, SECONDS = 2048, 500, 64, 5.0
K1 = 512                   # 19 x 26 = 494 context features, padded to 512
FIXTURE_SCORER = os.path.join(FIX, "pruned_lm.scorer")


def synth_scorer(scorer_dir):
    from stt_amd import scorertools
    lm, vocab = os.path.join(scorer_dir, "lm.binary"), os.path.join(scorer_dir, "vocab.txt")
    path = os.path.join(scorer_dir, "synthetic_500k.scorer")
    t_s = time.perf_counter()
    scorertools.synth_lm(lm, vocab, words=500000, order=5, seed=7, avg={2: 24, 3: 1.2, 4: 0.7, 5: 0.5})
    scorertools.generate_scorer_package(lm, vocab, path, alphabet=os.path.join(FIX, "alphabet.txt"),
                                        default_alpha=0.931289039105002, default_beta=1.1834137581510284)   # doc/LANGUAGE_MODEL.rst:80-81
    desc = ("synthetic huge-vocabulary scorer (500 k words, order 5, quant-array-trie `-a 255 -q 8`, %.0f MB, built in %.0f s)"
            % (os.path.getsize(path) / 1e6, time.perf_counter() - t_s))
    return path, desc


def cpu_baseline(model, weights, audio, scorer_path):
    """SURVEY.md 8d "CPU baseline timed beside it" on the GPU box's host cores, a bounded sample of the timed workload:
    (3) the evaluate_export.py:65-80 pattern -- worker processes over ALL 64 utterances of the batch, each running the whole CPU
        path at batch 1: MFCC + acoustic model as torch-CPU f32 with 4 threads (tflitemodelstate.cc:200; a RESTATEMENT, TensorFlow
        Lite is not in the tree: kind "port") and the REAL reference DecoderState (oracle/_ref: kind "reference"), same scorer, beam;
    (1) the reference's own multi-core decoder entry point, ctc_beam_search_decoder_batch(num_processes = host cores)
        (ctc_beam_search_decoder.cpp:608-652), on the same 64 emission matrices (the GPU's): kind "reference";
    (2) the acoustic restatement alone on a quiet host: ONE worker, 4 threads, one utterance."""
    from oracle import cpu_harness, ref
    cores = os.cpu_count() or 1
    unit = "audio-seconds/sec"
    if not ref.available():
        return {"value": None, "unit": unit, "cores": cores, "kind": "port", "sample": "oracle/_ref not built: no CPU baseline"}
    alphabet = os.path.join(FIX, "alphabet.txt")
    secs = len(audio) * SECONDS
    workers = max(1, min(len(audio), cores // 4))
    r = cpu_harness.run(weights, audio, scorer_path, alphabet, BEAM, workers, threads=4)
    # (1): the decoder's own multi-core path on the emissions of the timed batch
    probs = np.stack(model.acousticProbs(audio)).astype(np.float64)
    A = ref.Alphabet(alphabet)
    S = ref.Scorer(scorer_path, A)
    t0 = time.perf_counter()
    ref.decode_batch(probs, [probs.shape[1]] * len(audio), A, BEAM, cores, S)
    dec_wall = time.perf_counter() - t0
    # (2): the acoustic restatement with the host to itself
    quiet = cpu_harness.run(weights, audio[:1], None, alphabet, 1, 1, threads=4)
    return {"value": secs / r["wall_s"], "unit": unit, "cores": min(cores, workers * 4), "kind": "port",
            "kind_by_part": {"acoustic": "port (torch-CPU f32 restatement, 4 threads per worker, batch 1; not TFLite, not int8)",
                             "decoder": "reference (oracle/_ref DecoderState, beam %d, same scorer)" % BEAM},
            "acoustic_s_per_utterance": round(r["am_s_per_utt"], 3), "decoder_s_per_utterance": round(r["dec_s_per_utt"], 3),
            "sample": "all %d utterances of the timed batch (%.0f audio-s): %d worker processes x 4 threads on %d host cores (evaluate_export.py:65-80 "
                      "pattern), wall %.2f s after the workers reported ready" % (len(audio), secs, r["workers"], cores, r["wall_s"]),
            "decoder_batch": {"value": secs / dec_wall, "unit": unit, "cores": cores, "kind": "reference", "wall_s": round(dec_wall, 3),
                              "sample": "ctc_beam_search_decoder_batch(num_processes=%d) on the %d emission matrices of the timed batch (decoder stage only)" % (cores, len(audio))},
            "acoustic_quiet": {"s_per_utterance": round(quiet["am_s_per_utt"], 3), "value": SECONDS / max(1e-9, quiet["am_s_per_utt"]), "unit": unit, "cores": 4,
                               "kind": "port", "sample": "one 5 s utterance, one worker, 4 threads, nothing else on the host"}}


class Ctx:
    """What the workloads share: device, process group, the English model + scorer."""


def make_model(C, beam, labels):
    from stt_amd import Model, modelfile, synth
    weights = synth.synth_weights(0, n_hidden=H, n_classes=C)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synth.sttw")
        modelfile.write_model(path, weights, labels, beam_width=beam)
        return Model(path), weights


def measure(wl, args, cx, steps, warmup):
    """One workload, timed as the contract says: W untimed steps, barrier + synchronize, K steps, barrier + synchronize."""
    import torch
    from stt_amd import dist as sdist
    from stt_amd import model as M
    from stt_amd import native, synth
    rank, world, dev, cdev, dist = cx.rank, cx.world, cx.dev, cx.cdev, cx.dist
    C = 256 if wl == "bytes" else 29
    beam = 1024 if wl == "bytes" else BEAM
    if wl == "bytes":
        if cx.bytes_model is None:
            cx.bytes_model, _ = make_model(256, 1024, [bytes([i + 1]) for i in range(255)])   # UTF8Alphabet (alphabet.h:83-91)
            cx.bytes_model.enableExternalScorer(os.path.join(FIX, "pruned_lm.bytes.scorer"))
        model, scorer_desc = cx.bytes_model, "pruned_lm.bytes.scorer (codepoint-level, order 2)"
    else:
        model, scorer_desc = cx.model, cx.scorer_desc
    hop_lat, extra = [], {}
    n = int(SECONDS * 16000)
    if wl in ("batch", "bytes"):
        audio = [synth.synth_audio(n, seed=1000 * rank + i) for i in range(BATCH)]
        sizes, stride = [n] * BATCH, n
        host = np.stack(audio)
        # bytes: the code-point FullScore memo persists across batches -- a different batch every step keeps it honest
        variants = [host] if wl == "batch" else [np.roll(host, 3571 * k + 11, axis=1) for k in range(steps + warmup + 1)]
        d_audios = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in variants]     # int16 [B][stride]
        audio_s_step = BATCH * SECONDS
        desc = ("configs[1]: batch=64 synthetic 5 s 16 kHz utterances per GPU, English geometry (n_hidden 2048, 29 classes), beam_width=500, KenLM scorer = "
                if wl == "batch" else "configs[4]: batch=64 synthetic 5 s utterances per GPU (different audio every step), byte-output model (n_hidden 2048, 256 classes, alphabet-free), beam_width=1024, scorer = ") + scorer_desc
        gbatch = world * BATCH
    elif wl == "ragged":
        nu = args.utterances or 1250
        rng = np.random.RandomState(2 + rank)
        lens = (rng.uniform(1.0, 15.0, size=nu) * 16000).astype(np.int64)
        stride = int(lens.max())
        base = synth.synth_audio(stride, seed=5)
        host = np.zeros((nu, stride), dtype=np.int16)
        for i, ln in enumerate(lens):                      # cheap synthetic variety: rotated copies of one noise/tone mixture
            host[i, :ln] = np.roll(base, 977 * i)[:ln]
        d_audios = [torch.from_numpy(host).to(dev)]
        sizes = [int(x) for x in lens]
        audio_s_step = float(lens.sum()) / 16000.0
        desc = ("configs[3]: LibriSpeech-shaped job, %d utterances per GPU (lengths U(1,15) s, taken longest first in groups), English geometry, "
                "beam_width=500, scorer = %s" % (nu, scorer_desc))
        gbatch = world * nu
    elif wl == "stream":
        nu = args.utterances or 256
        rng = np.random.RandomState(1 + rank)
        base = synth.synth_audio(15 * 16000, seed=3)
        utts = [np.roll(base, 977 * u)[:int(rng.uniform(1, 15) * 16000)].copy() for u in range(nu)]
        audio_s_step = sum(len(a) for a in utts) / 16000.0
        desc = ("configs[2]: %d synthetic utterances (1-15 s) per GPU fed in 320 ms hops (5120 samples) with an intermediate decode after every hop, "
                "%d live streams advanced together, English geometry, beam_width=500, scorer = %s" % (nu, args.streams, scorer_desc))
        gbatch = world * nu
    else:  # peaky
        vocab = open(os.path.join(FIX, "vocab.pruned.txt")).read().split()
        rng = np.random.RandomState(7 + rank)
        T = 250
        em = []
        for i in range(BATCH):
            sent = ""
            while len(sent) < 48:
                sent += (" " if sent else "") + str(rng.choice(vocab))
            lab = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in sent[:56]]
            em.append(synth.peaky_emissions(lab, T, 29, 28, seed=int(rng.randint(1 << 30)), noise=0.02))
        em = np.stack(em).astype(np.float32)
        model.disableExternalScorer(); model.enableExternalScorer(FIXTURE_SCORER)   # the sentences' own vocabulary
        scorer_desc = "pruned_lm.scorer fixture (the sentences' vocabulary)"
        audio_s_step = BATCH * SECONDS
        desc = ("configs[1] decoder stage on peaky synthetic emissions (blank ~0.9, labels held 2 frames, noise 0.02): 64 streams x 250 frames, "
                "beam_width=500, scorer = " + scorer_desc)
        gbatch = world * BATCH
        decoders = [model.createDecoder(BATCH, BEAM) for _ in range(steps + warmup)]    # state slabs: allocated before the clock starts
    step_no = [0]

    def step():
        k = step_no[0]
        step_no[0] += 1
        if wl in ("batch", "bytes", "ragged"):
            texts = model.sttBatchDevice(d_audios[k % len(d_audios)].data_ptr(), stride, sizes)
        elif wl == "stream":
            texts = []
            for u0 in range(0, len(utts), args.streams):
                group = [(a, model.createStream()) for a in utts[u0:u0 + args.streams]]
                live, kk = list(group), 0
                while live:
                    t0 = time.perf_counter()
                    M.feedAudioContentBatch([s for _, s in live], [a[kk:kk + 5120] for a, _ in live])
                    M.intermediateDecodeBatch([s for _, s in live])
                    hop_lat.append(time.perf_counter() - t0)
                    kk += 5120
                    live = [(a, s) for a, s in live if kk < len(a)]
                texts += M.finishStreamBatch([s for _, s in group])
        else:
            d = decoders[k]
            tb = time.perf_counter()
            d.next(em)
            tc = time.perf_counter()
            res = d.decode(1, 256)
            td = time.perf_counter()
            for k_, v_ in (("next_ms", tc - tb), ("decode_ms", td - tc)):
                extra[k_] = extra.get(k_, 0.0) + 1e3 * v_
            texts = ["".join(" " if t == 0 else ("'" if t == 27 else chr(ord("a") + int(t) - 1)) for t in r[0][1]) if r else "" for r in res]
        return sdist.gather_transcripts(texts, device=cdev) if world > 1 else [texts]

    pipelined = wl in ("batch", "bytes") and not args.no_pipeline
    csz = (ctypes.c_uint * len(sizes))(*sizes) if pipelined else None
    if pipelined:
        # the W untimed warm-up steps go through the same pipeline as the timed ones (its first batches allocate the chunk rings and
        # capture the recurrence graphs: 13 ms that would otherwise land in the timed region), drained before the clock starts
        pend = []
        for k in range(warmup):
            if len(pend) == model.pipelineDepth():
                model.collectBatch(pend.pop(0))
            pend.append(model.submitBatchDevice(d_audios[k % len(d_audios)].data_ptr(), stride, csz))
        while pend:
            model.collectBatch(pend.pop(0))
        step_no[0] = warmup
    else:
        for _ in range(warmup):
            step()
    hop_lat.clear()
    extra.clear()
    profiled = wl in ("batch", "bytes", "ragged") and not args.no_profile
    model.setProfiling(profiled)
    stage, step_s, timed_texts = {}, [], []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    depth = model.pipelineDepth() if pipelined else 1
    host_submit_s = 0.0
    if pipelined:
        # K batches through the library's own pipeline (STTX_BatchSubmitDevice / STTX_BatchCollect, STTX_BatchPipelineDepthFor batches in
        # flight): a batch is submitted as soon as there is room, every batch is collected (and gathered) inside the timed region
        inflight = []
        for k in range(steps + 1):
            while inflight and (len(inflight) == depth or k == steps):
                tk, ts, kk = inflight.pop(0)
                texts = model.collectBatch(tk)
                out = sdist.gather_transcripts(texts, device=cdev) if world > 1 else [texts]
                step_s.append(time.perf_counter() - ts)           # submit -> transcripts of that batch
                timed_texts.append((kk, texts))
            if k < steps:
                ts = time.perf_counter()
                kk = (warmup + k) % len(d_audios)
                inflight.append((model.submitBatchDevice(d_audios[kk].data_ptr(), stride, csz), ts, kk))
                host_submit_s += time.perf_counter() - ts
        if profiled:
            stage = dict(model.stageTimes())                      # (summed over the K batches when the pipeline drained)
    else:
        for _ in range(steps):
            ts = time.perf_counter()
            kk = step_no[0] % max(1, len(d_audios)) if wl in ("batch", "bytes", "ragged") else 0
            out = step()
            step_s.append(time.perf_counter() - ts)
            timed_texts.append((kk, out[rank] if world > 1 else out[0]))
            if profiled:
                for k, v in model.stageTimes().items():
                    stage[k] = stage.get(k, 0.0) + v
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.setProfiling(False)
    # ---- after the clock: every timed batch's transcripts against one blocking call on the same audio
    verified, verified_what = None, None
    if wl in ("batch", "bytes", "ragged"):
        want = {}
        ok = True
        for kk, texts in timed_texts:
            if kk not in want:
                want[kk] = model.sttBatchDevice(d_audios[kk].data_ptr(), stride, sizes)
            ok = ok and texts == want[kk]
        verified = bool(ok)
        verified_what = ("transcripts of all %d timed batches == a blocking STTX_SpeechToTextBatchDevice call on the same audio (no decoder error bits); "
                         "%d of %d transcripts non-empty" % (len(timed_texts), sum(1 for w_ in want.values() for s in w_ if s), sum(len(w_) for w_ in want.values())))
    elif wl == "peaky":
        verified = all(t for _, t in timed_texts) and len({tuple(t) for _, t in timed_texts}) == 1
        verified_what = "all timed steps give the same non-empty transcripts"
    dstats, dphase, dstamps = {}, {}, []
    if profiled:
        model.setProfiling(2)            # one extra, untimed step with the search kernel's phase cycle counters on
        step()
        dstats = model.decoderStats()
        dphase = model.decoderPhaseCycles()
        dstamps = model.decoderStamps()
    model.setProfiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if wl == "peaky":
        for d in decoders:
            d.close()
        model.disableExternalScorer(); model.enableExternalScorer(cx.scorer_path)
    if rank != 0:
        return None
    K = steps
    if args.no_profile:
        return {"experiment": "no-profile", "workload": wl, "ms_per_step": 1e3 * elapsed / K, "value": world * audio_s_step * K / elapsed,
                "host_enqueue_ms_per_step": 1e3 * host_submit_s / K, "verified": verified}
    res = {
        "metric": "audio-seconds/sec (RTF)", "value": world * audio_s_step * K / elapsed, "unit": "audio-seconds/sec", "n_gpus": world,
        "steps": K, "warmup": warmup, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 (MFMA operands, f32 accumulate/state; decoder f32+f64)", "data": "synthetic",
        "config": {"workload": desc, "global_batch": gbatch, "parallelism": "dp%d (utterance shards, RCCL transcript gather)" % world,
                   "batches_in_flight": depth,
                   # two 64-utterance batches share one recurrence where the step is acoustic-bound (tunable `pair`; not the search-bound bytes setup)
                   "rows_per_recurrent_step": (128 if (native.get_tuning("pair") and wl != "bytes" and (pipelined or wl == "ragged")) else 64)},
        "verified": verified, "verified_what": verified_what,
        # a batch completes together (submit -> all transcripts on the host): per-utterance latency = that span; median over the timed
        # batches (with several batches in flight it is longer than ms_per_step: the next batches' acoustic models run beside this one's search)
        "p50_utterance_latency_ms": 1e3 * float(np.median(step_s)),
    }
    if pipelined:
        res["host_enqueue_ms_per_step"] = 1e3 * host_submit_s / K     # host time inside STTX_BatchSubmitDevice
    if wl == "stream":
        lat = np.array(hop_lat) * 1e3
        res["p50_utterance_latency_ms"] = None
        res["hop_latency_ms"] = {"p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)), "max": float(lat.max()),
                                 "what": "feed 320 ms + intermediate decode of ALL live streams (STTX_*Batch), host wall clock", "hops": int(len(lat))}
        # per hop: 16 recurrent steps re-stream the 33.5 MB f16 recurrent matrix (shared by the live streams) + the dense weights once
        hop_bytes = 16 * H * 4 * H * 2 + 60.9e6
        ach = hop_bytes / (np.percentile(lat, 50) * 1e-3) / 1e9
        res["roofline"] = {"kernel": "one 320 ms hop of all live streams (16 x lstm_step_kernel + dense + ctc_next_kernel + ctc_decode_kernel)", "bound": "hbm",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "note": "host-timed whole hop, not a single kernel: launch-bound (about 45 kernels per hop)"}
    elif wl == "peaky":
        ms = extra.get("next_ms", 0.0) / K      # DecoderState::next alone: H2D of 1.9 MB of emissions + the search launch, host-timed
        res["stage_ms_per_step"] = {k_: v_ / K for k_, v_ in extra.items()}
        by = BATCH * 250 * (29 * 4 + 2 * BEAM * 40)
        res["roofline"] = {"kernel": "ctc_next_kernel (+ H2D of 1.9 MB emissions)", "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "us_per_stream_timestep": 1e3 * ms / 250.0,
                           "note": "host-timed STTX_DecoderNext of 64 streams x 250 frames (state slabs allocated before the clock)"}
    else:
        T = 250
        rows = res["config"]["rows_per_recurrent_step"]
        lstm_launches = max(1.0, stage["lstm_launches"])
        lstm_avg_ms = stage["lstm_ms"] / lstm_launches
        # recurrent matrix H x 4H f16 once per launch + per row: x-projection (f32 4H) in, h (f16 H) in/out, c (f32 H) in/out
        lstm_bytes = H * 4 * H * 2 + rows * (4 * H * 4 + 2 * H * 2 + 2 * H * 4)
        lstm_name = "lstm_step8_kernel<1>" if rows == 128 else "lstm_step_kernel<4, 2, 4, 3>"      # as rocprofv3 names them
        dec_ms = stage["decoder_next_ms"] / K
        steps_total = max(1, dstats["steps"])
        tsteps = stage["timesteps"]                  # utterance-timesteps through the acoustic model in the timed region
        # per utterance-timestep C*4 B of probabilities in, beam state ~ beam*40 B read + written, 8 B per counted LM probe
        dec_bytes = steps_total * (C * 4 + 2 * beam * 40) + 8.0 * dstats["lm_probes"]
        # dense layers (MFMA roofline): flops per utterance-timestep, SURVEY.md 8(d)
        fl_in = 2.0 * (K1 * H + 2 * H * H + H * 4 * H) * tsteps        # layers 1-3 + x-projection
        fl_out = 2.0 * (H * H + H * C) * tsteps                       # layer 5 + output layer
        feat_bytes = 744.0 * tsteps                                    # 640 B of int16 samples in, 104 B of MFCC out per timestep
        kernels = {
            lstm_name: {"avg_ms": lstm_avg_ms, "bytes": lstm_bytes, "share_ms": stage["lstm_ms"] / K},
            "ctc_next_kernel": {"avg_ms": dec_ms, "bytes": dec_bytes, "share_ms": dec_ms},
        }
        dom = max(kernels, key=lambda k: kernels[k]["share_ms"])
        ach = kernels[dom]["bytes"] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        # HBM traffic from the committed rocprofv3 --pmc passes of this round's build (FETCH_SIZE and WRITE_SIZE cannot share a pass, and
        # counters are never collected inside a timed run).  Only valid for the batch workload's shapes.
        pmc, pmc_file = {}, None
        if wl == "batch":
            for prof in ("r03_l_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))["kernels"]
                    pmc_file = prof
                    break
                except Exception:
                    pass

        def pmc_entry(prefix):
            ks = [k for k in pmc if k.startswith(prefix)]
            return pmc[ks[0]] if ks else None

        # per LAUNCH, like `achieved`: a search launch covers one time-chunk of the group's streams (16 + 48 ... frames x 64 or 128 streams)
        launches = {lstm_name: lstm_launches / K, "ctc_next_kernel": stage["timesteps"] / K / (rows * 250.0) * 6.0 if wl == "batch" else None}
        traffic, traffic_note = None, None
        e = pmc_entry(dom.split("<")[0])
        if e:
            wide = dom.startswith("lstm")   # 16 B/lane coalesced streams: FETCH_SIZE reads 1/2 on gfx950 (MI355X_MICROARCH.md, HBM)
            traffic = (e["fetch_kb_per_launch"] * (2.0 if wide else 1.0) + e["write_kb_per_launch"]) * 1024.0
            traffic_note = "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (separate passes, profiles/%s), bytes per launch%s" % (pmc_file, " (FETCH_SIZE x2: 16 B/lane streams)" if wide else "")
        # The search kernel is bound by instruction issue and dependent-latency chains inside one CU per stream, not by bytes
        # (DESIGN.md 8.2): shader cycles per stream-timestep is the figure that tracks its speed.
        cyc = sum(v for n_, v in dphase.items() if not n_.startswith("lm_wave")) / steps_total if dphase else None
        allk = {k: {"bound": "hbm", "GB/s": v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9, "frac": v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "avg_ms": v["avg_ms"], "ms_per_step": v["share_ms"]} for k, v in kernels.items()}
        din, dout = stage["dense_in_ms"], stage["dense_out_ms"]
        allk["dense_kernel (layers 1-3 + LSTM x-projection; GEMM engine stream)"] = {
            "bound": "mfma", "TFLOP/s": fl_in / (din * 1e-3) / 1e12, "frac": fl_in / (din * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "ms_per_step": din / K,
            "note": "flops of the timed region / busy time of the GEMM engine's stream (HIP events); the kernels run one workgroup per CU beside the recurrent step"}
        allk["dense_kernel + logits_softmax_kernel (layer 5, output layer; output engine stream)"] = {
            "bound": "mfma", "TFLOP/s": fl_out / (dout * 1e-3) / 1e12, "frac": fl_out / (dout * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "ms_per_step": dout / K}
        if stage.get("features_ms"):
            fg = feat_bytes / (stage["features_ms"] * 1e-3) / 1e9
            allk["mfcc_kernel (+ tables, decoder init on the same stream)"] = {"bound": "hbm", "GB/s": fg, "frac": fg / HBM_PEAK_GBS, "ms_per_step": stage["features_ms"] / K}
        # the co-tenant GEMM form of this build (128 x 256 eight-wave tile); algorithmic bytes of a 48-frame chunk of `rows` streams
        Mc = 48 * rows
        for name, pref, alg in (("dense_wide_kernel<1> (x-projection)", "dense_wide_kernel<1>", Mc * H * 2 + 4 * H * H * 2 + Mc * 4 * H * 4),
                                ("dense_wide_kernel<0> (layers 2, 3, 5)", "dense_wide_kernel<0>", Mc * H * 2 + H * H * 2 + Mc * H * 2)):
            e = pmc_entry(pref)
            if e:
                allk.setdefault("pmc", {})[name] = {"fetch_MB_per_launch_raw": e["fetch_kb_per_launch"] / 1024.0, "write_MB_per_launch": e["write_kb_per_launch"] / 1024.0,
                                                    "algorithmic_MB_per_launch": alg / 1e6, "note": "FETCH_SIZE raw (x2 for 16 B/lane streams on gfx950)", "source": "profiles/" + pmc_file}
        n_l = launches.get(dom)
        roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "launches_per_step": n_l, "algorithmic_bytes_per_launch": (kernels[dom]["bytes"] / n_l if (n_l and dom != lstm_name) else kernels[dom]["bytes"]),
                    "avg_launch_ms": (kernels[dom]["avg_ms"] / n_l if (n_l and dom != lstm_name) else kernels[dom]["avg_ms"]),
                    "traffic": traffic, "traffic_note": traffic_note,
                    "search_cycles_per_stream_timestep": cyc, "search_us_per_stream_timestep": 1e3 * dec_ms / T if wl != "ragged" else None,
                    "all": allk}
        if pipelined:
            # With batches in flight the searches of neighbouring groups overlap, the recurrences cannot: the stream that is busy for
            # most of a step is the recurrence's.  Its roofline is the one that bounds the step.
            lg = kernels[lstm_name]["bytes"] / (kernels[lstm_name]["avg_ms"] * 1e-3) / 1e9
            roofline["critical_path"] = {"kernel": lstm_name, "stream_busy_frac_of_step": kernels[lstm_name]["share_ms"] / (1e3 * elapsed / K), "bound": "hbm",
                                         "achieved": lg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lg / HBM_PEAK_GBS, "us_per_launch": 1e3 * lstm_avg_ms,
                                         "rows_per_launch": rows,
                                         "note": "33.5 MB of recurrent weights + the rows' state per launch / HIP-event time per launch, measured beside the GEMM and search kernels of the other engines"}
        res.update({
            "stage_ms_per_step": {k: v / K for k, v in stage.items() if k.endswith("_ms")},
            "decoder_counters_last_step": dstats,
            # (the LM wave runs beside the expand phases: not part of the serial sum)
            "decoder_phase_cycles_per_stream_step": {k: round(v / steps_total, 1) for k, v in dphase.items()},
            # profiling level 2 (the extra untimed step): [0..15] arrival of each wave at the end of the expand phase (cycles since the step began),
            # [16..31] its wait there, [32..47] how often that wave was the last, [48] / [49] last / second-last arrival minus first
            "decoder_stamp_cycles_per_stream_step": [round(v / steps_total, 1) for v in dstamps[:50]] if any(dstamps) else None,
            "roofline": roofline,
        })
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="batch", choices=["batch", "stream", "ragged", "bytes", "peaky"])
    ap.add_argument("--utterances", type=int, default=0, help="stream: utterances per step (default 256); ragged: per rank (default 1250)")
    ap.add_argument("--streams", type=int, default=128, help="stream: live streams advanced together (one recurrent launch covers 128 rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="batch: do not append the other workloads' sub-lines")
    ap.add_argument("--scorer", default="synthetic", choices=["synthetic", "fixture"])
    ap.add_argument("--no-profile", action="store_true", help="experiment: no HIP-event stage timing inside the timed region")
    ap.add_argument("--no-pipeline", action="store_true", help="batch / bytes: one blocking call per step instead of several batches in flight")
    args = ap.parse_args()

    import torch
    cx = Ctx()
    cx.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the engine has no CPU path")
    # STT_BENCH_BACKEND=gloo: plumbing check of the N>1 path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives on host tensors); the measured configuration is always nccl (= RCCL), one rank per GPU
    backend = os.environ.get("STT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    cx.dev = dev = torch.device("cuda", local_rank)
    cx.cdev = dev if backend == "nccl" else None      # where the collectives' tensors live
    cx.dist = None
    if world > 1:
        import torch.distributed as dist
        cx.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from stt_amd import dist as _sd
        if args.workload in ("batch", "bytes"):
            _sd.assume_equal_batches()      # weak scaling: every rank decodes BATCH utterances -> the gather is one collective

    from stt_amd import native, synth
    native.lib().STTX_SetDevice(local_rank)
    wl = args.workload
    cx.bytes_model = None
    cx.model, cx.scorer_path, cx.scorer_desc, weights = None, None, None, None
    scorer_dir = None
    if wl != "bytes" or not args.no_extras:
        cx.model, weights = make_model(29, BEAM, synth.ENGLISH_LABELS)
        if args.scorer == "synthetic":
            scorer_dir = tempfile.TemporaryDirectory()
            cx.scorer_path, cx.scorer_desc = synth_scorer(scorer_dir.name)
        else:
            cx.scorer_path, cx.scorer_desc = FIXTURE_SCORER, "pruned_lm.scorer fixture (quant-array-trie order 4)"
        cx.model.enableExternalScorer(cx.scorer_path)

    res = measure(wl, args, cx, args.steps, args.warmup)
    if rank == 0 and wl == "batch" and world == 1 and not args.no_extras and not args.no_profile:
        # the other configs, same process, same build: short runs (a few seconds each), each with its own roofline
        sub = {}
        for w, k, wu, kw in (("ragged", 2, 1, {}), ("stream", 1, 1, {"utterances": 256}), ("bytes", 8, 5, {}), ("peaky", 10, 2, {})):   # (bytes: four batches in flight -- the warm-up covers every slot's first use)
            a2 = argparse.Namespace(**vars(args))
            a2.utterances = kw.get("utterances", 0)
            try:
                r = measure(w, a2, cx, k, wu)
                sub[w] = {key: r[key] for key in ("value", "unit", "ms_per_step", "steps", "warmup", "verified", "p50_utterance_latency_ms", "hop_latency_ms",
                                                  "stage_ms_per_step", "roofline", "config") if key in r}
                if "roofline" in sub[w] and "all" in sub[w]["roofline"]:
                    sub[w]["roofline"] = {kk: vv for kk, vv in sub[w]["roofline"].items() if kk != "all"}
            except Exception as ex:      # a failing side workload must not take the driver's line with it; it is reported, not hidden
                sub[w] = {"error": repr(ex)}
        res["workloads"] = sub
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and wl == "batch" and not args.no_profile:
            audio = [synth.synth_audio(int(SECONDS * 16000), seed=1000 * rank + i) for i in range(BATCH)]
            res["cpu_baseline"] = cpu_baseline(cx.model, weights, audio, cx.scorer_path)
        print(json.dumps(res))
    if cx.dist is not None:
        cx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
