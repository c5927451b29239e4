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
         bytes  configs[4]: byte-output model (256 classes), pruned_lm.bytes.scorer, beam 1024, 64 x 5 s
         peaky  configs[1]'s decoder stage alone on peaky synthetic emissions (SURVEY.md 8d Config 2: blank ~0.9, labels held two
                frames) of sentences drawn from vocab.pruned.txt, 64 streams x 250 frames: the beam search on speech-like input
weights  seeded random init of the reference architecture (no checkpoint exists offline); scorer = a synthetic
         huge-vocabulary package written at start-up by stt_amd/tools (500 k pseudo-words, order 5, 30 M n-grams, KenLM
         `-a 255 -q 8 trie` layout = the release recipe of doc/LANGUAGE_MODEL.rst:52-62; no corpus or lmplz offline).
         --scorer fixture switches to the reference's small data/smoke_test/pruned_lm.scorer.
scaling  weak: every rank decodes its own utterances; one RCCL gather of the transcripts per step

One JSON line on rank 0, including `roofline` (dominant kernel, algorithmic bytes / measured HIP-event time on the
engine's own stream) and `cpu_baseline` (the reference's CPU evaluation pattern on the host cores, rank 0 at N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch initialises HIP: the engine's streams each get a hardware queue (api.cpp)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FIX = os.path.join(ROOT, "tests", "golden", "fixtures")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
H, BEAM, BATCH, SECONDS = 2048, 500, 64, 5.0
SCORER_PATH = os.path.join(FIX, "pruned_lm.scorer")
SCORER_DESC = "pruned_lm.scorer fixture (quant-array-trie order 4)"


def cpu_baseline(model_weights, audio):
    """SURVEY.md 8d "CPU baseline timed beside it", item (3): the evaluate_export.py:65-80 pattern -- worker processes over the
    utterances, each running the whole CPU path at batch 1: MFCC + acoustic model as torch-CPU f32 with 4 threads
    (tflitemodelstate.cc:200; a restatement, TensorFlow Lite is not in the tree) and the REAL reference beam search
    (oracle/_ref) with the same scorer and beam -- plus item (1), the reference's own multi-core decoder entry point
    ctc_beam_search_decoder_batch on the same emissions.  All 64 utterances of the timed batch."""
    from oracle import cpu_harness, ref
    cores = os.cpu_count() or 1
    if not ref.available():
        return {"value": None, "unit": "audio-seconds/sec", "cores": cores, "kind": "port", "sample": "oracle/_ref not built: no CPU baseline"}
    workers = max(1, min(len(audio), cores // 4))
    r = cpu_harness.run(model_weights, audio, SCORER_PATH, os.path.join(FIX, "alphabet.txt"), BEAM, workers, threads=4)
    secs = len(audio) * SECONDS
    return {"value": secs / r["wall_s"], "unit": "audio-seconds/sec", "cores": min(cores, workers * 4), "kind": "port",
            "parts": {"acoustic": "restatement (torch-CPU f32, 4 threads per worker, batch 1), not TFLite", "decoder": "reference (oracle/_ref DecoderState, beam %d, same scorer)" % BEAM},
            "acoustic_s_per_utterance": round(r["am_s_per_utt"], 3), "decoder_s_per_utterance": round(r["dec_s_per_utt"], 3),
            "sample": "all %d utterances of the timed batch (%.0f audio-s): %d worker processes x 4 threads on %d host cores (evaluate_export.py:65-80 "
                      "pattern), wall %.2f s after the workers reported ready" % (len(audio), secs, r["workers"], cores, r["wall_s"])}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="batch", choices=["batch", "stream", "ragged", "bytes", "peaky"])
    ap.add_argument("--utterances", type=int, default=0, help="stream: utterances per step (default 256); ragged: per rank (default 1250)")
    ap.add_argument("--streams", type=int, default=64, help="stream: live streams advanced together")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scorer", default="synthetic", choices=["synthetic", "fixture"])
    ap.add_argument("--no-profile", action="store_true", help="experiment: no HIP-event stage timing inside the timed region")
    ap.add_argument("--no-pipeline", action="store_true", help="batch / bytes: one blocking call per step instead of several batches in flight")
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
        if args.workload in ("batch", "bytes"):
            _sd.assume_equal_batches()      # weak scaling: every rank decodes BATCH utterances -> the gather is one collective

    from stt_amd import Model, modelfile, native, synth
    from stt_amd import dist as sdist
    from stt_amd import model as M
    native.lib().STTX_SetDevice(local_rank)

    wl = args.workload
    C = 256 if wl == "bytes" else 29
    beam = 1024 if wl == "bytes" else BEAM
    labels = [bytes([i + 1]) for i in range(255)] if wl == "bytes" else synth.ENGLISH_LABELS   # UTF8Alphabet (alphabet.h:83-91)
    weights = synth.synth_weights(0, n_hidden=H, n_classes=C)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synth.sttw")
        modelfile.write_model(path, weights, labels, beam_width=beam)
        model = Model(path)
    global SCORER_PATH, SCORER_DESC
    scorer_dir = None
    if wl == "bytes":
        SCORER_PATH, SCORER_DESC = os.path.join(FIX, "pruned_lm.bytes.scorer"), "pruned_lm.bytes.scorer (codepoint-level, order 2)"
    elif args.scorer == "synthetic":
        scorer_dir = tempfile.TemporaryDirectory()
        SCORER_PATH, SCORER_DESC = synth_scorer(scorer_dir.name)
    model.enableExternalScorer(SCORER_PATH)

    # ---- workload: audio resident in HBM before the clock starts (peaky: emissions on the host, the decoder entry takes host buffers)
    hop_lat = []
    extra = {}
    if wl in ("batch", "bytes"):
        n = int(SECONDS * 16000)
        audio = [synth.synth_audio(n, seed=1000 * rank + i) for i in range(BATCH)]
        sizes, stride = [n] * BATCH, n
        d_audio = torch.from_numpy(np.stack(audio)).to(dev)     # int16 [B][stride]
        audio_s_step = BATCH * SECONDS
        desc = ("configs[1]: batch=64 synthetic 5 s 16 kHz utterances per GPU, English geometry (n_hidden 2048, 29 classes), beam_width=500, KenLM scorer = "
                if wl == "batch" else "configs[4]: batch=64 synthetic 5 s utterances per GPU, byte-output model (n_hidden 2048, 256 classes, alphabet-free), beam_width=1024, scorer = ") + SCORER_DESC
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
        d_audio = torch.from_numpy(host).to(dev)
        sizes = [int(x) for x in lens]
        audio_s_step = float(lens.sum()) / 16000.0
        desc = ("configs[3]: LibriSpeech-shaped job, %d utterances per GPU (lengths U(1,15) s, taken longest first in groups of 64), English geometry, "
                "beam_width=500, scorer = %s" % (nu, SCORER_DESC))
        gbatch = world * nu
    elif wl == "stream":
        nu = args.utterances or 256
        rng = np.random.RandomState(1 + rank)
        base = synth.synth_audio(15 * 16000, seed=3)
        utts = [np.roll(base, 977 * u)[:int(rng.uniform(1, 15) * 16000)].copy() for u in range(nu)]
        audio_s_step = sum(len(a) for a in utts) / 16000.0
        desc = ("configs[2]: %d synthetic utterances (1-15 s) per GPU fed in 320 ms hops (5120 samples) with an intermediate decode after every hop, "
                "%d live streams advanced together, English geometry, beam_width=500, scorer = %s" % (nu, args.streams, SCORER_DESC))
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
        model.disableExternalScorer(); model.enableExternalScorer(os.path.join(FIX, "pruned_lm.scorer"))   # the sentences' own vocabulary
        SCORER_DESC = "pruned_lm.scorer fixture (the sentences' vocabulary)"
        audio_s_step = BATCH * SECONDS
        desc = ("configs[1] decoder stage on peaky synthetic emissions (blank ~0.9, labels held 2 frames, noise 0.02): 64 streams x 250 frames, "
                "beam_width=500, scorer = " + SCORER_DESC)
        gbatch = world * BATCH

    def step():
        if wl in ("batch", "bytes", "ragged"):
            texts = model.sttBatchDevice(d_audio.data_ptr(), stride, sizes)
        elif wl == "stream":
            texts = []
            for u0 in range(0, len(utts), args.streams):
                group = [(a, model.createStream()) for a in utts[u0:u0 + args.streams]]
                live, k = list(group), 0
                while live:
                    t0 = time.perf_counter()
                    M.feedAudioContentBatch([s for _, s in live], [a[k:k + 5120] for a, _ in live])
                    M.intermediateDecodeBatch([s for _, s in live])
                    hop_lat.append(time.perf_counter() - t0)
                    k += 5120
                    live = [(a, s) for a, s in live if k < len(a)]
                texts += M.finishStreamBatch([s for _, s in group])
        else:
            ta = time.perf_counter()
            d = model.createDecoder(BATCH, BEAM)
            tb = time.perf_counter()
            d.next(em)
            tc = time.perf_counter()
            res = d.decode(1, 256)
            td = time.perf_counter()
            d.close()
            te = time.perf_counter()
            for k_, v_ in (("create_ms", tb - ta), ("next_ms", tc - tb), ("decode_ms", td - tc), ("free_ms", te - td)):
                extra[k_] = extra.get(k_, 0.0) + 1e3 * v_
            texts = ["".join(" " if t == 0 else ("'" if t == 27 else chr(ord("a") + int(t) - 1)) for t in r[0][1]) if r else "" for r in res]
        return sdist.gather_transcripts(texts, device=cdev) if world > 1 else [texts]

    pipelined = wl in ("batch", "bytes") and not args.no_pipeline
    if pipelined:
        # the W untimed warm-up steps go through the same pipeline as the timed ones (its first batches allocate the chunk rings and
        # capture the recurrence graphs: 13 ms that would otherwise land in the timed region), drained before the clock starts
        csz = (ctypes.c_uint * len(sizes))(*sizes)
        pend = []
        for k in range(args.warmup):
            if len(pend) == model.pipelineDepth():
                model.collectBatch(pend.pop(0))
            pend.append(model.submitBatchDevice(d_audio.data_ptr(), stride, csz))
        while pend:
            model.collectBatch(pend.pop(0))
    else:
        for _ in range(args.warmup):
            step()
    hop_lat.clear()
    extra.clear()
    profiled = wl in ("batch", "bytes", "ragged") and not args.no_profile
    model.setProfiling(profiled)
    stage, step_s = {}, []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    depth = model.pipelineDepth() if pipelined else 1
    host_submit_s = 0.0
    if pipelined:
        # K batches through the library's own pipeline (STTX_BatchSubmitDevice / STTX_BatchCollect, STTX_BatchPipelineDepth batches in
        # flight): a batch is submitted as soon as there is room, every batch is collected (and gathered) inside the timed region
        inflight = []
        for k in range(args.steps + 1):
            while inflight and (len(inflight) == depth or k == args.steps):
                tk, ts = inflight.pop(0)
                texts = model.collectBatch(tk)
                out = sdist.gather_transcripts(texts, device=cdev) if world > 1 else [texts]
                step_s.append(time.perf_counter() - ts)           # submit -> transcripts of that batch
            if k < args.steps:
                ts = time.perf_counter()
                inflight.append((model.submitBatchDevice(d_audio.data_ptr(), stride, csz), ts))
                host_submit_s += time.perf_counter() - ts
        if profiled:
            stage = dict(model.stageTimes())                      # (summed over the K batches when the pipeline drained)
    else:
        for _ in range(args.steps):
            ts = time.perf_counter()
            out = step()
            step_s.append(time.perf_counter() - ts)
            if profiled:
                for k, v in model.stageTimes().items():
                    stage[k] = stage.get(k, 0.0) + v
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
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

    if rank == 0 and args.no_profile:
        print(json.dumps({"experiment": "no-profile", "workload": wl, "ms_per_step": 1e3 * elapsed / args.steps, "value": world * audio_s_step * args.steps / elapsed,
                          "host_enqueue_ms_per_step": 1e3 * host_submit_s / args.steps}))
    elif rank == 0:
        K = args.steps
        res = {
            "metric": "audio-seconds/sec (RTF)", "value": world * audio_s_step * K / elapsed, "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (MFMA operands, f32 accumulate/state; decoder f32+f64)", "data": "synthetic",
            "config": {"workload": desc, "global_batch": gbatch, "parallelism": "dp%d (utterance shards, RCCL transcript gather)" % world,
                       "batches_in_flight": depth},
            # a batch completes together (submit -> all transcripts on the host): per-utterance latency = that span; median over the timed
            # batches (with several batches in flight it is longer than ms_per_step: the next batches' acoustic models run beside this one's search)
            "p50_utterance_latency_ms": 1e3 * float(np.median(step_s)),
        }
        if pipelined:
            res["host_enqueue_ms_per_step"] = 1e3 * host_submit_s / K     # host time inside STTX_BatchSubmitDevice (about 300 launches per batch)
        if wl == "stream":
            lat = np.array(hop_lat) * 1e3
            res["p50_utterance_latency_ms"] = None
            res["hop_latency_ms"] = {"p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)), "max": float(lat.max()),
                                     "what": "feed 320 ms + intermediate decode of ALL live streams (STTX_*Batch), host wall clock", "hops": int(len(lat))}
            # per hop and stream: 16 recurrent steps re-stream the 33.5 MB f16 recurrent matrix (shared by the live streams) + the dense weights once
            hop_bytes = 16 * H * 4 * H * 2 + 60.9e6
            ach = hop_bytes / (np.percentile(lat, 50) * 1e-3) / 1e9
            res["roofline"] = {"kernel": "one 320 ms hop of all live streams (16 x lstm_step_kernel + dense + ctc_next_kernel + ctc_decode_kernel)", "bound": "hbm",
                               "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "note": "host-timed whole hop, not a single kernel: launch-bound (about 45 kernels per hop)"}
        elif wl == "peaky":
            ms = extra.get("next_ms", 0.0) / K      # DecoderState::next alone: H2D of 1.9 MB of emissions + the search launch, host-timed
            res["stage_ms_per_step"] = {k_: v_ / K for k_, v_ in extra.items()}
            res["roofline"] = {"kernel": "ctc_next_kernel (+ H2D of 1.9 MB emissions)", "bound": "hbm",
                               "achieved": BATCH * 250 * (29 * 4 + 2 * BEAM * 40) / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": BATCH * 250 * (29 * 4 + 2 * BEAM * 40) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "us_per_stream_timestep": 1e3 * ms / 250.0,
                               "note": "host-timed STTX_DecoderNext of 64 streams x 250 frames; ms_per_step also holds create (slab hipMalloc), decode and free"}
        else:
            T = 250
            lstm_launches = stage["lstm_launches"]
            lstm_avg_ms = stage["lstm_ms"] / max(1.0, lstm_launches)
            # recurrent matrix H x 4H f16 once per batch-timestep + per-row x-projection (f32 4H) in and h (f16 H) in/out, c (f32 H) in/out
            lstm_bytes = H * 4 * H * 2 + BATCH * (4 * H * 4 + 2 * H * 2 + 2 * H * 4)
            dec_ms = stage["decoder_next_ms"] / K
            steps_total = max(1, dstats["steps"])
            # per utterance-timestep C*4 B of probabilities in, beam state ~ beam*40 B read + written, 8 B per counted LM probe
            dec_bytes = steps_total * (C * 4 + 2 * beam * 40) + 8.0 * dstats["lm_probes"]
            kernels = {
                "lstm_step_kernel<4, 2, 4>": {"avg_ms": lstm_avg_ms, "bytes": lstm_bytes, "share_ms": stage["lstm_ms"] / K},
                "ctc_next_kernel": {"avg_ms": dec_ms, "bytes": dec_bytes, "share_ms": dec_ms},
            }
            dom = max(kernels, key=lambda k: kernels[k]["share_ms"])
            ach = kernels[dom]["bytes"] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
            # HBM traffic of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass, and
            # counters are never collected inside a timed run); bytes per batch, like `achieved`.  Only valid for the batch workload's shapes.
            traffic, traffic_note = None, None
            if wl == "batch":
                for prof in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                    try:
                        pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))["kernels"]
                        key = [k for k in pmc if k.startswith(dom.split("<")[0])][0]
                        e = pmc[key]
                        wide = dom.startswith("lstm")   # 16 B/lane coalesced streams: FETCH_SIZE reads 1/2 on gfx950 (MI355X_MICROARCH.md, HBM)
                        traffic = (e["fetch_kb_per_launch"] * (2.0 if wide else 1.0) + e["write_kb_per_launch"]) * 1024.0 * (e["launches_per_batch"] if not wide else 1.0)
                        traffic_note = "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (separate passes, profiles/%s), bytes per %s" % (prof, "launch" if wide else "batch (%d chunk launches)" % round(e["launches_per_batch"]))
                        break
                    except Exception:
                        pass
            # The search kernel is bound by instruction issue and dependent-latency chains inside one CU per stream, not by bytes
            # (DESIGN.md 8.2): shader cycles per stream-timestep is the figure that tracks its speed.
            cyc = sum(v for n_, v in dphase.items() if not n_.startswith("lm_wave")) / steps_total if dphase else None
            roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_note": traffic_note,
                        "search_cycles_per_stream_timestep": cyc, "search_us_per_stream_timestep": 1e3 * dec_ms / T if wl != "ragged" else None,
                        "all": {k: {"GB/s": v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9, "avg_ms": v["avg_ms"], "ms_per_step": v["share_ms"]}
                                for k, v in kernels.items()}}
            if pipelined:
                # With batches in flight the searches of neighbouring batches overlap (64 CUs each), the recurrences cannot: the stream that is
                # busy for most of a step is the recurrence's.  Its roofline is the one that bounds the step.
                lk = "lstm_step_kernel<4, 2, 4>"
                lg = kernels[lk]["bytes"] / (kernels[lk]["avg_ms"] * 1e-3) / 1e9
                roofline["critical_path"] = {"kernel": lk, "stream_busy_frac_of_step": kernels[lk]["share_ms"] / (1e3 * elapsed / K), "bound": "hbm",
                                             "achieved": lg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lg / HBM_PEAK_GBS,
                                             "note": "33.5 MB of recurrent weights + state per launch / HIP-event time per launch, measured beside the GEMM and search kernels of the other engines (alone: 11.5 us per launch, 2.9 TB/s)"}
            res.update({
                "stage_ms_per_step": {k: v / K for k, v in stage.items() if k.endswith("_ms")},
                "decoder_counters_last_step": dstats,
                # (the LM wave runs beside the expand phases: not part of the serial sum)
                "decoder_phase_cycles_per_stream_step": {k: round(v / steps_total, 1) for k, v in dphase.items()},
                "decoder_stamp_cycles_per_stream_step": [round(v / steps_total, 1) for v in dstamps] if any(dstamps) else None,
                "roofline": roofline,
            })
        if world == 1 and not args.no_cpu_baseline and wl == "batch":
            res["cpu_baseline"] = cpu_baseline(weights, audio)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
