"""stt_amd/model.py -- Python host-side mirror of the reference's `stt` package API.

Same class and method names, argument meaning and error behaviour as native_client/python/__init__.py:26-430
(Model, Stream, Metadata wrappers), implemented over the C-ABI of include/coqui-stt.h; the `*Batch`,
stage-level and decoder helpers wrap include/stt_amd.h.  No compute happens in Python.
"""
import ctypes as C

import numpy as np

from . import native


def _text(s):
    """char* result -> str; NULL (the call failed, coqui-stt.h: "NULL on error") -> None"""
    return None if s is None else s.decode("utf-8", "replace")


def _audio(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a, a.ctypes.data, a.shape[0]


def _metadata_to_py(mp, free=True):
    """Metadata* -> list of dicts (confidence, tokens[(text, timestep, start_time)]) (+ emissions)."""
    if not mp:
        return None
    m = mp.contents
    out = []
    for i in range(m.num_transcripts):
        tr = m.transcripts[i]
        toks = [(tr.tokens[j].text.decode("utf-8", "replace"), tr.tokens[j].timestep, tr.tokens[j].start_time)
                for j in range(tr.num_tokens)]
        out.append({"confidence": tr.confidence, "tokens": toks, "text": "".join(t[0] for t in toks)})
    res = {"transcripts": out}
    if m.emissions:
        e = m.emissions.contents
        n = e.num_timesteps * (e.num_symbols + 1)
        res["emissions"] = np.ctypeslib.as_array(e.emissions, shape=(n,)).reshape(e.num_timesteps, e.num_symbols + 1).copy()
        res["symbols"] = [e.symbols[i].decode("utf-8", "replace") for i in range(e.num_symbols + 1)]
    if free:
        native.lib().STT_FreeMetadata(mp)
    return res


class Model(object):
    """native_client/python/__init__.py:26-221"""

    def __init__(self, model_path=None, model_bytes=None):
        L = native.lib()
        self._impl = None
        h = C.c_void_p()
        if model_bytes is not None:
            self._buf = C.create_string_buffer(model_bytes, len(model_bytes))  # caller keeps the buffer alive (client.cc:491)
            status = L.STT_CreateModelFromBuffer(C.cast(self._buf, C.c_void_p), len(model_bytes), C.byref(h))
        else:
            status = L.STT_CreateModel(str(model_path).encode(), C.byref(h))
        if status != 0:
            raise RuntimeError("CreateModel failed with '{}' (0x{:X})".format(native.error_message(status), status))
        self._impl = h

    def __del__(self):
        if getattr(self, "_impl", None):
            native.lib().STT_FreeModel(self._impl)
            self._impl = None

    def beamWidth(self):
        return native.lib().STT_GetModelBeamWidth(self._impl)

    def setBeamWidth(self, beam_width):
        return native.lib().STT_SetModelBeamWidth(self._impl, beam_width)

    def sampleRate(self):
        return native.lib().STT_GetModelSampleRate(self._impl)

    def enableExternalScorer(self, scorer_path=None, scorer_bytes=None):
        if scorer_bytes is not None:
            buf = C.create_string_buffer(scorer_bytes, len(scorer_bytes))
            status = native.lib().STT_EnableExternalScorerFromBuffer(self._impl, C.cast(buf, C.c_void_p), len(scorer_bytes))
        else:
            status = native.lib().STT_EnableExternalScorer(self._impl, str(scorer_path).encode())
        if status != 0:
            raise RuntimeError("EnableExternalScorer failed with '{}' (0x{:X})".format(native.error_message(status), status))

    def disableExternalScorer(self):
        return native.lib().STT_DisableExternalScorer(self._impl)

    def addHotWord(self, word, boost):
        status = native.lib().STT_AddHotWord(self._impl, word.encode(), boost)
        if status != 0:
            raise RuntimeError("AddHotWord failed with '{}' (0x{:X})".format(native.error_message(status), status))

    def eraseHotWord(self, word):
        status = native.lib().STT_EraseHotWord(self._impl, word.encode())
        if status != 0:
            raise RuntimeError("EraseHotWord failed with '{}' (0x{:X})".format(native.error_message(status), status))

    def clearHotWords(self):
        status = native.lib().STT_ClearHotWords(self._impl)
        if status != 0:
            raise RuntimeError("ClearHotWords failed with '{}' (0x{:X})".format(native.error_message(status), status))

    def setScorerAlphaBeta(self, alpha, beta):
        return native.lib().STT_SetScorerAlphaBeta(self._impl, alpha, beta)

    def stt(self, audio_buffer):
        a, p, n = _audio(audio_buffer)
        s = native.take_string(native.lib().STT_SpeechToText(self._impl, p, n))
        if s is None:
            raise RuntimeError("STT_SpeechToText failed")
        return s.decode("utf-8", "replace")

    def sttWithMetadata(self, audio_buffer, num_results=1):
        a, p, n = _audio(audio_buffer)
        return _metadata_to_py(native.lib().STT_SpeechToTextWithMetadata(self._impl, p, n, num_results))

    def sttWithEmissions(self, audio_buffer, num_results=1):
        a, p, n = _audio(audio_buffer)
        return _metadata_to_py(native.lib().STT_SpeechToTextWithEmissions(self._impl, p, n, num_results))

    def createStream(self):
        h = C.c_void_p()
        status = native.lib().STT_CreateStream(self._impl, C.byref(h))
        if status != 0:
            raise RuntimeError("CreateStream failed with '{}' (0x{:X})".format(native.error_message(status), status))
        return Stream(h, self)

    # ---- stt_amd.h ----------------------------------------------------------------------------
    def geometry(self):
        g = (C.c_int * 10)()
        native.lib().STTX_GetGeometry(self._impl, g)
        keys = ["n_input", "n_context", "n_hidden", "n_classes", "n_steps", "sample_rate", "win_len", "win_step", "beam_width", "space"]
        return dict(zip(keys, list(g)))

    def sttBatch(self, audio_buffers):
        arrs = [np.ascontiguousarray(a, dtype=np.int16) for a in audio_buffers]
        B = len(arrs)
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        sizes = (C.c_uint * B)(*[a.shape[0] for a in arrs])
        r = native.lib().STTX_SpeechToTextBatch(self._impl, ptrs, sizes, B)
        if not r:
            raise RuntimeError("STTX_SpeechToTextBatch failed")
        out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(B)]
        native.lib().STTX_FreeStrings(r, B)
        return out

    def sttBatchWithMetadata(self, audio_buffers, num_results=1):
        arrs = [np.ascontiguousarray(a, dtype=np.int16) for a in audio_buffers]
        B = len(arrs)
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        sizes = (C.c_uint * B)(*[a.shape[0] for a in arrs])
        r = native.lib().STTX_SpeechToTextBatchWithMetadata(self._impl, ptrs, sizes, B, num_results)
        if not r:
            raise RuntimeError("STTX_SpeechToTextBatchWithMetadata failed")
        out = [_metadata_to_py(r[i], free=False) for i in range(B)]
        native.lib().STTX_FreeMetadataArray(r, B)
        return out

    def sttBatchDevice(self, device_ptr, stride, sizes):
        """device_ptr: integer address of int16 [B][stride] already in HBM (e.g. torch tensor .data_ptr())."""
        B = len(sizes)
        sz = (C.c_uint * B)(*[int(s) for s in sizes])
        r = native.lib().STTX_SpeechToTextBatchDevice(self._impl, C.c_void_p(device_ptr), stride, sz, B)
        if not r:
            raise RuntimeError("STTX_SpeechToTextBatchDevice failed")
        out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(B)]
        native.lib().STTX_FreeStrings(r, B)
        return out

    def submitBatchDevice(self, device_ptr, stride, sizes):
        """Enqueue one batch of 1..64 utterances (audio resident in HBM) without waiting; returns a ticket for collectBatch().
        At most pipelineDepth() batches may be in flight (STTX_BatchSubmitDevice)."""
        B = len(sizes)
        sz = sizes if isinstance(sizes, C.Array) else (C.c_uint * B)(*[int(s) for s in sizes])
        t = native.lib().STTX_BatchSubmitDevice(self._impl, C.c_void_p(device_ptr), stride, sz, B)
        if t < 0:
            raise RuntimeError("STTX_BatchSubmitDevice failed (%d)" % t)
        return t

    def submitBatch(self, audio_buffers):
        """STTX_BatchSubmit: 1..64 host int16 buffers -> ticket (collectBatch*).  `audio_buffers`: a list of int16 arrays, or a prepared
        (pointer array, size array, count, keep-alive) tuple from prepareBatch() -- a caller that submits the same buffers again and again
        (a benchmark) builds the two small ctypes arrays once."""
        ptrs, sizes, n, _ = audio_buffers if isinstance(audio_buffers, tuple) else self.prepareBatch(audio_buffers)
        t = native.lib().STTX_BatchSubmit(self._impl, ptrs, sizes, n)
        if t < 0:
            raise RuntimeError("STTX_BatchSubmit failed 0x%X" % -t)
        return t

    @staticmethod
    def prepareBatch(audio_buffers):
        arrs = [np.ascontiguousarray(a, dtype=np.int16) for a in audio_buffers]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        sizes = (C.c_uint * len(arrs))(*[a.shape[0] for a in arrs])
        return ptrs, sizes, len(arrs), arrs

    def pipelineDepth(self):
        """Batches STTX_BatchSubmitDevice accepts for this model before the oldest must be collected (STTX_BatchPipelineDepthFor)."""
        return int(native.lib().STTX_BatchPipelineDepthFor(self._impl))

    def collectBatch(self, ticket):
        """Wait for a submitted batch; its transcripts in submission order (STTX_BatchCollect)."""
        n = C.c_uint(0)
        r = native.lib().STTX_BatchCollect(self._impl, int(ticket), C.byref(n))
        if not r:
            raise RuntimeError("STTX_BatchCollect failed")
        out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(n.value)]
        native.lib().STTX_FreeStrings(r, n.value)
        return out

    def collectBatchScored(self, ticket):
        """Wait for a submitted batch; (transcripts, confidences of the best transcripts) (STTX_BatchCollectScored)."""
        n = C.c_uint(0)
        conf = (C.c_double * 64)()
        r = native.lib().STTX_BatchCollectScored(self._impl, int(ticket), C.byref(n), conf)
        if not r:
            raise RuntimeError("STTX_BatchCollectScored failed")
        out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(n.value)]
        native.lib().STTX_FreeStrings(r, n.value)
        return out, list(conf[:n.value])

    def collectBatchWithMetadata(self, ticket):
        """Wait for a submitted batch; per utterance the best transcript's metadata (STTX_BatchCollectWithMetadata)."""
        n = C.c_uint(0)
        r = native.lib().STTX_BatchCollectWithMetadata(self._impl, int(ticket), C.byref(n))
        if not r:
            raise RuntimeError("STTX_BatchCollectWithMetadata failed")
        out = [_metadata_to_py(r[i], free=False) for i in range(n.value)]
        native.lib().STTX_FreeMetadataArray(r, n.value)
        return out

    def batchProbs(self, ticket, n_utterances):
        """STTX_DebugBatchProbs: the probabilities the pipelined path computed for a submitted, not yet collected batch."""
        g = self.geometry()
        tmax = 1 << 12
        nfr = (C.c_uint * n_utterances)()
        probs = np.zeros((n_utterances, tmax, g["n_classes"]), dtype=np.float32)
        status = native.lib().STTX_DebugBatchProbs(self._impl, int(ticket), probs.ctypes.data, tmax, nfr)
        if status != 0:
            raise RuntimeError("STTX_DebugBatchProbs failed 0x%X" % status)
        return [probs[b, :nfr[b]].copy() for b in range(n_utterances)]

    def lstmSteps(self, xproj, batch, steps, graph=False, timing=False):
        """STTX_TestLstmSteps: `steps` recurrent steps from a zero state; xproj f32 [period * batch][4 * n_hidden].
        Returns (c, h, h_all bits) -- final state [batch][H] and the f16 bits of h over the last `period` steps."""
        g = self.geometry()
        H = g["n_hidden"]
        x = np.ascontiguousarray(xproj, dtype=np.float32)
        period = x.shape[0] // batch
        assert x.shape == (period * batch, 4 * H)
        c = np.zeros((batch, H), np.float32); h = np.zeros((batch, H), np.float32); hall = np.zeros((period * batch, H), np.uint16)
        ms = C.c_float(0.0)
        status = native.lib().STTX_TestLstmSteps(self._impl, batch, steps, period, int(bool(graph)), x.ctypes.data, c.ctypes.data, h.ctypes.data,
                                                 hall.ctypes.data, C.byref(ms))
        if status != 0:
            raise RuntimeError("STTX_TestLstmSteps failed 0x%X" % status)
        return (c, h, hall, ms.value) if timing else (c, h, hall)

    def setProfiling(self, level):
        """0/False = off, 1/True = stage events + decoder counters, 2 = also the search kernel's phase cycle counters, 3 = only the events around
        the recurrence's launches (lstm_ms, lstm_launches, timesteps)."""
        native.lib().STTX_SetProfiling(self._impl, int(level))

    def stageTimes(self):
        ms = (C.c_float * 8)()
        native.lib().STTX_GetStageTimes(self._impl, ms, 8)
        keys = ["features_ms", "dense_in_ms", "lstm_ms", "dense_out_ms", "decoder_next_ms", "decoder_decode_ms", "lstm_launches", "timesteps"]
        return dict(zip(keys, list(ms)))

    def decoderStats(self):
        st = (C.c_ulonglong * 4)()
        native.lib().STTX_GetDecoderStats(self._impl, st)
        return dict(steps=st[0], candidates=st[1], lm_queries=st[2], lm_probes=st[3])

    def decoderPhaseCycles(self):
        st = (C.c_ulonglong * 8)()
        native.lib().STTX_GetDecoderPhaseCycles(self._impl, st)
        return dict(zip(["setup", "expand_events", "expand_items", "lm", "merge", "select", "rank+write", "lm_wave (parallel to expand)"], [int(x) for x in st]))

    def decoderStamps(self):
        st = (C.c_ulonglong * 64)()
        native.lib().STTX_GetDecoderStamps(self._impl, st)
        return [int(x) for x in st]

    def computeMfcc(self, audio_buffer):
        a, p, n = _audio(audio_buffer)
        g = self.geometry()
        cap = n // g["win_step"] + 4
        out = np.zeros((cap, g["n_input"]), dtype=np.float32)
        nf = C.c_uint(0)
        status = native.lib().STTX_ComputeMfcc(self._impl, p, n, out.ctypes.data, cap, C.byref(nf))
        if status != 0:
            raise RuntimeError("STTX_ComputeMfcc failed 0x%X" % status)
        return out[:nf.value].copy()

    def acousticProbs(self, audio_buffers):
        arrs = [np.ascontiguousarray(a, dtype=np.int16) for a in audio_buffers]
        B = len(arrs)
        g = self.geometry()
        tmax = max(a.shape[0] for a in arrs) // g["win_step"] + 4
        probs = np.zeros((B, tmax, g["n_classes"]), dtype=np.float32)
        nfr = (C.c_uint * B)()
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        sizes = (C.c_uint * B)(*[a.shape[0] for a in arrs])
        status = native.lib().STTX_AcousticProbs(self._impl, ptrs, sizes, B, probs.ctypes.data, tmax, nfr)
        if status != 0:
            raise RuntimeError("STTX_AcousticProbs failed 0x%X" % status)
        return [probs[b, :nfr[b]].copy() for b in range(B)]

    def inferChunk(self, windows, state_c, state_h):
        g = self.geometry()
        w = np.ascontiguousarray(windows, dtype=np.float32)
        T = w.shape[0]
        c = np.ascontiguousarray(state_c, dtype=np.float32); h = np.ascontiguousarray(state_h, dtype=np.float32)
        probs = np.zeros((T, g["n_classes"]), dtype=np.float32)
        nc = np.zeros_like(c); nh = np.zeros_like(h)
        status = native.lib().STTX_InferChunk(self._impl, w.ctypes.data, T, c.ctypes.data, h.ctypes.data, probs.ctypes.data,
                                              nc.ctypes.data, nh.ctypes.data)
        if status != 0:
            raise RuntimeError("STTX_InferChunk failed 0x%X" % status)
        return probs, nc, nh

    def acousticMode(self):
        """0 = f16 MFMA operands / f32 accumulate, 1 = TFLite's hybrid int8 arithmetic end to end (include/stt_amd.h: STTX_GetAcousticMode)."""
        return int(native.lib().STTX_GetAcousticMode(self._impl))

    def slowRows(self):
        """int8 path, test hook: rows that took the recurrent step's slow path so far (include/stt_amd.h: STTX_DebugSlowRows)."""
        import ctypes
        n = ctypes.c_uint(0)
        status = native.lib().STTX_DebugSlowRows(self._impl, ctypes.byref(n))
        if status != 0:
            raise RuntimeError("STTX_DebugSlowRows failed with '{}' (0x{:X})".format(native.error_message(status), status))
        return int(n.value)

    def hybridChain(self, windows, state_c=None, state_h=None):
        """Test hook (int8-path models): windows f32 [T][B][n_in1] -> dict of the chain's intermediate results (STTX_TestHybridChain)."""
        g = self.geometry()
        w = np.ascontiguousarray(windows, dtype=np.float32)
        T, B, _ = w.shape
        H, Cn = g["n_hidden"], g["n_classes"]
        c = None if state_c is None else np.ascontiguousarray(state_c, dtype=np.float32)
        h = None if state_h is None else np.ascontiguousarray(state_h, dtype=np.float32)
        out = {"l3": np.zeros((T, B, H), np.float32), "accx": np.zeros((T, B, 4 * H), np.int32), "h_all": np.zeros((T, B, H), np.float32),
               "logits": np.zeros((T, B, Cn), np.float32), "probs": np.zeros((B, T, Cn), np.float32), "c": np.zeros((B, H), np.float32), "h": np.zeros((B, H), np.float32)}
        slow = C.c_uint(0)
        ms = C.c_float(0)
        status = native.lib().STTX_TestHybridChain(self._impl, w.ctypes.data, B, T, None if c is None else c.ctypes.data, None if h is None else h.ctypes.data,
                                                   out["l3"].ctypes.data, out["accx"].ctypes.data, out["h_all"].ctypes.data, out["logits"].ctypes.data,
                                                   out["probs"].ctypes.data, out["c"].ctypes.data, out["h"].ctypes.data, C.byref(slow), C.byref(ms))
        if status != 0:
            raise RuntimeError("STTX_TestHybridChain failed 0x%X" % status)
        out["slow_rows"] = int(slow.value)
        out["lstm_ms"] = float(ms.value)
        return out

    def createDecoder(self, n_streams=1, beam_width=None, cutoff_prob=1.0, cutoff_top_n=40):
        return Decoder(self, n_streams, beam_width or self.beamWidth(), cutoff_prob, cutoff_top_n)


def _stream_ptrs(streams):
    for st in streams:
        st._check()
    return (C.c_void_p * len(streams))(*[st._impl for st in streams])


def feedAudioContentBatch(streams, audio_buffers, last=None):
    """STTX_FeedAudioContentBatch: stream i receives audio_buffers[i]; the ready windows of all streams run as one batch.
    last (optional, one flag per stream): this is the stream's final audio -- its flush rides in the same pass
    (STTX_FeedAudioContentBatchEx) and finishStream() / finishStreamBatch() then only decode."""
    arrs = [np.ascontiguousarray(a, dtype=np.int16) for a in audio_buffers]
    n = len(streams)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else 0 for a in arrs])
    sizes = (C.c_uint * n)(*[a.shape[0] for a in arrs])
    if last is None:
        native.lib().STTX_FeedAudioContentBatch(_stream_ptrs(streams), ptrs, sizes, n)
    else:
        flags = (C.c_ubyte * n)(*[int(f) if f else 0 for f in last])   # 1 = final audio, flush now; 2 = final audio, the flush's tail rides in the next call
        native.lib().STTX_FeedAudioContentBatchEx(_stream_ptrs(streams), ptrs, sizes, flags, n)


class StreamBatchCall(object):
    """Preallocated argument tables for the batched stream calls of a server's hop loop (a cohort of at most `capacity` streams): the caller
    fills streams / audio addresses / sizes / flags by index and calls feed() and decode() -- no per-stream numpy or ctypes objects per hop
    (128 streams: 0.24 ms of Python per hop with feedAudioContentBatch's conveniences, a seventh of the hop)."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.streams = (C.c_void_p * capacity)()
        self.audio = (C.c_void_p * capacity)()
        self.sizes = (C.c_uint * capacity)()
        self.last = (C.c_ubyte * capacity)()
        self.finish = (C.c_ubyte * capacity)()
        self._objs = [None] * capacity      # the Stream objects of the rows: kept alive between set() and feed() / decode()

    def set(self, i, stream, audio_address, n_samples, last=0, finish=0):
        """row i: `stream` (a Stream) gets n_samples int16 samples at audio_address (0 / None with n_samples 0: nothing) in feed();
        last as STTX_FeedAudioContentBatchEx's aLast; finish != 0: decode() finishes (destroys) the stream"""
        stream._check()
        self.streams[i] = stream._impl
        self._objs[i] = stream
        self.audio[i] = audio_address if n_samples else None
        self.sizes[i] = n_samples
        self.last[i] = last
        self.finish[i] = finish

    def feed(self, n):
        native.lib().STTX_FeedAudioContentBatchEx(self.streams, self.audio, self.sizes, self.last, n)

    def decode(self, n, finished_streams=()):
        """STTX_DecodeStreamsBatch over rows [0, n): -> n strings.  Every row flagged finish is destroyed by the call, whether it succeeds or
        not (include/stt_amd.h): its Stream object is marked so here -- a later freeStream() / __del__ must not hand the native stream back
        a second time.  (`finished_streams` is accepted for callers of round 4 and ignored: the rows themselves say which streams went.)"""
        if n == 0:
            return []
        r = native.lib().STTX_DecodeStreamsBatch(self.streams, self.finish, n)
        for i in range(n):
            if self.finish[i] and self._objs[i] is not None:
                self._objs[i]._impl = None
                self._objs[i] = None
        if not r:
            raise RuntimeError("STTX_DecodeStreamsBatch failed")
        out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(n)]
        native.lib().STTX_FreeStrings(r, n)
        return out


def intermediateDecodeBatch(streams):
    n = len(streams)
    if n == 0:
        return []
    r = native.lib().STTX_IntermediateDecodeBatch(_stream_ptrs(streams), n)
    if not r:
        raise RuntimeError("STTX_IntermediateDecodeBatch failed")
    out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(n)]
    native.lib().STTX_FreeStrings(r, n)
    return out


def decodeStreamsBatch(streams, finish):
    """STTX_DecodeStreamsBatch: one launch for a hop's intermediate results (finish[i] false) and its finishes (true: stream destroyed)."""
    n = len(streams)
    if n == 0:
        return []
    flags = (C.c_ubyte * n)(*[1 if f else 0 for f in finish])
    r = native.lib().STTX_DecodeStreamsBatch(_stream_ptrs(streams), flags, n)
    for st, f in zip(streams, finish):
        if f:
            st._impl = None
    if not r:
        raise RuntimeError("STTX_DecodeStreamsBatch failed")
    out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(n)]
    native.lib().STTX_FreeStrings(r, n)
    return out


def finishStreamBatch(streams):
    n = len(streams)
    if n == 0:
        return []
    r = native.lib().STTX_FinishStreamBatch(_stream_ptrs(streams), n)
    for st in streams:
        st._impl = None   # destroyed by the call, like STT_FinishStream
    if not r:
        raise RuntimeError("STTX_FinishStreamBatch failed")
    out = [C.string_at(r[i]).decode("utf-8", "replace") for i in range(n)]
    native.lib().STTX_FreeStrings(r, n)
    return out


class Stream(object):
    """native_client/python/__init__.py:223-384"""

    def __init__(self, native_stream, model):
        self._impl = native_stream
        self._model = model  # keep the model alive

    def __del__(self):
        if getattr(self, "_impl", None):
            self.freeStream()

    def _check(self):
        if not self._impl:
            raise RuntimeError("Stream object is not valid. Trying to use it after finishStream or freeStream?")

    def feedAudioContent(self, audio_buffer):
        self._check()
        a, p, n = _audio(audio_buffer)
        native.lib().STT_FeedAudioContent(self._impl, p, n)

    def intermediateDecode(self):
        self._check()
        return _text(native.take_string(native.lib().STT_IntermediateDecode(self._impl)))

    def intermediateDecodeWithMetadata(self, num_results=1):
        self._check()
        return _metadata_to_py(native.lib().STT_IntermediateDecodeWithMetadata(self._impl, num_results))

    def intermediateDecodeFlushBuffers(self):
        self._check()
        return _text(native.take_string(native.lib().STT_IntermediateDecodeFlushBuffers(self._impl)))

    def intermediateDecodeWithMetadataFlushBuffers(self, num_results=1):
        self._check()
        return _metadata_to_py(native.lib().STT_IntermediateDecodeWithMetadataFlushBuffers(self._impl, num_results))

    def finishStream(self):
        self._check()
        s = native.take_string(native.lib().STT_FinishStream(self._impl))
        self._impl = None
        return _text(s)

    def finishStreamWithMetadata(self, num_results=1):
        self._check()
        m = _metadata_to_py(native.lib().STT_FinishStreamWithMetadata(self._impl, num_results))
        self._impl = None
        return m

    def freeStream(self):
        self._check()
        native.lib().STT_FreeStream(self._impl)
        self._impl = None


class Decoder(object):
    """DecoderState (ctc_beam_search_decoder.h:14-87) for n independent streams, running on the GPU."""

    def __init__(self, model, n_streams, beam_width, cutoff_prob, cutoff_top_n):
        h = C.c_void_p()
        status = native.lib().STTX_DecoderCreate(model._impl, n_streams, beam_width, cutoff_prob, cutoff_top_n, C.byref(h))
        if status != 0:
            raise RuntimeError("STTX_DecoderCreate failed 0x%X" % status)
        self._impl, self._model, self.n, self.beam = h, model, n_streams, beam_width
        self.C = model.geometry()["n_classes"]

    def next(self, probs, n_frames=None):
        """probs: float32 [n_streams, T, C] (or [T, C] for one stream)."""
        p = np.ascontiguousarray(probs, dtype=np.float32)
        if p.ndim == 2:
            p = p[None]
        assert p.shape[0] == self.n and p.shape[2] == self.C
        nf = (C.c_uint * self.n)(*([p.shape[1]] * self.n if n_frames is None else [int(x) for x in n_frames]))
        status = native.lib().STTX_DecoderNext(self._impl, p.ctypes.data, p.shape[1], nf)
        if status != 0:
            raise RuntimeError("STTX_DecoderNext failed 0x%X" % status)

    def decode(self, num_results=1, max_len=2048):
        tok = np.zeros((self.n, num_results, max_len), dtype=np.uint32)
        ts = np.zeros((self.n, num_results, max_len), dtype=np.uint32)
        lens = np.zeros((self.n, num_results), dtype=np.int32)
        conf = np.zeros((self.n, num_results), dtype=np.float64)
        nres = np.zeros(self.n, dtype=np.int32)
        status = native.lib().STTX_DecoderDecode(self._impl, num_results, max_len, tok.ctypes.data, ts.ctypes.data,
                                                 lens.ctypes.data, conf.ctypes.data, nres.ctypes.data)
        if status != 0:
            raise RuntimeError("STTX_DecoderDecode failed 0x%X" % status)
        return [[(conf[i, r], tok[i, r, :lens[i, r]].copy(), ts[i, r, :lens[i, r]].copy()) for r in range(nres[i])]
                for i in range(self.n)]

    def raw_beam(self, stream=0):
        cap = self.beam + 8
        sc = np.zeros(cap, np.float32); pb = np.zeros(cap, np.float32); pnb = np.zeros(cap, np.float32); ch = np.zeros(cap, np.int32)
        n = native.lib().STTX_DecoderBeam(self._impl, stream, sc.ctypes.data, pb.ctypes.data, pnb.ctypes.data, ch.ctypes.data, cap)
        return sc[:n], pb[:n], pnb[:n], ch[:n]

    def setProfiling(self, level):
        native.lib().STTX_DecoderSetProfiling(self._impl, int(level))

    def profile(self):
        """(phase cycles by name, 64 stamps, HIP-event ms of the search launches) since setProfiling()."""
        ph, st, ms = (C.c_ulonglong * 8)(), (C.c_ulonglong * 64)(), C.c_float(0)
        native.lib().STTX_DecoderGetProfile(self._impl, ph, st, C.byref(ms))
        names = ["setup", "expand_events", "expand_items", "lm", "merge", "select", "rank+write", "lm_wave (parallel to expand)"]
        return dict(zip(names, [int(x) for x in ph])), [int(x) for x in st], float(ms.value)

    def stats(self):
        st = (C.c_ulonglong * 4)()
        status = native.lib().STTX_DecoderStats(self._impl, st)
        return dict(steps=st[0], candidates=st[1], lm_queries=st[2], lm_probes=st[3], error=status)

    def error_bits(self):
        """OR of the streams' search-state error bits (include/stt_amd.h: STTX_DecoderErrorBits); 0 = intact."""
        b = C.c_int(0)
        if native.lib().STTX_DecoderErrorBits(self._impl, C.byref(b)) != 0:
            raise RuntimeError("STTX_DecoderErrorBits failed")
        return int(b.value)

    def close(self):
        if self._impl:
            native.lib().STTX_DecoderFree(self._impl)
            self._impl = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Fleet(object):
    """Several GPUs of one node behind one handle (include/stt_amd.h: STTX_Fleet*): a replica per device, utterances dealt
    longest-processing-time-first, one host thread per device, transcripts gathered with RCCL.  The in-library form of the
    reference's one-process-per-GPU pattern (training/coqui_stt_training/transcribe.py:40-56)."""

    def __init__(self, model_path, devices):
        self._impl = None
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        status = native.lib().STTX_FleetCreate(model_path.encode("utf-8"), devs, len(devices), C.byref(h))
        if status != 0:
            raise RuntimeError("STTX_FleetCreate failed with '{}' (0x{:X})".format(native.error_message(status), status))
        self._impl = h

    def __del__(self):
        if getattr(self, "_impl", None):
            native.lib().STTX_FleetFree(self._impl)
            self._impl = None

    def size(self):
        return native.lib().STTX_FleetSize(self._impl)

    def enableExternalScorer(self, scorer_path):
        status = native.lib().STTX_FleetEnableExternalScorer(self._impl, scorer_path.encode("utf-8"))
        if status != 0:
            raise RuntimeError("STTX_FleetEnableExternalScorer failed with '{}' (0x{:X})".format(native.error_message(status), status))

    def setBeamWidth(self, beam_width):
        return native.lib().STTX_FleetSetBeamWidth(self._impl, beam_width)

    def sttBatch(self, audio_buffers):
        arrs = [_audio(a) for a in audio_buffers]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a[1] for a in arrs])
        sizes = (C.c_uint * n)(*[a[2] for a in arrs])
        res = native.lib().STTX_FleetSpeechToTextBatch(self._impl, ptrs, sizes, n)
        if not res:
            raise RuntimeError("STTX_FleetSpeechToTextBatch failed")
        out = [C.string_at(res[i]).decode("utf-8", "replace") for i in range(n)]
        native.lib().STTX_FreeStrings(res, n)
        return out


def shard_utterances_native(lengths, n_shards):
    """STTX_ShardUtterances: shard index of every utterance (the C++ twin of stt_amd.dist.shard_utterances)."""
    n = len(lengths)
    sizes = (C.c_uint * n)(*[int(x) for x in lengths])
    out = (C.c_uint * n)()
    status = native.lib().STTX_ShardUtterances(sizes, n, n_shards, out)
    if status != 0:
        raise RuntimeError("STTX_ShardUtterances failed 0x%X" % status)
    return [int(x) for x in out]
