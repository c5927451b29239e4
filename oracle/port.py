"""oracle/port.py -- TEST INFRASTRUCTURE ONLY (ctypes face of oracle/_build/libstt_oracle.so).

libstt_oracle.so is OUR plain-C restatement of the decoder half of the hot path
(oracle/stt_port.c + oracle/glibc_flt.h), built by `make -C oracle port`.  It is the
"port" oracle: the HIP kernels are compared against it bit for bit, and it is itself pinned
against the real reference (oracle/ref.py) by tests/test_oracle_port.py.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libstt_oracle.so")
_lib = None


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_LIB_PATH)
        vp, ci, cd, cf = C.c_void_p, C.c_int, C.c_double, C.c_float
        sig = {
            "port_murmur64a": (C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64]),
            "port_scorer_load": (vp, [vp, C.c_size_t, C.POINTER(ci)]),
            "port_kenlm_load": (vp, [vp, C.c_size_t, C.POINTER(ci)]),
            "port_scorer_free": (None, [vp]),
            "port_scorer_order": (ci, [vp]),
            "port_scorer_utf8": (ci, [vp]),
            "port_scorer_alpha": (cd, [vp]),
            "port_scorer_beta": (cd, [vp]),
            "port_scorer_set_alpha_beta": (None, [vp, cf, cf]),
            "port_scorer_lm_end": (C.c_uint64, [vp]),
            "port_scorer_model_type": (ci, [vp]),
            "port_kenlm_index": (C.c_uint, [vp, C.c_char_p, C.c_size_t]),
            "port_kenlm_score": (ci, [vp, C.POINTER(C.c_char_p), ci, ci, vp, vp]),
            "port_scorer_log_cond_prob": (cd, [vp, C.POINTER(C.c_char_p), ci, ci, ci]),
            "port_scorer_fst_dump": (C.c_long, [vp, C.POINTER(ci), vp, C.c_long, C.POINTER(C.c_long), vp, C.c_long]),
            "port_decoder_new": (vp, [ci, ci, ci, cd, ci, vp, vp, vp, C.POINTER(C.c_char_p), C.POINTER(cf), ci]),
            "port_decoder_free": (None, [vp]),
            "port_decoder_next": (None, [vp, vp, ci, ci]),
            "port_decoder_decode": (ci, [vp, ci, vp, vp, vp, vp, ci]),
            "port_decoder_beam": (ci, [vp, vp, vp, vp, vp, vp, ci]),
            "port_decoder_stats": (None, [vp, vp]),
            "port_expf": (cf, [cf]),
            "port_logf": (cf, [cf]),
            "port_expf_array": (None, [vp, vp, C.c_long]),
            "port_logf_array": (None, [vp, vp, C.c_long]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _cstrs(words):
    arr = (C.c_char_p * max(1, len(words)))()
    for i, w in enumerate(words):
        arr[i] = w if isinstance(w, bytes) else w.encode("utf-8")
    return arr


def parse_alphabet_file(path):
    """Alphabet::init (native_client/alphabet.cc:42-68) -> (labels: list[bytes], space_index)."""
    labels, space = [], -2
    with open(path, "rb") as f:
        data = f.read()
    for line in data.replace(b"\r\n", b"\n").replace(b"\r", b"\n").split(b"\n"):
        if line == b"\\#":
            line = b"#"
        elif line[:1] == b"#":
            continue
        if line == b" ":
            space = len(labels)
        if len(line) == 0:
            continue
        labels.append(line)
    return labels, space


def utf8_alphabet():
    """UTF8Alphabet (native_client/alphabet.h:80-100): label n <-> byte n+1, space = ' ' - 1."""
    return [bytes([i + 1]) for i in range(255)], ord(" ") - 1


class Scorer:
    def __init__(self, path=None, data=None, lm_only=False):
        if data is None:
            with open(path, "rb") as f:
                data = f.read()
        # 8-byte aligned, 16 bytes of slack for the 64-bit bit-packed reads
        self._buf = np.zeros((len(data) + 16 + 7) // 8 + 1, dtype=np.uint64)
        self._buf.view(np.uint8)[:len(data)] = np.frombuffer(data, dtype=np.uint8)
        err = C.c_int(0)
        self.h = (lib().port_kenlm_load if lm_only else lib().port_scorer_load)(self._buf.ctypes.data, len(data), C.byref(err))
        self.err = err.value
        if not self.h:
            raise RuntimeError("port_scorer_load failed: 0x%x" % self.err)
        self.lm_only = lm_only
        self.utf8 = bool(lib().port_scorer_utf8(self.h))
        self.order = lib().port_scorer_order(self.h)
        self.model_type = lib().port_scorer_model_type(self.h)

    @property
    def alpha(self):
        return lib().port_scorer_alpha(self.h)

    @property
    def beta(self):
        return lib().port_scorer_beta(self.h)

    def set_alpha_beta(self, a, b):
        lib().port_scorer_set_alpha_beta(self.h, a, b)

    def index(self, word):
        w = word if isinstance(word, bytes) else word.encode()
        return lib().port_kenlm_index(self.h, w, len(w))

    def score(self, words, bos=True):
        probs = np.zeros(len(words), np.float32)
        lens = np.zeros(len(words), np.int32)
        lib().port_kenlm_score(self.h, _cstrs(words), len(words), int(bos), probs.ctypes.data, lens.ctypes.data)
        return probs, lens

    def log_cond_prob(self, words, bos=False, eos=False):
        return lib().port_scorer_log_cond_prob(self.h, _cstrs(words), len(words), int(bos), int(eos))

    def fst(self):
        start = C.c_int(0)
        na = C.c_long(0)
        ns = lib().port_scorer_fst_dump(self.h, C.byref(start), None, 0, C.byref(na), None, 0)
        arcs = np.zeros((na.value, 3), dtype=np.int32)
        finals = np.zeros(ns, dtype=np.uint8)
        lib().port_scorer_fst_dump(self.h, C.byref(start), arcs.ctypes.data, na.value, C.byref(na), finals.ctypes.data, ns)
        return start.value, arcs, finals


class Decoder:
    """reference_order=False: the flat restatement the kernels share their tie rule with (ties of (score, character) broken by live before new,
    beam index).  reference_order=True: the pointer-trie restatement with libstdc++'s nth_element / partial_sort restated (stt_port.c, Part D):
    the compiled reference's output including the tie cases."""

    def __init__(self, labels, space, beam, scorer=None, cutoff_prob=1.0, cutoff_top_n=40, hot_words=None, reference_order=False):
        self.labels = labels
        self._t = bool(reference_order)
        self.C = len(labels) + 1
        self._bytes = np.frombuffer(b"".join(labels) + b"\0", dtype=np.uint8).copy()
        self._off = np.cumsum([len(l) for l in labels]).astype(np.int32)
        hot_words = hot_words or {}
        words = list(hot_words.keys())
        boosts = (C.c_float * max(1, len(words)))(*[hot_words[w] for w in words])
        L = lib()
        if self._t:
            L.port_tdecoder_new.restype = C.c_void_p
            L.port_tdecoder_new.argtypes = L.port_decoder_new.argtypes
            L.port_tdecoder_next.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            L.port_tdecoder_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            L.port_tdecoder_free.argtypes = [C.c_void_p]
        new = L.port_tdecoder_new if self._t else L.port_decoder_new
        self.h = new(self.C, space, beam, cutoff_prob, cutoff_top_n, scorer.h if scorer else None,
                     self._bytes.ctypes.data, self._off.ctypes.data, _cstrs(words), boosts, len(words))
        self.beam = beam
        self._keep = scorer

    def next(self, probs):
        p = np.ascontiguousarray(probs, dtype=np.float64)
        assert p.ndim == 2 and p.shape[1] == self.C
        (lib().port_tdecoder_next if self._t else lib().port_decoder_next)(self.h, p.ctypes.data, p.shape[0], p.shape[1])

    def decode(self, num_results=1, max_len=4096):
        tok = np.zeros((num_results, max_len), dtype=np.uint32)
        ts = np.zeros((num_results, max_len), dtype=np.uint32)
        lens = np.zeros(num_results, dtype=np.int32)
        conf = np.zeros(num_results, dtype=np.float64)
        n = (lib().port_tdecoder_decode if self._t else lib().port_decoder_decode)(self.h, num_results, tok.ctypes.data, ts.ctypes.data, lens.ctypes.data,
                                                                                   conf.ctypes.data, max_len)
        if n < 0:
            raise RuntimeError("result longer than max_len")
        return [(conf[i], tok[i, :lens[i]].copy(), ts[i, :lens[i]].copy()) for i in range(n)]

    def raw_beam(self):
        cap = self.beam + 8
        sc = np.zeros(cap, np.float32); pb = np.zeros(cap, np.float32); pnb = np.zeros(cap, np.float32)
        ch = np.zeros(cap, np.int32); ln = np.zeros(cap, np.int32)
        f = lib().port_tdecoder_beam if self._t else lib().port_decoder_beam
        f.argtypes = [C.c_void_p] * 6 + [C.c_int]
        n = f(self.h, sc.ctypes.data, pb.ctypes.data, pnb.ctypes.data, ch.ctypes.data, ln.ctypes.data, cap)
        return sc[:n], pb[:n], pnb[:n], ch[:n], ln[:n]

    def boundary_ties(self):
        """Steps at which a (score, character) tie straddled the beam boundary: there the reference's choice is libstdc++'s
        nth_element order (stt_port.c: stat_boundary_ties); 0 = the result is determined by the scores alone."""
        f = lib().port_decoder_boundary_ties
        f.restype, f.argtypes = C.c_uint64, [C.c_void_p]
        return int(f(self.h))

    def stats(self):
        out = np.zeros(3, np.uint64)
        lib().port_decoder_stats(self.h, out.ctypes.data)
        return dict(steps=int(out[0]), candidates=int(out[1]), lm_queries=int(out[2]))

    def close(self):
        if self.h:
            (lib().port_tdecoder_free if self._t else lib().port_decoder_free)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_text(labels, tokens):
    return b"".join(labels[t] for t in tokens)
