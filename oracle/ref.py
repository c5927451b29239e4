"""oracle/ref.py -- TEST INFRASTRUCTURE ONLY (ctypes face of oracle/_ref/libctcdecode_ref.so).

The shared object is the *real* reference decoder (native_client/ctcdecode + vendored KenLM
and OpenFst) compiled in place by oracle/Makefile; this module only marshals arguments.
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libctcdecode_ref.so")
_lib = None


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_LIB_PATH)
        vp, ci, cd, cf = C.c_void_p, C.c_int, C.c_double, C.c_float
        sig = {
            "ref_alphabet_from_file": (vp, [C.c_char_p]),
            "ref_alphabet_utf8": (vp, []),
            "ref_alphabet_free": (None, [vp]),
            "ref_alphabet_size": (ci, [vp]),
            "ref_alphabet_space": (ci, [vp]),
            "ref_alphabet_decode": (ci, [vp, vp, ci, C.c_char_p, ci]),
            "ref_alphabet_serialize": (ci, [vp, C.c_char_p, ci]),
            "ref_scorer_load": (vp, [C.c_char_p, vp, C.POINTER(ci)]),
            "ref_scorer_free": (None, [vp]),
            "ref_scorer_is_utf8": (ci, [vp]),
            "ref_scorer_order": (ci, [vp]),
            "ref_scorer_alpha": (cd, [vp]),
            "ref_scorer_beta": (cd, [vp]),
            "ref_scorer_set_alpha_beta": (None, [vp, cf, cf]),
            "ref_scorer_log_cond_prob": (cd, [vp, C.POINTER(C.c_char_p), ci, ci, ci]),
            "ref_scorer_fst_dump": (C.c_long, [vp, C.POINTER(ci), vp, C.c_long, C.POINTER(C.c_long), vp, C.c_long]),
            "ref_make_scorer": (ci, [C.c_char_p, C.c_char_p, C.c_char_p, cf, cf, C.c_char_p]),
            "ref_decoder_new": (vp, [vp, ci, cd, ci, vp, C.POINTER(C.c_char_p), C.POINTER(cf), ci]),
            "ref_decoder_free": (None, [vp]),
            "ref_decoder_next": (None, [vp, vp, ci, ci]),
            "ref_decoder_decode": (ci, [vp, ci, vp, vp, vp, vp, ci]),
            "ref_decoder_beam": (ci, [vp, vp, vp, vp, vp, vp, ci]),
            "ref_decode_batch": (ci, [vp, ci, ci, ci, vp, vp, ci, ci, cd, ci, vp, vp, vp, vp, ci]),
            "ref_kenlm_load": (vp, [C.c_char_p]),
            "ref_kenlm_free": (None, [vp]),
            "ref_kenlm_order": (ci, [vp]),
            "ref_kenlm_index": (C.c_uint, [vp, C.c_char_p]),
            "ref_kenlm_score": (ci, [vp, C.POINTER(C.c_char_p), ci, ci, vp, vp]),
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


class Alphabet:
    def __init__(self, path=None):
        self.h = lib().ref_alphabet_from_file(path.encode()) if path else lib().ref_alphabet_utf8()
        if not self.h:
            raise RuntimeError("reference Alphabet::init failed for %r" % path)
        self.size = lib().ref_alphabet_size(self.h)
        self.space = lib().ref_alphabet_space(self.h)

    def decode(self, labels):
        a = np.ascontiguousarray(labels, dtype=np.uint32)
        buf = C.create_string_buffer(4 * len(a) + 8)
        n = lib().ref_alphabet_decode(self.h, a.ctypes.data, len(a), buf, len(buf))
        return buf.raw[:n]

    def serialize(self):
        buf = C.create_string_buffer(1 << 20)
        n = lib().ref_alphabet_serialize(self.h, buf, len(buf))
        return buf.raw[:n]


class Scorer:
    def __init__(self, path, alphabet):
        err = C.c_int(0)
        self.h = lib().ref_scorer_load(path.encode(), alphabet.h, C.byref(err))
        self.err = err.value
        if not self.h:
            raise RuntimeError("reference Scorer::init_from_filepath failed: 0x%x" % self.err)
        self.alphabet = alphabet
        self.utf8 = bool(lib().ref_scorer_is_utf8(self.h))
        self.order = lib().ref_scorer_order(self.h)

    @property
    def alpha(self):
        return lib().ref_scorer_alpha(self.h)

    @property
    def beta(self):
        return lib().ref_scorer_beta(self.h)

    def set_alpha_beta(self, a, b):
        lib().ref_scorer_set_alpha_beta(self.h, a, b)

    def log_cond_prob(self, words, bos=False, eos=False):
        return lib().ref_scorer_log_cond_prob(self.h, _cstrs(words), len(words), int(bos), int(eos))

    def fst(self):
        """-> (start, arcs[n,3] = (state, ilabel, nextstate), finals[numstates])"""
        start = C.c_int(0)
        na = C.c_long(0)
        ns = lib().ref_scorer_fst_dump(self.h, C.byref(start), None, 0, C.byref(na), None, 0)
        arcs = np.zeros((na.value, 3), dtype=np.int32)
        finals = np.zeros(ns, dtype=np.uint8)
        lib().ref_scorer_fst_dump(self.h, C.byref(start), arcs.ctypes.data, na.value, C.byref(na), finals.ctypes.data, ns)
        return start.value, arcs, finals


def make_scorer(lm_binary, vocab_txt, alphabet_path, alpha, beta, out_path):
    rc = lib().ref_make_scorer(lm_binary.encode(), vocab_txt.encode(),
                               alphabet_path.encode() if alphabet_path else None, alpha, beta, out_path.encode())
    if rc != 0:
        raise RuntimeError("ref_make_scorer rc=%d" % rc)


class Decoder:
    """DecoderState (ctc_beam_search_decoder.h:14-87)."""

    def __init__(self, alphabet, beam, scorer=None, cutoff_prob=1.0, cutoff_top_n=40, hot_words=None):
        hot_words = hot_words or {}
        words = list(hot_words.keys())
        boosts = (C.c_float * max(1, len(words)))(*[hot_words[w] for w in words])
        self.h = lib().ref_decoder_new(alphabet.h, beam, cutoff_prob, cutoff_top_n, scorer.h if scorer else None,
                                       _cstrs(words), boosts, len(words))
        self.beam = beam
        self._keep = (alphabet, scorer)

    def next(self, probs):
        p = np.ascontiguousarray(probs, dtype=np.float64)
        assert p.ndim == 2
        lib().ref_decoder_next(self.h, p.ctypes.data, p.shape[0], p.shape[1])

    def decode(self, num_results=1, max_len=4096):
        tok = np.zeros((num_results, max_len), dtype=np.uint32)
        ts = np.zeros((num_results, max_len), dtype=np.uint32)
        lens = np.zeros(num_results, dtype=np.int32)
        conf = np.zeros(num_results, dtype=np.float64)
        n = lib().ref_decoder_decode(self.h, num_results, tok.ctypes.data, ts.ctypes.data, lens.ctypes.data,
                                     conf.ctypes.data, max_len)
        if n < 0:
            raise RuntimeError("result longer than max_len")
        return [(conf[i], tok[i, :lens[i]].copy(), ts[i, :lens[i]].copy()) for i in range(n)]

    def raw_beam(self):
        cap = self.beam + 8
        sc = np.zeros(cap, np.float32); pb = np.zeros(cap, np.float32); pnb = np.zeros(cap, np.float32)
        ch = np.zeros(cap, np.int32); ln = np.zeros(cap, np.int32)
        n = lib().ref_decoder_beam(self.h, sc.ctypes.data, pb.ctypes.data, pnb.ctypes.data, ch.ctypes.data,
                                   ln.ctypes.data, cap)
        return sc[:n], pb[:n], pnb[:n], ch[:n], ln[:n]

    def close(self):
        if self.h:
            lib().ref_decoder_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_batch(probs, seq_lengths, alphabet, beam, num_threads, scorer=None, cutoff_prob=1.0, cutoff_top_n=40,
                 max_len=4096):
    """ctc_beam_search_decoder_batch (ctc_beam_search_decoder.cpp:608-652), top-1 per utterance."""
    p = np.ascontiguousarray(probs, dtype=np.float64)
    B, T, Cc = p.shape
    sl = np.ascontiguousarray(seq_lengths, dtype=np.int32)
    tok = np.zeros((B, max_len), dtype=np.uint32)
    lens = np.zeros(B, dtype=np.int32)
    conf = np.zeros(B, dtype=np.float64)
    rc = lib().ref_decode_batch(p.ctypes.data, B, T, Cc, sl.ctypes.data, alphabet.h, beam, num_threads, cutoff_prob,
                                cutoff_top_n, scorer.h if scorer else None, tok.ctypes.data, lens.ctypes.data,
                                conf.ctypes.data, max_len)
    if rc != 0:
        raise RuntimeError("ref_decode_batch rc=%d" % rc)
    return [(conf[b], tok[b, :lens[b]].copy()) for b in range(B)]


class KenLM:
    def __init__(self, path):
        self.h = lib().ref_kenlm_load(path.encode())
        if not self.h:
            raise RuntimeError("kenlm load failed: %s" % path)
        self.order = lib().ref_kenlm_order(self.h)

    def index(self, word):
        return lib().ref_kenlm_index(self.h, word.encode())

    def score(self, words, bos=True):
        probs = np.zeros(len(words), np.float32)
        lens = np.zeros(len(words), np.int32)
        lib().ref_kenlm_score(self.h, _cstrs(words), len(words), int(bos), probs.ctypes.data, lens.ctypes.data)
        return probs, lens
