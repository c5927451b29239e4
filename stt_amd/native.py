"""stt_amd/native.py -- ctypes binding of stt_amd/lib/libstt.so (the C-ABI of include/coqui-stt.h + stt_amd.h).

This is the stub a maintainer of the reference's Python package would write instead of the SWIG module
(native_client/python/impl.i): every function is declared with the exact C signature.  There is no Python
or CPU implementation behind it -- if the shared library is missing, importing raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.path.join(_HERE, "lib", "libstt.so")          # coqui-stt.h + stt_amd.h: what a binding loads
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libstt_test.so")        # the same objects + the hooks of include/stt_amd_test.h
# tests/conftest.py and the benchmarks/ probes set STT_AMD_TEST_HOOKS=1; bench.py, __graft_entry__.smoke() and the `stt` client run the product
TEST_HOOKS = os.environ.get("STT_AMD_TEST_HOOKS", "0") not in ("", "0")
LIB_PATH = TEST_LIB_PATH if TEST_HOOKS else PRODUCT_LIB_PATH


class TokenMetadata(C.Structure):
    _fields_ = [("text", C.c_char_p), ("timestep", C.c_uint), ("start_time", C.c_float)]


class CandidateTranscript(C.Structure):
    _fields_ = [("tokens", C.POINTER(TokenMetadata)), ("num_tokens", C.c_uint), ("confidence", C.c_double)]


class AcousticModelEmissions(C.Structure):
    _fields_ = [("num_symbols", C.c_int), ("symbols", C.POINTER(C.c_char_p)), ("num_timesteps", C.c_int),
                ("emissions", C.POINTER(C.c_double))]


class ModelInfo(C.Structure):  # STTX_ModelInfo, include/stt_amd.h
    _fields_ = [(n, C.c_int) for n in ("n_input", "n_context", "n_hidden", "n_classes", "n_steps", "sample_rate", "win_len",
                                       "win_step", "beam_width")] + [("relu_clip", C.c_float), ("alphabet_bytes", C.c_uint),
                                                                      ("is_tflite", C.c_int), ("hybrid_int8", C.c_int),
                                                                      ("asymmetric_quantize_inputs", C.c_int)]


class Metadata(C.Structure):
    _fields_ = [("transcripts", C.POINTER(CandidateTranscript)), ("num_transcripts", C.c_uint),
                ("emissions", C.POINTER(AcousticModelEmissions))]


COQUI_STT_H = [
    "STT_CreateModel", "STT_CreateModelFromBuffer", "STT_GetModelBeamWidth", "STT_SetModelBeamWidth",
    "STT_GetModelSampleRate", "STT_FreeModel", "STT_EnableExternalScorer", "STT_EnableExternalScorerFromBuffer",
    "STT_AddHotWord", "STT_EraseHotWord", "STT_ClearHotWords", "STT_DisableExternalScorer", "STT_SetScorerAlphaBeta",
    "STT_SpeechToText", "STT_SpeechToTextWithMetadata", "STT_SpeechToTextWithEmissions", "STT_CreateStream",
    "STT_FeedAudioContent", "STT_IntermediateDecode", "STT_IntermediateDecodeWithMetadata",
    "STT_IntermediateDecodeFlushBuffers", "STT_IntermediateDecodeWithMetadataFlushBuffers", "STT_FinishStream",
    "STT_FinishStreamWithMetadata", "STT_FreeStream", "STT_FreeMetadata", "STT_FreeString", "STT_Version",
    "STT_ErrorCodeToErrorMessage",
]
STT_AMD_H = [
    "STTX_SetDevice", "STTX_GetDeviceCount", "STTX_SpeechToTextBatch", "STTX_SpeechToTextBatchWithMetadata", "STTX_SpeechToTextBatchDevice",
    "STTX_BatchPipelineDepth", "STTX_BatchPipelineDepthFor", "STTX_BatchSubmit", "STTX_BatchSubmitDevice", "STTX_BatchCollect",
    "STTX_BatchCollectWithMetadata", "STTX_BatchCollectScored", "STTX_SetTuning", "STTX_GetTuning", "STTX_ConfigureRuntime", "STTX_GetAcousticMode",
    "STTX_FeedAudioContentBatch", "STTX_FeedAudioContentBatchEx", "STTX_IntermediateDecodeBatch", "STTX_FinishStreamBatch",
    "STTX_DecodeStreamsBatch", "STTX_FreeStrings", "STTX_FreeMetadataArray", "STTX_SetProfiling", "STTX_GetStageTimes", "STTX_GetDecoderStats",
    "STTX_GetDecoderPhaseCycles", "STTX_GetDecoderStamps", "STTX_ComputeMfcc", "STTX_AcousticProbs", "STTX_InferChunk", "STTX_GetGeometry",
    "STTX_DecoderCreate", "STTX_DecoderNext", "STTX_DecoderDecode", "STTX_DecoderBeam", "STTX_DecoderStats", "STTX_DecoderErrorBits",
    "STTX_DecoderSetProfiling", "STTX_DecoderGetProfile", "STTX_DecoderFree", "STTX_PackLstmRecurrent", "STTX_InspectModel", "STTX_ReadModelTensor",
    "STTX_FleetCreate", "STTX_FleetSize", "STTX_FleetEnableExternalScorer", "STTX_FleetSetBeamWidth", "STTX_FleetSpeechToTextBatch",
    "STTX_FleetFree", "STTX_ShardUtterances",
]
STT_AMD_TEST_H = [      # include/stt_amd_test.h: libstt_test.so only
    "STTX_DebugBatchProbs", "STTX_TestLstmSteps", "STTX_TestDenseHybrid", "STTX_DebugSlowRows", "STTX_TestHybridChain", "STTX_TestDense",
    "STTX_TestMath", "STTX_TestLm", "STTX_TestDictionaryWalk", "STTX_DebugLimitArena", "STTX_TestFleetRecords", "STTX_DebugFleetFailShard",
]

_lib = None


def lib():
    """Loads libstt.so; raises if it has not been built (python -m stt_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("stt_amd: %s is missing -- build it with `python -m stt_amd.build`; "
                           "there is no fallback implementation." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.STTX_ConfigureRuntime()      # before the first HIP call of this process (include/stt_amd.h)
    vp, ci, cu, cf, cd, cs = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_double, C.c_char_p
    pp = C.POINTER
    sig = {
        "STT_CreateModel": (ci, [cs, pp(vp)]),
        "STT_CreateModelFromBuffer": (ci, [vp, cu, pp(vp)]),
        "STT_GetModelBeamWidth": (cu, [vp]),
        "STT_SetModelBeamWidth": (ci, [vp, cu]),
        "STT_GetModelSampleRate": (ci, [vp]),
        "STT_FreeModel": (None, [vp]),
        "STT_EnableExternalScorer": (ci, [vp, cs]),
        "STT_EnableExternalScorerFromBuffer": (ci, [vp, vp, cu]),
        "STT_AddHotWord": (ci, [vp, cs, cf]),
        "STT_EraseHotWord": (ci, [vp, cs]),
        "STT_ClearHotWords": (ci, [vp]),
        "STT_DisableExternalScorer": (ci, [vp]),
        "STT_SetScorerAlphaBeta": (ci, [vp, cf, cf]),
        "STT_SpeechToText": (vp, [vp, vp, cu]),
        "STT_SpeechToTextWithMetadata": (pp(Metadata), [vp, vp, cu, cu]),
        "STT_SpeechToTextWithEmissions": (pp(Metadata), [vp, vp, cu, cu]),
        "STT_CreateStream": (ci, [vp, pp(vp)]),
        "STT_FeedAudioContent": (None, [vp, vp, cu]),
        "STT_IntermediateDecode": (vp, [vp]),
        "STT_IntermediateDecodeWithMetadata": (pp(Metadata), [vp, cu]),
        "STT_IntermediateDecodeFlushBuffers": (vp, [vp]),
        "STT_IntermediateDecodeWithMetadataFlushBuffers": (pp(Metadata), [vp, cu]),
        "STT_FinishStream": (vp, [vp]),
        "STT_FinishStreamWithMetadata": (pp(Metadata), [vp, cu]),
        "STT_FreeStream": (None, [vp]),
        "STT_FreeMetadata": (None, [pp(Metadata)]),
        "STT_FreeString": (None, [vp]),
        "STT_Version": (vp, []),
        "STT_ErrorCodeToErrorMessage": (vp, [ci]),
        "STTX_SetDevice": (ci, [ci]),
        "STTX_GetDeviceCount": (ci, []),
        "STTX_SpeechToTextBatch": (pp(vp), [vp, pp(vp), pp(cu), cu]),
        "STTX_SpeechToTextBatchWithMetadata": (pp(pp(Metadata)), [vp, pp(vp), pp(cu), cu, cu]),
        "STTX_SpeechToTextBatchDevice": (pp(vp), [vp, vp, cu, pp(cu), cu]),
        "STTX_BatchPipelineDepth": (ci, []),
        "STTX_BatchPipelineDepthFor": (ci, [vp]),
        "STTX_BatchSubmit": (ci, [vp, pp(vp), pp(cu), cu]),
        "STTX_BatchSubmitDevice": (ci, [vp, vp, cu, pp(cu), cu]),
        "STTX_BatchCollect": (pp(vp), [vp, ci, pp(cu)]),
        "STTX_BatchCollectWithMetadata": (pp(pp(Metadata)), [vp, ci, pp(cu)]),
        "STTX_BatchCollectScored": (pp(vp), [vp, ci, pp(cu), pp(cd)]),
        "STTX_DebugBatchProbs": (ci, [vp, ci, vp, cu, pp(cu)]),
        "STTX_SetTuning": (ci, [cs, ci]),
        "STTX_GetTuning": (ci, [cs, pp(ci)]),
        "STTX_ConfigureRuntime": (None, []),
        "STTX_TestLstmSteps": (ci, [vp, cu, cu, cu, ci, vp, vp, vp, vp, pp(cf)]),
        "STTX_TestDenseHybrid": (ci, [vp, cu, cu, vp, vp, cu, vp, cu, vp, vp, vp, cu, pp(cf), ci, cf]),
        "STTX_GetAcousticMode": (ci, [vp]),
        "STTX_DebugSlowRows": (ci, [vp, C.POINTER(C.c_uint)]),
        "STTX_TestHybridChain": (ci, [vp, vp, cu, cu, vp, vp, vp, vp, vp, vp, vp, vp, vp, pp(cu), pp(cf)]),
        "STTX_FeedAudioContentBatch": (None, [pp(vp), pp(vp), pp(cu), cu]),
        "STTX_FeedAudioContentBatchEx": (None, [pp(vp), pp(vp), pp(cu), vp, cu]),
        "STTX_IntermediateDecodeBatch": (pp(vp), [pp(vp), cu]),
        "STTX_FinishStreamBatch": (pp(vp), [pp(vp), cu]),
        "STTX_DecodeStreamsBatch": (pp(vp), [pp(vp), vp, cu]),
        "STTX_FreeStrings": (None, [pp(vp), cu]),
        "STTX_FreeMetadataArray": (None, [pp(pp(Metadata)), cu]),
        "STTX_SetProfiling": (ci, [vp, ci]),
        "STTX_GetStageTimes": (ci, [vp, pp(cf), ci]),
        "STTX_GetDecoderStats": (ci, [vp, pp(C.c_ulonglong)]),
        "STTX_GetDecoderPhaseCycles": (ci, [vp, pp(C.c_ulonglong)]),
        "STTX_GetDecoderStamps": (ci, [vp, pp(C.c_ulonglong)]),
        "STTX_ComputeMfcc": (ci, [vp, vp, cu, vp, cu, pp(cu)]),
        "STTX_AcousticProbs": (ci, [vp, pp(vp), pp(cu), cu, vp, cu, pp(cu)]),
        "STTX_InferChunk": (ci, [vp, vp, cu, vp, vp, vp, vp, vp]),
        "STTX_GetGeometry": (ci, [vp, pp(ci)]),
        "STTX_DecoderCreate": (ci, [vp, cu, cu, cd, cu, pp(vp)]),
        "STTX_DecoderNext": (ci, [vp, vp, cu, pp(cu)]),
        "STTX_DecoderDecode": (ci, [vp, cu, cu, vp, vp, vp, vp, vp]),
        "STTX_DecoderBeam": (ci, [vp, cu, vp, vp, vp, vp, cu]),
        "STTX_DecoderStats": (ci, [vp, pp(C.c_ulonglong)]),
        "STTX_DecoderErrorBits": (ci, [vp, pp(ci)]),
        "STTX_TestDictionaryWalk": (ci, [vp, cu, ci, vp, cu, cu, vp]),
        "STTX_DecoderSetProfiling": (ci, [vp, ci]),
        "STTX_DecoderGetProfile": (ci, [vp, pp(C.c_ulonglong), pp(C.c_ulonglong), pp(cf)]),
        "STTX_DecoderFree": (None, [vp]),
        "STTX_TestDense": (ci, [ci, ci, ci, vp, vp, vp, cf, ci, vp]),
        "STTX_TestMath": (ci, [ci, vp, vp, vp, cu]),
        "STTX_PackLstmRecurrent": (ci, [vp, ci, vp]),
        "STTX_InspectModel": (ci, [vp, cu, pp(ModelInfo)]),
        "STTX_ReadModelTensor": (ci, [vp, cu, ci, vp, C.c_ulonglong, pp(C.c_ulonglong)]),
        "STTX_TestLm": (ci, [vp, cu, pp(cs), cu, ci, ci, vp, vp]),
        "STTX_DebugLimitArena": (ci, [ci]),
        "STTX_FleetCreate": (ci, [cs, pp(ci), cu, pp(vp)]),
        "STTX_FleetSize": (cu, [vp]),
        "STTX_FleetEnableExternalScorer": (ci, [vp, cs]),
        "STTX_FleetSetBeamWidth": (ci, [vp, cu]),
        "STTX_FleetSpeechToTextBatch": (pp(vp), [vp, pp(vp), pp(cu), cu]),
        "STTX_FleetFree": (None, [vp]),
        "STTX_ShardUtterances": (ci, [pp(cu), cu, cu, pp(cu)]),
        "STTX_TestFleetRecords": (pp(vp), [pp(cs), pp(cu), cu, cu]),
        "STTX_DebugFleetFailShard": (ci, [vp, ci]),
    }
    for name, (res, args) in sig.items():
        if name in STT_AMD_TEST_H and not TEST_HOOKS:
            continue                      # (the shipped library does not have them; a caller that needs one gets ctypes' AttributeError)
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def take_string(ptr):
    """char* returned by the library -> bytes, released with STT_FreeString."""
    if not ptr:
        return None
    s = C.string_at(ptr)
    lib().STT_FreeString(ptr)
    return s


def lm_score(lm_bytes, words, bos=True, mode=0):
    """STTX_TestLm: KenLM FullScore over `words` -> (log10 probs f32, matched n-gram lengths).  mode 0 = hashed index on the
    host (no GPU), 1 = device trie walk, 2 = device index lookup, 3 = the code-point bigram blocks on the host, 4 = the same on the device."""
    import numpy as np
    ws = [w if isinstance(w, bytes) else w.encode() for w in words]
    arr = (C.c_char_p * len(ws))(*ws)
    probs = np.zeros(len(ws), np.float32)
    lens = np.zeros(len(ws), np.int32)
    rc = lib().STTX_TestLm(lm_bytes, len(lm_bytes), arr, len(ws), int(bos), int(mode), probs.ctypes.data, lens.ctypes.data)
    if rc != 0:
        raise RuntimeError("STTX_TestLm failed: 0x%x" % rc)
    return probs, lens


def set_tuning(name, value):
    """STTX_SetTuning: one of the engine's tunables (stt_amd/csrc/tuning.h); between calls, nothing in flight."""
    if lib().STTX_SetTuning(name.encode(), int(value)) != 0:
        raise KeyError("no tunable named %r" % name)


def get_tuning(name):
    v = C.c_int(0)
    if lib().STTX_GetTuning(name.encode(), C.byref(v)) != 0:
        raise KeyError("no tunable named %r" % name)
    return v.value


def error_message(code):
    return take_string(lib().STT_ErrorCodeToErrorMessage(code)).decode()
