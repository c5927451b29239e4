"""`.tflite` model files: the FlatBuffer reader of stt_amd/csrc/tflite_reader.cpp against files laid
out by stt_amd/tflitefile.py -- host only.  (No TensorFlow / real export exists offline: see the reader's parity note.)"""
import ctypes as C
import struct

import numpy as np
import pytest

from stt_amd import convert, modelfile, native, synth, tflitefile

VARIANTS = [
    {},                                                   # float32, fused bias, metadata behind a node
    {"quantize": True},                                   # hybrid int8, one scale per tensor (export.py:139-140)
    {"quantize": True, "per_channel": True},              # one scale per output row
    {"f16_weights": True},                                # float16 constants behind DEQUANTIZE
    {"metadata_behind_op": False, "fuse_bias": False, "legacy_opcodes": True},
]


@pytest.fixture(scope="module")
def weights():
    return synth.synth_weights(3, n_hidden=128)


@pytest.mark.parametrize("kw", VARIANTS, ids=["f32", "int8", "int8-per-channel", "f16", "legacy"])
def test_reader_returns_what_the_interpreter_would_compute_with(weights, kw):
    data, eff = tflitefile.tflite_bytes(weights, synth.ENGLISH_LABELS, n_steps=16, beam_width=777, relu_clip=17.5, **kw)
    info, t, blob = convert.read_tensors(data)
    assert info["is_tflite"] == 1
    assert (info["n_input"], info["n_context"], info["n_hidden"], info["n_classes"], info["n_steps"]) == (26, 9, 128, 29, 16)
    assert (info["sample_rate"], info["win_len"], info["win_step"], info["beam_width"]) == (16000, 512, 320, 777)   # tflitemodelstate.cc:283-287
    assert info["relu_clip"] == 17.5
    assert convert.parse_alphabet(blob) == synth.ENGLISH_LABELS
    for name in modelfile.TENSOR_ORDER:
        assert np.array_equal(t[name], np.asarray(eff[name], dtype=np.float32).reshape(t[name].shape)), name
    if kw.get("quantize"):                                # dynamic-range quantisation keeps matrices within half a step
        w = np.asarray(weights["lstm/kernel"], dtype=np.float32)
        assert 0 < np.abs(t["lstm/kernel"] - w).max() <= np.abs(w).max() / 127 * 0.5001


def test_container_and_tflite_agree(weights):
    data, eff = tflitefile.tflite_bytes(weights, synth.ENGLISH_LABELS)
    sttw = modelfile.model_bytes(weights, synth.ENGLISH_LABELS)
    i1, t1, b1 = convert.read_tensors(data)
    i2, t2, b2 = convert.read_tensors(sttw)
    assert i2["is_tflite"] == 0 and {k: v for k, v in i1.items() if k != "is_tflite"} == {k: v for k, v in i2.items() if k != "is_tflite"}
    assert b1 == b2 and all(np.array_equal(t1[n], t2[n]) for n in modelfile.TENSOR_ORDER)


def test_converter_round_trip(weights, tmp_path):
    src, dst = str(tmp_path / "m.tflite"), str(tmp_path / "m.sttw")
    eff = tflitefile.write_tflite(src, weights, synth.ENGLISH_LABELS, quantize=True, beam_width=321)
    assert convert.main([src, dst]) == 0
    want = modelfile.model_bytes({n: eff[n] for n in modelfile.TENSOR_ORDER}, synth.ENGLISH_LABELS, beam_width=321)
    assert open(dst, "rb").read() == want


def _inspect_rc(data):
    info = native.ModelInfo()
    return native.lib().STTX_InspectModel(data, len(data), C.byref(info))


def test_error_codes(weights):
    # graph version below ds_graph_version (tflitemodelstate.cc:256-264) -> STT_ERR_MODEL_INCOMPATIBLE
    data, _ = tflitefile.tflite_bytes(weights, synth.ENGLISH_LABELS, graph_version=5)
    assert _inspect_rc(data) == 0x2003
    # a FlatBuffer that is not an STT graph
    w = tflitefile._Writer().finish(tflitefile.Table(f0=("I", 3), f2=("o", tflitefile.Vec("o", [tflitefile.Table()]))))
    assert _inspect_rc(w) == 0x3002                       # STT_ERR_FAIL_INTERPRETER
    assert _inspect_rc(b"\x10\0\0\0TFL3") != 0
    assert _inspect_rc(b"garbage that is no model at all, longer than a header of sixty-four bytes ........") == 0x3005


def test_truncated_and_corrupted_files_are_rejected_not_crashed_on(weights):
    data, _ = tflitefile.tflite_bytes(weights, synth.ENGLISH_LABELS, quantize=True)
    rng = np.random.RandomState(0)
    for cut in [9, 24, 100, 1000, len(data) // 2, len(data) - 1000, len(data) - 1]:
        assert _inspect_rc(data[:cut]) != 0, cut
    bad = 0
    for trial in range(300):
        b = bytearray(data)
        for _ in range(4):
            pos = int(rng.randint(8, 60000))
            b[pos] = int(rng.randint(0, 256))
        bad += _inspect_rc(bytes(b)) != 0
    assert bad >= 0                                           # the point: no crash, no out-of-bounds read


def test_string_tensor_layout():
    blob = tflitefile._string_tensor([b"abc", b"de"])
    assert struct.unpack_from("<4i", blob) == (2, 16, 19, 21) and blob[16:] == b"abcde"


def test_asymmetric_input_quantisation_is_not_claimed(weights):
    """FullyConnectedOptions.asymmetric_quantize_inputs (converters newer than the reference's set it; export.py's tfv1 converter does not):
    TFLite would then quantise the inputs with a zero point -- an arithmetic the int8 path does not restate.  The reader sees the flag, drops
    the int8 operands (the model takes the de-quantised f16 path) and says so; the same file without the flag is a hybrid int8 model."""
    for flag in (False, True):
        data, _ = tflitefile.tflite_bytes(weights, synth.ENGLISH_LABELS, quantize=True, asymmetric_quantize_inputs=flag)
        info = native.ModelInfo()
        assert native.lib().STTX_InspectModel(data, len(data), C.byref(info)) == 0
        assert info.is_tflite == 1 and info.asymmetric_quantize_inputs == int(flag) and info.hybrid_int8 == int(not flag)
