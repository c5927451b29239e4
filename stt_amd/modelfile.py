"""stt_amd/modelfile.py -- writer of the engine's model container (format: stt_amd/csrc/model.cpp header).

The tensors are the checkpoint variables of the reference graph
(training/coqui_stt_training/deepspeech_model.py:66-75,145-163) in float32.
"""
import struct

import numpy as np

TENSOR_ORDER = ["layer_1/weights", "layer_1/bias", "layer_2/weights", "layer_2/bias", "layer_3/weights", "layer_3/bias",
                "lstm/kernel", "lstm/bias", "layer_5/weights", "layer_5/bias", "layer_6/weights", "layer_6/bias"]


def serialize_alphabet(labels):
    """Alphabet::Serialize (native_client/alphabet.cc:102-131): u16 count; {u16 key; u16 len; bytes}."""
    out = struct.pack("<H", len(labels))
    for i, l in enumerate(labels):
        out += struct.pack("<HH", i, len(l)) + l
    return out


def model_bytes(weights, labels, n_input=26, n_context=9, n_steps=16, sample_rate=16000, win_len=512, win_step=320,
                beam_width=500, relu_clip=20.0):
    n_hidden = weights["layer_1/bias"].shape[0]
    n_classes = weights["layer_6/bias"].shape[0]
    assert n_classes == len(labels) + 1
    alpha = serialize_alphabet(labels)
    hdr = b"STTAMDW1" + struct.pack("<10I", 1, n_input, n_context, n_hidden, n_classes, n_steps, sample_rate, win_len,
                                    win_step, beam_width)
    hdr += struct.pack("<fI2I", relu_clip, len(alpha), 0, 0)
    assert len(hdr) == 64
    pad = (-len(alpha)) % 8
    parts = [hdr, alpha, b"\0" * pad]
    for name in TENSOR_ORDER:
        parts.append(np.ascontiguousarray(weights[name], dtype="<f4").tobytes())
    return b"".join(parts)


def write_model(path, weights, labels, **kw):
    with open(path, "wb") as f:
        f.write(model_bytes(weights, labels, **kw))
