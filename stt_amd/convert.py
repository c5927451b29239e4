"""python -m stt_amd.convert model.tflite model.sttw -- rewrites a reference `.tflite` export as this engine's raw container.

Not needed to *use* a `.tflite` (STT_CreateModel reads it directly, stt_amd/csrc/tflite_reader.cpp); the container loads
faster (no de-quantisation / transposition at start-up) and makes the tensors inspectable.  Host only."""
import ctypes as C
import sys

import numpy as np

from . import modelfile, native

NAMES = modelfile.TENSOR_ORDER


def inspect(data):
    info = native.ModelInfo()
    rc = native.lib().STTX_InspectModel(data, len(data), C.byref(info))
    if rc != 0:
        raise RuntimeError("STTX_InspectModel failed: 0x%x" % rc)
    return {n: getattr(info, n) for n, _ in native.ModelInfo._fields_}


def read_tensors(data):
    """-> (info dict, {name: f32 array}, alphabet blob)"""
    L = native.lib()
    info = inspect(data)
    H, Cn, K1 = info["n_hidden"], info["n_classes"], info["n_input"] * (2 * info["n_context"] + 1)
    shapes = [(K1, H), (H,), (H, H), (H,), (H, H), (H,), (2 * H, 4 * H), (4 * H,), (H, H), (H,), (H, Cn), (Cn,)]
    out = {}
    for i, (name, shp) in enumerate(zip(NAMES, shapes)):
        a = np.empty(shp, dtype=np.float32)
        n = C.c_ulonglong()
        rc = L.STTX_ReadModelTensor(data, len(data), i, a.ctypes.data, a.nbytes, C.byref(n))
        if rc != 0 or n.value != a.nbytes:
            raise RuntimeError("tensor %s: rc 0x%x, %d bytes (expected %d)" % (name, rc, n.value, a.nbytes))
        out[name] = a
    blob = C.create_string_buffer(info["alphabet_bytes"])
    L.STTX_ReadModelTensor(data, len(data), 12, blob, info["alphabet_bytes"], None)
    return info, out, blob.raw


def parse_alphabet(blob):
    import struct
    n, = struct.unpack_from("<H", blob, 0)
    off, labels = 2, []
    for _ in range(n):
        _, ln = struct.unpack_from("<HH", blob, off)
        labels.append(blob[off + 4:off + 4 + ln])
        off += 4 + ln
    return labels


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print(__doc__)
        return 2
    data = open(argv[0], "rb").read()
    info, tensors, blob = read_tensors(data)
    modelfile.write_model(argv[1], tensors, parse_alphabet(blob), n_input=info["n_input"], n_context=info["n_context"],
                          n_steps=info["n_steps"], sample_rate=info["sample_rate"], win_len=info["win_len"],
                          win_step=info["win_step"], beam_width=info["beam_width"], relu_clip=info["relu_clip"])
    print("%s: n_hidden %d, %d classes, %d steps -> %s" % (argv[0], info["n_hidden"], info["n_classes"], info["n_steps"], argv[1]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
