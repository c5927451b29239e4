"""stt_amd/tflitefile.py -- writer of `.tflite` acoustic-model files with the tensor inventory of the reference's exporter.

What `training/coqui_stt_training/export.py:44-153` produces with `--export_tflite` is a TensorFlow Lite FlatBuffer of
the inference graph of `deepspeech_model.py:266-403`: inputs `input_node` [1, n_steps, 2*n_context+1, n_input],
`previous_state_c/h` [1, n_cell], `input_samples` [window]; outputs `logits`, `new_state_c/h`, `mfccs` and the
`metadata_*` constants; the LSTM unrolled n_steps times over one shared kernel (rnn_impl_static_rnn).  TensorFlow is not
available offline, so this module lays such a file out by hand (public schema tensorflow/lite/schema/schema.fbs, field
numbers as in stt_amd/csrc/tflite_reader.cpp).  It exists so that the loader can be exercised; real models come from the
reference's exporter.  `quantize=True` stores matrices the way the converter's dynamic-range quantisation does
(export.py:139-140): INT8, symmetric, one scale per tensor (or per output row with per_channel=True), zero point 0.

No FlatBuffers library: objects are laid out front to back (every offset points forward, vtables directly precede their
tables), which is a valid encoding of the format.
"""
import struct

import numpy as np

from .modelfile import serialize_alphabet

FLOAT32, FLOAT16, INT32, UINT8, INT64, STRING, INT8 = 0, 1, 2, 3, 4, 5, 9
ADD, CONCATENATION, DEQUANTIZE, FULLY_CONNECTED, LOGISTIC, MUL, RELU, RESHAPE, SOFTMAX, TANH, CUSTOM, SPLIT, MINIMUM = \
    0, 2, 6, 9, 14, 18, 19, 22, 25, 28, 32, 49, 57


class Table:
    """fields: {id: (fmt, value)}; fmt a struct code for scalars, 'o' for a child object (Table, Vec, Str or None)."""

    def __init__(self, **fields):
        self.fields = {int(k[1:]): v for k, v in fields.items() if v is not None and v[1] is not None}


class Vec:
    def __init__(self, fmt, items, align=4):
        self.fmt, self.items, self.align = fmt, items, align   # fmt 'o': vector of offsets; else scalar code / 'raw' bytes


class Str:
    def __init__(self, s):
        self.s = s if isinstance(s, bytes) else s.encode()


class _Writer:
    def __init__(self):
        self.buf = bytearray()
        self.fix = []   # (position of the u32 offset field, child object)

    def pad(self, align, bias=0):
        while (len(self.buf) + bias) % align:
            self.buf.append(0)

    def emit(self, obj):
        if isinstance(obj, Str):
            self.pad(4)
            pos = len(self.buf)
            self.buf += struct.pack("<I", len(obj.s)) + obj.s + b"\0"
            return pos
        if isinstance(obj, Vec):
            if obj.fmt == "raw":
                data = bytes(obj.items)
                self.pad(max(4, obj.align), bias=4)        # the elements (after the length) carry the alignment
                pos = len(self.buf)
                self.buf += struct.pack("<I", len(data)) + data
                return pos
            if obj.fmt == "o":
                self.pad(4)
                pos = len(self.buf)
                self.buf += struct.pack("<I", len(obj.items))
                for it in obj.items:
                    self.fix.append((len(self.buf), it))
                    self.buf += b"\0\0\0\0"
                return pos
            size = struct.calcsize("<" + obj.fmt)
            self.pad(max(4, size), bias=4)
            pos = len(self.buf)
            self.buf += struct.pack("<I", len(obj.items)) + struct.pack("<%d%s" % (len(obj.items), obj.fmt), *obj.items)
            return pos
        # table: vtable, then the table (soffset + fields by decreasing size)
        ids = sorted(obj.fields)
        sized = sorted(ids, key=lambda i: -(4 if obj.fields[i][0] == "o" else struct.calcsize("<" + obj.fields[i][0])))
        off, lay = 4, {}
        for i in sized:
            sz = 4 if obj.fields[i][0] == "o" else struct.calcsize("<" + obj.fields[i][0])
            off = (off + sz - 1) // sz * sz
            lay[i] = off
            off += sz
        tsize = off
        nslots = (max(ids) + 1) if ids else 0
        vsize = 4 + 2 * nslots
        self.pad(8, bias=vsize)                            # the table starts 8-aligned, its vtable directly before it
        vpos = len(self.buf)
        self.buf += struct.pack("<HH", vsize, tsize) + b"".join(struct.pack("<H", lay.get(i, 0)) for i in range(nslots))
        tpos = len(self.buf)
        self.buf += struct.pack("<i", tpos - vpos) + b"\0" * (tsize - 4)
        for i in ids:
            fmt, val = obj.fields[i]
            if fmt == "o":
                self.fix.append((tpos + lay[i], val))
            else:
                struct.pack_into("<" + fmt, self.buf, tpos + lay[i], val)
        return tpos

    def finish(self, root, ident=b"TFL3"):
        self.buf += b"\0" * 8
        self.fix.append((0, root))
        k = 0
        while k < len(self.fix):                           # breadth first: children always land after their referrers
            at, obj = self.fix[k]
            k += 1
            pos = self.emit(obj)
            struct.pack_into("<I", self.buf, at, pos - at)
        self.buf[4:8] = ident
        return bytes(self.buf)


def _string_tensor(items):
    """TFLite string tensor payload: i32 count, i32 offsets[count+1], bytes."""
    head = 4 * (len(items) + 2)
    offs, cur = [], head
    for it in items:
        offs.append(cur)
        cur += len(it)
    offs.append(cur)
    return struct.pack("<i%di" % len(offs), len(items), *offs) + b"".join(items)


def tflite_bytes(weights, labels, n_input=26, n_context=9, n_steps=16, sample_rate=16000, win_len_ms=32, win_step_ms=20,
                 beam_width=500, relu_clip=20.0, graph_version=6, quantize=False, per_channel=False, f16_weights=False,
                 metadata_behind_op=True, fuse_bias=True, legacy_opcodes=False, asymmetric_quantize_inputs=False):
    """Serialises the inference graph with `weights` (checkpoint orientation, stt_amd/modelfile.py names).
    Returns (bytes, effective_weights): effective_weights are the f32 values a TFLite interpreter would compute with
    (de-quantised when quantize / f16_weights)."""
    H = weights["layer_1/bias"].shape[0]
    C = weights["layer_6/bias"].shape[0]
    win = 2 * n_context + 1
    window_samples = int(sample_rate * (win_len_ms / 1000.0))
    tensors, buffers, ops, opcodes = [], [Table()], [], []
    eff = {}

    def buf(data):
        buffers.append(Table(f0=("o", Vec("raw", data, align=16))))
        return len(buffers) - 1

    def tensor(name, shape, ttype=FLOAT32, data=None, quant=None):
        q = None
        if quant is not None:
            scale, qdim = quant
            q = Table(f2=("o", Vec("f", [float(x) for x in scale])), f3=("o", Vec("q", [0] * len(scale))),
                      f6=("i", qdim) if qdim else None)
        tensors.append(Table(f0=("o", Vec("i", list(shape))), f1=("b", ttype) if ttype else None,
                             f2=("I", buf(data)) if data is not None else None, f3=("o", Str(name)), f4=("o", q)))
        return len(tensors) - 1

    def opcode(code, custom=None):
        key = (code, custom)
        if key not in opcodes:
            opcodes.append(key)
        return opcodes.index(key)

    def op(code, ins, outs, custom=None):
        opts = {}
        if code == FULLY_CONNECTED and asymmetric_quantize_inputs:      # BuiltinOptions 8 = FullyConnectedOptions, field 3 = asymmetric_quantize_inputs
            opts = dict(f3=("B", 8), f4=("o", Table(f3=("B", 1))))
        ops.append(Table(f0=("I", opcode(code, custom)), f1=("o", Vec("i", ins)), f2=("o", Vec("i", outs)), **opts))

    def const_matrix(name, w_in_out):
        w = np.ascontiguousarray(np.asarray(w_in_out, dtype=np.float32).T)          # FULLY_CONNECTED weights are [out][in]
        if quantize and w.size >= 1024:                                              # the converter leaves small tensors in float
            amax = np.abs(w).max(axis=1) if per_channel else np.array([np.abs(w).max()])
            scale = (np.maximum(amax, 1e-30) / 127.0).astype(np.float32)
            # tensor_utils::SymmetricQuantizeFloats, as the converter's quantize_weights pass calls it: q = round(w * (127 / range)), halves away from zero
            inv = (np.float32(127.0) / np.maximum(amax, 1e-30).astype(np.float32)).astype(np.float32)
            t = (w * (inv[:, None] if per_channel else inv[0])).astype(np.float32)
            q = np.clip(np.sign(t) * np.floor(np.abs(t).astype(np.float64) + 0.5), -127, 127).astype(np.int8)
            eff_w = (q.astype(np.float32) * (scale[:, None] if per_channel else scale[0])).astype(np.float32)
            return tensor(name, w.shape, INT8, q.tobytes(), quant=(scale, 0)), eff_w.T
        if f16_weights:
            h = w.astype(np.float16)
            src = tensor(name + "_f16", w.shape, FLOAT16, h.tobytes())
            dst = tensor(name + "_dequantized", w.shape)
            op(DEQUANTIZE, [src], [dst])
            return dst, h.astype(np.float32).T
        return tensor(name, w.shape, FLOAT32, w.tobytes()), w.T

    def dense(x, name_w, name_b, rows, n_out, out_name, clip=True):
        wt, eff_w = const_matrix(name_w + "/transpose", weights[name_w])
        eff[name_w] = eff_w
        eff[name_b] = np.asarray(weights[name_b], dtype=np.float32)
        bt = tensor(name_b, [n_out], FLOAT32, eff[name_b].tobytes())
        y = tensor(out_name + "/MatMul", [rows, n_out])
        if fuse_bias:
            op(FULLY_CONNECTED, [x, wt, bt], [y])
        else:
            y0 = tensor(out_name + "/MatMul_nobias", [rows, n_out])
            op(FULLY_CONNECTED, [x, wt, -1], [y0])
            op(ADD, [y0, bt], [y])
        if not clip:
            return y
        r = tensor(out_name + "/Relu", [rows, n_out])
        op(RELU, [y], [r])
        m = tensor(out_name + "/Minimum", [rows, n_out])
        op(MINIMUM, [r, clip_t], [m])
        return m

    # ---- interface tensors
    t_samples = tensor("input_samples", [window_samples])
    t_input = tensor("input_node", [1, n_steps, win, n_input])
    t_c = tensor("previous_state_c", [1, H])
    t_h = tensor("previous_state_h", [1, H])
    clip_t = tensor("Minimum/y", [], FLOAT32, struct.pack("<f", relu_clip))
    # features: AudioSpectrogram + Mfcc are custom operators with built-in TFLite kernels (export.py:142-143)
    t_spec = tensor("AudioSpectrogram", [1, 1, 257])
    op(CUSTOM, [t_samples], [t_spec], custom="AudioSpectrogram")
    t_rate = tensor("Mfcc/sample_rate", [], INT32, struct.pack("<i", sample_rate))
    t_mfcc = tensor("mfccs", [1, n_input])
    op(CUSTOM, [t_spec, t_rate], [t_mfcc], custom="Mfcc")
    # layers 1-3 over all n_steps rows at once
    t_shape = tensor("Reshape/shape", [2], INT32, struct.pack("<2i", n_steps, win * n_input))
    x = tensor("Reshape", [n_steps, win * n_input])
    op(RESHAPE, [t_input, t_shape], [x])
    x = dense(x, "layer_1/weights", "layer_1/bias", n_steps, H, "layer_1")
    x = dense(x, "layer_2/weights", "layer_2/bias", n_steps, H, "layer_2")
    x = dense(x, "layer_3/weights", "layer_3/bias", n_steps, H, "layer_3")
    # unrolled LSTM: concat([x_t, h]) . kernel + bias -> i, j, f, o (deepspeech_model.py:143-168)
    kt, eff_k = const_matrix("cudnn_lstm/rnn/multi_rnn_cell/cell_0/cudnn_compatible_lstm_cell/kernel/transpose", weights["lstm/kernel"])
    eff["lstm/kernel"] = eff_k
    eff["lstm/bias"] = np.asarray(weights["lstm/bias"], dtype=np.float32)
    bt = tensor("cudnn_lstm/rnn/multi_rnn_cell/cell_0/cudnn_compatible_lstm_cell/bias", [4 * H], FLOAT32, eff["lstm/bias"].tobytes())
    t_axis0 = tensor("split/split_dim", [], INT32, struct.pack("<i", 0))
    t_axis1 = tensor("lstm/split_dim", [], INT32, struct.pack("<i", 1))
    rows = [tensor("unstack:%d" % t, [1, H]) for t in range(n_steps)]
    op(SPLIT, [t_axis0, x], rows)
    c, h, outs = t_c, t_h, []
    for t in range(n_steps):
        pre = "cell_0/step_%d/" % t
        xc = tensor(pre + "concat", [1, 2 * H]); op(CONCATENATION, [rows[t], h], [xc])
        z = tensor(pre + "BiasAdd", [1, 4 * H]); op(FULLY_CONNECTED, [xc, kt, bt], [z])
        gi, gj, gf, go = [tensor(pre + "split:%d" % k, [1, H]) for k in range(4)]
        op(SPLIT, [t_axis1, z], [gi, gj, gf, go])
        si = tensor(pre + "Sigmoid", [1, H]); op(LOGISTIC, [gi], [si])
        tj = tensor(pre + "Tanh", [1, H]); op(TANH, [gj], [tj])
        sf = tensor(pre + "Sigmoid_1", [1, H]); op(LOGISTIC, [gf], [sf])
        so = tensor(pre + "Sigmoid_2", [1, H]); op(LOGISTIC, [go], [so])
        m1 = tensor(pre + "mul", [1, H]); op(MUL, [sf, c], [m1])
        m2 = tensor(pre + "mul_1", [1, H]); op(MUL, [si, tj], [m2])
        last = t == n_steps - 1
        c = tensor("new_state_c" if last else pre + "add_1", [1, H]); op(ADD, [m1, m2], [c])
        tc = tensor(pre + "Tanh_1", [1, H]); op(TANH, [c], [tc])
        h = tensor("new_state_h" if last else pre + "mul_2", [1, H]); op(MUL, [so, tc], [h])
        outs.append(h)
    x = tensor("concat", [n_steps, H]); op(CONCATENATION, outs, [x])
    x = dense(x, "layer_5/weights", "layer_5/bias", n_steps, H, "layer_5")
    x = dense(x, "layer_6/weights", "layer_6/bias", n_steps, C, "layer_6", clip=False)
    t_logits = tensor("logits", [n_steps, C]); op(SOFTMAX, [x], [t_logits])
    # metadata constants (export.py:57-77)
    meta = []
    for name, ttype, payload, shape in (
            ("metadata_version", INT32, struct.pack("<i", graph_version), [1]),
            ("metadata_sample_rate", INT32, struct.pack("<i", sample_rate), [1]),
            ("metadata_feature_win_len", INT32, struct.pack("<i", win_len_ms), [1]),
            ("metadata_feature_win_step", INT32, struct.pack("<i", win_step_ms), [1]),
            ("metadata_beam_width", INT32, struct.pack("<i", beam_width), [1]),
            ("metadata_alphabet", STRING, _string_tensor([serialize_alphabet(labels)]), [1])):
        if metadata_behind_op:      # released models give every metadata output a producing node (tflitemodelstate.cc:235-241)
            src = tensor(name + "/value", shape, ttype, payload)
            shp = tensor(name + "/shape", [1], INT32, struct.pack("<i", 1))
            dst = tensor(name, shape, ttype)
            op(RESHAPE, [src, shp], [dst])
            meta.append(dst)
        else:
            meta.append(tensor(name, shape, ttype, payload))
    sub = Table(f0=("o", Vec("o", tensors)), f1=("o", Vec("i", [t_input, t_c, t_h, t_samples])),
                f2=("o", Vec("i", [t_logits, c, h, t_mfcc] + meta)), f3=("o", Vec("o", ops)), f4=("o", Str("main")))
    codes = []
    for code, custom in opcodes:
        if legacy_opcodes:          # files written before the int32 builtin_code field existed
            codes.append(Table(f0=("b", code), f1=("o", Str(custom) if custom else None)))
        else:
            codes.append(Table(f0=("b", min(code, 127)), f1=("o", Str(custom) if custom else None), f2=("i", 1), f3=("i", code)))
    model = Table(f0=("I", 3), f1=("o", Vec("o", codes)), f2=("o", Vec("o", [sub])),
                  f3=("o", Str("stt_amd.tflitefile (layout of coqui_stt_training.export)")), f4=("o", Vec("o", buffers)))
    return _Writer().finish(model), eff


def write_tflite(path, weights, labels, **kw):
    data, eff = tflite_bytes(weights, labels, **kw)
    with open(path, "wb") as f:
        f.write(data)
    return eff
