// stt_amd/csrc/tflite_reader.cpp -- reads the reference's exported `.tflite` acoustic model without TensorFlow Lite.
//
// Replaces what TFLiteModelState::init (native_client/tflitemodelstate.cc:161-338) gets from the TFLite interpreter:
// the interface tensors by name (input_node, previous_state_c, ... :211-226), the metadata constants
// (training/coqui_stt_training/export.py:57-77) and -- because this engine runs its own kernels instead of interpreting
// the graph -- the weights of the six matrix products of deepspeech_model.py:171-263.
//
// The file is a FlatBuffer of the public TensorFlow Lite schema (tensorflow/lite/schema/schema.fbs, file identifier
// "TFL3"; upstream TensorFlow is an un-vendored submodule of the reference, .gitmodules:5-7, so the schema is restated
// here from its published form -- field numbers below).  Only the parts of the schema the path needs are read:
//
//   Model      : 0 version, 1 operator_codes, 2 subgraphs, 3 description, 4 buffers
//   SubGraph   : 0 tensors, 1 inputs, 2 outputs, 3 operators, 4 name
//   Tensor     : 0 shape, 1 type (byte), 2 buffer, 3 name, 4 quantization
//   Quantization: 0 min, 1 max, 2 scale, 3 zero_point, 6 quantized_dimension
//   OperatorCode: 0 deprecated_builtin_code (byte), 1 custom_code, 2 version, 3 builtin_code (int32)
//   Operator   : 0 opcode_index, 1 inputs, 2 outputs
//   Buffer     : 0 data, 1 offset, 2 size      (offset/size: data stored outside the FlatBuffer, same file)
//
// Weights are recognised by graph position, not by tensor name (converter versions rename constants): the distinct weight
// operands of the FULLY_CONNECTED operators, in execution order, are layer 1, 2, 3, the LSTM kernel (shared by the
// n_steps unrolled cells of rnn_impl_static_rnn, deepspeech_model.py:143-168), layer 5 and layer 6; their shapes must be
// [H,K1] [H,H] [H,H] [4H,2H] [H,H] [C,H].  Hybrid ("dynamic range") models store these as INT8/UINT8 with per-tensor or
// per-channel scales (export.py:139-140 `converter.optimizations`); FLOAT16 constants behind DEQUANTIZE are followed too.
//
// Parity note: no `.tflite` file, TFLite runtime or schema compiler exists offline, so this reader is exercised against
// files written by stt_amd/tflitefile.py (same schema restatement) -- "unpinned" against real exports, see DESIGN.md.
#include <cstdint>
#include <cstring>
#include <map>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/coqui-stt.h"
#include "engine.h"

namespace {

enum : int { T_FLOAT32 = 0, T_FLOAT16 = 1, T_INT32 = 2, T_UINT8 = 3, T_INT64 = 4, T_STRING = 5, T_INT8 = 9 };
enum : int { OP_ADD = 0, OP_DEQUANTIZE = 6, OP_FULLY_CONNECTED = 9, OP_RESHAPE = 22, OP_MINIMUM = 57, OP_RELU_N1_TO_1 = 20, OP_RELU6 = 21 };

struct Fb {  // bounds-checked FlatBuffer access; any violation clears `ok` and yields zeros
  const uint8_t* b; size_t n; bool ok = true;
  template <typename T> T rd(size_t o) {
    T v{};
    if (o > n || n - o < sizeof(T)) { ok = false; return v; }
    memcpy(&v, b + o, sizeof(T));
    return v;
  }
  size_t field(size_t t, int id) {  // absolute position of field `id` of the table at t, 0 if absent
    if (!t) return 0;
    const int32_t so = rd<int32_t>(t);
    const int64_t vt = (int64_t)t - so;
    if (vt < 0 || (size_t)vt + 4 > n) { ok = false; return 0; }
    const uint16_t vsz = rd<uint16_t>((size_t)vt);
    const size_t slot = 4 + 2 * (size_t)id;
    if (slot + 2 > vsz) return 0;
    const uint16_t fo = rd<uint16_t>((size_t)vt + slot);
    return fo ? t + fo : 0;
  }
  size_t ref(size_t p) {  // follow a uoffset stored at p
    if (!p) return 0;
    const uint32_t o = rd<uint32_t>(p);
    if (!o || p + o >= n) { ok = false; return 0; }
    return p + o;
  }
  size_t sub(size_t t, int id) { return ref(field(t, id)); }          // table / vector / string field
  template <typename T> T scalar(size_t t, int id, T dflt) { const size_t p = field(t, id); return p ? rd<T>(p) : dflt; }
  uint32_t vlen(size_t v) { return v ? rd<uint32_t>(v) : 0; }
  size_t velem(size_t v, uint32_t i, size_t esz) { return v + 4 + (size_t)i * esz; }
  size_t vtable_elem(size_t v, uint32_t i) { return ref(velem(v, i, 4)); }
  std::string str(size_t s) {
    const uint32_t l = vlen(s);
    if (!s || s + 4 + (size_t)l > n) { if (s) ok = false; return std::string(); }
    return std::string(reinterpret_cast<const char*>(b + s + 4), l);
  }
  std::vector<int32_t> ints(size_t v) {
    std::vector<int32_t> out(vlen(v));
    for (uint32_t i = 0; i < out.size() && ok; ++i) out[i] = rd<int32_t>(velem(v, i, 4));
    return out;
  }
};

struct TensorInfo {
  std::vector<int32_t> shape;
  int type = 0;
  uint32_t buffer = 0;
  std::string name;
  std::vector<float> scale;
  std::vector<int64_t> zero_point;
  int qdim = 0;
  const uint8_t* data = nullptr;  // constant payload, if any
  size_t nbytes = 0;
  size_t count() const { size_t c = 1; for (int32_t d : shape) c *= (size_t)(d > 0 ? d : 0); return c; }
};
struct OpInfo { int code = -1; std::string custom; std::vector<int32_t> in, out; bool asym_inputs = false; };  // asym_inputs: FullyConnectedOptions.asymmetric_quantize_inputs

float half_to_float(uint16_t h) {
  _Float16 v;
  memcpy(&v, &h, 2);
  return (float)v;
}

struct Graph {
  std::vector<TensorInfo> tensors;
  std::vector<OpInfo> ops;
  std::vector<int32_t> inputs, outputs;
  std::vector<int> producer;  // tensor -> op index, -1 = none

  int by_name(const std::vector<int32_t>& set, const char* name) const {  // tflitemodelstate.cc:24-50: exact name match
    for (int32_t t : set)
      if (t >= 0 && (size_t)t < tensors.size() && tensors[t].name == name) return t;
    return -1;
  }
  // constant tensor behind shape-only / dequantising operators (the metadata outputs have a parent node in released
  // models, tflitemodelstate.cc:235-241; f16 weights sit behind DEQUANTIZE)
  int resolve_const(int t) const {
    for (int hop = 0; hop < 6 && t >= 0 && (size_t)t < tensors.size(); ++hop) {
      if (tensors[t].data) return t;
      const int op = producer[t];
      if (op < 0 || ops[op].in.empty()) return -1;
      t = ops[op].in[0];
    }
    return -1;
  }
  // element (flat index i) of a constant numeric tensor as float, dequantised
  bool to_float(int t, std::vector<float>& out, std::string& err) const {
    const TensorInfo& ti = tensors[t];
    const size_t cnt = ti.count();
    out.resize(cnt);
    const size_t esz = ti.type == T_FLOAT32 || ti.type == T_INT32 ? 4 : ti.type == T_FLOAT16 ? 2 : (ti.type == T_INT8 || ti.type == T_UINT8) ? 1 : 0;
    if (!esz || ti.nbytes < cnt * esz) { err = "tensor '" + ti.name + "': unsupported type or short buffer"; return false; }
    if (ti.type == T_FLOAT32) { memcpy(out.data(), ti.data, cnt * 4); return true; }
    if (ti.type == T_FLOAT16) {
      for (size_t i = 0; i < cnt; ++i) { uint16_t h; memcpy(&h, ti.data + 2 * i, 2); out[i] = half_to_float(h); }
      return true;
    }
    if (ti.type == T_INT32) { for (size_t i = 0; i < cnt; ++i) { int32_t v; memcpy(&v, ti.data + 4 * i, 4); out[i] = (float)v; } return true; }
    if (ti.scale.empty()) { err = "tensor '" + ti.name + "': quantised without scale"; return false; }
    // per-tensor, or per-slice along quantized_dimension
    size_t inner = 1;
    const int rank = (int)ti.shape.size();
    const int qd = ti.scale.size() > 1 ? ti.qdim : -1;
    if (qd >= 0) {
      if (qd >= rank || (size_t)ti.shape[qd] != ti.scale.size()) { err = "tensor '" + ti.name + "': per-channel scales do not match the shape"; return false; }
      for (int d = qd + 1; d < rank; ++d) inner *= (size_t)ti.shape[d];
    }
    for (size_t i = 0; i < cnt; ++i) {
      const size_t ch = qd >= 0 ? (i / inner) % ti.scale.size() : 0;
      const int64_t zp = ti.zero_point.empty() ? 0 : ti.zero_point[ti.zero_point.size() > 1 ? ch : 0];
      const int q = ti.type == T_INT8 ? (int)(int8_t)ti.data[i] : (int)ti.data[i];
      out[i] = ti.scale[ch] * (float)(q - (int)zp);
    }
    return true;
  }
};

bool parse_graph(Fb& fb, Graph& gr, std::string& err) {
  if (fb.n < 8) { err = "file too short"; return false; }
  const uint32_t root = fb.rd<uint32_t>(0);
  const size_t model = (root >= 8 && root < fb.n) ? (size_t)root : 0;
  const size_t sgs = fb.sub(model, 2);
  if (!fb.ok || !model || fb.vlen(sgs) < 1) { err = "no subgraph"; return false; }
  const size_t sg = fb.vtable_elem(sgs, 0);
  // buffers
  const size_t bufs = fb.sub(model, 4);
  const uint32_t n_buf = fb.vlen(bufs);
  std::vector<std::pair<const uint8_t*, size_t>> bdata(n_buf, {nullptr, 0});
  for (uint32_t i = 0; i < n_buf && fb.ok; ++i) {
    const size_t bt = fb.vtable_elem(bufs, i);
    const size_t d = fb.sub(bt, 0);
    if (d) {
      const uint32_t l = fb.vlen(d);
      if (d + 4 + (size_t)l > fb.n) { fb.ok = false; break; }
      if (l) bdata[i] = {fb.b + d + 4, l};
    } else {
      const uint64_t off = fb.scalar<uint64_t>(bt, 1, 0), sz = fb.scalar<uint64_t>(bt, 2, 0);
      if (off > 1 && sz && off <= fb.n && sz <= fb.n - off) bdata[i] = {fb.b + off, (size_t)sz};
    }
  }
  // operator codes
  const size_t ocs = fb.sub(model, 1);
  std::vector<std::pair<int, std::string>> codes(fb.vlen(ocs));
  for (uint32_t i = 0; i < codes.size() && fb.ok; ++i) {
    const size_t oc = fb.vtable_elem(ocs, i);
    const int dep = fb.scalar<int8_t>(oc, 0, 0), full = fb.scalar<int32_t>(oc, 3, 0);
    codes[i] = {full > dep ? full : dep, fb.str(fb.sub(oc, 1))};  // schema: the larger of the two is the operator
  }
  // tensors
  const size_t ts = fb.sub(sg, 0);
  gr.tensors.resize(fb.vlen(ts));
  for (uint32_t i = 0; i < gr.tensors.size() && fb.ok; ++i) {
    const size_t t = fb.vtable_elem(ts, i);
    TensorInfo& ti = gr.tensors[i];
    ti.shape = fb.ints(fb.sub(t, 0));
    ti.type = fb.scalar<int8_t>(t, 1, 0);
    ti.buffer = fb.scalar<uint32_t>(t, 2, 0);
    ti.name = fb.str(fb.sub(t, 3));
    const size_t q = fb.sub(t, 4);
    if (q) {
      const size_t sc = fb.sub(q, 2), zp = fb.sub(q, 3);
      ti.scale.resize(fb.vlen(sc));
      for (uint32_t k = 0; k < ti.scale.size() && fb.ok; ++k) ti.scale[k] = fb.rd<float>(fb.velem(sc, k, 4));
      ti.zero_point.resize(fb.vlen(zp));
      for (uint32_t k = 0; k < ti.zero_point.size() && fb.ok; ++k) ti.zero_point[k] = fb.rd<int64_t>(fb.velem(zp, k, 8));
      ti.qdim = fb.scalar<int32_t>(q, 6, 0);
    }
    if (ti.buffer > 0 && ti.buffer < n_buf) { ti.data = bdata[ti.buffer].first; ti.nbytes = bdata[ti.buffer].second; }  // buffer 0 = "no data"
  }
  gr.inputs = fb.ints(fb.sub(sg, 1));
  gr.outputs = fb.ints(fb.sub(sg, 2));
  const size_t ops = fb.sub(sg, 3);
  gr.ops.resize(fb.vlen(ops));
  gr.producer.assign(gr.tensors.size(), -1);
  for (uint32_t i = 0; i < gr.ops.size() && fb.ok; ++i) {
    const size_t o = fb.vtable_elem(ops, i);
    const uint32_t ci = fb.scalar<uint32_t>(o, 0, 0);
    if (ci < codes.size()) { gr.ops[i].code = codes[ci].first; gr.ops[i].custom = codes[ci].second; }
    gr.ops[i].in = fb.ints(fb.sub(o, 1));
    gr.ops[i].out = fb.ints(fb.sub(o, 2));
    // schema: Operator.builtin_options_type (field 3) = 8 for FullyConnectedOptions (field 4), whose field 3 is asymmetric_quantize_inputs --
    // the runtime then quantises the op's inputs with a zero point instead of symmetrically (converters newer than the reference's set it)
    if (gr.ops[i].code == OP_FULLY_CONNECTED && fb.scalar<uint8_t>(o, 3, 0) == 8) gr.ops[i].asym_inputs = fb.scalar<uint8_t>(fb.sub(o, 4), 3, 0) != 0;
    for (int32_t t : gr.ops[i].out)
      if (t >= 0 && (size_t)t < gr.tensors.size()) gr.producer[t] = (int)i;
  }
  if (!fb.ok) { err = "malformed FlatBuffer"; return false; }
  return true;
}

// [out][in] -> [in][out] (the container's orientation: checkpoint variables, rows = inputs)
void transpose_oi(const std::vector<float>& w, int out, int in, std::vector<float>& dst) {
  dst.resize((size_t)out * in);
  for (int o = 0; o < out; ++o)
    for (int i = 0; i < in; ++i) dst[(size_t)i * out + o] = w[(size_t)o * in + i];
}

}  // namespace

bool looks_like_tflite(const char* buf, size_t len) { return len >= 8 && memcmp(buf + 4, "TFL3", 4) == 0; }

int read_tflite_model(const char* buf, size_t len, ModelTensors& m, std::string& err) {
  Fb fb{reinterpret_cast<const uint8_t*>(buf), len};
  Graph gr;
  if (!parse_graph(fb, gr, err)) return STT_ERR_FAIL_INTERPRETER;

  // ---- interface tensors (tflitemodelstate.cc:211-226)
  const int t_in = gr.by_name(gr.inputs, "input_node"), t_c = gr.by_name(gr.inputs, "previous_state_c"),
            t_h = gr.by_name(gr.inputs, "previous_state_h"), t_samples = gr.by_name(gr.inputs, "input_samples"),
            t_logits = gr.by_name(gr.outputs, "logits");
  if (t_in < 0 || t_c < 0 || t_h < 0 || t_samples < 0 || t_logits < 0 || gr.by_name(gr.outputs, "new_state_c") < 0 ||
      gr.by_name(gr.outputs, "new_state_h") < 0 || gr.by_name(gr.outputs, "mfccs") < 0) {
    err = "not an STT acoustic model: interface tensors (input_node, previous_state_c/h, input_samples, logits, new_state_c/h, mfccs) missing";
    return STT_ERR_FAIL_INTERPRETER;
  }
  auto meta_i32 = [&](const char* name, int& v) {
    const int t = gr.resolve_const(gr.by_name(gr.outputs, name));
    if (t < 0 || gr.tensors[t].type != T_INT32 || gr.tensors[t].nbytes < 4) return false;
    int32_t x; memcpy(&x, gr.tensors[t].data, 4); v = x;
    return true;
  };
  int version = 0, sr = 0, win_len_ms = 0, win_step_ms = 0, beam = 0;
  if (!meta_i32("metadata_version", version)) { err = "Unable to read model file version."; return STT_ERR_MODEL_INCOMPATIBLE; }
  if (version < 6) {  // ds_graph_version(), native_client/ds_graph_version.h / training GRAPH_VERSION
    err = "Specified model file version (" + std::to_string(version) + ") is incompatible with minimum version supported by this client (6).";
    return STT_ERR_MODEL_INCOMPATIBLE;
  }
  if (!meta_i32("metadata_sample_rate", sr)) { err = "Unable to read model sample rate."; return STT_ERR_MODEL_INCOMPATIBLE; }
  if (!meta_i32("metadata_feature_win_len", win_len_ms) || !meta_i32("metadata_feature_win_step", win_step_ms)) {
    err = "Unable to read model feature window informations."; return STT_ERR_MODEL_INCOMPATIBLE;
  }
  if (!meta_i32("metadata_beam_width", beam)) { err = "Unable to read model beam width."; return STT_ERR_MODEL_INCOMPATIBLE; }
  {  // string tensor: i32 count, i32 offsets[count+1], bytes (tflite string_util; GetString(tensor, 0), :296)
    const int t = gr.resolve_const(gr.by_name(gr.outputs, "metadata_alphabet"));
    if (t < 0 || gr.tensors[t].type != T_STRING || gr.tensors[t].nbytes < 12) { err = "Unable to read model alphabet."; return STT_ERR_INVALID_ALPHABET; }
    const uint8_t* d = gr.tensors[t].data;
    int32_t cnt, o0, o1;
    memcpy(&cnt, d, 4); memcpy(&o0, d + 4, 4); memcpy(&o1, d + 8, 4);
    if (cnt < 1 || o0 < 0 || o1 < o0 || (size_t)o1 > gr.tensors[t].nbytes) { err = "Unable to read model alphabet."; return STT_ERR_INVALID_ALPHABET; }
    m.alphabet.assign(reinterpret_cast<const char*>(d + o0), (size_t)(o1 - o0));
  }
  const auto& s_in = gr.tensors[t_in].shape;
  const auto& s_c = gr.tensors[t_c].shape;
  const auto& s_lg = gr.tensors[t_logits].shape;
  if (s_in.size() != 4 || s_c.size() != 2 || s_lg.size() != 2 || gr.tensors[t_h].shape != s_c) { err = "unexpected interface tensor ranks"; return STT_ERR_INVALID_SHAPE; }
  Geometry& g = m.g;
  g.n_steps = s_in[1]; g.n_context = (s_in[2] - 1) / 2; g.n_input = s_in[3];   // :306-309
  g.n_hidden = s_c[1]; g.n_classes = s_lg[1];
  g.sample_rate = sr; g.beam_width = beam;
  g.win_len = (int)(sr * (win_len_ms / 1000.0)); g.win_step = (int)(sr * (win_step_ms / 1000.0));  // :283-284
  if ((size_t)g.win_len != gr.tensors[t_samples].count()) { err = "input_samples does not hold one feature window"; return STT_ERR_INVALID_SHAPE; }
  if (g.n_hidden < 1 || g.n_hidden > 16384 || g.n_classes < 2 || g.n_classes > 16384 || g.n_steps < 1 || g.n_steps > 4096 || s_in[2] < 1 ||
      s_in[2] > 255 || g.n_input < 1 || g.n_input > 1024) {
    err = "interface tensor shapes out of range"; return STT_ERR_INVALID_SHAPE;
  }
  const int H = g.n_hidden, C = g.n_classes, K1 = g.n_in1();

  // ---- the matrix products, in execution order
  struct Fc { int w, b, out; };
  std::vector<Fc> fcs;
  bool asym = false;
  for (size_t i = 0; i < gr.ops.size(); ++i) {
    const OpInfo& op = gr.ops[i];
    if (op.code != OP_FULLY_CONNECTED || op.in.size() < 2 || op.out.empty()) continue;
    asym = asym || op.asym_inputs;
    const int w = gr.resolve_const(op.in[1]);
    if (w < 0) { err = "FULLY_CONNECTED with non-constant weights"; return STT_ERR_MODEL_INCOMPATIBLE; }
    bool seen = false;
    for (const Fc& f : fcs) seen |= f.w == w || (gr.tensors[f.w].data == gr.tensors[w].data && gr.tensors[f.w].shape == gr.tensors[w].shape);
    if (seen) continue;
    int b = op.in.size() > 2 && op.in[2] >= 0 ? gr.resolve_const(op.in[2]) : -1;
    if (b < 0) {  // bias not fused: an ADD of the product with a constant vector
      for (const OpInfo& a : gr.ops)
        if (a.code == OP_ADD && a.in.size() == 2 && (a.in[0] == op.out[0] || a.in[1] == op.out[0])) {
          b = gr.resolve_const(a.in[0] == op.out[0] ? a.in[1] : a.in[0]);
          break;
        }
    }
    fcs.push_back({w, b, op.out[0]});
  }
  const int want[6][2] = {{H, K1}, {H, H}, {H, H}, {4 * H, 2 * H}, {H, H}, {C, H}};
  if (fcs.size() != 6) { err = "expected 6 distinct weight matrices, found " + std::to_string(fcs.size()); return STT_ERR_MODEL_INCOMPATIBLE; }
  std::vector<float>* wdst[6] = {&m.l1w, &m.l2w, &m.l3w, &m.lk, &m.l5w, &m.l6w};
  std::vector<float>* bdst[6] = {&m.l1b, &m.l2b, &m.l3b, &m.lb, &m.l5b, &m.l6b};
  for (int l = 0; l < 6; ++l) {
    const TensorInfo& wt = gr.tensors[fcs[l].w];
    if (wt.shape.size() != 2 || wt.shape[0] != want[l][0] || wt.shape[1] != want[l][1]) {
      err = "weight matrix " + std::to_string(l + 1) + " ('" + wt.name + "') has an unexpected shape"; return STT_ERR_INVALID_SHAPE;
    }
    std::vector<float> w;
    if (!gr.to_float(fcs[l].w, w, err)) return STT_ERR_MODEL_INCOMPATIBLE;
    transpose_oi(w, want[l][0], want[l][1], *wdst[l]);
    // dynamic-range quantised weights as the file holds them (symmetric int8, per tensor or per output row): the hybrid path's operands
    bool zp0 = true;
    for (int64_t z : wt.zero_point) zp0 = zp0 && z == 0;
    if (wt.type == T_INT8 && zp0 && (wt.scale.size() == 1 || (wt.scale.size() == (size_t)want[l][0] && wt.qdim == 0))) {
      m.wq[l].assign(reinterpret_cast<const int8_t*>(wt.data), reinterpret_cast<const int8_t*>(wt.data) + (size_t)want[l][0] * want[l][1]);
      m.wq_scale[l] = wt.scale;
    }
    if (fcs[l].b >= 0) {
      if (!gr.to_float(fcs[l].b, *bdst[l], err)) return STT_ERR_MODEL_INCOMPATIBLE;
      if ((int)bdst[l]->size() != want[l][0]) { err = "bias " + std::to_string(l + 1) + " has an unexpected size"; return STT_ERR_INVALID_SHAPE; }
    } else bdst[l]->assign((size_t)want[l][0], 0.0f);
  }
  if (asym && m.all_int8()) {
    // The engine's int8 path is TFLite's SYMMETRIC hybrid kernel (what export.py's tfv1 converter produces).  A re-exported file that asks for
    // asymmetric input quantisation would run a different arithmetic in TFLite: do not claim it -- take the de-quantised f16 path and say so.
    std::cerr << "stt_amd: FULLY_CONNECTED with asymmetric_quantize_inputs: the int8 weights are de-quantised and the f16 path is taken "
                 "(the hybrid int8 path restates the symmetric kernel only)" << std::endl;
    for (int l = 0; l < 6; ++l) { m.wq[l].clear(); m.wq_scale[l].clear(); }
    m.asymmetric_inputs = true;
  }
  // ---- ReLU clip (deepspeech_model.py:80-82 `minimum(relu(x), relu_clip)`): the constant operand of the first MINIMUM
  g.relu_clip = 20.0f;
  for (const OpInfo& op : gr.ops)
    if (op.code == OP_MINIMUM && op.in.size() == 2) {
      for (int k = 0; k < 2; ++k) {
        const int t = gr.resolve_const(op.in[k]);
        if (t >= 0 && gr.tensors[t].type == T_FLOAT32 && gr.tensors[t].nbytes >= 4 && gr.tensors[t].count() == 1) { memcpy(&g.relu_clip, gr.tensors[t].data, 4); break; }
      }
      break;
    } else if (op.code == OP_RELU6) { g.relu_clip = 6.0f; break; }
  return STT_ERR_OK;
}
