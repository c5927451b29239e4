// stt_amd/csrc/model.cpp -- model container reader and weight preparation for HBM.
//
// Replaces TFLiteModelState::init (native_client/tflitemodelstate.cc:161-338): instead of building a
// TFLite interpreter, the weights of the exported graph (training/coqui_stt_training/deepspeech_model.py:
// 171-263, variable names SURVEY.md A.4) are converted once to the layouts the kernels stream:
// f16, transposed to [N][K] for the dense layers, fragment-packed for the recurrent matrix.
//
// Container ("STTAMDW1", little endian; written by stt_amd/modelfile.py):
//   char magic[8]; u32 version(=1), n_input, n_context, n_hidden, n_classes, n_steps, sample_rate,
//   win_len, win_step, beam_width; f32 relu_clip; u32 alphabet_bytes; u32 reserved[2]          (64 bytes)
//   alphabet blob (Alphabet::Serialize format, alphabet.cc:102-131), zero padded to 8 bytes
//   f32 tensors, row-major, in this order:
//     layer_1/weights [n_input*(2*n_context+1)][H], layer_1/bias [H], layer_2/weights [H][H], layer_2/bias,
//     layer_3/weights [H][H], layer_3/bias, lstm/kernel [2H][4H] (rows x then h; columns i|j|f|o), lstm/bias [4H],
//     layer_5/weights [H][H], layer_5/bias, layer_6/weights [H][C], layer_6/bias [C]
// `.tflite` exports of the reference (float or hybrid int8) are recognised by their file identifier and read by
// tflite_reader.cpp into the same twelve tensors.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/coqui-stt.h"
#include "engine.h"

namespace {
struct SttwHeader {
  char magic[8];
  uint32_t version, n_input, n_context, n_hidden, n_classes, n_steps, sample_rate, win_len, win_step, beam_width;
  float relu_clip;
  uint32_t alphabet_bytes;
  uint32_t reserved[2];
};
static_assert(sizeof(SttwHeader) == 64, "header is 64 bytes");

// W [K][N] f32 -> WT [N_pad][K_pad] f16 (zero padded)
std::vector<_Float16> transpose_f16(const float* w, int K, int N, int K_pad, int N_pad, int ldw) {
  std::vector<_Float16> out((size_t)N_pad * K_pad, (_Float16)0.0f);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) out[(size_t)n * K_pad + k] = (_Float16)w[(size_t)k * ldw + n];
  return out;
}
}  // namespace

int n_frames_for(const Geometry& g, int n_samples) {  // stt.cc:105-128 + flushBuffers :236-254
  const int full = n_samples >= g.win_len ? (n_samples - g.win_len) / g.win_step + 1 : 0;
  return full + 1;
}

// Recurrent half of the LSTM kernel, packed for lstm_step_kernel:
//   out[((wg*4 + q)*ksteps + s)*2 + mt][lane][e]  with  ksteps = H/128,
//   row r = mt*16 + (lane&15): gate g = r>>3, unit u = wg*8 + (r&7)  -> column n = g*H + u
//   k = q*(H/4) + s*32 + (lane>>4)*8 + e                             -> kernel[H + k][n]
void pack_lstm_recurrent_host(const float* kernel, int H, _Float16* out) {
  const int ksteps = H / 128;
  const size_t ld = (size_t)4 * H;
  const int UPW = lstm_units_per_wg(H), MT = UPW / 4;  // units and 16-row gate tiles per workgroup (kernels_am.hip: lstm_step_kernel)
  for (int wg = 0; wg < H / UPW; ++wg)
    for (int q = 0; q < 4; ++q)
      for (int s = 0; s < ksteps; ++s)
        for (int mt = 0; mt < MT; ++mt)
          for (int lane = 0; lane < 64; ++lane) {
            const int r = mt * 16 + (lane & 15);
            const int n = (r / UPW) * H + wg * UPW + (r % UPW);
            const int k0 = q * (H / 4) + s * 32 + (lane >> 4) * 8;
            _Float16* o = out + ((((size_t)(wg * 4 + q) * ksteps + s) * MT + mt) * 64 + lane) * 8;
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)kernel[(size_t)(H + k0 + e) * ld + n];
          }
}

// Recurrent half of the int8 kernel ([4H][2H] as the file holds it: rows = gate columns i|j|f|o, columns = [x(H); h(H)]), packed for
// lstm_i8_step_kernel (kernels_i8.hip): 16 hidden units x 4 gates per workgroup, k-steps of 64,
//   out[(((wg*KS + s)*4 + mt)*64 + lane)*16 + e] = kernel_q[mt*H + wg*16 + (lane & 15)][H + s*64 + (lane >> 4)*16 + e],  KS = H / 64
void pack_lstm_recurrent_i8_host(const int8_t* kq, int H, int8_t* out) {
  const int KS = H / 64;
  for (int wg = 0; wg < H / 16; ++wg)
    for (int s = 0; s < KS; ++s)
      for (int mt = 0; mt < 4; ++mt)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = mt * H + wg * 16 + (lane & 15);
          const int k0 = s * 64 + (lane >> 4) * 16;
          memcpy(out + ((((size_t)wg * KS + s) * 4 + mt) * 64 + lane) * 16, kq + (size_t)n * 2 * H + H + k0, 16);
        }
}

void quantize_weights_host(const float* w, int n_in, int n_out, std::vector<int8_t>& q, std::vector<float>& scale) {
  float amax = 0.0f;
  for (size_t i = 0; i < (size_t)n_in * n_out; ++i) amax = std::max(amax, std::fabs(w[i]));
  // the converter's weight quantisation is tensor_utils::SymmetricQuantizeFloats (tools/optimize/quantize_weights.cc -> SymmetricQuantizeTensor):
  // scale = range / 127, q = TfLiteRound(w * (127 / range)) -- a multiply by the inverse, halves away from zero; oracle/am_hybrid.py and
  // stt_amd/tflitefile.py write the same form
  const float sc = std::max(amax, 1e-30f) / 127.0f;
  const float inv = 127.0f / std::max(amax, 1e-30f);
  scale.assign(1, sc);
  q.resize((size_t)n_in * n_out);
  for (int k = 0; k < n_in; ++k)
    for (int n = 0; n < n_out; ++n) {
      const float r = std::roundf(w[(size_t)k * n_out + n] * inv);
      q[(size_t)n * n_in + k] = (int8_t)std::min(127.0f, std::max(-127.0f, r));
    }
}

ModelState::~ModelState() {
  tuning_model_count(-1);
  for (StreamingState* s : stream_pool_) delete s;
  if (stream) (void)hipStreamDestroy(stream);
  if (stream_dec) (void)hipStreamDestroy(stream_dec);
  for (hipStream_t s : decoder_streams_) if (s) (void)hipStreamDestroy(s);
  for (auto& e : ev_chunk) if (e) (void)hipEventDestroy(e);
  for (auto& kv : lstm_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  for (auto& kv : hop_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  if (stream_h2d) (void)hipStreamDestroy(stream_h2d);
  for (auto& st : stage_) if (st.copied) (void)hipEventDestroy(st.copied);
  for (auto& e : ev_watch) if (e) (void)hipEventDestroy(e);
  if (stream_l) (void)hipStreamDestroy(stream_l);
  if (stream_o) (void)hipStreamDestroy(stream_o);
  for (int i = 0; i < kAmRing; ++i)
    for (hipEvent_t e : {ev_x_ready[i], ev_x_free[i], ev_h_ready[i], ev_h_free[i]}) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_audio) if (e) (void)hipEventDestroy(e);
  for (auto& sl : slots_) if (sl.done) (void)hipEventDestroy(sl.done);
  for (int i = 1; i < kSlots; ++i) if (slots_[i].stream_dec) (void)hipStreamDestroy(slots_[i].stream_dec);
}

int parse_model_file(const char* buf, size_t len, ModelTensors& tfl, ModelView& v, std::string& err) {
  Geometry& g = v.g;
  if (looks_like_tflite(buf, len)) {
    const int rc = read_tflite_model(buf, len, tfl, err);
    if (rc != STT_ERR_OK) return rc;
    g = tfl.g;
    v.alphabet = tfl.alphabet.data(); v.alphabet_bytes = tfl.alphabet.size();
    const std::vector<float>* src[12] = {&tfl.l1w, &tfl.l1b, &tfl.l2w, &tfl.l2b, &tfl.l3w, &tfl.l3b, &tfl.lk, &tfl.lb, &tfl.l5w, &tfl.l5b, &tfl.l6w, &tfl.l6b};
    for (int i = 0; i < 12; ++i) { v.t[i] = src[i]->data(); v.count[i] = src[i]->size(); }
    v.quant = tfl.all_int8() ? &tfl : nullptr;
  } else {
    if (len < sizeof(SttwHeader)) { err = "model file too short"; return STT_ERR_FAIL_READ_PROTOBUF; }
    SttwHeader h;
    memcpy(&h, buf, sizeof(h));
    if (memcmp(h.magic, "STTAMDW1", 8) != 0) { err = "unknown model file format (neither STTAMDW1 nor TFL3)"; return STT_ERR_FAIL_READ_PROTOBUF; }
    if (h.version != 1) { err = "unsupported container version"; return STT_ERR_MODEL_INCOMPATIBLE; }
    g.n_input = h.n_input; g.n_context = h.n_context; g.n_hidden = h.n_hidden; g.n_classes = h.n_classes; g.n_steps = h.n_steps;
    g.sample_rate = h.sample_rate; g.win_len = h.win_len; g.win_step = h.win_step; g.beam_width = h.beam_width; g.relu_clip = h.relu_clip;
    size_t off = sizeof(SttwHeader);
    if (off + h.alphabet_bytes > len) { err = "alphabet runs past the end of the file"; return STT_ERR_INVALID_ALPHABET; }
    v.alphabet = buf + off; v.alphabet_bytes = h.alphabet_bytes;
    off += (h.alphabet_bytes + 7) & ~(size_t)7;
    const size_t H = g.n_hidden, C = g.n_classes, K1 = g.n_in1();
    const size_t cnt[12] = {K1 * H, H, H * H, H, H * H, H, 2 * H * 4 * H, 4 * H, H * H, H, H * C, C};
    size_t n_f32 = 0;
    for (size_t c : cnt) n_f32 += c;
    if (H == 0 || H > 65536 || C > 65536 || off + n_f32 * 4 > len) { err = "tensor data runs past the end of the file"; return STT_ERR_INVALID_SHAPE; }
    const float* f = reinterpret_cast<const float*>(buf + off);
    for (int i = 0; i < 12; ++i) { v.t[i] = f; v.count[i] = cnt[i]; f += cnt[i]; }
  }
  const int H = g.n_hidden, C = g.n_classes;
  // The feature kernel takes FFT lengths of 128 .. 2048 (TF's AudioSpectrogram: fft_length = NextPowerOfTwo(window), util/feeding.py:51-73
  // -> spectrogram.cc): windows of 65 .. 2048 samples -- 32 ms at 8, 16, 22.05, 32, 44.1 or 48 kHz (tflitemodelstate.cc:287-307 takes any
  // window; one workgroup holds at most 1024 butterflies, so windows beyond 2048 samples are refused rather than silently framed
  // differently).  The DCT has 40 mel inputs, so at most 40 coefficients exist.
  if (H % 128 != 0 || H < 128 || C < 2 || C > STT_MAX_CLASSES || g.win_len > 2048 || g.win_len <= 64 || g.win_step < 1 || g.sample_rate < 1000 || g.n_input > 40 || g.n_input < 1 ||
      g.n_steps < 1 || g.n_context < 0 || g.beam_width < 1) {
    err = "model geometry outside what the engine supports";
    return STT_ERR_INVALID_SHAPE;
  }
  return STT_ERR_OK;
}

int ModelState::InitFromBuffer(const char* buf, size_t len) {
  ModelTensors storage;
  ModelView v;
  std::string err;
  const int rc = parse_model_file(buf, len, storage, v, err);
  if (rc != STT_ERR_OK) { fprintf(stderr, "%s\n", err.c_str()); return rc; }
  g = v.g;
  if (alphabet_.Deserialize(v.alphabet, (int)v.alphabet_bytes) != 0) return STT_ERR_INVALID_ALPHABET;
  if ((int)alphabet_.GetSize() + 1 != g.n_classes) {  // tflitemodelstate.cc:319-329
    fprintf(stderr, "Error: Alphabet size does not match loaded model: alphabet has size %d, but model has %d classes in its output. "
                    "Make sure you're passing an alphabet file with the same size as the one used for training.\n",
            (int)alphabet_.GetSize(), g.n_classes - 1);
    return STT_ERR_INVALID_ALPHABET;
  }
  if (g.n_classes > STT_MAX_CLASSES) {
    fprintf(stderr, "Error: %d output classes; the beam search handles alphabets of up to %d labels.\n", g.n_classes, STT_MAX_CLASSES - 1);
    return STT_ERR_INVALID_ALPHABET;
  }
  const float *l1w = v.t[0], *l1b = v.t[1], *l2w = v.t[2], *l2b = v.t[3], *l3w = v.t[4], *l3b = v.t[5], *lk = v.t[6], *lb = v.t[7],
              *l5w = v.t[8], *l5b = v.t[9], *l6w = v.t[10], *l6b = v.t[11];
  beam_width_ = g.beam_width;
  const int H = g.n_hidden, C = g.n_classes, K1 = g.n_in1();

  HIP_CHECK(hipSetDevice(device));
  if (!stream) create_engine_stream(&stream, 0);
  if (!stream_dec) create_engine_stream(&stream_dec, 3);
  {
    auto t = transpose_f16(l1w, K1, H, g.k1_pad(), H, H); w1t.upload(t.data(), t.size() * 2, stream);
    t = transpose_f16(l2w, H, H, H, H, H); w2t.upload(t.data(), t.size() * 2, stream);
    t = transpose_f16(l3w, H, H, H, H, H); w3t.upload(t.data(), t.size() * 2, stream);
    t = transpose_f16(lk, H, 4 * H, H, 4 * H, 4 * H); wxt.upload(t.data(), t.size() * 2, stream);  // x rows of the kernel
    t = transpose_f16(l5w, H, H, H, H, H); w5t.upload(t.data(), t.size() * 2, stream);
    t = transpose_f16(l6w, H, C, H, g.c_pad(), C); w6t.upload(t.data(), t.size() * 2, stream);
    std::vector<_Float16> packed((size_t)H * 4 * H);
    pack_lstm_recurrent_host(lk, H, packed.data());
    whp.upload(packed.data(), packed.size() * 2, stream);
    b1.upload(l1b, H * 4, stream); b2.upload(l2b, H * 4, stream); b3.upload(l3b, H * 4, stream);
    bl.upload(lb, 4 * H * 4, stream); b5.upload(l5b, H * 4, stream);
    std::vector<float> b6p(g.c_pad(), 0.0f);
    memcpy(b6p.data(), l6b, C * 4);
    b6.upload(b6p.data(), b6p.size() * 4, stream);
  }
  // ---- the released models' own arithmetic: the int8 matrices as the file holds them (or quantised here as the converter would)
  {
    const int want = tune().am_i8;
    i8 = (want > 0 || (want < 0 && v.quant != nullptr)) && H % 256 == 0 && H <= 4096;
    if ((want > 0 || (want < 0 && v.quant)) && !i8)
      fprintf(stderr, "Note: n_hidden = %d is outside the int8 path's shapes (multiples of 256 up to 4096): int8 weights are de-quantised to f16.\n", H);
  }
  if (i8) {
    std::vector<int8_t> q[6];
    std::vector<float> qs[6];
    const float* wf[6] = {l1w, l2w, l3w, lk, l5w, l6w};
    const int kin[6] = {K1, H, H, 2 * H, H, H}, nout[6] = {H, H, H, 4 * H, H, C};
    for (int l = 0; l < 6; ++l) {
      if (v.quant) { q[l] = v.quant->wq[l]; qs[l] = v.quant->wq_scale[l]; }
      else quantize_weights_host(wf[l], kin[l], nout[l], q[l], qs[l]);
      sn[l] = (int)qs[l].size();
    }
    auto up_rows = [&](DevBuf& dst, const int8_t* src, int N, int K, int ld, int Np, int Kp) {   // [N][K] (row stride ld) -> [Np][Kp], zero padded
      std::vector<int8_t> t((size_t)Np * Kp, 0);
      for (int n = 0; n < N; ++n) memcpy(t.data() + (size_t)n * Kp, src + (size_t)n * ld, (size_t)K);
      dst.upload(t.data(), t.size(), stream);
    };
    up_rows(w1q, q[0].data(), H, K1, K1, H, g.k1_pad8());
    up_rows(w2q, q[1].data(), H, H, H, H, H);
    up_rows(w3q, q[2].data(), H, H, H, H, H);
    up_rows(wxq, q[3].data(), 4 * H, H, 2 * H, 4 * H, H);
    up_rows(whq, q[3].data() + H, 4 * H, H, 2 * H, 4 * H, H);
    up_rows(w5q, q[4].data(), H, H, H, H, H);
    up_rows(w6q, q[5].data(), C, H, H, g.c_pad8(), H);
    {
      std::vector<int8_t> packed((size_t)4 * H * H);
      pack_lstm_recurrent_i8_host(q[3].data(), H, packed.data());
      whpq.upload(packed.data(), packed.size(), stream);
    }
    DevBuf* sd[6] = {&s1, &s2, &s3, &sk, &s5, &s6};
    for (int l = 0; l < 6; ++l) {
      std::vector<float> sc = qs[l];
      if (l == 5 && sc.size() > 1) sc.resize(g.c_pad8(), 1.0f);     // (padded output rows: zero weights, any scale)
      sd[l]->upload(sc.data(), sc.size() * 4, stream);
    }
    std::vector<float> b6p(g.c_pad8(), 0.0f);
    memcpy(b6p.data(), l6b, C * 4);
    b6q.upload(b6p.data(), b6p.size() * 4, stream);
  }
  // ---- feature tables (oracle/am_ref.py MfccSpec; upstream tensorflow spectrogram.cc / mfcc_mel_filterbank.cc / mfcc_dct.cc)
  {
    const int n_mel = 40, nfft = g.fft_len(), n_bins = nfft / 2 + 1;
    std::vector<double> window(g.win_len);
    for (int i = 0; i < g.win_len; ++i) window[i] = 0.5 - 0.5 * cos(2.0 * M_PI * i / g.win_len);
    std::vector<double2> tw(nfft / 2);
    for (int m = 0; m < nfft / 2; ++m) tw[m] = make_double2(cos(2.0 * M_PI * m / nfft), -sin(2.0 * M_PI * m / nfft));
    auto mel = [](double fq) { return 1127.0 * log1p(fq / 700.0); };
    const double lower = 20.0, upper = g.sample_rate / 2.0;
    const double mel_low = mel(lower), mel_hi = mel(upper), spacing = (mel_hi - mel_low) / (n_mel + 1);
    std::vector<double> center(n_mel + 1);
    for (int i = 0; i <= n_mel; ++i) center[i] = mel_low + spacing * (i + 1);
    const double hz_per_sbin = 0.5 * g.sample_rate / (n_bins - 1);
    const int start_index = (int)(1.5 + lower / hz_per_sbin), end_index = (int)(upper / hz_per_sbin);
    std::vector<int> mapper(n_bins, -2);
    std::vector<double> wts(n_bins, 0.0);
    int channel = 0;
    for (int b = 0; b < n_bins; ++b) {
      const double melf = mel(b * hz_per_sbin);
      if (b < start_index || b > end_index) { mapper[b] = -2; continue; }
      while (channel < n_mel && center[channel] < melf) ++channel;
      mapper[b] = channel - 1;
    }
    for (int b = 0; b < n_bins; ++b) {
      const int ch = mapper[b];
      if (b < start_index || b > end_index) wts[b] = 0.0;
      else if (ch >= 0) wts[b] = (center[ch + 1] - mel(b * hz_per_sbin)) / (center[ch + 1] - center[ch]);
      else wts[b] = (center[0] - mel(b * hz_per_sbin)) / (center[0] - mel_low);
    }
    // band c receives (amp - amp*w) from bins mapped to c-1 and amp*w from bins mapped to c; both are contiguous bin ranges
    std::vector<int> idx(4 * n_mel, 0);
    for (int c = 0; c < n_mel; ++c) {
      int lo_b = -1, lo_e = -1, hi_b = -1, hi_e = -1;
      for (int b = start_index; b <= end_index && b < n_bins; ++b) {
        if (mapper[b] == c - 1) { if (lo_b < 0) lo_b = b; lo_e = b + 1; }
        if (mapper[b] == c) { if (hi_b < 0) hi_b = b; hi_e = b + 1; }
      }
      idx[c] = lo_b < 0 ? 0 : lo_b; idx[n_mel + c] = lo_b < 0 ? 0 : lo_e;
      idx[2 * n_mel + c] = hi_b < 0 ? 0 : hi_b; idx[3 * n_mel + c] = hi_b < 0 ? 0 : hi_e;
    }
    std::vector<double> dct((size_t)g.n_input * n_mel);
    const double fnorm = sqrt(2.0 / n_mel), arg = M_PI / n_mel;
    for (int i = 0; i < g.n_input; ++i)
      for (int j = 0; j < n_mel; ++j) dct[(size_t)i * n_mel + j] = fnorm * cos(i * arg * (j + 0.5));
    t_window.upload(window.data(), window.size() * 8, stream);
    t_twiddle.upload(tw.data(), tw.size() * sizeof(double2), stream);
    t_melw.upload(wts.data(), wts.size() * 8, stream);
    t_mel_idx.upload(idx.data(), idx.size() * 4, stream);
    t_dct.upload(dct.data(), dct.size() * 8, stream);
  }
  // ---- alphabet label bytes for on-device word hashing
  {
    std::string bytes;
    std::vector<int> offs;
    for (const auto& l : alphabet_.labels()) { bytes += l; offs.push_back((int)bytes.size()); }
    bytes.push_back('\0');
    al_bytes.upload(bytes.data(), bytes.size(), stream);
    al_off.upload(offs.data(), offs.size() * 4, stream);
    dev_alphabet.n_labels = (int)alphabet_.GetSize();
    dev_alphabet.space_id = alphabet_.GetSpaceLabel();
    dev_alphabet.label_bytes = al_bytes.as<uint8_t>();
    dev_alphabet.label_off = al_off.as<int>();
    dev_alphabet.byte_labels = 1;
    for (size_t c = 0; c < alphabet_.labels().size(); ++c) {
      const std::string& l = alphabet_.labels()[c];
      if (l.size() != 1 || (unsigned char)l[0] != (unsigned char)(c + 1)) { dev_alphabet.byte_labels = 0; break; }
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  return STT_ERR_OK;
}

MfccArgs ModelState::mfcc_args() const {
  MfccArgs a{};
  a.win_len = g.win_len; a.win_step = g.win_step; a.n_coef = g.n_input; a.n_mel = 40; a.fft_len = g.fft_len();
  a.window = t_window.as<double>(); a.twiddle = t_twiddle.as<double2>(); a.mel_w = t_melw.as<double>();
  const int* idx = t_mel_idx.as<int>();
  a.mel_lo_begin = idx; a.mel_lo_end = idx + 40; a.mel_hi_begin = idx + 80; a.mel_hi_end = idx + 120;
  a.dct = t_dct.as<double>();
  return a;
}
