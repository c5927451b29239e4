// stt_amd/csrc/engine.cpp -- pipeline orchestration on one HIP stream: features -> acoustic model -> decoder.
//
// Replaces TFLiteModelState::{compute_mfcc,infer} (tflitemodelstate.cc:369-436), ModelState::decode*
// (modelstate.cc:32-76) and StreamingState (stt.cc:60-334).  Nothing here computes on the CPU: host code
// only sizes buffers, enqueues kernels and turns token ids into strings.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>

#include "../../include/coqui-stt.h"
#include "engine.h"
#include "tuning.h"

uint64_t stt_murmur64a(const void* key, size_t len);
int g_debug_arena_frames = 0;  // STTX_DebugLimitArena

// ------------------------------------------------------------------------------------------- features
void ModelState::run_mfcc(const int16_t* d_audio, const int* h_nsamples, int B, int n_max, int t_max, std::vector<int>& n_frames) {
  n_frames.resize(B);
  for (int b = 0; b < B; ++b) n_frames[b] = n_frames_for(g, h_nsamples[b]);
  ws_nsamp.reserve(B * 4); ws_nframes.reserve(B * 4);
  HIP_CHECK(hipMemcpyAsync(ws_nsamp.p, h_nsamples, B * 4, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipMemcpyAsync(ws_nframes.p, n_frames.data(), B * 4, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipStreamSynchronize(stream));  // n_frames is a caller-owned vector; keep the copy simple and safe
  ws_feats.reserve((size_t)B * t_max * g.n_input * 4);
  MfccArgs a = mfcc_args();
  a.audio = d_audio; a.n_samples = ws_nsamp.as<int>(); a.n_frames = ws_nframes.as<int>();
  a.feats = ws_feats.as<float>(); a.n_max = n_max; a.t_max = t_max;
  launch_mfcc(a, B * t_max, stream);
}

// ------------------------------------------------------------------------------------------- acoustic model
// rows: x1 [T*B][k1_pad] f16, row = t*B + b.  B <= 64.
// carry: 0 = zero state, 1 = state from the f32 vectors d_c/d_h (streaming), 2 = continue from the engine's own
// buffers (next time-chunk of the same batch; `t_par` = number of steps already run, for the h ping-pong parity).
// The same chain in the released models' own arithmetic (ModelState::i8; oracle/am_hybrid.py: HybridModel.forward_batch): every
// FULLY_CONNECTED as TFLite's hybrid kernel -- the f32 input rows quantised to int8 one by one (launch_quantize_rows), int32 sums on
// v_mfma_i32_16x16x64_i8, rescale + bias (+ clipped ReLU) in f32 -- and the cell with the joint [x_t, h_(t-1)] row scale (kernels_i8.hip).
// x1: f32 [T*B][k1_pad8], row = t*B + b.  carry as acoustic_rows(); the carried h lives in f32 (d_h, or ws_hlast between the chunks of a batch).
static void acoustic_rows_i8(ModelState& m, const float* d_x1, int B, int T, float* d_c, float* d_h, int carry, float* d_probs_out, int probs_t_max,
                             const int* d_nframes = nullptr, int t0 = 0) {
  const Geometry& g = m.g;
  hipStream_t st = m.stream;
  const int H = g.n_hidden, M = T * B, C = g.n_classes, K1 = g.k1_pad8(), CP = g.c_pad8();
  const int NT = lstm_nt_for_batch(B);
  if (NT < 0) throw std::runtime_error("run_acoustic_rows: more batch rows than one recurrent launch covers");
  const int NTR = NT * 16, NWG = H / 16;
  m.ws_a.reserve((size_t)M * H * 4); m.ws_b.reserve((size_t)M * H * 4);
  m.ws_xproj.reserve((size_t)M * 4 * H * 4); m.ws_hall.reserve((size_t)M * H * 4);
  m.ws_logits.reserve((size_t)M * CP * 4);
  m.q_x.reserve((size_t)M * std::max(K1, H)); m.q_s.reserve((size_t)M * 4); m.q_rng.reserve((size_t)M * 8);
  const size_t hq_bytes = lstm_i8_hq_bytes(H, NT);
  m.ws_hq0.reserve(hq_bytes); m.ws_hq1.reserve(hq_bytes);
  m.ws_hprev0.reserve((size_t)B * H * 4); m.ws_hlast.reserve((size_t)B * H * 4);
  m.ws_pmax.reserve((size_t)2 * NTR * NWG * 4); m.ws_flag.reserve((size_t)2 * NTR * 4); m.ws_zslow.reserve((size_t)NWG * NTR * 64 * 4);
  m.ws_c.reserve((size_t)B * H * 4);
  if (!m.ws_slow.p) { m.ws_slow.reserve(4); HIP_CHECK(hipMemsetAsync(m.ws_slow.p, 0, 4, st)); }
  signed char* qx = m.q_x.as<signed char>();
  float* qs = m.q_s.as<float>();
  float *xs3 = m.q_rng.as<float>(), *xr3 = xs3 + M;       // layer 3's rows: scaling factor and range (max |x_t|)
  float *act_a = m.ws_a.as<float>(), *act_b = m.ws_b.as<float>();
  stt_prof_mark(&m, 1);
  // layers 1-3 (deepspeech_model.py:204-224)
  launch_quantize_rows(d_x1, qx, qs, M, K1, st);
  launch_dense_hybrid_i8(qx, qs, m.w1q.as<signed char>(), m.s1.as<float>(), m.sn[0], m.b1.as<float>(), act_a, M, H, K1, st, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(act_a, qx, qs, M, H, st);
  launch_dense_hybrid_i8(qx, qs, m.w2q.as<signed char>(), m.s2.as<float>(), m.sn[1], m.b2.as<float>(), act_b, M, H, H, st, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(act_b, qx, qs, M, H, st);
  launch_dense_hybrid_i8(qx, qs, m.w3q.as<signed char>(), m.s3.as<float>(), m.sn[2], m.b3.as<float>(), act_a, M, H, H, st, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  // x half of the cell's int32 sums for all timesteps at once, from layer 3's rows quantised at their own scale
  launch_quantize_rows(act_a, qx, xs3, M, H, st, xr3);
  launch_dense_hybrid_i8(qx, xs3, m.wxq.as<signed char>(), m.sk.as<float>(), m.sn[3], m.bl.as<float>(), m.ws_xproj.p, M, 4 * H, H, st, DENSE_EPI_I8_RAW);
  stt_prof_mark(&m, 2);
  // recurrence
  float* cbuf = (carry != 2 && d_c) ? d_c : m.ws_c.as<float>();
  if (carry == 0 || (carry == 1 && !d_c)) HIP_CHECK(hipMemsetAsync(cbuf, 0, (size_t)B * H * 4, st));
  const float* h_src = carry == 1 ? d_h : (carry == 2 ? m.ws_hlast.as<float>() : nullptr);
  LstmI8Args l{};
  l.whp = m.whpq.as<signed char>(); l.accx = m.ws_xproj.as<int>(); l.bias = m.bl.as<float>(); l.wscale = m.sk.as<float>(); l.wscale_n = m.sn[3];
  l.xscale = xs3; l.xrange = xr3; l.c = cbuf; l.h_all = m.ws_hall.as<float>();
  l.h_last = d_h ? d_h : m.ws_hlast.as<float>();     // (a stream's first chunk starts from zeros, carry 0, and still hands its state back)
  l.pmax = m.ws_pmax.as<float>(); l.flag = m.ws_flag.as<int>(); l.y3 = act_a; l.h_prev0 = m.ws_hprev0.as<float>();
  l.wxq = m.wxq.as<signed char>(); l.whq = m.whq.as<signed char>(); l.zslow = m.ws_zslow.as<float>();
  l.n_hidden = H; l.batch = B; l.T = T; l.slow_count = m.ws_slow.as<unsigned>();
  l.probe = m.dbg_ev_[0] ? tune().lstm_probe : 0;     // (only a timed test-hook call may ask for a probe kernel)
  l.row_frames = d_nframes; l.t0 = t0;                // (batch path: rows past their utterance's end are never worth the slow path)
  l.hq_in = m.ws_hq0.as<signed char>(); l.hq_out = m.ws_hq1.as<signed char>();
  if (m.dbg_ev_[0]) HIP_CHECK(hipEventRecord(m.dbg_ev_[0], st));
  auto steps = [&]() {
    launch_lstm_i8_prep(l, h_src, NT, st);
    for (int t = 0; t < T; ++t) {
      l.t = t;
      l.hq_in = (t & 1) ? m.ws_hq1.as<signed char>() : m.ws_hq0.as<signed char>();
      l.hq_out = (t & 1) ? m.ws_hq0.as<signed char>() : m.ws_hq1.as<signed char>();
      launch_lstm_i8_step(l, NT, st, tune().lstm_i8_rows);
    }
  };
  steps();   // (launch by launch: replaying these chunks as hipGraphs on the shared acoustic stream measured SLOWER -- 22.6 against 20.0 us per step)
  if (m.dbg_ev_[1]) HIP_CHECK(hipEventRecord(m.dbg_ev_[1], st));
  stt_prof_mark(&m, 3);
  // layer 5, layer 6, softmax (deepspeech_model.py:241-252, 357)
  launch_quantize_rows(m.ws_hall.as<float>(), qx, qs, M, H, st);
  launch_dense_hybrid_i8(qx, qs, m.w5q.as<signed char>(), m.s5.as<float>(), m.sn[4], m.b5.as<float>(), act_b, M, H, H, st, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(act_b, qx, qs, M, H, st);
  launch_dense_hybrid_i8(qx, qs, m.w6q.as<signed char>(), m.s6.as<float>(), m.sn[5], m.b6q.as<float>(), m.ws_logits.p, M, CP, H, st, DENSE_EPI_I8_F32);
  SoftmaxArgs s{};
  s.logits = m.ws_logits.as<float>(); s.probs = d_probs_out; s.M = M; s.C = C; s.ldl = CP; s.batch = B; s.t_max = probs_t_max; s.exact = 1;
  launch_softmax(s, st);
}

static void acoustic_rows(ModelState& m, const void* d_x1v, int B, int T, float* d_c, float* d_h, int carry, int t_par,
                          float* d_probs_out, int probs_t_max, const int* d_nframes = nullptr) {
  if (m.i8) { acoustic_rows_i8(m, static_cast<const float*>(d_x1v), B, T, d_c, d_h, carry, d_probs_out, probs_t_max, d_nframes, t_par); return; }
  const _Float16* d_x1 = static_cast<const _Float16*>(d_x1v);
  const Geometry& g = m.g;
  hipStream_t stream = m.stream;
  const int H = g.n_hidden, M = T * B, C = g.n_classes;
  const int NT = lstm_nt_for_batch(B);
  if (NT < 0 || B > lstm_max_rows(H)) throw std::runtime_error("run_acoustic_rows: more batch rows than one recurrent launch covers");
  m.ws_a.reserve((size_t)M * H * 2); m.ws_b.reserve((size_t)M * H * 2);
  m.ws_xproj.reserve((size_t)M * 4 * H * 4); m.ws_hall.reserve((size_t)M * H * 2);
  m.ws_logits.reserve((size_t)M * g.c_pad() * 4);
  const size_t hp_bytes = (size_t)(H / 32) * NT * 64 * 16;
  m.ws_hp0.reserve(hp_bytes); m.ws_hp1.reserve(hp_bytes);
  m.ws_c.reserve((size_t)B * H * 4);
  DenseArgs d{};
  d.relu_clip = g.relu_clip; d.M = M;
  stt_prof_mark(&m, 1);
  // layers 1-3 (deepspeech_model.py:204-224)
  d.wt = m.w1t.as<_Float16>(); d.x = d_x1; d.bias = m.b1.as<float>(); d.y = m.ws_a.p; d.N = H; d.K = g.k1_pad(); d.ldx = g.k1_pad(); d.ldy = H;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  d.wt = m.w2t.as<_Float16>(); d.x = m.ws_a.as<_Float16>(); d.bias = m.b2.as<float>(); d.y = m.ws_b.p; d.K = H; d.ldx = H;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  d.wt = m.w3t.as<_Float16>(); d.x = m.ws_b.as<_Float16>(); d.bias = m.b3.as<float>(); d.y = m.ws_a.p;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  // x-projection of all timesteps at once: [M][H] x [H][4H] + lstm bias
  d.wt = m.wxt.as<_Float16>(); d.x = m.ws_a.as<_Float16>(); d.bias = m.bl.as<float>(); d.y = m.ws_xproj.p; d.N = 4 * H; d.ldy = 4 * H;
  launch_dense(d, DENSE_EPI_BIAS_F32, stream);
  stt_prof_mark(&m, 2);
  // recurrence
  float* cbuf = (carry != 2 && d_c) ? d_c : m.ws_c.as<float>();
  if (carry == 0 || (carry == 1 && !d_c)) HIP_CHECK(hipMemsetAsync(cbuf, 0, (size_t)B * H * 4, stream));
  void* hp_a = (t_par & 1) ? m.ws_hp1.p : m.ws_hp0.p;  // holds h_{t-1} for the first step of this call
  if (carry == 1 && d_h) launch_pack_h(d_h, hp_a, B, H, NT, stream);
  else if (carry != 2) HIP_CHECK(hipMemsetAsync(hp_a, 0, hp_bytes, stream));
  LstmArgs l{};
  l.whp = m.whp.as<_Float16>(); l.xproj = m.ws_xproj.as<float>(); l.c = cbuf; l.h_all = m.ws_hall.as<_Float16>();
  l.n_hidden = H; l.batch = B;
  for (int t = 0; t < T; ++t) {
    const bool odd = ((t_par + t) & 1) != 0;
    l.hp_in = odd ? m.ws_hp1.as<_Float16>() : m.ws_hp0.as<_Float16>();
    l.hp_out = odd ? m.ws_hp0.as<_Float16>() : m.ws_hp1.as<_Float16>();
    l.t = t;
    l.h_f32 = (t == T - 1) ? d_h : nullptr;
    launch_lstm_step(l, NT, stream);
  }
  stt_prof_mark(&m, 3);
  // layer 5, layer 6, softmax (deepspeech_model.py:241-252, 357)
  d.wt = m.w5t.as<_Float16>(); d.x = m.ws_hall.as<_Float16>(); d.bias = m.b5.as<float>(); d.y = m.ws_b.p; d.N = H; d.K = H; d.ldx = H; d.ldy = H;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  if (!launch_logits_softmax(m.ws_b.as<_Float16>(), m.w6t.as<_Float16>(), m.b6.as<float>(), d_probs_out, M, H, C, B, probs_t_max, stream)) {
    d.wt = m.w6t.as<_Float16>(); d.x = m.ws_b.as<_Float16>(); d.bias = m.b6.as<float>(); d.y = m.ws_logits.p; d.N = g.c_pad(); d.ldy = g.c_pad();
    launch_dense(d, DENSE_EPI_BIAS_F32, stream);
    SoftmaxArgs s{};
    s.logits = m.ws_logits.as<float>(); s.probs = d_probs_out; s.M = M; s.C = C; s.ldl = g.c_pad(); s.batch = B; s.t_max = probs_t_max;
    launch_softmax(s, stream);
  }
}

void ModelState::run_acoustic_rows(const void* d_x1, int B, int T, float* d_c, float* d_h, bool carry_in, float* d_probs_out, int probs_t_max) {
  acoustic_rows(*this, d_x1, B, T, d_c, d_h, carry_in ? 1 : 0, 0, d_probs_out, probs_t_max);
}

void ModelState::run_acoustic(const float* d_feats, const int* d_nframes, int B, int t_max, float* d_c, float* d_h, bool carry_in) {
  const int M = t_max * B;
  ws_x1.reserve(x1_bytes(M));
  ws_probs.reserve((size_t)B * t_max * g.n_classes * 4);
  ContextArgs c{};
  c.feats = d_feats; c.n_frames = d_nframes; c.x1 = ws_x1.as<_Float16>(); c.x1_f32 = i8 ? ws_x1.as<float>() : nullptr;
  c.batch = B; c.t_max = t_max; c.n_coef = g.n_input; c.n_context = g.n_context; c.k_pad = x1_cols(); c.t0 = 0;
  launch_context(c, M, stream);
  run_acoustic_rows(ws_x1.p, B, t_max, d_c, d_h, carry_in, ws_probs.as<float>(), t_max);
}

void ModelState::run_acoustic_chunk(const float* d_feats, const int* d_nframes, int B, int t_max, int t0, int T, float* d_probs) {
  const int M = T * B;
  ws_x1.reserve(x1_bytes(M));
  ContextArgs c{};
  c.feats = d_feats; c.n_frames = d_nframes; c.x1 = ws_x1.as<_Float16>(); c.x1_f32 = i8 ? ws_x1.as<float>() : nullptr;
  c.batch = B; c.t_max = t_max; c.n_coef = g.n_input; c.n_context = g.n_context; c.k_pad = x1_cols(); c.t0 = t0;
  launch_context(c, M, stream);
  // probs[b][t0 + t][:]: the softmax writes row (t, b) at probs + ((b*t_max + t)*C), so offsetting the base by t0*C lands it
  acoustic_rows(*this, ws_x1.p, B, T, nullptr, nullptr, t0 == 0 ? 0 : 2, t0, d_probs + (size_t)t0 * g.n_classes, t_max, d_nframes);
}

// ------------------------------------------------------------------------------------------- acoustic model, three engines
// (ModelState::stream / stream_l / stream_o, see engine.h.)  The same kernels on the same operands in the same order per
// chunk as acoustic_rows(): results are bit-identical; only what runs beside what changes.

// Which of `cands` wait for a dispatch in flight on `hog`?  One oversubscribed launch on `hog`, two one-wave launches on every
// candidate right behind it: a chain on the hog's pipe (or on its hardware queue) finishes only when the hog's last workgroup has been
// placed, every other one within microseconds (benchmarks/pipe_probe.hip: 43 us per launch against 3).
static std::vector<char> streams_behind(hipStream_t hog, const std::vector<hipStream_t>& cands, unsigned* scratch) {
  std::vector<char> behind(cands.size(), 0);
  hipEvent_t h0 = nullptr, h1 = nullptr;
  std::vector<hipEvent_t> done(cands.size(), nullptr);
  HIP_CHECK(hipEventCreate(&h0)); HIP_CHECK(hipEventCreate(&h1));
  for (auto& e : done) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipStreamSynchronize(hog));      // (whatever the stream still had to do or to wait for must not stand between h0 and the hog)
  HIP_CHECK(hipEventRecord(h0, hog));
  launch_placement_hog(scratch, hog);
  HIP_CHECK(hipEventRecord(h1, hog));
  for (size_t c = 0; c < cands.size(); ++c) {
    for (int k = 0; k < 2; ++k) launch_placement_tick(scratch + 16 + c, cands[c]);
    HIP_CHECK(hipEventRecord(done[c], cands[c]));
  }
  HIP_CHECK(hipEventSynchronize(h1));
  for (auto& e : done) HIP_CHECK(hipEventSynchronize(e));
  float hog_ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&hog_ms, h0, h1));
  for (size_t c = 0; c < cands.size(); ++c) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, h0, done[c]));
    behind[c] = ms > 0.75f * hog_ms ? 1 : 0;    // (the hog's dispatch lasts ~1.6 ms, issuing all the chains ~0.2 ms: a free candidate is done long before)
  }
  (void)hipEventDestroy(h0); (void)hipEventDestroy(h1);
  for (auto& e : done) (void)hipEventDestroy(e);
  return behind;
}

// The recurrence is 250 short dependent launches per batch; the GEMM engine and the output engine launch grids of hundreds to thousands of
// workgroups that take their whole run time to dispatch; a search launch waits for CUs the search before it still holds.  Any of those on the
// recurrence's pipe and every recurrent step is picked up late (3.0 or 6.1 ms per batch, round 5).  Eight candidate streams are created (two
// per pipe, if nobody else creates streams meanwhile), each probed against the GEMM engine's stream and the group slots' search streams;
// the recurrence takes the candidate that waited for none of them (failing that: for a little-used search stream only), the output engine
// one that shares with neither the recurrence nor the GEMM engine.  The rest are destroyed.  ~2 ms, once per model (and per watch move).
void ModelState::place_engine_streams(hipStream_t* out_l, hipStream_t* out_o) {
  constexpr int NC = 8;
  std::vector<hipStream_t> cands(NC, nullptr);
  for (int c = 0; c < NC; ++c) create_engine_stream(&cands[c], c == 0 ? 1 : 2, /*high_priority=*/false);
  placement_scratch_.reserve(4096);
  unsigned* scratch = placement_scratch_.as<unsigned>();
  HIP_CHECK(hipMemsetAsync(scratch, 0, 4096, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  for (hipStream_t c : cands) { launch_placement_tick(scratch + 8, c); HIP_CHECK(hipStreamSynchronize(c)); }   // (a stream takes its hardware queue at first use)
  std::vector<int> penalty(NC, 0);
  const std::vector<char> with_gemm = streams_behind(stream, cands, scratch);
  int n_gemm = 0;
  for (int c = 0; c < NC; ++c) if (with_gemm[c]) { penalty[c] += 1000; ++n_gemm; }
  for (size_t a = 0; a < placement_avoid_.size() && a < 4; ++a) {
    if (!placement_avoid_[a]) continue;
    const std::vector<char> w = streams_behind(placement_avoid_[a], cands, scratch);
    for (int c = 0; c < NC; ++c) if (w[c]) penalty[c] += a < 2 ? 100 : 10;
  }
  int il = 0;
  for (int c = 1; c < NC; ++c) if (penalty[c] < penalty[il]) il = c;
  // the output engine: not behind the recurrence (its own GEMMs would hold the recurrence's pipe), not behind the GEMM engine
  std::vector<hipStream_t> rest;
  std::vector<int> rest_idx;
  for (int c = 0; c < NC; ++c) if (c != il) { rest.push_back(cands[c]); rest_idx.push_back(c); }
  const std::vector<char> with_l = streams_behind(cands[il], rest, scratch);
  int io = -1, best = 1 << 30;
  for (size_t r = 0; r < rest.size(); ++r) {
    const int pen = penalty[rest_idx[r]] + (with_l[r] ? 100000 : 0);
    if (pen < best) { best = pen; io = rest_idx[r]; }
  }
  if (io < 0) io = (il + 1) % NC;
  if (tune().dump_marks) fprintf(stderr, "stt_amd: engine streams placed: recurrence = candidate %d (penalty %d), output engine = candidate %d (penalty %d); %d of %d candidates behind the GEMM engine's stream\n",
                                 il, penalty[il], io, penalty[io], n_gemm, NC);
  // the recurrence's stream keeps its high priority where the runtime allows one to be re-created on the same queue: it does not -- the chosen
  // candidate is used as it is (measured in round 5: the recurrence's queue at normal priority 3.095 against 3.09 ms per batch)
  *out_l = cands[il]; *out_o = cands[io];
  for (int c = 0; c < NC; ++c) if (c != il && c != io) (void)hipStreamDestroy(cands[c]);
  __atomic_fetch_add(&tune().am_placed, 1, __ATOMIC_RELAXED);
  tune().am_placed = (tune().am_placed & 0xff) | (n_gemm << 8);
}

// The whole batch pipeline at once, before its first group is enqueued (api.cpp: batch_init_slots; nothing in flight): sixteen candidate
// streams are sorted into pipe classes by probing (which candidates wait behind the GEMM engine's stream; then, for each class not yet seen,
// which wait behind its first member), and every role gets a class of its own where there are enough: the recurrence one, the output
// engine one, ALL group slots' search streams a third (searches run one or two at a time and only ever wait for CUs, never for each other's
// dispatch) -- so that no search stream can sit on the recurrence's pipe either (the first cut probed the existing search streams and found,
// with eight idle streams created before the model, every pipe already taken by one of them: 6.3 ms per batch again).
// spread_searches (a search-bound setup -- code-point scorer, beam > 512: four searches side by side, each waiting for its own chunk events):
// the group slots' search streams take FOUR DIFFERENT classes instead of one.  Four event-gated queues on one pipe run a batch in 71 ms
// instead of 47 (`bytes`, round 6): a queue whose head packet waits for an event holds up the queues behind it on that pipe.
void ModelState::place_batch_streams(hipStream_t* slot_streams, int n_slots, bool spread_searches) {
  constexpr int NC = 16;
  std::vector<hipStream_t> cands(NC, nullptr);
  for (int c = 0; c < NC; ++c) create_engine_stream(&cands[c], 2, false);
  placement_scratch_.reserve(4096);
  unsigned* scratch = placement_scratch_.as<unsigned>();
  HIP_CHECK(hipMemsetAsync(scratch, 0, 4096, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  for (hipStream_t c : cands) { launch_placement_tick(scratch + 8, c); HIP_CHECK(hipStreamSynchronize(c)); }   // (a stream takes its hardware queue at first use)
  std::vector<int> cls(NC, -1);
  {
    const std::vector<char> g = streams_behind(stream, cands, scratch);
    for (int c = 0; c < NC; ++c) if (g[c]) cls[c] = 0;                       // class 0: the GEMM engine's pipe (or queue)
  }
  int n_cls = 1;
  for (int c = 0; c < NC && n_cls < 8; ++c) {
    if (cls[c] >= 0) continue;
    std::vector<hipStream_t> open; std::vector<int> open_idx;
    for (int d = c + 1; d < NC; ++d) if (cls[d] < 0) { open.push_back(cands[d]); open_idx.push_back(d); }
    cls[c] = n_cls;
    if (!open.empty()) {
      const std::vector<char> b = streams_behind(cands[c], open, scratch);
      for (size_t r = 0; r < open.size(); ++r) if (b[r]) cls[open_idx[r]] = n_cls;
    }
    ++n_cls;
  }
  std::vector<std::vector<int>> members(n_cls);
  for (int c = 0; c < NC; ++c) members[cls[c]].push_back(c);
  std::vector<int> free_cls;                                                    // classes other than the GEMM engine's, largest first
  for (int k = 1; k < n_cls; ++k) if (!members[k].empty()) free_cls.push_back(k);
  std::stable_sort(free_cls.begin(), free_cls.end(), [&](int a, int b) { return members[a].size() > members[b].size(); });
  std::vector<char> used(NC, 0);
  auto take = [&](int k) -> hipStream_t {
    for (int c : members[k]) if (!used[c]) { used[c] = 1; return cands[c]; }
    return nullptr;
  };
  // searches get the largest class (they need n_slots streams), the recurrence the next, the output engine the third (or shares the searches')
  const int ks = free_cls.size() > 0 ? free_cls[0] : 0, kr = free_cls.size() > 1 ? free_cls[1] : ks, ko = free_cls.size() > 2 ? free_cls[2] : ks;
  hipStream_t nl = take(kr), no = take(ko);
  if (!nl || !no) {   // (cannot happen with sixteen candidates unless the runtime hands out one queue for all of them: keep what exists)
    for (hipStream_t c : cands) if (c != nl && c != no) (void)hipStreamDestroy(c);
    if (nl) (void)hipStreamDestroy(nl);
    if (no) (void)hipStreamDestroy(no);
    return;
  }
  stream_l = nl; stream_o = no;
  for (int i = 0; i < n_slots; ++i) {
    hipStream_t st = nullptr;
    if (spread_searches) {
      const int order[4] = {ks, ko, kr, 0};
      for (int k = 0; k < 4 && !st; ++k) st = take(order[(i + k) & 3]);
    } else {
      st = take(ks);
      if (!st && ko != ks) st = take(ko);
    }
    if (!st) continue;                            // (this slot keeps the stream it has)
    if (slot_streams[i]) { HIP_CHECK(hipStreamSynchronize(slot_streams[i])); (void)hipStreamDestroy(slot_streams[i]); }
    slot_streams[i] = st;
  }
  for (int c = 0; c < NC; ++c) if (!used[c]) (void)hipStreamDestroy(cands[c]);
  if (tune().dump_marks) {
    fprintf(stderr, "stt_amd: batch streams placed: %d pipe classes among %d candidates (sizes", n_cls, NC);
    for (int k = 0; k < n_cls; ++k) fprintf(stderr, " %zu", members[k].size());
    fprintf(stderr, "; class 0 = behind the GEMM engine); searches class %d, recurrence class %d, output engine class %d\n", ks, kr, ko);
  }
  __atomic_fetch_add(&tune().am_placed, 1, __ATOMIC_RELAXED);
  tune().am_placed = (tune().am_placed & 0xff) | ((int)members[0].size() << 8) | (n_cls << 16);
}

bool ModelState::am_pipe_init() {
  if (!tune().am_pipe) return false;
  if (ev_x_ready[0]) return true;
  if (stream_l) {}     // (placed with the group slots' search streams: place_batch_streams)
  else if (tune().am_place && tune().search_cus <= 0) place_engine_streams(&stream_l, &stream_o);
  else {
    // the recurrence is the critical path: its workgroups go first whenever a CU has room
    create_engine_stream(&stream_l, 1, /*high_priority=*/true);
    create_engine_stream(&stream_o, 2);
  }
  for (int i = 0; i < kAmRing; ++i)
    for (hipEvent_t* e : {&ev_x_ready[i], &ev_x_free[i], &ev_h_ready[i], &ev_h_free[i]}) HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  return true;
}

// Which hardware queues the recurrence's and the output engine's streams land on decides whether the three engines work at all: the same
// process, the same kernels, one idle stream created at another moment -- 3.0 or 6.5 ms per batch (int8 form; the f16 form 3.08 or 6.1:
// profiles/r05_queue_placement.json).  In a bad placement every recurrent step (a fresh dispatch behind the one before it, 250 per batch)
// is picked up tens of microseconds late, as if its queue were served in turns with another one.  HIP offers no say in the placement -- a
// stream takes the process's next hardware queue -- but the symptom is unmistakable in the pipeline's own timing: a chunk's recurrence,
// bracketed by two events on its stream, takes 17 - 23 us per step when the queue is served at once and 36 - 46 when it is not.  So the
// engine watches one chunk at a time (no synchronisation: the events are read when they have long been reached) and, after two slow
// readings in a row, moves the recurrence and the output engine to fresh streams -- created BEFORE the old ones are destroyed, so that they
// take other hardware queues -- at most `am_moves` times per model.  A pipeline that is slow for another reason pays a few stream
// synchronisations and stays where it was last.
void ModelState::am_watch_begin(int T) {
  if (tune().am_moves <= 0 || watch_armed || watch_search_bound || T < 32 || watch_moves > tune().am_moves) return;
  if (!ev_watch[0]) for (auto& e : ev_watch) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipEventRecord(ev_watch[0], stream_l));
  watch_steps = -T;                                    // (negative: begun, not ended)
}
void ModelState::am_watch_end() {
  if (watch_steps >= 0 || watch_armed) return;
  HIP_CHECK(hipEventRecord(ev_watch[1], stream_l));
  watch_steps = -watch_steps;
  watch_armed = true;
}
void ModelState::am_replace_if_slow() {
  if (!watch_armed || hipEventQuery(ev_watch[1]) != hipSuccess) { (void)hipGetLastError(); return; }
  watch_armed = false;
  if (watch_search_bound) { watch_slow = 0; return; }   // (a reading taken before the scorer / beam changed)
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, ev_watch[0], ev_watch[1]) != hipSuccess) { (void)hipGetLastError(); return; }
  const float us = 1e3f * ms / (float)std::max(1, watch_steps);
  const float hh = (float)g.n_hidden / 2048.0f;
  const float limit = tune().am_slow_us > 0 ? (float)tune().am_slow_us : 31.0f * std::max(1.0f, hh * hh);   // (the bench's model: 17 - 23 us per step as run when served at once)
  tune().am_step_us_x10 = (int)(us * 10.0f);
  watch_slow = us > limit ? watch_slow + 1 : 0;
  if (watch_slow < 2 || watch_moves >= tune().am_moves) return;
  watch_slow = 0;
  ++watch_moves;
  __atomic_fetch_add(&tune().am_moved, 1, __ATOMIC_RELAXED);
  hipStream_t nl = nullptr, no = nullptr;
  HIP_CHECK(hipStreamSynchronize(stream_l)); HIP_CHECK(hipStreamSynchronize(stream_o));   // (rare: nothing of the old queues is left in flight)
  if (tune().am_place && tune().search_cus <= 0) {
    HIP_CHECK(hipStreamSynchronize(stream));       // (the probe's dispatches should meet an idle GEMM engine)
    place_engine_streams(&nl, &no);
  } else {
    create_engine_stream(&nl, 1, /*high_priority=*/true);
    create_engine_stream(&no, 2);
  }
  (void)hipStreamDestroy(stream_l); (void)hipStreamDestroy(stream_o);
  stream_l = nl; stream_o = no;
}

static int dense_lds_floor() {  // bytes; > 80 KiB = one GEMM workgroup per CU while the recurrence runs beside it
  const int kb = tune().dense_lds_kb;
  return kb <= 0 ? 0 : kb * 1024;
}

// The three engines in the released models' own arithmetic (ModelState::i8): the same hand-over (rings of kAmRing chunk buffers, one event
// pair per slot), the kernels of acoustic_rows_i8().  Engine 1 leaves per ring slot: the x half of the cell's int32 sums (am_xproj), layer
// 3's f32 rows (am_y3: the recurrent step's slow path) and their scaling factors / ranges (am_xs); engine 2 (prep + T steps, one hipGraph)
// leaves h_t in f32 (am_hall); engine 3 quantises those rows for layer 5.
void ModelState::run_acoustic_chunk_piped_i8(const float* d_feats, const int* d_nframes, int B, int t_max, int t0, int T, float* d_probs, hipEvent_t done) {
  const int H = g.n_hidden, M = T * B, C = g.n_classes, K1 = g.k1_pad8(), CP = g.c_pad8();
  const int NT = lstm_nt_for_batch(B);
  if (NT < 0) throw std::runtime_error("run_acoustic_chunk: more batch rows than one recurrent launch covers");
  const int NTR = NT * 16, NWG = H / 16;
  const int slot = (int)(am_seq % kAmRing);
  const bool wrapped = am_seq >= (unsigned long long)kAmRing;
  ++am_seq;
  ws_x1.reserve(x1_bytes(M));
  ws_a.reserve((size_t)M * H * 4); ws_b.reserve((size_t)M * H * 4); ws_o.reserve((size_t)M * H * 4);
  am_xproj[slot].reserve((size_t)M * 4 * H * 4); am_hall[slot].reserve((size_t)M * H * 4); am_y3[slot].reserve((size_t)M * H * 4); am_xs[slot].reserve((size_t)M * 8);
  am_logits.reserve((size_t)M * CP * 4);
  am_qx.reserve((size_t)M * std::max(K1, H)); am_qs.reserve((size_t)M * 4); am_qo.reserve((size_t)M * H); am_qos.reserve((size_t)M * 4);
  const size_t hq_bytes = lstm_i8_hq_bytes(H, NT);
  am_hq0.reserve(hq_bytes); am_hq1.reserve(hq_bytes);
  am_c.reserve((size_t)B * H * 4); am_hprev0.reserve((size_t)B * H * 4); am_hlast.reserve((size_t)B * H * 4);
  am_pmax.reserve((size_t)2 * NTR * NWG * 4); am_flag.reserve((size_t)2 * NTR * 4); am_zslow.reserve((size_t)NWG * NTR * 64 * 4);
  if (!ws_slow.p) { ws_slow.reserve(4); HIP_CHECK(hipMemsetAsync(ws_slow.p, 0, 4, stream)); HIP_CHECK(hipStreamSynchronize(stream)); }

  // ---- engine 1 (`stream`)
  if (wrapped) HIP_CHECK(hipStreamWaitEvent(stream, ev_x_free[slot], 0));
  ContextArgs c{};
  c.feats = d_feats; c.n_frames = d_nframes; c.x1 = nullptr; c.x1_f32 = ws_x1.as<float>();
  c.batch = B; c.t_max = t_max; c.n_coef = g.n_input; c.n_context = g.n_context; c.k_pad = K1; c.t0 = t0;
  launch_context(c, M, stream);
  stt_prof_mark_on(this, 1, 0, stream);
  signed char* qx = am_qx.as<signed char>();
  float* qs = am_qs.as<float>();
  float *xs3 = am_xs[slot].as<float>(), *xr3 = xs3 + M;
  float *act_a = ws_a.as<float>(), *act_b = ws_b.as<float>(), *y3 = am_y3[slot].as<float>();
  launch_quantize_rows(ws_x1.as<float>(), qx, qs, M, K1, stream);
  launch_dense_hybrid_i8(qx, qs, w1q.as<signed char>(), s1.as<float>(), sn[0], b1.as<float>(), act_a, M, H, K1, stream, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(act_a, qx, qs, M, H, stream);
  launch_dense_hybrid_i8(qx, qs, w2q.as<signed char>(), s2.as<float>(), sn[1], b2.as<float>(), act_b, M, H, H, stream, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(act_b, qx, qs, M, H, stream);
  launch_dense_hybrid_i8(qx, qs, w3q.as<signed char>(), s3.as<float>(), sn[2], b3.as<float>(), y3, M, H, H, stream, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(y3, qx, xs3, M, H, stream, xr3);
  launch_dense_hybrid_i8(qx, xs3, wxq.as<signed char>(), sk.as<float>(), sn[3], bl.as<float>(), am_xproj[slot].p, M, 4 * H, H, stream, DENSE_EPI_I8_RAW);
  stt_prof_mark_on(this, -1, 0, stream);
  HIP_CHECK(hipEventRecord(ev_x_ready[slot], stream));

  // ---- engine 2 (`stream_l`): prep + T steps; c in am_c, the carried h in am_hlast (f32)
  HIP_CHECK(hipStreamWaitEvent(stream_l, ev_x_ready[slot], 0));
  if (wrapped) HIP_CHECK(hipStreamWaitEvent(stream_l, ev_h_free[slot], 0));
  stt_prof_mark_on(this, 2, 5, stream_l);
  if (t0 == 0) HIP_CHECK(hipMemsetAsync(am_c.p, 0, (size_t)B * H * 4, stream_l));
  LstmI8Args l{};
  l.whp = whpq.as<signed char>(); l.accx = am_xproj[slot].as<int>(); l.bias = bl.as<float>(); l.wscale = sk.as<float>(); l.wscale_n = sn[3];
  l.xscale = xs3; l.xrange = xr3; l.c = am_c.as<float>(); l.h_all = am_hall[slot].as<float>(); l.h_last = am_hlast.as<float>();
  l.pmax = am_pmax.as<float>(); l.flag = am_flag.as<int>(); l.y3 = y3; l.h_prev0 = am_hprev0.as<float>();
  l.wxq = wxq.as<signed char>(); l.whq = whq.as<signed char>(); l.zslow = am_zslow.as<float>();
  l.n_hidden = H; l.batch = B; l.T = T; l.prio = tune().lstm_prio; l.slow_count = ws_slow.as<unsigned>();
  l.row_frames = d_nframes; l.t0 = t0;
  l.probe = tune().lstm_probe >= 100 ? tune().lstm_probe - 100 : 0;     // (experiments only, lstm_probe = 100 + probe: a timing-probe kernel in the timed path -- wrong results)
  const float* h_src = t0 == 0 ? nullptr : am_hlast.as<float>();
  auto steps = [&]() {
    l.t = 0; l.hq_in = am_hq0.as<signed char>(); l.hq_out = am_hq1.as<signed char>();
    launch_lstm_i8_prep(l, h_src, NT, stream_l);
    for (int t = 0; t < T; ++t) {
      l.t = t;
      l.hq_in = (t & 1) ? am_hq1.as<signed char>() : am_hq0.as<signed char>();
      l.hq_out = (t & 1) ? am_hq0.as<signed char>() : am_hq1.as<signed char>();
      launch_lstm_i8_step(l, NT, stream_l, tune().lstm_i8_rows);
    }
  };
  am_watch_begin(T);
  if (tune().lstm_graph) {
    LstmGraphKey key;
    memset(&key, 0, sizeof(key));
    key.xproj = am_xproj[slot].p; key.hall = am_hall[slot].p; key.c = am_c.p; key.hp0 = am_hq0.p; key.hp1 = am_hq1.p; key.whp = am_y3[slot].p;
    key.T = T; key.par = t0; key.B = B; key.NT = NT; key.passes = 100; key.prio = l.prio; key.H = H;   // (t0 and the frame table's address are baked into the launches)
    key.first = tune().lstm_i8_rows * 1000 + l.probe; key.nframes = d_nframes;
    run_lstm_graph(key, steps);
  } else steps();
  am_watch_end();
  stt_prof_mark_on(this, -1, 5, stream_l);
  HIP_CHECK(hipEventRecord(ev_x_free[slot], stream_l));
  HIP_CHECK(hipEventRecord(ev_h_ready[slot], stream_l));

  // ---- engine 3 (`stream_o`): layer 5, layer 6, softmax -> probs[b][t0 + t][:]
  HIP_CHECK(hipStreamWaitEvent(stream_o, ev_h_ready[slot], 0));
  stt_prof_mark_on(this, 3, 6, stream_o);
  float* probs_out = d_probs + (size_t)t0 * C;
  signed char* qo = am_qo.as<signed char>();
  float* qos = am_qos.as<float>();
  launch_quantize_rows(am_hall[slot].as<float>(), qo, qos, M, H, stream_o);
  HIP_CHECK(hipEventRecord(ev_h_free[slot], stream_o));
  launch_dense_hybrid_i8(qo, qos, w5q.as<signed char>(), s5.as<float>(), sn[4], b5.as<float>(), ws_o.p, M, H, H, stream_o, DENSE_EPI_I8_RELU_F32, g.relu_clip);
  launch_quantize_rows(ws_o.as<float>(), qo, qos, M, H, stream_o);
  launch_dense_hybrid_i8(qo, qos, w6q.as<signed char>(), s6.as<float>(), sn[5], b6q.as<float>(), am_logits.p, M, CP, H, stream_o, DENSE_EPI_I8_F32);
  SoftmaxArgs sm{};
  sm.logits = am_logits.as<float>(); sm.probs = probs_out; sm.M = M; sm.C = C; sm.ldl = CP; sm.batch = B; sm.t_max = t_max; sm.exact = 1;
  launch_softmax(sm, stream_o);
  stt_prof_mark_on(this, -1, 6, stream_o);
  HIP_CHECK(hipEventRecord(done, stream_o));
}

// The recurrence of a chunk as one hipGraph (engine.h: LstmGraphKey): `steps` enqueues the launches on stream_l.
void ModelState::run_lstm_graph(const LstmGraphKey& key, const std::function<void()>& steps, hipStream_t st) {
  hipStream_t stream_l = st ? st : this->stream_l;   // (the stream the launches go to)
  // A combination is captured the SECOND time it comes up (the first ran eagerly: module load, function attributes).  First
  // sightings live in their own small set, so a ragged job's many one-off shapes never push the graphs out of the cache; when
  // the cache is full (or holds graphs of buffers that have since been reallocated) it is emptied and refills with what recurs.
  auto found = lstm_graphs_.find(key);
  if (found != lstm_graphs_.end()) {
    if (found->second.exec) HIP_CHECK(hipGraphLaunch(found->second.exec, stream_l));
    else steps();                                       // instantiation failed once: this combination stays on plain launches
  } else if (!lstm_seen_.count(key)) {
    if (lstm_seen_.size() >= 1024) lstm_seen_.clear();
    lstm_seen_.insert(key);
    steps();
  } else {
    if (lstm_graphs_.size() >= 256) {
      HIP_CHECK(hipStreamSynchronize(stream_l));        // (rare) nothing may still be replaying what is destroyed
      for (auto& kv : lstm_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      lstm_graphs_.clear();
    }
    LstmGraph gr;
    hipGraph_t graph = nullptr;
    std::lock_guard<std::recursive_mutex> capturing(hip_capture_mutex());   // (no other thread's device-wide synchronisation or free meanwhile)
    HIP_CHECK(hipStreamBeginCapture(stream_l, hipStreamCaptureModeRelaxed));
    try { steps(); }
    catch (...) { (void)hipStreamEndCapture(stream_l, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }  // never leave the stream capturing
    HIP_CHECK(hipStreamEndCapture(stream_l, &graph));
    const hipError_t ie = hipGraphInstantiate(&gr.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { gr.exec = nullptr; (void)hipGetLastError(); }
    lstm_graphs_[key] = gr;
    lstm_seen_.erase(key);
    if (gr.exec) HIP_CHECK(hipGraphLaunch(gr.exec, stream_l));
    else steps();
  }
}

void ModelState::run_acoustic_chunk_piped(const float* d_feats, const int* d_nframes, int B, int t_max, int t0, int T, float* d_probs, hipEvent_t done) {
  if (i8) { run_acoustic_chunk_piped_i8(d_feats, d_nframes, B, t_max, t0, T, d_probs, done); return; }
  const int H = g.n_hidden, M = T * B, C = g.n_classes;
  const int NT = lstm_nt_for_batch(B);
  if (NT < 0 || B > lstm_max_rows(H)) throw std::runtime_error("run_acoustic_chunk: more batch rows than one recurrent launch covers");
  const int slot = (int)(am_seq % kAmRing);
  const bool wrapped = am_seq >= (unsigned long long)kAmRing;
  ++am_seq;
  // Buffers may only grow while nothing is in flight on them: DevBuf::reserve frees the old allocation (hipFree waits for the
  // device), so a growing chunk size costs one stall, never a dangling pointer.
  ws_x1.reserve((size_t)M * g.k1_pad() * 2);
  ws_a.reserve((size_t)M * H * 2); ws_b.reserve((size_t)M * H * 2); ws_o.reserve((size_t)M * H * 2);
  am_xproj[slot].reserve((size_t)M * 4 * H * 4); am_hall[slot].reserve((size_t)M * H * 2);
  am_logits.reserve((size_t)M * g.c_pad() * 4);
  const size_t hp_bytes = (size_t)(H / 32) * NT * 64 * 16;
  am_hp0.reserve(hp_bytes); am_hp1.reserve(hp_bytes);
  am_c.reserve((size_t)B * H * 4);

  // ---- engine 1 (`stream`): windows, layers 1-3, x-projection of this chunk into ring slot `slot`
  if (wrapped) HIP_CHECK(hipStreamWaitEvent(stream, ev_x_free[slot], 0));  // the recurrence of chunk seq - kAmRing has read the slot
  ContextArgs c{};
  c.feats = d_feats; c.n_frames = d_nframes; c.x1 = ws_x1.as<_Float16>();
  c.batch = B; c.t_max = t_max; c.n_coef = g.n_input; c.n_context = g.n_context; c.k_pad = g.k1_pad(); c.t0 = t0;
  launch_context(c, M, stream);
  DenseArgs d{};
  d.relu_clip = g.relu_clip; d.M = M;
  const int lstm_passes = tune().lstm_passes;  // form of the recurrent step (kernels.h); 1 = 64 KiB of LDS: no room beside the solo GEMM
  {  // GEMMs as co-tenants of the recurrence: the three-stage one-per-CU form (default), or the two-stage form padded to one per CU
    const int solo = tune().dense_solo;  // 2: eight waves, 1: four, 0: padded two-stage form
    d.solo = lstm_passes >= 2 ? solo : 0; d.lds_floor = d.solo ? 0 : dense_lds_floor();  // (96 KiB beside the one-pass step's 66 would not fit)
  }
  stt_prof_mark_on(this, 1, 0, stream);
  d.wt = w1t.as<_Float16>(); d.x = ws_x1.as<_Float16>(); d.bias = b1.as<float>(); d.y = ws_a.p; d.N = H; d.K = g.k1_pad(); d.ldx = g.k1_pad(); d.ldy = H;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  d.wt = w2t.as<_Float16>(); d.x = ws_a.as<_Float16>(); d.bias = b2.as<float>(); d.y = ws_b.p; d.K = H; d.ldx = H;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  d.wt = w3t.as<_Float16>(); d.x = ws_b.as<_Float16>(); d.bias = b3.as<float>(); d.y = ws_a.p;
  launch_dense(d, DENSE_EPI_RELU_F16, stream);
  d.wt = wxt.as<_Float16>(); d.x = ws_a.as<_Float16>(); d.bias = bl.as<float>(); d.y = am_xproj[slot].p; d.N = 4 * H; d.ldy = 4 * H;
  launch_dense(d, DENSE_EPI_BIAS_F32, stream);
  stt_prof_mark_on(this, -1, 0, stream);
  HIP_CHECK(hipEventRecord(ev_x_ready[slot], stream));

  // ---- engine 2 (`stream_l`): the recurrence; state carried in am_c / am_hp* from the previous chunk of the batch
  HIP_CHECK(hipStreamWaitEvent(stream_l, ev_x_ready[slot], 0));
  if (wrapped) HIP_CHECK(hipStreamWaitEvent(stream_l, ev_h_free[slot], 0));  // layer 5 of chunk seq - kAmRing has read the slot
  stt_prof_mark_on(this, 2, 5, stream_l);
  void* hp_a = (t0 & 1) ? am_hp1.p : am_hp0.p;  // holds h_{t-1} for the first step of this chunk
  if (t0 == 0) {
    HIP_CHECK(hipMemsetAsync(am_c.p, 0, (size_t)B * H * 4, stream_l));
    HIP_CHECK(hipMemsetAsync(hp_a, 0, hp_bytes, stream_l));
  }
  LstmArgs l{};
  l.whp = whp.as<_Float16>(); l.xproj = am_xproj[slot].as<float>(); l.c = am_c.as<float>(); l.h_all = am_hall[slot].as<_Float16>();
  l.n_hidden = H; l.batch = B; l.h_f32 = nullptr; l.passes = lstm_passes;
  l.prio = tune().lstm_prio;
  auto steps = [&]() {
    for (int t = 0; t < T; ++t) {
      const bool odd = ((t0 + t) & 1) != 0;
      l.hp_in = odd ? am_hp1.as<_Float16>() : am_hp0.as<_Float16>();
      l.hp_out = odd ? am_hp0.as<_Float16>() : am_hp1.as<_Float16>();
      l.t = t;
      launch_lstm_step(l, NT, stream_l);
    }
  };
  am_watch_begin(T);
  if (tune().lstm_graph) {
    LstmGraphKey key;
    memset(&key, 0, sizeof(key));  // (padding bytes take part in the comparison)
    key.xproj = am_xproj[slot].p; key.hall = am_hall[slot].p; key.c = am_c.p; key.hp0 = am_hp0.p; key.hp1 = am_hp1.p; key.whp = whp.p;
    key.T = T; key.par = t0 & 1; key.B = B; key.NT = NT; key.passes = l.passes; key.prio = l.prio; key.H = H; key.first = tune().lstm_form * 16 + tune().lstm_prefetch;  // (what else selects the kernel instance)
    run_lstm_graph(key, steps);
  } else steps();
  am_watch_end();
  stt_prof_mark_on(this, -1, 5, stream_l);
  HIP_CHECK(hipEventRecord(ev_x_free[slot], stream_l));
  HIP_CHECK(hipEventRecord(ev_h_ready[slot], stream_l));

  // ---- engine 3 (`stream_o`): layer 5, layer 6 + softmax -> probs[b][t0 + t][:]
  HIP_CHECK(hipStreamWaitEvent(stream_o, ev_h_ready[slot], 0));
  stt_prof_mark_on(this, 3, 6, stream_o);
  float* probs_out = d_probs + (size_t)t0 * C;
  d.wt = w5t.as<_Float16>(); d.x = am_hall[slot].as<_Float16>(); d.bias = b5.as<float>(); d.y = ws_o.p; d.N = H; d.K = H; d.ldx = H; d.ldy = H;
  launch_dense(d, DENSE_EPI_RELU_F16, stream_o);
  HIP_CHECK(hipEventRecord(ev_h_free[slot], stream_o));
  if (!launch_logits_softmax(ws_o.as<_Float16>(), w6t.as<_Float16>(), b6.as<float>(), probs_out, M, H, C, B, t_max, stream_o)) {
    d.wt = w6t.as<_Float16>(); d.x = ws_o.as<_Float16>(); d.bias = b6.as<float>(); d.y = am_logits.p; d.N = g.c_pad(); d.ldy = g.c_pad();
    launch_dense(d, DENSE_EPI_BIAS_F32, stream_o);
    SoftmaxArgs sm{};
    sm.logits = am_logits.as<float>(); sm.probs = probs_out; sm.M = M; sm.C = C; sm.ldl = g.c_pad(); sm.batch = B; sm.t_max = t_max;
    launch_softmax(sm, stream_o);
  }
  stt_prof_mark_on(this, -1, 6, stream_o);
  HIP_CHECK(hipEventRecord(done, stream_o));
}

// ------------------------------------------------------------------------------------------- decoder state
DevScorer ModelState::current_scorer(std::shared_ptr<ScorerDev> sc, const std::map<std::string, float>& hot, HotTables& ht, bool in_flight) {
  DevScorer s{};
  if (!sc) return s;
  s = sc->dev;
  s.n_hot = 0;
  if (!hot.empty()) {
    if (!ht.valid || ht.loaded != hot) {
      std::vector<uint64_t> hs; std::vector<float> bs;
      for (const auto& kv : hot) { hs.push_back(stt_murmur64a(kv.first.data(), kv.first.size())); bs.push_back(kv.second); }
      if (in_flight && ht.hash) { retired_bufs_.push_back(std::move(ht.hash)); retired_bufs_.push_back(std::move(ht.boost)); }  // still read by the searches in flight
      if (!ht.hash) { ht.hash.reset(new DevBuf()); ht.boost.reset(new DevBuf()); }
      ht.hash->upload(hs.data(), hs.size() * 8, stream); ht.boost->upload(bs.data(), bs.size() * 4, stream);  // (synchronises `stream`: only when the words changed)
      ht.loaded = hot; ht.valid = true;
    }
    s.n_hot = (int)hot.size(); s.hot_hash = ht.hash->as<uint64_t>(); s.hot_boost = ht.boost->as<float>();
  }
  return s;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Per-stream slab: [fixed: beam arrays + per-step candidate workspace][path arena][time arena][pq][boundary entries].
// Every timestep appends at most `beam` path nodes, `beam` time nodes and `beam` boundary entries (one per scored word end), so
// (frames + 2) x beam of each is the bound that can never overflow -- 84 B per node: 40 GB for a group of 64 five-minute
// utterances at beam 500.  Measured fill of that bound (profiles/r03_arena_fill.txt; 64 x 5 s, the 1-15 s job, byte mode, peaky):
// time nodes <= 68 %, path nodes <= 14 %, boundary entries <= 8 % (64 B each -- the bulk of the bound).
//   * a live stream checks before every chunk that the chunk's own worst case fits and grows the arena that does not
//     (decoder_reserve: capacity doubles) -- it can never overflow and holds about twice what it uses;
//   * a batch group is enqueued without a host round trip per chunk, so it is sized optimistically (`optimistic`: time nodes in
//     full, path nodes 1/2, boundary entries 1/4 of the bound: 30 B per node instead of 84) and a group that does overflow --
//     the search kernel flags it, nothing is written out of bounds -- is decoded again with the full bound (api.cpp).
namespace {
struct SlabLayout {
  size_t fixed, o_pa, o_ta, o_pq, o_be, per;
  uint32_t pa_cap, ta_cap, be_cap;
  size_t o_dpd, o_dtd, o_chain;   // incremental back-tracking (chain_cap > 0): ctc.h, DecStream::dpd / dtd / chain
  uint32_t chain_cap;
};
// chain_cap = tokens / timesteps the cached best path may hold: one per frame the time arena was sized for (0 = no cache)
SlabLayout slab_layout(size_t fixed, uint32_t pa_cap, uint32_t ta_cap, uint32_t be_cap, uint32_t chain_cap = 0) {
  SlabLayout l{};
  l.fixed = fixed; l.pa_cap = pa_cap; l.ta_cap = ta_cap; l.be_cap = be_cap; l.chain_cap = chain_cap;
  l.o_pa = fixed; l.o_ta = l.o_pa + al256((size_t)pa_cap * 8); l.o_pq = l.o_ta + al256((size_t)ta_cap * 8);
  l.o_be = l.o_pq + al256((size_t)pa_cap * 4); l.per = l.o_be + al256((size_t)be_cap * sizeof(BEntry));
  if (chain_cap) {
    l.o_dpd = l.per; l.o_dtd = l.o_dpd + al256((size_t)pa_cap * 4); l.o_chain = l.o_dtd + al256((size_t)ta_cap * 4);
    l.per = l.o_chain + al256(((size_t)8 * chain_cap + 2) * 4);
  }
  return l;
}
uint32_t chain_cap_for(uint32_t ta_cap, int beam) { return ta_cap / (uint32_t)std::max(1, beam) + 8u; }
void point_arenas(DecStream& S, uint8_t* base, const SlabLayout& l) {
  S.pa = (uint2*)(base + l.o_pa); S.ta = (uint2*)(base + l.o_ta); S.pq = (uint32_t*)(base + l.o_pq); S.be = (BEntry*)(base + l.o_be);
  S.pa_cap = l.pa_cap; S.ta_cap = l.ta_cap; S.be_cap = l.be_cap;
  S.dpd = l.chain_cap ? (uint32_t*)(base + l.o_dpd) : nullptr; S.dtd = l.chain_cap ? (uint32_t*)(base + l.o_dtd) : nullptr;
  S.chain = l.chain_cap ? (uint32_t*)(base + l.o_chain) : nullptr; S.chain_cap = l.chain_cap; S.pad_ = 0;
}
}  // namespace

void ModelState::decoder_create(DecoderBatch& db, int n_streams, int beam, int expected_frames, std::shared_ptr<ScorerDev> sc, PinnedBuf* staging,
                                bool optimistic, bool decode_cache) {
  const int C = g.n_classes;
  if (beam < 1 || beam > STT_MAX_BEAM) throw std::runtime_error("beam width must be in [1, 1024]");
  db.n_streams = n_streams; db.beam = beam; db.C = C;
  const uint32_t cap = (uint32_t)((beam + 63) & ~63);
  const uint32_t cand_cap = (uint32_t)beam * (uint32_t)(C - 1);
  if (g_debug_arena_frames > 0) expected_frames = g_debug_arena_frames;  // test hook: arenas that cannot hold the utterance
  const uint32_t arena = (uint32_t)(expected_frames + 2) * (uint32_t)beam + 2;
  const size_t fixed = al256(cap * 8) + 8 * al256(cap * 4) + al256(cand_cap * 4) * 3 + al256(cand_cap * 8) + al256(((size_t)cap + cand_cap) * 8);
  const uint32_t shrink = (uint32_t)std::max(1, tune().arena_shrink);
  const uint32_t floor_n = (uint32_t)beam * 16u / shrink + 2u;  // (short inputs: never less than 16 frames' worth)
  const bool opt = optimistic && g_debug_arena_frames <= 0;
  uint32_t want_pa = opt ? std::min(arena, std::max(arena / 2 / shrink, floor_n)) : arena, want_ta = arena,
           want_be = opt ? std::min(arena, std::max(arena / 4 / shrink, floor_n)) : arena;
  if (!opt && db.keep_n == n_streams && db.keep_fixed == fixed && g_debug_arena_frames <= 0) {   // (see DecoderBatch::keep_*)
    want_pa = std::max(want_pa, db.keep_pa); want_ta = std::max(want_ta, db.keep_ta); want_be = std::max(want_be, db.keep_be);
  }
  decode_cache = decode_cache && !optimistic && tune().decode_cache != 0;
  const SlabLayout l = slab_layout(fixed, want_pa, want_ta, want_be, decode_cache ? chain_cap_for(want_ta, beam) : 0u);
  db.keep_pa = l.pa_cap; db.keep_ta = l.ta_cap; db.keep_be = l.be_cap; db.keep_fixed = fixed; db.keep_n = n_streams;
  db.decode_cache = decode_cache;
  db.per_stream_fixed = fixed;
  db.slab.reserve(l.per * n_streams);
  db.host.assign(n_streams, DecStream{});
  db.pa_cap.assign(n_streams, l.pa_cap); db.ta_cap.assign(n_streams, l.ta_cap);
  uint8_t* base = db.slab.as<uint8_t>();
  for (int i = 0; i < n_streams; ++i) {
    uint8_t* p = base + l.per * i;
    DecStream& S = db.host[i];
    auto take = [&](size_t bytes) { uint8_t* r = p; p += al256(bytes); return r; };
    S.key = (uint64_t*)take(cap * 8);
    S.score = (float*)take(cap * 4); S.pb = (float*)take(cap * 4); S.pnb = (float*)take(cap * 4);
    S.ch = (uint32_t*)take(cap * 4); S.node = (uint32_t*)take(cap * 4); S.ts = (uint32_t*)take(cap * 4); S.fst = (int*)take(cap * 4);
    S.bnd = (uint32_t*)take(cap * 4);
    S.c_logp = (float*)take(cand_cap * 4); S.c_pi = (uint32_t*)take(cand_cap * 4); S.c_fst = (int*)take(cand_cap * 4);
    S.c_key = (uint64_t*)take(cand_cap * 8); S.sel_keys = (uint64_t*)take(((size_t)cap + cand_cap) * 8);
    point_arenas(S, base + l.per * i, l);
    S.cand_cap = cand_cap;
    // no node has been on a decoded path yet, the cached path is empty (the slab may be a parked stream's: whatever it holds is stale)
    if (l.chain_cap) HIP_CHECK(hipMemsetAsync(base + l.per * i + l.o_dpd, 0, l.per - l.o_dpd, stream));
  }
  if (staging) {
    staging->reserve(sizeof(DecStream) * n_streams);
    memcpy(staging->p, db.host.data(), sizeof(DecStream) * n_streams);
    db.table.reserve(sizeof(DecStream) * n_streams);
    copy_h2d(db.table.p, *staging, sizeof(DecStream) * n_streams, stream);
  } else {
    db.table.upload(db.host.data(), sizeof(DecStream) * n_streams, stream);
  }
  launch_ctc_init(db.table.as<DecStream>(), n_streams, sc ? &sc->dev : nullptr, stream);
}

// Streaming use: make sure stream i can append `more_frames[i]` further timesteps.  Grows the whole slab
// (copying the live state) when an arena would overflow; rare (capacity doubles).
void ModelState::decoder_reserve(DecoderBatch& db, const std::vector<int>& more_frames, hipStream_t st_in) {
  if (g_debug_arena_frames > 0) return;  // test hook: no growth
  hipStream_t stream = st_in ? st_in : this->stream;   // (shadows the member on purpose: everything below runs on the caller's stream)
  HIP_CHECK(hipMemcpyAsync(db.host.data(), db.table.p, sizeof(DecStream) * db.n_streams, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  bool grow = false;
  uint32_t need_pa = 0, need_ta = 0, need_be = 0;
  for (int i = 0; i < db.n_streams; ++i) {
    const uint32_t add = (uint32_t)(more_frames[i] + 1) * db.beam + 2;  // this chunk's worst case, per arena
    const DecStream& S = db.host[i];
    if (S.pa_n + add > S.pa_cap || S.ta_n + add > S.ta_cap || S.be_n + add > S.be_cap) grow = true;
    need_pa = std::max(need_pa, S.pa_n + add); need_ta = std::max(need_ta, S.ta_n + add); need_be = std::max(need_be, S.be_n + add);
  }
  if (!grow) return;
  if (tune().dump_marks) fprintf(stderr, "ARENA GROW %d streams: need path %u time %u entries %u (caps %u %u %u)\n", db.n_streams, need_pa, need_ta, need_be, db.host[0].pa_cap, db.host[0].ta_cap, db.host[0].be_cap);
  const DecStream& S0 = db.host[0];
  const SlabLayout ol = slab_layout(db.per_stream_fixed, S0.pa_cap, S0.ta_cap, S0.be_cap, db.decode_cache ? S0.chain_cap : 0u);
  // only the arena that is short grows (to twice what it needs now); the others keep their size
  const uint32_t new_ta = need_ta > S0.ta_cap ? need_ta * 2 : S0.ta_cap;
  const SlabLayout nl = slab_layout(db.per_stream_fixed, need_pa > S0.pa_cap ? need_pa * 2 : S0.pa_cap, new_ta,
                                    need_be > S0.be_cap ? need_be * 2 : S0.be_cap, db.decode_cache ? std::max(S0.chain_cap, chain_cap_for(new_ta, db.beam)) : 0u);
  DevBuf ns;
  ns.reserve(nl.per * db.n_streams);
  std::vector<DecStream> nh = db.host;
  for (int i = 0; i < db.n_streams; ++i) {
    uint8_t* ob = db.slab.as<uint8_t>() + ol.per * i;
    uint8_t* nb = ns.as<uint8_t>() + nl.per * i;
    HIP_CHECK(hipMemcpyAsync(nb, ob, db.per_stream_fixed, hipMemcpyDeviceToDevice, stream));
    const ptrdiff_t delta = nb - ob;
    DecStream& S = nh[i];
    auto mv = [&](auto*& ptr) { ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<uint8_t*>(ptr) + delta); };
    mv(S.key); mv(S.score); mv(S.pb); mv(S.pnb); mv(S.ch); mv(S.node); mv(S.ts); mv(S.fst); mv(S.bnd);
    mv(S.c_logp); mv(S.c_pi); mv(S.c_fst); mv(S.c_key); mv(S.sel_keys);
    const DecStream& O = db.host[i];
    point_arenas(S, nb, nl);
    HIP_CHECK(hipMemcpyAsync(S.pa, O.pa, (size_t)std::min(O.pa_n, O.pa_cap) * 8, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(S.ta, O.ta, (size_t)std::min(O.ta_n, O.ta_cap) * 8, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(S.pq, O.pq, (size_t)std::min(O.pa_n, O.pa_cap) * 4, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(S.be, O.be, (size_t)std::min(O.be_n, O.be_cap) * sizeof(BEntry), hipMemcpyDeviceToDevice, stream));
    if (nl.chain_cap) {   // the cached best path and the depths of the nodes on it move along (new room: no node seen yet)
      HIP_CHECK(hipMemsetAsync(nb + nl.o_dpd, 0, nl.per - nl.o_dpd, stream));
      HIP_CHECK(hipMemcpyAsync(S.dpd, O.dpd, (size_t)std::min(O.pa_n, O.pa_cap) * 4, hipMemcpyDeviceToDevice, stream));
      HIP_CHECK(hipMemcpyAsync(S.dtd, O.dtd, (size_t)std::min(O.ta_n, O.ta_cap) * 4, hipMemcpyDeviceToDevice, stream));
      HIP_CHECK(hipMemcpyAsync(S.chain, O.chain, 8, hipMemcpyDeviceToDevice, stream));                     // the two lengths
      for (int a4 = 0; a4 < 4; ++a4)   // tokens, their nodes, timesteps, their nodes (the scratch quarter-blocks hold nothing between decodes)
        HIP_CHECK(hipMemcpyAsync(S.chain + 2 + (size_t)a4 * nl.chain_cap, O.chain + 2 + (size_t)a4 * ol.chain_cap, (size_t)ol.chain_cap * 4, hipMemcpyDeviceToDevice, stream));
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  std::swap(db.slab.p, ns.p); std::swap(db.slab.cap, ns.cap);
  db.keep_pa = nl.pa_cap; db.keep_ta = nl.ta_cap; db.keep_be = nl.be_cap;
  db.host = nh;
  db.table.upload(db.host.data(), sizeof(DecStream) * db.n_streams, stream);
}

// A stream whose search state overflowed an arena (or lost an invariant) has a damaged beam: no transcript is better than
// a wrong one.  The callers' guarded() turns this into NULL / STT_ERR_FAIL_RUN_SESS.
void check_decoder_errors(const int* errors, int n) {
  int err = 0;
  for (int i = 0; i < n; ++i) err |= errors[i];
  if (err) {
    char hex[16];
    snprintf(hex, sizeof(hex), "0x%x", (unsigned)err);
    throw std::runtime_error(std::string("decoder state error bits ") + hex + " (0x1 path arena, 0x2 time arena, 0x4 candidates, 0x8 scorer cache, 0x10 intra-workgroup counter wait timed out, 0x20 two prefixes with one path key)");
  }
}

hipStream_t ModelState::decoder_stream() {
  const int n = std::min(tune().decoder_streams, (tune().debug_scribble & 2) ? (int)kDecoderStreamsDebug : (int)kDecoderStreams);
  if (n <= 1) return stream;      // (the default: every decoder on the model's own stream, as in rounds 1 - 5)
  std::lock_guard<std::mutex> lk(decoder_stream_mu_);
  hipStream_t& s = decoder_streams_[decoder_stream_next_++ % (unsigned)n];
  if (!s) {
    create_engine_stream(&s, 3);
    if (tune().debug_scribble & 64) { launch_debug_scribble(s, 1 | 8 | 16); HIP_CHECK(hipStreamSynchronize(s)); }   // (experiment: the queue's scratch sized once, when the stream is made)
  }
  return s;
}

std::vector<std::vector<Output>> decode_streams(const ModelState& mc, const DecoderBatch& db, std::shared_ptr<ScorerDev> sc,
                                                const std::map<std::string, float>& hot, HotTables& ht, unsigned num_results, int max_len) {
  ModelState& m = const_cast<ModelState&>(mc);  // workspaces only
  return decode_table(m, db.table.as<DecStream>(), db.n_streams, db.beam, db.C, sc, hot, ht, num_results, max_len);
}

std::vector<std::vector<Output>> decode_table(ModelState& m, const DecStream* d_table, int n, int beam, int C, std::shared_ptr<ScorerDev> sc,
                                              const std::map<std::string, float>& hot, HotTables& ht, unsigned num_results, int max_len,
                                              hipStream_t st, DevBuf* ws, PinnedBuf* ho) {
  const int nr = (int)std::max(1u, std::min<unsigned>(num_results, (unsigned)beam));
  const DecodeBlock blk = DecodeBlock::layout(n, nr, max_len);
  DevBuf& ws_out = ws ? *ws : m.ws_out;
  PinnedBuf& h_out = ho ? *ho : m.h_out;
  hipStream_t stream = st ? st : m.stream;
  ws_out.reserve(blk.bytes); h_out.reserve(blk.bytes);
  const DecodeOut o = blk.view(ws_out.p, nr, max_len);
  DecParams p{};
  p.C = C; p.blank = C - 1; p.beam = beam; p.cutoff_top_n = 40; p.cutoff_prob = 1.0; p.t_max = 0;
  DevScorer ds = m.current_scorer(sc, hot, ht);
  launch_ctc_decode(p, ds, m.dev_alphabet, d_table, n, o, stream);
  copy_d2h(h_out, ws_out.p, blk.bytes, stream);  // one block, page-locked destination
  HIP_CHECK(hipStreamSynchronize(stream));
  const DecodeOut h = blk.view(h_out.p, nr, max_len);
  const uint32_t *tok = h.tokens, *ts = h.timesteps;
  const int *lens = h.lens, *nres = h.n_results;
  const double* conf = h.confidence;
  check_decoder_errors(h.errors, n);
  std::vector<std::vector<Output>> out(n);
  for (int i = 0; i < n; ++i) {
    for (int r = 0; r < nres[i]; ++r) {
      Output ou;
      const size_t ob = (size_t)i * nr + r;
      const int len = std::min(lens[ob], max_len);
      ou.confidence = conf[ob];
      // the kernel walked each chain once, newest entry first, into ring slot k % max_len; the (oldest) `len` kept
      // entries come back in order here: entry j is k = total-1-j
      const int total = lens[ob];
      ou.tokens.resize(len); ou.timesteps.resize(len);
      for (int j = 0; j < len; ++j) {
        const size_t slot = ob * max_len + (size_t)((total - 1 - j) % max_len);
        ou.tokens[j] = tok[slot]; ou.timesteps[j] = ts[slot];
      }
      out[i].push_back(std::move(ou));
    }
  }
  return out;
}

// ------------------------------------------------------------------------------------------- streaming
// Equivalent of the three nested buffers of stt.cc:105-334, expressed in counts:
//   frames_        MFCC frames pushed (starts at n_context zero frames, stt.cc:533)
//   windows ready  = frames_ - 2*n_context  (a 19-frame window completes with every frame beyond the 18th)
//   windows_done_  windows already sent through the model in batches of n_steps
void StreamingState::recycle() {
  scorer_.reset(); hot_words_.clear(); hot_tables_.valid = false; beam_width_ = 0; keep_emissions_ = false;
  audio_buffer_.clear(); frames_ = 0; windows_done_ = 0; state_nonzero = false; flushed_ = false; arena_bound_[0] = arena_bound_[1] = arena_bound_[2] = 2; probs_.clear();
}
void StreamingState::pushZeroFrames(int n) {
  ModelState& m = *model_;
  const int need = frames_ + n;
  if (need > frames_cap) { frames_cap = std::max(need * 2, 256); d_frames.reserve((size_t)frames_cap * m.g.n_input * 4, true, m.stream); }
  HIP_CHECK(hipMemsetAsync(d_frames.as<float>() + (size_t)frames_ * m.g.n_input, 0, (size_t)n * m.g.n_input * 4, m.stream));
  frames_ += n;
}

// span = int16 samples covering n_new_frames windows starting every win_step (the last one may be short -> zero padded)
void StreamingState::pushFrames(const int16_t* span, int n_span, int n_new_frames) {
  ModelState& m = *model_;
  const int need = frames_ + n_new_frames;
  if (need > frames_cap) { frames_cap = std::max(need * 2, 256); d_frames.reserve((size_t)frames_cap * m.g.n_input * 4, true, m.stream); }
  m.ws_audio.reserve((size_t)std::max(n_span, 1) * 2);
  if (n_span) {  // span is a host temporary: stage it in page-locked memory so that the feed need not wait for the copy
    const unsigned slot = m.audio_slot++ & 3u;
    if (!m.ev_audio[slot]) HIP_CHECK(hipEventCreateWithFlags(&m.ev_audio[slot], hipEventDisableTiming));
    else HIP_CHECK(hipEventSynchronize(m.ev_audio[slot]));
    m.h_audio[slot].reserve((size_t)n_span * 2);
    memcpy(m.h_audio[slot].p, span, (size_t)n_span * 2);
    copy_h2d(m.ws_audio.p, m.h_audio[slot], (size_t)n_span * 2, m.stream);
    HIP_CHECK(hipEventRecord(m.ev_audio[slot], m.stream));
  }
  MfccArgs a = m.mfcc_args();
  a.audio = m.ws_audio.as<int16_t>(); a.n_samples = nullptr; a.n_frames = nullptr;
  a.all_n_samples = n_span; a.all_n_frames = n_new_frames;  // (in the kernel arguments: no table upload, no host sync)
  a.feats = d_frames.as<float>() + (size_t)frames_ * m.g.n_input; a.n_max = std::max(n_span, 1); a.t_max = n_new_frames;
  launch_mfcc(a, n_new_frames, m.stream);
  frames_ += n_new_frames;
}

void StreamingState::feedAudioContent(const short* buffer, unsigned int buffer_size) {
  const Geometry& g = model_->g;
  if (flushed_) {   // the stream's last audio went in with aLast (include/stt_amd.h): only decode and finish are meaningful now.  Audio fed anyway
    processReady(true, true);   // is ignored -- appending frames behind the trailing zero-context frames would silently change the transcript --
    return;                     // and what a deferred tail left is drained
  }
  // identical to filling audio_buffer_ sample by sample and firing a window whenever it holds win_len samples (stt.cc:105-128)
  std::vector<int16_t> all(audio_buffer_);
  all.insert(all.end(), buffer, buffer + buffer_size);
  const int len = (int)all.size();
  const int W = len >= g.win_len ? (len - g.win_len) / g.win_step + 1 : 0;
  if (W > 0) {
    pushFrames(all.data(), (W - 1) * g.win_step + g.win_len, W);
    audio_buffer_.assign(all.begin() + (size_t)W * g.win_step, all.end());
    processReady(false, false);
  } else {
    audio_buffer_.swap(all);
  }
}

void StreamingState::flushBuffers(bool addZeroMfccVectors) {
  if (flushed_) {                          // its last audio came through STTX_FeedAudioContentBatchEx with the last flag: already flushed ...
    processReady(true, true);              // ... except for a tail that was deferred and has not ridden along yet.  Also for the
    return;                                // ...FlushBuffers decodes (addZeroMfccVectors false): no further partial-window frame behind the final flush
  }
  if (addZeroMfccVectors) flushed_ = true;
  // stt.cc:236-254: the partial audio window goes through the feature graph as is (zero padded), audio_buffer_ is kept
  pushFrames(audio_buffer_.data(), (int)audio_buffer_.size(), 1);
  if (addZeroMfccVectors) pushZeroFrames(model_->g.n_context);
  processReady(true, addZeroMfccVectors);
}

// Runs every complete batch of n_steps windows; with flush_partial also the remaining partial batch
// (zero padded to n_steps through the LSTM exactly like tflitemodelstate.cc:341-355 unless it is the final flush,
// where the padded steps cannot influence anything that is still observable).
void StreamingState::processReady(bool flush_partial, bool final_flush) {
  ModelState& m = *model_;
  const Geometry& g = m.g;
  const int H = g.n_hidden, C = g.n_classes, kp = m.x1_cols(), kw = g.n_in1();
  for (;;) {
    const int ready = std::max(0, frames_ - 2 * g.n_context) - windows_done_;
    int take = 0;
    if (ready >= g.n_steps) take = g.n_steps;
    else if (flush_partial && ready > 0) take = ready;
    if (take == 0) break;
    const int T = (take < g.n_steps && !final_flush) ? g.n_steps : take;  // padded steps perturb the carried state (coqui-stt.h:393-399 of the reference)
    // windows are contiguous slices of the frame list: window w = frames[w .. w+19) flattened (stt.cc:292-309)
    m.ws_x1.reserve(m.x1_bytes(T));
    // raw gather: x1[t][k] = frames_flat[(windows_done_ + t) * n_input + k], k < 494; rows >= take are written as zeros
    launch_window_rows(d_frames.as<float>() + (size_t)windows_done_ * g.n_input, m.ws_x1.p, take, T, g.n_input, kw, kp, m.stream, m.i8);
    d_c.reserve((size_t)H * 4); d_h.reserve((size_t)H * 4);
    m.ws_probs.reserve((size_t)T * C * 4);
    m.run_acoustic_rows(m.ws_x1.p, 1, T, d_c.as<float>(), d_h.as<float>(), state_nonzero, m.ws_probs.as<float>(), T);
    state_nonzero = true;
    if (keep_emissions_) {  // stt.cc:326-329: probs_ is *replaced* by the last batch
      std::vector<float> pr((size_t)take * C);
      HIP_CHECK(hipMemcpyAsync(pr.data(), m.ws_probs.p, pr.size() * 4, hipMemcpyDeviceToHost, m.stream));
      HIP_CHECK(hipStreamSynchronize(m.stream));
      probs_.assign(pr.begin(), pr.end());
    }
    // decoder_state_.next(inputs, n_frames, num_classes)
    reserveArena(take);
    DecParams p{};
    p.C = C; p.blank = C - 1; p.beam = dec.beam; p.cutoff_top_n = 40; p.cutoff_prob = 1.0; p.t_max = T;  // stt.cc:539-540
    p.all_begin = 0; p.all_count = take;  // the frame range rides in the kernel arguments: no table upload, no host sync mid-hop
    DevScorer ds = m.current_scorer(scorer_, hot_words_, hot_tables_);
    m.ws_wide.reserve(ctc_rows_ws_bytes(p, 1, take));
    launch_ctc_next(p, ds, m.dev_alphabet, dec.table.as<DecStream>(), 1, m.ws_probs.as<float>(), nullptr, nullptr, m.stream,
                    take, m.ws_wide.p);
    windows_done_ += take;
  }
}

// Room for `take` more timesteps in the arenas.  The exact fill lives on the device; a host-side bound (every step appends
// at most beam nodes) avoids reading it back on every chunk.
void StreamingState::reserveArena(int take) {
  const uint32_t add = (uint32_t)(take + 1) * (uint32_t)dec.beam + 2;
  const DecStream& S = dec.host[0];
  if (arena_bound_[0] + add > S.pa_cap || arena_bound_[1] + add > S.ta_cap || arena_bound_[2] + add > S.be_cap) {
    model_->decoder_reserve(dec, std::vector<int>{take});  // reads the table back, grows what is short
    arena_bound_[0] = dec.host[0].pa_n; arena_bound_[1] = dec.host[0].ta_n; arena_bound_[2] = dec.host[0].be_n;  // (the real fill: far below the bound)
  }
  for (uint32_t& b : arena_bound_) b += (uint32_t)take * (uint32_t)dec.beam;
}

// ------------------------------------------------------------------------------------------- batched streaming
bool streams_batchable(const std::vector<StreamingState*>& ss) {
  if (ss.empty()) return false;
  const StreamingState* a = ss[0];
  for (const StreamingState* s : ss)
    if (!s || s->model_ != a->model_ || s->beam_width_ != a->beam_width_ || s->scorer_ != a->scorer_ || s->hot_words_ != a->hot_words_ || s->keep_emissions_)
      return false;
  for (size_t i = 0; i < ss.size(); ++i)
    for (size_t j = i + 1; j < ss.size(); ++j)
      if (ss[i] == ss[j]) return false;
  return true;
}

namespace {
// One read-back for ALL streams whose host-side arena bound says "may not fit" (the bound adds `beam` nodes per step, the real fill
// is a fraction of it): their DecStream entries are gathered into one block and copied with one transfer.  Stream by stream
// (StreamingState::reserveArena) this was a table read-back + synchronisation EACH -- 128 streams that started together all reach
// their bound in the same hop: 8 ms in one hop of a 128-stream set (benchmarks/stream_rolling.py).
void streams_check_arenas(ModelState& m, const std::vector<StreamingState*>& R, const std::vector<int>& takes) {
  std::vector<int> idx;
  for (size_t i = 0; i < R.size(); ++i) {
    StreamingState* s = R[i];
    const uint32_t add = (uint32_t)(takes[i] + 1) * (uint32_t)s->dec.beam + 2;
    const DecStream& S = s->dec.host[0];
    if (s->arena_bound_[0] + add > S.pa_cap || s->arena_bound_[1] + add > S.ta_cap || s->arena_bound_[2] + add > S.be_cap) idx.push_back((int)i);
  }
  if (idx.empty()) return;
  if (tune().dump_marks) fprintf(stderr, "ARENA CHECK %zu of %zu streams\n", idx.size(), R.size());
  const size_t n = idx.size();
  m.sb_htab3.reserve(n * 8); m.sb_tab3.reserve(n * 8);
  void** hp = m.sb_htab3.as<void*>();
  for (size_t k = 0; k < n; ++k) hp[k] = R[idx[k]]->dec.table.p;
  HIP_CHECK(hipMemcpyAsync(m.sb_tab3.p, hp, n * 8, hipMemcpyHostToDevice, m.stream));
  m.sb_table.reserve(sizeof(DecStream) * n);
  launch_gather_streams(reinterpret_cast<const DecStream* const*>(m.sb_tab3.p), m.sb_table.as<DecStream>(), (int)n, m.stream);
  std::vector<DecStream> fill(n);
  HIP_CHECK(hipMemcpyAsync(fill.data(), m.sb_table.p, sizeof(DecStream) * n, hipMemcpyDeviceToHost, m.stream));
  HIP_CHECK(hipStreamSynchronize(m.stream));
  for (size_t k = 0; k < n; ++k) {
    StreamingState* s = R[idx[k]];
    s->arena_bound_[0] = fill[k].pa_n; s->arena_bound_[1] = fill[k].ta_n; s->arena_bound_[2] = fill[k].be_n;   // the real fill
  }
}

// runs every batch of n_steps windows that is ready in any of the streams (and the partial ones of the streams that flush);
// returns whether the stream was synchronised after the last launch (the page-locked staging of the caller is then free again)
bool streams_process(const std::vector<StreamingState*>& ss, bool flush_partial, const std::vector<uint8_t>* flush_each = nullptr) {
  bool synced = false;
  ModelState& m = *ss[0]->model_;
  const Geometry& g = m.g;
  const int H = g.n_hidden, C = g.n_classes, kp = m.x1_cols(), kw = g.n_in1(), T = g.n_steps;
  DevScorer ds = m.current_scorer(ss[0]->scorer_, ss[0]->hot_words_, ss[0]->hot_tables_);
  for (int pass = 0;; ++pass) {
    std::vector<StreamingState*> R;
    std::vector<int> takes;
    for (size_t si = 0; si < ss.size(); ++si) {
      StreamingState* s = ss[si];
      // flush_each: 1 = this stream's partial batch of windows goes through as well; 2 = only in the first pass of this call -- what
      // is left after it waits for the stream's next call (or its finish): no pass of its own for a handful of rows
      const uint8_t fe = flush_each ? (*flush_each)[si] : 0;
      const bool fl = flush_partial || fe == 1 || (fe == 2 && pass == 0);
      const int ready = std::max(0, s->frames_ - 2 * g.n_context) - s->windows_done_;
      const int take = ready >= T ? T : ((fl && ready > 0) ? ready : 0);
      if (take) { R.push_back(s); takes.push_back(take); }
    }
    if (R.empty()) break;
    streams_check_arenas(m, R, takes);
    const size_t rows = (size_t)lstm_max_rows(H);  // streams advanced by one recurrent launch: 128 with 16 units per workgroup, else 64
    for (size_t g0 = 0; g0 < R.size(); g0 += rows) {
      const int B = (int)std::min<size_t>(rows, R.size() - g0);
      for (int b = 0; b < B; ++b) {
        StreamingState* s = R[g0 + b];
        s->d_c.reserve((size_t)H * 4); s->d_h.reserve((size_t)H * 4);
        s->reserveArena(takes[g0 + b]);
      }
      // one page-locked table: [frames ptr | c ptr | h ptr | stream-table ptr] (8 bytes each) then [win_off | take | zero] ints, then valid bytes
      const size_t n_ptr = (size_t)4 * B, bytes = n_ptr * 8 + (size_t)3 * B * 4 + B;
      m.sb_htab.reserve(bytes); m.sb_tab.reserve(bytes);
      uint8_t* hb = m.sb_htab.as<uint8_t>();
      void** hp = reinterpret_cast<void**>(hb);
      int* hi = reinterpret_cast<int*>(hb + n_ptr * 8);
      uint8_t* hv = hb + n_ptr * 8 + (size_t)3 * B * 4;
      for (int b = 0; b < B; ++b) {
        StreamingState* s = R[g0 + b];
        hp[b] = s->d_frames.p; hp[B + b] = s->d_c.p; hp[2 * B + b] = s->d_h.p; hp[3 * B + b] = s->dec.table.p;
        hi[b] = s->windows_done_; hi[B + b] = takes[g0 + b]; hi[2 * B + b] = 0;
        hv[b] = s->state_nonzero ? 1 : 0;
      }
      uint8_t* db = m.sb_tab.as<uint8_t>();
      const float* const* d_frames = reinterpret_cast<const float* const*>(db);
      float* const* d_cp = reinterpret_cast<float* const*>(db + (size_t)B * 8);
      float* const* d_hp = reinterpret_cast<float* const*>(db + (size_t)2 * B * 8);
      DecStream* const* d_tp = reinterpret_cast<DecStream* const*>(db + (size_t)3 * B * 8);
      const int* d_int = reinterpret_cast<const int*>(db + n_ptr * 8);
      const unsigned char* d_valid = db + n_ptr * 8 + (size_t)3 * B * 4;
      DecParams p{};
      p.C = C; p.blank = C - 1; p.beam = R[g0]->dec.beam; p.cutoff_top_n = 40; p.cutoff_prob = 1.0; p.t_max = T;  // stt.cc:539-540
      // everything the pass enqueues: the table upload (from the page-locked block filled above), windows, state gather, the acoustic
      // model, state scatter, the search.  Nothing in it depends on this hop except through that table.
      auto enqueue = [&]() {
        HIP_CHECK(hipMemcpyAsync(m.sb_tab.p, hb, bytes, hipMemcpyHostToDevice, m.stream));
        m.ws_x1.reserve(m.x1_bytes(T * B));
        launch_window_rows_batch(d_frames, d_int, d_int + B, m.ws_x1.p, B, T, g.n_input, kw, kp, m.stream, m.i8);
        m.sb_c.reserve((size_t)B * H * 4); m.sb_h.reserve((size_t)B * H * 4);
        launch_gather_rows(d_cp, d_valid, m.sb_c.as<float>(), B, H, m.stream);
        launch_gather_rows(d_hp, d_valid, m.sb_h.as<float>(), B, H, m.stream);
        m.ws_probs.reserve((size_t)B * T * C * 4);
        m.run_acoustic_rows(m.ws_x1.p, B, T, m.sb_c.as<float>(), m.sb_h.as<float>(), true, m.ws_probs.as<float>(), T);
        launch_scatter_rows(d_cp, m.sb_c.as<float>(), B, H, m.stream);
        launch_scatter_rows(d_hp, m.sb_h.as<float>(), B, H, m.stream);
        m.sb_table.reserve(sizeof(DecStream) * B);
        launch_gather_streams(d_tp, m.sb_table.as<DecStream>(), B, m.stream);
        m.ws_wide.reserve(ctc_rows_ws_bytes(p, B, T));
        launch_ctc_next(p, ds, m.dev_alphabet, m.sb_table.as<DecStream>(), B, m.ws_probs.as<float>(), d_int + 2 * B, d_int + B, m.stream, T, m.ws_wide.p);
        launch_scatter_streams(d_tp, m.sb_table.as<DecStream>(), B, m.stream);
      };
      bool replayed = false;
      if (tune().stream_graph && !m.prof_.on) {
        // key: the shape of the live set and everything baked into the launches by VALUE (the scorer description, the search parameters);
        // addresses are covered by layout_generation().  First sighting: launched one by one (allocations, function attributes, module
        // loads happen there, never inside a capture); second: captured; from then on: one graph launch per hop.
        if (m.hop_generation_ != layout_generation()) {
          for (auto& kv : m.hop_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
          m.hop_graphs_.clear(); m.hop_seen_.clear();
          m.hop_generation_ = layout_generation();
        }
        uint64_t key = 0xCBF29CE484222325ULL;
        auto mix = [&](const void* q, size_t nb) { const uint8_t* b8 = static_cast<const uint8_t*>(q); for (size_t k = 0; k < nb; ++k) { key ^= b8[k]; key *= 0x100000001B3ULL; } };
        const int shape[6] = {B, T, p.beam, C, m.i8 ? 1 : 0, (int)bytes};
        mix(shape, sizeof(shape)); mix(&ds, sizeof(ds)); mix(&m.dev_alphabet, sizeof(m.dev_alphabet));
        auto found = m.hop_graphs_.find(key);
        if (found != m.hop_graphs_.end()) {
          if (found->second.exec) { HIP_CHECK(hipGraphLaunch(found->second.exec, m.stream)); replayed = true; __atomic_fetch_add(&tune().hop_replays, 1, __ATOMIC_RELAXED); }
        } else if (!m.hop_seen_.count(key)) {
          if (m.hop_seen_.size() >= 512) m.hop_seen_.clear();
          m.hop_seen_.insert(key);
        } else {
          const unsigned long long gen0 = layout_generation();
          ModelState::HopGraph gr;
          hipGraph_t graph = nullptr;
          std::lock_guard<std::recursive_mutex> capturing(hip_capture_mutex());   // (two cohorts on two threads: the other one may be growing a buffer right now)
          HIP_CHECK(hipStreamBeginCapture(m.stream, hipStreamCaptureModeRelaxed));
          try { enqueue(); }
          catch (...) { (void)hipStreamEndCapture(m.stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
          HIP_CHECK(hipStreamEndCapture(m.stream, &graph));
          if (layout_generation() != gen0 || hipGraphInstantiate(&gr.exec, graph, nullptr, nullptr, 0) != hipSuccess) { gr.exec = nullptr; (void)hipGetLastError(); }   // (a buffer moved inside the capture: not a graph to keep)
          (void)hipGraphDestroy(graph);
          if (m.hop_graphs_.size() >= 128) { for (auto& kv : m.hop_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec); m.hop_graphs_.clear(); }
          m.hop_graphs_[key] = gr;
          m.hop_seen_.erase(key);
          if (gr.exec) { HIP_CHECK(hipGraphLaunch(gr.exec, m.stream)); replayed = true; }
        }
      }
      if (!replayed) enqueue();
      HIP_CHECK(hipStreamSynchronize(m.stream));  // the page-locked table is reused by the next group
      synced = true;
      for (int b = 0; b < B; ++b) { R[g0 + b]->windows_done_ += takes[g0 + b]; R[g0 + b]->state_nonzero = true; }
    }
  }
  return synced;
}

// MFCC of the new windows of every stream in one launch, written straight behind each stream's frame list
void streams_push_frames(const std::vector<StreamingState*>& ss, const std::vector<std::vector<int16_t>>& spans, const std::vector<int>& n_frames) {
  ModelState& m = *ss[0]->model_;
  const Geometry& g = m.g;
  const int n = (int)ss.size();
  int max_span = 1, max_w = 0;
  for (int i = 0; i < n; ++i) { max_span = std::max(max_span, (int)spans[i].size()); max_w = std::max(max_w, n_frames[i]); }
  if (max_w == 0) return;
  for (int i = 0; i < n; ++i) {
    StreamingState* s = ss[i];
    const int need = s->frames_ + n_frames[i];
    if (need > s->frames_cap) { s->frames_cap = std::max(need * 2, 256); s->d_frames.reserve((size_t)s->frames_cap * g.n_input * 4, true, m.stream); }
  }
  m.sb_haudio.reserve((size_t)n * max_span * 2); m.sb_audio.reserve((size_t)n * max_span * 2);
  int16_t* ha = m.sb_haudio.as<int16_t>();
  memset(ha, 0, (size_t)n * max_span * 2);
  for (int i = 0; i < n; ++i) if (!spans[i].empty()) memcpy(ha + (size_t)i * max_span, spans[i].data(), spans[i].size() * 2);
  HIP_CHECK(hipMemcpyAsync(m.sb_audio.p, ha, (size_t)n * max_span * 2, hipMemcpyHostToDevice, m.stream));
  // (its own table: the acoustic pass that follows fills sb_htab / sb_tab while this one may still be in flight -- the caller
  // synchronises once, behind that pass)
  const size_t bytes = (size_t)n * 8 + (size_t)2 * n * 4;
  m.sb_htab2.reserve(bytes); m.sb_tab2.reserve(bytes);
  uint8_t* hb = m.sb_htab2.as<uint8_t>();
  void** hp = reinterpret_cast<void**>(hb);
  int* hi = reinterpret_cast<int*>(hb + (size_t)n * 8);
  for (int i = 0; i < n; ++i) {
    hp[i] = ss[i]->d_frames.as<float>() + (size_t)ss[i]->frames_ * g.n_input;
    hi[i] = (int)spans[i].size(); hi[n + i] = n_frames[i];
  }
  HIP_CHECK(hipMemcpyAsync(m.sb_tab2.p, hb, bytes, hipMemcpyHostToDevice, m.stream));
  MfccArgs a = m.mfcc_args();
  a.audio = m.sb_audio.as<int16_t>(); a.n_max = max_span; a.t_max = max_w;
  a.feats = nullptr; a.feats_ptrs = reinterpret_cast<float* const*>(m.sb_tab2.p);
  a.n_samples = reinterpret_cast<const int*>(m.sb_tab2.as<uint8_t>() + (size_t)n * 8); a.n_frames = a.n_samples + n;
  launch_mfcc(a, n * max_w, m.stream);
  for (int i = 0; i < n; ++i) ss[i]->frames_ += n_frames[i];
}
}  // namespace

// `last` (or null): last[i] != 0 = this is the final audio of stream i -- what StreamingState::flushBuffers(true) would do in an acoustic
// pass of its own (the partial window as one more frame, n_context zero frames, the partial batch of windows: stt.cc:236-254) rides in
// THIS pass, beside the live streams' rows; the stream is marked flushed and its finish only decodes.
void streams_feed_batch(const std::vector<StreamingState*>& ss, const short* const* buffers, const unsigned int* sizes, const unsigned char* last) {
  ModelState& m = *ss[0]->model_;
  const Geometry& g = m.g;
  const int n = (int)ss.size();
  std::vector<std::vector<int16_t>> spans(n);
  std::vector<int> nf(n, 0);
  std::vector<uint8_t> fl(n, 0);
  bool any_last = false;
  for (int i = 0; i < n; ++i) {  // the window arithmetic of StreamingState::feedAudioContent, per stream
    StreamingState* s = ss[i];
    if (s->flushed_) { fl[i] = 2; any_last = true; continue; }   // flushed by an earlier call that deferred its tail (last = 2): the tail rides in this pass; audio passed with it is ignored
    std::vector<int16_t> all(s->audio_buffer_);
    all.insert(all.end(), buffers[i], buffers[i] + sizes[i]);
    const int len = (int)all.size();
    const int W = len >= g.win_len ? (len - g.win_len) / g.win_step + 1 : 0;
    const bool is_last = last && last[i] && !s->flushed_;
    if (is_last) {   // W full windows and the partial one behind them (zero padded by the feature kernel), from one span
      fl[i] = last[i] == 2 ? 2 : 1; any_last = true;
      spans[i] = all;
      s->audio_buffer_.assign(all.begin() + std::min((size_t)len, (size_t)W * g.win_step), all.end());   // (flushBuffers keeps it: stt.cc:236-254)
      nf[i] = W + 1;
    } else if (W > 0) {
      spans[i].assign(all.begin(), all.begin() + (size_t)(W - 1) * g.win_step + g.win_len);
      s->audio_buffer_.assign(all.begin() + (size_t)W * g.win_step, all.end());
      nf[i] = W;
    } else {
      s->audio_buffer_.swap(all);
    }
  }
  streams_push_frames(ss, spans, nf);
  if (any_last)
    for (int i = 0; i < n; ++i) if (fl[i] && !ss[i]->flushed_) { ss[i]->pushZeroFrames(g.n_context); ss[i]->flushed_ = true; }
  if (!streams_process(ss, false, any_last ? &fl : nullptr)) HIP_CHECK(hipStreamSynchronize(m.stream));   // page-locked staging is reused by the next call
}

void streams_flush_batch(const std::vector<StreamingState*>& ss_all, bool addZeroMfccVectors) {
  std::vector<StreamingState*> ss, left;
  for (StreamingState* s : ss_all) {
    if (!(addZeroMfccVectors && s->flushed_)) ss.push_back(s);   // (else: flushed with its last audio, streams_feed_batch ...
    else if (std::max(0, s->frames_ - 2 * s->model_->g.n_context) - s->windows_done_ > 0) left.push_back(s);   // ... but its tail was deferred and never rode along)
  }
  if (!left.empty() && !streams_process(left, true)) HIP_CHECK(hipStreamSynchronize(left[0]->model_->stream));
  if (ss.empty()) return;
  const int n = (int)ss.size();
  std::vector<std::vector<int16_t>> spans(n);
  std::vector<int> nf(n, 1);
  for (int i = 0; i < n; ++i) spans[i] = ss[i]->audio_buffer_;  // stt.cc:236-254: the partial window as is (zero padded)
  streams_push_frames(ss, spans, nf);
  if (addZeroMfccVectors) for (StreamingState* s : ss) { s->pushZeroFrames(s->model_->g.n_context); s->flushed_ = true; }
  if (!streams_process(ss, true)) HIP_CHECK(hipStreamSynchronize(ss[0]->model_->stream));
}

std::vector<std::vector<Output>> streams_decode_batch(const std::vector<StreamingState*>& ss, unsigned num_results) {
  ModelState& m = *ss[0]->model_;
  const int n = (int)ss.size();
  const size_t bytes = (size_t)n * 8;
  m.sb_htab.reserve(bytes); m.sb_tab.reserve(bytes);
  void** hp = m.sb_htab.as<void*>();
  int max_len = 2;
  for (int i = 0; i < n; ++i) { hp[i] = ss[i]->dec.table.p; max_len = std::max(max_len, ss[i]->windows_done_ + 1); }
  HIP_CHECK(hipMemcpyAsync(m.sb_tab.p, hp, bytes, hipMemcpyHostToDevice, m.stream));
  m.sb_table.reserve(sizeof(DecStream) * n);
  launch_gather_streams(reinterpret_cast<const DecStream* const*>(m.sb_tab.p), m.sb_table.as<DecStream>(), n, m.stream);
  return decode_table(m, m.sb_table.as<DecStream>(), n, ss[0]->dec.beam, ss[0]->dec.C, ss[0]->scorer_, ss[0]->hot_words_, ss[0]->hot_tables_, num_results, max_len);
}

std::vector<Output> StreamingState::decode(unsigned num_results) {
  auto r = decode_streams(*model_, dec, scorer_, hot_words_, hot_tables_, num_results, std::max(1, windows_done_) + 1);  // <= one token per processed window
  return r.empty() ? std::vector<Output>() : r[0];
}
