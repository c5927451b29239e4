// stt_amd/csrc/api.cpp -- the exported C ABI: coqui-stt.h (29 functions) + stt_amd.h (STTX_*).
//
// Function-by-function replacement of native_client/stt.cc:336-737, native_client/modelstate.cc:32-76 and
// native_client/stt_errors.cc:4-19.  All recognition work is enqueued on the GPU; an unusable GPU runtime or a
// missing kernel image surfaces as an error code / NULL, never as a silent CPU path.
#include <unistd.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../include/stt_amd.h"
#ifdef STT_TEST_HOOKS
#include "../../include/stt_amd_test.h"   // libstt_test.so only (stt_amd/build.py)
#endif
#include "engine.h"
#include "scorer_host.h"
#include "tuning.h"

namespace {
int g_device = 0;
const char* kVersion = "1.4.0";  // training/coqui_stt_training/VERSION of the reference this ABI mirrors

// Stage timing with HIP events on the engine's own streams.  A mark says "stage `id` starts now on this stream"; the time
// up to the next mark on the same stream is charged to that stage (id < 0 = idle / end).  Stage ids: 0 features,
// 1 dense layers 1-3 + x-projection, 2 LSTM recurrence, 3 layers 5-6 + softmax, 4 decoder next, 5 decoder decode + D2H.
// (struct Prof lives in ModelState, engine.h: two models may run on two threads)
// live models: a stream freed after its model (a caller error the reference happens to survive) must not touch the model
std::mutex g_models_mu;
std::unordered_map<ModelState*, int> g_models;

Prof& prof_of(ModelState* m) { return m->prof_; }
void mark_on(ModelState* m, int id, int which, hipStream_t st) {
  Prof& p = prof_of(m);
  if (!p.on || (p.only >= 0 && which != p.only)) return;
  if (p.used == p.pool.size()) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); p.pool.push_back(e); }
  hipEvent_t e = p.pool[p.used++];
  HIP_CHECK(hipEventRecord(e, st));
  p.marks[which].push_back({id, e});
}
void mark(ModelState* m, int id) { mark_on(m, id, 0, m->stream); }
void prof_reset(Prof& p) { p.used = 0; for (auto& mk : p.marks) mk.clear(); }
void prof_collect(Prof& p) {  // every stream must be idle
  // tunable dump_marks: the raw timeline (stream list, stage id, microseconds since the first mark of the acoustic stream) on
  // stderr -- what ran beside what, without a tracer's overhead on the host
  const bool dump = tune().dump_marks != 0;
  if (dump && !p.marks[0].empty()) {
    hipEvent_t base = p.marks[0][0].second;
    for (int w = 0; w < 7; ++w)
      for (auto& m : p.marks[w]) {
        float t = 0;
        if (hipEventElapsedTime(&t, base, m.second) == hipSuccess) fprintf(stderr, "MARK %d %d %.1f\n", w, m.first, t * 1e3f);
      }
  }
  for (auto& mk : p.marks)
    for (size_t i = 0; i + 1 < mk.size(); ++i) {
      if (mk[i].first < 0 || mk[i].first >= 6) continue;
      float t = 0;
      HIP_CHECK(hipEventElapsedTime(&t, mk[i].second, mk[i + 1].second));
      p.ms[mk[i].first] += t;
    }
  prof_reset(p);
}

template <class F> int guarded(F&& f, int fail_code) {
  try { return f(); }
  catch (const std::exception& e) { std::cerr << "stt_amd: " << e.what() << std::endl; return fail_code; }
}

// ModelState::decode_metadata, modelstate.cc:39-76
Metadata* make_metadata(const ModelState* m, const std::vector<Output>& out) {
  const unsigned num_returned = (unsigned)out.size();
  CandidateTranscript* transcripts = (CandidateTranscript*)malloc(sizeof(CandidateTranscript) * std::max(1u, num_returned));
  for (unsigned i = 0; i < num_returned; ++i) {
    const size_t nt = out[i].tokens.size();
    TokenMetadata* tokens = (TokenMetadata*)malloc(sizeof(TokenMetadata) * std::max<size_t>(1, nt));
    for (size_t j = 0; j < nt; ++j) {
      const unsigned ts = j < out[i].timesteps.size() ? out[i].timesteps[j] : 0;
      TokenMetadata token{strdup(m->alphabet_.DecodeSingle(out[i].tokens[j]).c_str()), ts,
                          ts * ((float)m->g.win_step / m->g.sample_rate)};
      memcpy(&tokens[j], &token, sizeof(TokenMetadata));
    }
    CandidateTranscript tr{tokens, (unsigned)nt, out[i].confidence};
    memcpy(&transcripts[i], &tr, sizeof(CandidateTranscript));
  }
  Metadata* ret = (Metadata*)malloc(sizeof(Metadata));
  Metadata md{transcripts, num_returned, NULL};
  memcpy(ret, &md, sizeof(Metadata));
  return ret;
}

// stt.cc:136-175: attach the emissions of the last processed batch
Metadata* with_emissions(const StreamingState* s, Metadata* m) {
  const size_t alphabet_size = s->model_->alphabet_.GetSize();
  const int num_timesteps = (int)(s->probs_.size() / (alphabet_size + 1));
  AcousticModelEmissions* em = (AcousticModelEmissions*)malloc(sizeof(AcousticModelEmissions));
  em->num_symbols = (int)alphabet_size;
  em->num_timesteps = num_timesteps;
  em->symbols = (const char**)malloc(sizeof(char*) * (alphabet_size + 1));
  for (size_t i = 0; i < alphabet_size; ++i) em->symbols[i] = strdup(s->model_->alphabet_.DecodeSingle((unsigned)i).c_str());
  em->symbols[alphabet_size] = strdup("\t");
  double* probs = (double*)malloc(sizeof(double) * std::max<size_t>(1, (alphabet_size + 1) * num_timesteps));
  memcpy(probs, s->probs_.data(), sizeof(double) * (alphabet_size + 1) * num_timesteps);
  em->emissions = probs;
  Metadata* ret = (Metadata*)malloc(sizeof(Metadata));
  Metadata md{m->transcripts, m->num_transcripts, em};
  memcpy(ret, &md, sizeof(Metadata));
  free(m);
  return ret;
}

char* decode_string(const StreamingState* cs) {  // ModelState::decode, modelstate.cc:32-37
  StreamingState* s = const_cast<StreamingState*>(cs);  // (workspaces and the uploaded hot-word table only)
  std::vector<Output> out = s->decode(1);
  if (out.empty()) return strdup("");
  return strdup(s->model_->alphabet_.Decode(out[0].tokens.data(), (int)out[0].tokens.size()).c_str());
}
Metadata* decode_metadata(const StreamingState* cs, unsigned n) {
  StreamingState* s = const_cast<StreamingState*>(cs);
  Metadata* m = make_metadata(s->model_, s->decode(n));
  return s->keep_emissions_ ? with_emissions(s, m) : m;
}

int create_stream(ModelState* aCtx, StreamingState** retval, bool keep_emissions) {  // stt.cc:519-593
  *retval = nullptr;
  return guarded([&]() {
    std::unique_ptr<StreamingState> ctx;
    {
      std::lock_guard<std::mutex> lk(aCtx->stream_pool_mu_);
      if (!aCtx->stream_pool_.empty()) { ctx.reset(aCtx->stream_pool_.back()); aCtx->stream_pool_.pop_back(); }
    }
    if (!ctx) ctx.reset(new StreamingState());
    ctx->model_ = aCtx;
    ctx->scorer_ = aCtx->scorer_;
    ctx->hot_words_ = aCtx->hot_words_;
    ctx->beam_width_ = aCtx->beam_width_;
    ctx->keep_emissions_ = keep_emissions;
    HIP_CHECK(hipSetDevice(aCtx->device));
    ctx->pushZeroFrames(aCtx->g.n_context);                         // stt.cc:533
    ctx->d_c.reserve((size_t)aCtx->g.n_hidden * 4); ctx->d_h.reserve((size_t)aCtx->g.n_hidden * 4);
    // stt.cc:535-536 (previous_state_c / _h = 0): nothing is written here -- a stream whose state_nonzero is false enters its first pass
    // with a zero state by construction (acoustic_rows: carry 0 clears the cell state and the h fragments; the batched pass gathers
    // zeros for such a row), and a recycled stream's flag was reset (StreamingState::recycle)
    aCtx->decoder_create(ctx->dec, 1, (int)aCtx->beam_width_, std::max(16, tune().stream_frames), ctx->scorer_, nullptr, false, /*decode_cache=*/true);  // arenas sized for `stream_frames` (256: 11 MB at beam 500) up front; cutoff_top_n = 40, cutoff_prob = 1.0 fixed (stt.cc:539-540)
    *retval = ctx.release();
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_CREATE_STREAM);
}

// ---- batch path: every utterance goes through exactly the arithmetic of STT_SpeechToText ---------------------
// Groups of <= 64 utterances, or <= 128 where the recurrent kernel covers 128 rows (tunable `pair`): the 33.5 MB recurrent matrix is
// then streamed once per 128 rows.  Within a group the acoustic model runs in time-chunks on `stream` and the beam search of
// chunk k runs on the group's search stream while chunk k+1 is being computed (the search only occupies one workgroup per
// utterance).  Chunk schedule of a blocking call (latency of ONE group matters): a short first chunk (tunable chunk0,
// default 16 frames) so the beam search starts early, then chunks of `chunk` (48) frames.  A group submitted to the caller-driven
// pipeline (STTX_BatchSubmitDevice: other groups fill the chip meanwhile) takes pchunk0 / pchunk.
struct ChunkPlan { int first, rest; };
ChunkPlan chunk_plan(bool pipelined) {
  ChunkPlan c{pipelined ? tune().pchunk0 : tune().chunk0, pipelined ? tune().pchunk : tune().chunk};
  if (c.rest < 1) c.rest = 1 << 30;
  if (c.first < 1) c.first = 1 << 30;
  return c;
}
// One batch handed to a group: rows idx[] of the caller's [.][stride] int16 array.  A group takes one part (blocking calls) or
// two (two submitted batches advanced together).
struct BatchPart {
  const int16_t* d_audio;
  unsigned stride;
  const unsigned* sizes;
  std::vector<unsigned> idx;
  hipEvent_t ready = nullptr;   // the audio is there once this event has been reached (STTX_BatchSubmit's copy); null: it is there now
};
// Enqueue everything one group needs, on both streams, without waiting for anything: features + acoustic chunks on
// `stream`, the beam search of every chunk + the final ranking + the copy of the results to page-locked memory on
// `stream_dec`, then the slot's `done` event.
bool search_bound(const ModelState* m);
void batch_enqueue_group(ModelState* m, ModelState::GroupSlot& sl, const std::vector<BatchPart>& parts, unsigned num_results, const DevScorer& ds,
                         bool pipelined, hipEvent_t gate = nullptr, bool optimistic = true, const std::shared_ptr<ScorerDev>* scorer_then = nullptr) {
  const std::shared_ptr<ScorerDev> scorer = scorer_then ? *scorer_then : m->scorer_;   // (a retry: the scorer of the first attempt)
  int Bg = 0;
  for (const BatchPart& pt : parts) Bg += (int)pt.idx.size();
  {  // remember the group (the callers' size arrays do not outlive their call); `parts` may itself be built on a moved-out copy of this
    std::vector<ModelState::GroupSlot::SavedPart> keep;
    for (const BatchPart& pt : parts) {
      ModelState::GroupSlot::SavedPart sp{pt.d_audio, pt.stride, {}, pt.idx};   // (not `ready`: see the retry)
      unsigned mx = 0;
      for (unsigned i : pt.idx) mx = std::max(mx, i);
      sp.sizes.assign(pt.sizes, pt.sizes + (pt.idx.empty() ? 0 : mx + 1));
      keep.push_back(std::move(sp));
    }
    sl.saved_parts = std::move(keep);
  }
  sl.saved_num_results = num_results; sl.saved_pipelined = pipelined; sl.optimistic = optimistic;
  sl.saved_ds = ds; sl.saved_scorer = scorer;
  if (gate) HIP_CHECK(hipStreamWaitEvent(sl.stream_dec, gate, 0));  // this group's SEARCH starts behind group g - active; its acoustic model does not wait
  const int which = 1 + (int)(&sl - &m->slots_[0]);  // profiling mark list of this group's search stream
  // small integer tables in one page-locked block: [n_samples | n_frames | audio row | per chunk: begin, count]
  std::vector<int> ns(Bg), nf(Bg), row(Bg);
  int t_max = 1;
  {
    int b = 0;
    for (const BatchPart& pt : parts)
      for (unsigned i : pt.idx) { ns[b] = (int)pt.sizes[i]; nf[b] = n_frames_for(m->g, ns[b]); row[b] = (int)i; t_max = std::max(t_max, nf[b]); ++b; }
  }
  std::vector<int> cb;  // chunk boundaries
  {
    const ChunkPlan cp = chunk_plan(pipelined);
    for (int t = 0, k = 0; t < t_max; ++k) {
      cb.push_back(t);
      t += (k == 0) ? std::min(cp.first, cp.rest) : cp.rest;
    }
  }
  cb.push_back(t_max);
  const int n_chunks = (int)cb.size() - 1;
  const size_t n_ints = (size_t)(3 + 2 * n_chunks) * Bg;
  sl.h_ints.reserve(n_ints * 4); sl.ints.reserve(n_ints * 4);
  int* hi = sl.h_ints.as<int>();
  int *h_ns = hi, *h_nf = hi + Bg, *h_row = hi + 2 * Bg, *h_tab = hi + 3 * Bg;
  for (int b = 0; b < Bg; ++b) { h_ns[b] = ns[b]; h_nf[b] = nf[b]; h_row[b] = row[b]; }
  for (int k = 0; k < n_chunks; ++k)
    for (int b = 0; b < Bg; ++b) {
      h_tab[(size_t)(2 * k) * Bg + b] = cb[k];
      h_tab[(size_t)(2 * k + 1) * Bg + b] = std::max(0, std::min(cb[k + 1], h_nf[b]) - cb[k]);
    }
  copy_h2d(sl.ints.p, sl.h_ints, n_ints * 4, m->stream);
  const int* d_ns = sl.ints.as<int>(); const int* d_nf = d_ns + Bg; const int* d_row = d_ns + 2 * Bg; const int* d_tab = d_ns + 3 * Bg;
  sl.Bg = Bg; sl.t_max = t_max;
  sl.idx.clear();
  for (const BatchPart& pt : parts) sl.idx.insert(sl.idx.end(), pt.idx.begin(), pt.idx.end());
  // features: one launch per part (each part has its own audio array), all into the group's [Bg][t_max][n_input] block
  mark(m, 0);
  m->ws_feats.reserve((size_t)Bg * t_max * m->g.n_input * 4);
  {
    int off = 0;
    for (const BatchPart& pt : parts) {
      const int Bp = (int)pt.idx.size();
      if (pt.ready) HIP_CHECK(hipStreamWaitEvent(m->stream, pt.ready, 0));   // (a retry passes none: the first attempt's copy has long landed)
      MfccArgs fa = m->mfcc_args();
      fa.audio = pt.d_audio; fa.rows = d_row + off; fa.n_samples = d_ns + off; fa.n_frames = d_nf + off;
      fa.feats = m->ws_feats.as<float>() + (size_t)off * t_max * m->g.n_input; fa.n_max = (int)pt.stride; fa.t_max = t_max;
      if (Bp) launch_mfcc(fa, Bp * t_max, m->stream);
      off += Bp;
    }
  }
  // decoder streams of the group
  m->decoder_create(sl.dec, Bg, (int)m->beam_width_, t_max, scorer, &sl.h_table, optimistic);
  sl.probs.reserve((size_t)Bg * t_max * m->g.n_classes * 4);
  DecParams p{};
  p.C = m->g.n_classes; p.blank = p.C - 1; p.beam = sl.dec.beam; p.cutoff_top_n = 40; p.cutoff_prob = 1.0; p.t_max = t_max;
  p.phase_cycles = prof_of(m).phase_cycles ? 1 : 0;
  if (p.phase_cycles) {
    sl.stamps.reserve((size_t)Bg * 64 * 8);
    HIP_CHECK(hipMemsetAsync(sl.stamps.p, 0, (size_t)Bg * 64 * 8, sl.stream_dec));
    p.stamps = sl.stamps.as<unsigned long long>();
  }
  int max_chunk = 1;
  for (int k = 0; k < n_chunks; ++k) max_chunk = std::max(max_chunk, cb[k + 1] - cb[k]);
  sl.wide.reserve(ctc_rows_ws_bytes(p, Bg, max_chunk));
  // the acoustic model as three engines (engine.h) when other groups are in flight beside this one; a lone group (blocking call,
  // latency matters) keeps everything on `stream`: its own GEMMs would only slow its own recurrence (6.5 vs 5.9 ms)
  // (the int8 path takes the same form: tunable am_i8_pipe; 0 = one acoustic stream, 3.8 ms per batch wherever the engines' queues sit)
  const bool piped = pipelined && (!m->i8 || tune().am_i8_pipe != 0) && m->am_pipe_init();
  m->watch_search_bound = search_bound(m);   // four searches side by side hold every CU: the recurrence waits for THEM, whatever its queue (round 5: `bytes` tripped the watch)
  if (piped) m->am_replace_if_slow();   // (a bad placement of the engines' hardware queues shows in the pipeline's own timing: engine.cpp)
  for (int k = 0; k < n_chunks; ++k) {
    hipEvent_t ev = m->ev_chunk[k % 2];  // an event may be re-recorded once the wait on it has been enqueued
    if (piped) {
      m->run_acoustic_chunk_piped(m->ws_feats.as<float>(), d_nf, Bg, t_max, cb[k], cb[k + 1] - cb[k], sl.probs.as<float>(), ev);  // marks 1, 2, 3
    } else {
      m->run_acoustic_chunk(m->ws_feats.as<float>(), d_nf, Bg, t_max, cb[k], cb[k + 1] - cb[k], sl.probs.as<float>());  // marks 1, 2, 3
      mark(m, -1);
      HIP_CHECK(hipEventRecord(ev, m->stream));
    }
    HIP_CHECK(hipStreamWaitEvent(sl.stream_dec, ev, 0));
    mark_on(m, 4, which, sl.stream_dec);
    const int* fb = d_tab + (size_t)(2 * k) * Bg;
    launch_ctc_next(p, ds, m->dev_alphabet, sl.dec.table.as<DecStream>(), Bg, sl.probs.as<float>(), fb, fb + Bg, sl.stream_dec, max_chunk, sl.wide.p);
    mark_on(m, -1, which, sl.stream_dec);
  }
  // ranking + back-tracking, results to page-locked memory (a token needs its own timestep: <= t_max tokens)
  const int nr = (int)std::max(1u, std::min<unsigned>(num_results, (unsigned)sl.dec.beam));
  const int max_len = t_max + 1;
  sl.nr = nr; sl.max_len = max_len;
  sl.out_layout = DecodeBlock::layout(Bg, nr, max_len);
  sl.out.reserve(sl.out_layout.bytes); sl.h_out.reserve(sl.out_layout.bytes);
  const DecodeOut o = sl.out_layout.view(sl.out.p, nr, max_len);
  mark_on(m, 5, which, sl.stream_dec);
  launch_ctc_decode(p, ds, m->dev_alphabet, sl.dec.table.as<DecStream>(), Bg, o, sl.stream_dec);
  copy_d2h(sl.h_out, sl.out.p, sl.out_layout.bytes, sl.stream_dec);  // all results, one block
  mark_on(m, -1, which, sl.stream_dec);
  sl.prof_enqueued = prof_of(m).on && prof_of(m).only < 0; sl.prof_stamp_bytes = 0;   // (level 3: no counter block behind the results either)
  if (sl.prof_enqueued) {  // the search counters ride behind the results (no extra synchronisation when they are read)
    const size_t tb = sizeof(DecStream) * (size_t)Bg, sb = p.stamps ? (size_t)Bg * 64 * 8 : 0;
    sl.h_prof.reserve(tb + sb);
    copy_d2h(sl.h_prof, sl.dec.table.p, tb, sl.stream_dec);
    if (sb) copy_d2h(sl.h_prof, sl.stamps.p, sb, sl.stream_dec, tb);
    sl.prof_stamp_bytes = sb;
  }
  HIP_CHECK(hipEventRecord(sl.done, sl.stream_dec));
}

// Wait for a group and turn its page-locked result block into Output lists: all[idx[i]] = results of the group's stream i.
void batch_collect_group(ModelState* m, ModelState::GroupSlot& sl, std::vector<std::vector<Output>>& all, Prof& pr) {
  HIP_CHECK(hipEventSynchronize(sl.done));
  if (sl.optimistic) {  // an arena of the optimistic sizing overflowed (bits 0, 1, 3: flagged by the kernel, nothing written out of bounds): decode the group again with the bound that cannot
    const DecodeOut h0 = sl.out_layout.view(sl.h_out.p, sl.nr, sl.max_len);
    int err = 0;
    for (int i = 0; i < sl.Bg; ++i) err |= h0.errors[i];
    if (err & 0xB) {
      __atomic_fetch_add(&tune().arena_retries, 1, __ATOMIC_RELAXED);   // (a fleet collects on one thread per device)
      std::vector<unsigned> keep_idx = sl.idx;
      const std::vector<ModelState::GroupSlot::SavedPart> saved = std::move(sl.saved_parts);  // (alive while the group is enqueued again)
      std::vector<BatchPart> parts;
      for (const auto& sp : saved) parts.push_back(BatchPart{sp.d_audio, sp.stride, sp.sizes.data(), sp.idx});
      // under the scorer, hot words and alpha / beta the group was SUBMITTED with -- not what the model holds now (STT_AddHotWord or a
      // scorer swap between submit and collect must not change a batch only when its arenas happened to overflow)
      const DevScorer ds = sl.saved_ds;
      const std::shared_ptr<ScorerDev> scorer_then = sl.saved_scorer;
      batch_enqueue_group(m, sl, parts, sl.saved_num_results, ds, sl.saved_pipelined, nullptr, /*optimistic=*/false, &scorer_then);
      sl.idx = keep_idx;
      HIP_CHECK(hipEventSynchronize(sl.done));
    }
  }
  const DecodeOut h = sl.out_layout.view(sl.h_out.p, sl.nr, sl.max_len);
  const uint32_t *tok = h.tokens, *ts = h.timesteps;
  const int *lens = h.lens, *nres = h.n_results;
  const double* conf = h.confidence;
  check_decoder_errors(h.errors, sl.Bg);
  for (int i = 0; i < sl.Bg; ++i) {
    std::vector<Output>& dst = all[sl.idx[i]];
    for (int r = 0; r < nres[i]; ++r) {
      Output ou;
      const size_t ob = (size_t)i * sl.nr + r;
      const int total = lens[ob], len = std::min(total, sl.max_len);
      ou.confidence = conf[ob];
      ou.tokens.resize(len); ou.timesteps.resize(len);
      for (int j = 0; j < len; ++j) {  // ring slots of the single-pass back-tracking (ctc_decode_kernel)
        const size_t slot = ob * sl.max_len + (size_t)((total - 1 - j) % sl.max_len);
        ou.tokens[j] = tok[slot]; ou.timesteps[j] = ts[slot];
      }
      dst.push_back(std::move(ou));
    }
  }
  if (pr.on && pr.only >= 0) { pr.ms[6] += (float)sl.t_max; pr.ms[7] += (float)sl.t_max * sl.Bg; }   // (level 3: the launch and timestep counts still)
  if (pr.on && sl.prof_enqueued) {  // (only what was enqueued with profiling on carries a profiling block)
    pr.ms[6] += (float)sl.t_max; pr.ms[7] += (float)sl.t_max * sl.Bg;
    const DecStream* tb = sl.h_prof.as<DecStream>();
    for (int i = 0; i < sl.Bg; ++i) { for (int k = 0; k < 4; ++k) pr.dec_stats[k] += tb[i].stat[k]; for (int k = 0; k < 8; ++k) pr.dec_phase[k] += tb[i].phase[k]; }
    if (tune().dump_marks) {  // how much of the (worst-case sized) arenas this group really used
      double fp = 0, ft = 0, fb = 0;
      for (int i = 0; i < sl.Bg; ++i) { fp = std::max(fp, (double)tb[i].pa_n / tb[i].pa_cap); ft = std::max(ft, (double)tb[i].ta_n / tb[i].ta_cap); fb = std::max(fb, (double)tb[i].be_n / tb[i].be_cap); }
      fprintf(stderr, "ARENA group of %d streams, t_max %d: max fill path %.3f time %.3f boundary-entries %.3f\n", sl.Bg, sl.t_max, fp, ft, fb);
    }
    if (sl.prof_stamp_bytes) {
      const unsigned long long* st = reinterpret_cast<const unsigned long long*>((const char*)sl.h_prof.p + sizeof(DecStream) * (size_t)sl.Bg);
      for (int i = 0; i < sl.Bg; ++i) for (int k = 0; k < 64; ++k) pr.dec_stamps[k] += st[(size_t)i * 64 + k];
    }
  }
}

void sync_acoustic_streams(ModelState* m, bool check) {
  for (hipStream_t st : {m->stream, m->stream_l, m->stream_o}) {
    if (!st) continue;
    if (check) HIP_CHECK(hipStreamSynchronize(st)); else (void)hipStreamSynchronize(st);
  }
}
void sync_everything(ModelState* m) {  // failure path: nothing of a call may still run on the slots' buffers when the caller sees the error
  sync_acoustic_streams(m, false);
  for (auto& sl : m->slots_) if (sl.stream_dec) (void)hipStreamSynchronize(sl.stream_dec);
}
bool search_bound(const ModelState* m);
void batch_init_slots(ModelState* m) {
  const bool place = tune().am_pipe && tune().am_place && tune().search_cus <= 0;
  if (m->ev_chunk[0]) {
    // placed for the other kind of setup (the scorer or the beam width changed since): place again, with nothing in flight
    if (place && m->placed_search_bound_ >= 0 && m->placed_search_bound_ != (int)search_bound(m) && !m->async_any()) {
      sync_everything(m);
      if (m->stream_l) { (void)hipStreamDestroy(m->stream_l); m->stream_l = nullptr; }
      if (m->stream_o) { (void)hipStreamDestroy(m->stream_o); m->stream_o = nullptr; }
      hipStream_t ss[ModelState::kSlots];
      for (int i = 0; i < ModelState::kSlots; ++i) ss[i] = m->slots_[i].stream_dec;
      m->place_batch_streams(ss, ModelState::kSlots, search_bound(m));
      for (int i = 0; i < ModelState::kSlots; ++i) m->slots_[i].stream_dec = ss[i];
      m->stream_dec = ss[0];
      m->placed_search_bound_ = (int)search_bound(m);
      m->placement_avoid_.clear();
      for (auto& sl : m->slots_) m->placement_avoid_.push_back(sl.stream_dec);
    }
    return;
  }
  for (auto& e : m->ev_chunk) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& sl : m->slots_) HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  m->slots_[0].stream_dec = m->stream_dec;
  for (int i = 1; i < ModelState::kSlots; ++i) create_engine_stream(&m->slots_[i].stream_dec, 3);
  if (place && !m->stream_l) {
    // every stream of the batch pipeline on a dispatch pipe chosen for its role (engine.cpp: place_batch_streams)
    hipStream_t ss[ModelState::kSlots];
    for (int i = 0; i < ModelState::kSlots; ++i) ss[i] = m->slots_[i].stream_dec;
    m->place_batch_streams(ss, ModelState::kSlots, search_bound(m));
    m->placed_search_bound_ = (int)search_bound(m);
    for (int i = 0; i < ModelState::kSlots; ++i) m->slots_[i].stream_dec = ss[i];
    m->stream_dec = ss[0];           // (slot 0's stream is the model's: ~ModelState destroys it)
  }
  m->placement_avoid_.clear();      // (engine.cpp: place_engine_streams -- the recurrence must not sit behind a search stream's dispatches either)
  for (auto& sl : m->slots_) m->placement_avoid_.push_back(sl.stream_dec);
}
// Group slots in flight (tunable `pipeline`, 1..kSlots; default 2).  With more slots than `active` (default 2) the beam search of
// group g starts behind the `done` event of group g - active while its acoustic model starts as soon as the previous group's
// has finished: the recurrence -- the longest dependent chain of a batch -- then runs back to back across batches while at
// most `active` searches run beside it.  Measured: with two slots 4.1-4.8 ms per batch depending on the
// machine; three slots were never better.
// A search-bound setup -- code-point scorer or a beam beyond 512: the search of a batch takes 45+ ms on its 64 compute units, the
// acoustic model 5 -- gets four slots and all four searches side by side (bytes workload: 24.7 -> 16.3 ms per batch); everything
// else two and two (a third group crowds the recurrence).
bool search_bound(const ModelState* m) { return (m->scorer_ && m->scorer_->is_utf8) || m->beam_width_ > 512; }
bool pairing_cfg(const ModelState* m);
int active_groups(const ModelState* m) {
  const int d = tune().active;
  if (d > 0) return d;
  if (search_bound(m)) return ModelState::kSlots;
  return pairing_cfg(m) ? 1 : 2;  // a 128-stream group's search holds 128 CUs: one at a time (3.19 against 3.25 ms per batch), two 64-stream ones side by side
}
int pipeline_slots_cfg(const ModelState* m) {
  const int d = tune().pipeline;
  if (d > 0) return std::min(d, (int)ModelState::kSlots);
  return m && search_bound(m) ? ModelState::kSlots : 2;
}
// Two submitted batches per slot (one recurrence over 128 rows) where the recurrent kernel covers them and the acoustic model is
// what bounds the step (a search-bound setup gains nothing from it and wants its four slots' searches side by side).
bool pairing_cfg(const ModelState* m) { return tune().pair != 0 && m && lstm_max_rows(m->g.n_hidden) >= 128 && !search_bound(m); }
int group_rows(const ModelState* m) { return pairing_cfg(m) ? 128 : 64; }
// (while batches are in flight the configuration they were submitted under stays in force: ModelState::async_depth_ / async_pair_)
int pipeline_slots(const ModelState* m) { return (m && m->async_any() && m->async_depth_ > 0) ? m->async_depth_ : pipeline_slots_cfg(m); }
bool pairing(const ModelState* m) { return (m && m->async_any() && m->async_depth_ > 0) ? m->async_pair_ : pairing_cfg(m); }
int pipeline_depth(const ModelState* m) { return pipeline_slots(m) * (pairing(m) ? 2 : 1); }  // tickets the caller may hold

void prof_begin(Prof& pr) {
  for (float& x : pr.ms) x = 0; for (auto& x : pr.dec_stats) x = 0; for (auto& x : pr.dec_phase) x = 0; for (auto& x : pr.dec_stamps) x = 0;
  prof_reset(pr);
}

// Utterances are taken longest first in groups (length-homogeneous groups: the LSTM runs every group to its longest member),
// several groups in flight (see ModelState::GroupSlot).
std::vector<std::vector<Output>> batch_run(ModelState* m, const int16_t* d_audio, unsigned stride, const unsigned* sizes, unsigned B, unsigned num_results) {
  std::vector<std::vector<Output>> all(B);
  HIP_CHECK(hipSetDevice(m->device));
  Prof& pr = prof_of(m);
  if (m->async_any()) throw std::runtime_error("a batch submitted with STTX_BatchSubmitDevice has not been collected yet");
  if (pr.on) prof_begin(pr);
  batch_init_slots(m);
  std::vector<unsigned> order(B);
  for (unsigned i = 0; i < B; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return sizes[x] > sizes[y]; });
  const DevScorer ds = m->current_scorer(m->scorer_, m->hot_words_, m->hot_tables_);
  const int depth = pipeline_slots_cfg(m);
  const unsigned R = B > 64 ? (unsigned)group_rows(m) : 64u;  // rows per group
  int oldest = 0, gi = 0;  // groups [oldest, gi) are in flight, group g in slot g % depth
  try {
    for (unsigned g0 = 0; g0 < B; g0 += R, ++gi) {
      if (gi - oldest == depth) { batch_collect_group(m, m->slots_[oldest % depth], all, pr); ++oldest; }
      std::vector<BatchPart> parts(1);
      parts[0] = BatchPart{d_audio, stride, sizes, std::vector<unsigned>(order.begin() + g0, order.begin() + std::min(B, g0 + R))};
      const int act = active_groups(m);
      hipEvent_t gate = (act < depth && gi - act >= oldest) ? m->slots_[(gi - act) % depth].done : nullptr;
      batch_enqueue_group(m, m->slots_[gi % depth], parts, num_results, ds, B > R * (unsigned)act, gate);
    }
    for (; oldest < gi; ++oldest) batch_collect_group(m, m->slots_[oldest % depth], all, pr);
  } catch (...) {
    sync_everything(m);
    if (pr.on) prof_reset(pr);
    throw;
  }
  if (pr.on) {
    sync_acoustic_streams(m, true);
    for (auto& sl : m->slots_) HIP_CHECK(hipStreamSynchronize(sl.stream_dec));
    prof_collect(pr);
  }
  return all;
}

// The group slots as a caller-driven pipeline (STTX_BatchSubmitDevice / STTX_BatchCollect): a batch of <= 64 utterances
// is enqueued without waiting, so the acoustic model of batch k+1 runs while the beam search of batch k finishes and the
// host turns batch k-1's results into strings -- what batch_run() does between the groups of one call, across calls.
// With pairing, the first batch of a pair is only noted; the caller's next submit sends both through the acoustic model as one
// 128-row group (or the first one's collect sends it through alone).
void submit_enqueue(ModelState* m, const std::vector<BatchPart>& parts, const int* tickets) {
  const int depth = m->async_depth_;
  const int slot = m->async_groups_ % depth;
  ModelState::GroupSlot& sl = m->slots_[slot];
  if (sl.busy()) throw std::runtime_error("the pipeline is full (STTX_BatchPipelineDepthFor batches in flight): collect the oldest one first");
  const DevScorer ds = m->current_scorer(m->scorer_, m->hot_words_, m->hot_tables_, /*in_flight=*/true);
  hipEvent_t gate = nullptr;
  {
    const int act = active_groups(m), older = m->async_groups_ - act;
    if (act < depth && older >= 0 && m->slots_[older % depth].busy()) gate = m->slots_[older % depth].done;
  }
  try { batch_enqueue_group(m, sl, parts, 1, ds, true, gate); }
  catch (...) { sync_everything(m); throw; }  // half-enqueued work must not meet the next submit on the slot's buffers
  sl.n_parts = (int)parts.size();
  sl.part_begin[0] = 0;
  for (int p = 0; p < sl.n_parts; ++p) {
    sl.part_begin[p + 1] = sl.part_begin[p] + (int)parts[p].idx.size();
    sl.part_ticket[p] = tickets[p]; sl.part_open[p] = true;
  }
  for (int i = 0; i < sl.Bg; ++i) sl.idx[i] = (unsigned)i;  // results by position in the group; collect cuts them by part
  sl.results_ready = false; sl.results.clear();
  ++m->async_groups_;
}
int batch_submit(ModelState* m, const int16_t* d_audio, unsigned stride, const unsigned* sizes, unsigned B, hipEvent_t ready = nullptr) {
  if (B == 0 || B > 64) throw std::runtime_error("STTX_BatchSubmit / STTX_BatchSubmitDevice take 1..64 utterances (one batch) per call");
  HIP_CHECK(hipSetDevice(m->device));
  batch_init_slots(m);
  Prof& pr = prof_of(m);
  if (!m->async_any()) {
    m->retired_bufs_.clear();  // (the pipeline has drained: nothing reads a replaced hot-word table any more)
    m->async_depth_ = pipeline_slots_cfg(m); m->async_pair_ = pairing_cfg(m); m->async_groups_ = 0;
    if (pr.on) prof_begin(pr);
  }
  std::vector<unsigned> idx(B);
  for (unsigned i = 0; i < B; ++i) idx[i] = i;
  {
    int held = m->pending_.valid ? 1 : 0;  // tickets the caller holds
    for (const auto& sl : m->slots_) held += (sl.part_open[0] ? 1 : 0) + (sl.part_open[1] ? 1 : 0);
    if (held >= m->async_depth_ * (m->async_pair_ ? 2 : 1))
      throw std::runtime_error("the pipeline is full (STTX_BatchPipelineDepthFor batches in flight): collect the oldest one first");
  }
  if (m->async_pair_ && !m->pending_.valid) {  // first half of a pair: noted, not enqueued (its slot need only be free when its partner arrives)
    m->pending_.valid = true; m->pending_.d_audio = d_audio; m->pending_.stride = stride;
    m->pending_.sizes.assign(sizes, sizes + B); m->pending_.ticket = m->async_next_; m->pending_.ready = ready;
    return m->async_next_++;
  }
  std::vector<BatchPart> parts;
  int tickets[2] = {-1, -1};
  if (m->pending_.valid) {
    std::vector<unsigned> pidx(m->pending_.sizes.size());
    for (size_t i = 0; i < pidx.size(); ++i) pidx[i] = (unsigned)i;
    parts.push_back(BatchPart{m->pending_.d_audio, m->pending_.stride, m->pending_.sizes.data(), pidx, m->pending_.ready});
    tickets[0] = m->pending_.ticket;
  }
  parts.push_back(BatchPart{d_audio, stride, sizes, idx, ready});
  tickets[parts.size() - 1] = m->async_next_;
  submit_enqueue(m, parts, tickets);
  m->pending_.valid = false;
  return m->async_next_++;
}
// the slot and part that hold `ticket` (enqueueing a noted first half alone if that is what the ticket names)
ModelState::GroupSlot& slot_of_ticket(ModelState* m, int ticket, int& part) {
  if (ticket >= 0 && m->pending_.valid && m->pending_.ticket == ticket) {
    std::vector<unsigned> pidx(m->pending_.sizes.size());
    for (size_t i = 0; i < pidx.size(); ++i) pidx[i] = (unsigned)i;
    std::vector<BatchPart> parts{BatchPart{m->pending_.d_audio, m->pending_.stride, m->pending_.sizes.data(), pidx, m->pending_.ready}};
    const int tickets[2] = {ticket, -1};
    submit_enqueue(m, parts, tickets);
    m->pending_.valid = false;
  }
  if (ticket >= 0)
    for (auto& sl : m->slots_)
      for (int p = 0; p < sl.n_parts; ++p)
        if (sl.part_open[p] && sl.part_ticket[p] == ticket) { part = p; return sl; }
  throw std::runtime_error("no such batch in flight");
}
std::vector<std::vector<Output>> batch_collect(ModelState* m, int ticket) {
  HIP_CHECK(hipSetDevice(m->device));
  int part = 0;
  ModelState::GroupSlot& sl = slot_of_ticket(m, ticket, part);
  Prof& pr = prof_of(m);
  sl.part_open[part] = false;
  try {
    if (!sl.results_ready) {
      sl.results.assign((size_t)sl.Bg, {});
      batch_collect_group(m, sl, sl.results, pr);
      sl.results_ready = true;
    }
  } catch (...) {
    sync_everything(m);
    for (auto& s2 : m->slots_) { s2.part_open[0] = s2.part_open[1] = false; s2.results.clear(); s2.results_ready = false; }  // whatever else was in flight has finished; its results are dropped
    m->pending_.valid = false;
    if (pr.on) prof_reset(pr);
    throw;
  }
  std::vector<std::vector<Output>> out(sl.results.begin() + sl.part_begin[part], sl.results.begin() + sl.part_begin[part + 1]);
  if (!sl.busy()) { sl.results.clear(); sl.results_ready = false; sl.n_parts = 0; }
  if (pr.on && !m->async_any()) {  // the pipeline has drained: every mark has been reached
    sync_acoustic_streams(m, true);
    for (auto& s2 : m->slots_) HIP_CHECK(hipStreamSynchronize(s2.stream_dec));
    prof_collect(pr);
  }
  return out;
}
// A few host threads that stay around for STTX_BatchSubmit's gather (round 5 spawned three std::threads per submit): run(n, f) calls f(k)
// for k = 0 .. n-1, k = 0 on the caller's thread, and returns when all are done.  One submit at a time per process is the common case; a
// second model's submit that arrives meanwhile simply gathers on its own thread.
class GatherPool {
 public:
  static GatherPool& get() { static GatherPool* p = new GatherPool(3); return *p; }   // (never destroyed: no join at process exit)
  void run(unsigned n, const std::function<void(unsigned)>& f) {
    std::unique_lock<std::mutex> busy(busy_, std::try_to_lock);
    if (!busy.owns_lock() || n <= 1) { for (unsigned k = 0; k < n; ++k) f(k); return; }
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; next_ = 1; n_ = n; left_ = n - 1; ++gen_;
    }
    cv_.notify_all();
    f(0);
    for (;;) {   // the caller takes what the workers have not started yet, then waits for those that have
      unsigned k;
      { std::lock_guard<std::mutex> g(m_); if (next_ >= n_) break; k = next_++; }
      f(k);
      std::lock_guard<std::mutex> g(m_);
      --left_;
    }
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return left_ == 0; });
    job_ = nullptr;
  }

 private:
  explicit GatherPool(unsigned workers) {
    for (unsigned i = 0; i < workers; ++i) std::thread([this] { work(); }).detach();
  }
  void work() {
    unsigned long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> g(m_);
      cv_.wait(g, [&] { return gen_ != seen && job_ && next_ < n_; });
      seen = gen_;
      while (job_ && next_ < n_) {
        const unsigned k = next_++;
        const std::function<void(unsigned)>* f = job_;
        g.unlock();
        (*f)(k);
        g.lock();
        if (--left_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex busy_, m_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)>* job_ = nullptr;
  unsigned next_ = 0, n_ = 0, left_ = 0;
  unsigned long long gen_ = 0;
};

// STTX_BatchSubmit: host buffers -> one page-locked block [B][stride] -> HBM on the copy queue -> batch_submit() gated by the copy's event.
// The host pays one pass over the samples (the gather into page-locked memory: the ABI hands over ordinary pageable buffers, which no
// DMA engine may read in place); the transfer itself overlaps whatever the GPU is doing for the batches before this one.
int batch_submit_host(ModelState* m, const short* const* bufs, const unsigned* sizes, unsigned B) {
  if (B == 0 || B > 64) throw std::runtime_error("STTX_BatchSubmit takes 1..64 utterances (one batch) per call");
  if (!bufs || !sizes) throw std::runtime_error("STTX_BatchSubmit: NULL buffer / size array");
  for (unsigned i = 0; i < B; ++i)
    if (sizes[i] && !bufs[i]) throw std::runtime_error("STTX_BatchSubmit: utterance " + std::to_string(i) + " has " + std::to_string(sizes[i]) + " samples and a NULL buffer");
  HIP_CHECK(hipSetDevice(m->device));
  if (!m->stream_h2d) HIP_CHECK(hipStreamCreateWithFlags(&m->stream_h2d, hipStreamNonBlocking));
  ModelState::HostStage& hs = m->stage_[m->stage_seq_ % ModelState::kStage];
  unsigned stride = 8;
  for (unsigned i = 0; i < B; ++i) stride = std::max(stride, sizes[i]);
  stride = (stride + 7) & ~7u;
  const size_t bytes = (size_t)B * stride * 2;
  if (!hs.copied) HIP_CHECK(hipEventCreateWithFlags(&hs.copied, hipEventDisableTiming));
  else HIP_CHECK(hipEventSynchronize(hs.copied));          // (kStage submits ago: reached long since)
  hs.pin.reserve(bytes); hs.dev.reserve(bytes);
  int16_t* hp = hs.pin.as<int16_t>();
  auto gather = [&](unsigned lo, unsigned hi) {
    for (unsigned i = lo; i < hi; ++i) {
      if (sizes[i]) memcpy(hp + (size_t)i * stride, bufs[i], (size_t)sizes[i] * 2);
      if (sizes[i] < stride) memset(hp + (size_t)i * stride + sizes[i], 0, (size_t)(stride - sizes[i]) * 2);
    }
  };
  // 10 MB per 64 x 5 s batch: one thread moves it in ~1 ms, which is a third of what the GPU needs for the batch and sits between a
  // collect and the enqueue of the next group; four threads share it (below ~1 MB the threads cost more than they save)
  if (bytes >= (1u << 20) && B >= 8) {
    constexpr unsigned NPART = 8;    // (more parts than threads: whoever is free takes the next one)
    GatherPool::get().run(NPART, [&](unsigned k) { gather(B * k / NPART, B * (k + 1) / NPART); });
  } else gather(0, B);
  HIP_CHECK(hipMemcpyAsync(hs.dev.p, hs.pin.p, bytes, hipMemcpyHostToDevice, m->stream_h2d));
  HIP_CHECK(hipEventRecord(hs.copied, m->stream_h2d));
  const int ticket = batch_submit(m, hs.dev.as<int16_t>(), stride, sizes, B, hs.copied);
  ++m->stage_seq_;     // (only a submit that went through takes the entry)
  return ticket;
}
// Debug: the acoustic probabilities of a submitted batch exactly as the pipelined path computed them (three engines, graph-replayed
// recurrence, ring slots, 64 or 128 rows per step) -- the block the group's beam search reads.  Before the batch is collected.
void batch_probs(ModelState* m, int ticket, float* out, unsigned max_frames, unsigned* n_frames) {
  HIP_CHECK(hipSetDevice(m->device));
  int part = 0;
  ModelState::GroupSlot& sl = slot_of_ticket(m, ticket, part);
  HIP_CHECK(hipEventSynchronize(sl.done));
  const int C = m->g.n_classes;
  const int* h_nf = sl.h_ints.as<int>() + sl.Bg;
  for (int i = sl.part_begin[part]; i < sl.part_begin[part + 1]; ++i) {
    const int j = i - sl.part_begin[part], nf = h_nf[i];
    n_frames[j] = (unsigned)nf;
    if ((unsigned)nf > max_frames) throw std::runtime_error("STTX_DebugBatchProbs: aMaxFrames too small");
    HIP_CHECK(hipMemcpy(out + (size_t)j * max_frames * C, sl.probs.as<float>() + (size_t)i * sl.t_max * C, (size_t)nf * C * 4, hipMemcpyDeviceToHost));
  }
}
}  // namespace

void launch_test_math(int op, const float* a, const float* b, float* out, unsigned n, hipStream_t st);  // ctc.hip
void launch_test_lm(const DevScorer& s, const uint64_t* hashes, int n, int bos, int use_index, float* probs, int* lens, hipStream_t st);
uint64_t stt_murmur64a(const void* key, size_t len);

// hooks used by engine.cpp for the profiling marks inside run_acoustic_rows
void stt_prof_mark(ModelState* m, int i) { mark(m, i); }
void stt_prof_mark_on(ModelState* m, int id, int which, hipStream_t st) { mark_on(m, id, which, st); }

// The one-by-one fallback of the batched string calls (streams that do not share model / beam / scorer): all of the strings or none --
// a decode that throws partway frees what was built and the caller sees NULL, never an array with holes ("aCount strings or NULL").
template <class F>
char** strings_each(const std::vector<StreamingState*>& ss, F&& one) {
  std::vector<char*> tmp;
  try {
    for (StreamingState* s : ss) tmp.push_back(one(s));
  } catch (...) {
    for (char* c : tmp) free(c);
    throw;
  }
  char** r = (char**)malloc(sizeof(char*) * std::max<size_t>(1, tmp.size()));
  for (size_t i = 0; i < tmp.size(); ++i) r[i] = tmp[i];
  return r;
}

extern "C" {

// ================================================================ coqui-stt.h
int STT_CreateModelFromBuffer(const char* aModelBuffer, unsigned int aBufferSize, ModelState** retval) {
  *retval = nullptr;
  // stt.cc:344-345 prints two version lines on stderr; CI scripts grep for them (ci_scripts/asserts.sh:284-321)
  std::cerr << "TensorFlow: none (MI355X HIP engine, gfx950)" << std::endl;
  std::cerr << " Coqui STT: " << kVersion << "-mi355x" << std::endl;
  if (!aModelBuffer || !aBufferSize) {
    std::cerr << "No model specified, cannot continue." << std::endl;
    return STT_ERR_NO_MODEL;
  }
  return guarded([&]() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      std::cerr << "stt_amd: no HIP device available; this engine has no CPU path." << std::endl;
      return (int)STT_ERR_FAIL_INIT_SESS;
    }
    std::unique_ptr<ModelState> model(new ModelState());
    model->device = g_device;
    int err = model->InitFromBuffer(aModelBuffer, aBufferSize);
    if (err != STT_ERR_OK) return err;
    *retval = model.release();
    { std::lock_guard<std::mutex> lk(g_models_mu); g_models[*retval] = 1; }
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_CREATE_MODEL);
}

int STT_CreateModel(const char* aModelPath, ModelState** retval) {
  *retval = nullptr;
  if (!aModelPath || !strlen(aModelPath)) {
    std::cerr << "TensorFlow: none (MI355X HIP engine, gfx950)" << std::endl;
    std::cerr << " Coqui STT: " << kVersion << "-mi355x" << std::endl;
    std::cerr << "No model specified, cannot continue." << std::endl;
    return STT_ERR_NO_MODEL;
  }
  std::ifstream in(aModelPath, std::ios::binary | std::ios::ate);
  if (!in) {
    std::cerr << "TensorFlow: none (MI355X HIP engine, gfx950)" << std::endl;
    std::cerr << " Coqui STT: " << kVersion << "-mi355x" << std::endl;
    return STT_ERR_FAIL_INIT_MMAP;
  }
  const std::streamsize sz = in.tellg();
  in.seekg(0);
  std::vector<char> data((size_t)std::max<std::streamsize>(sz, 0));
  if (sz > 0) in.read(data.data(), sz);
  return STT_CreateModelFromBuffer(data.data(), (unsigned)data.size(), retval);
}

unsigned int STT_GetModelBeamWidth(const ModelState* aCtx) { return aCtx->beam_width_; }
int STT_SetModelBeamWidth(ModelState* aCtx, unsigned int aBeamWidth) {
  // (the reference stores any value; here the beam lives in LDS, so the width is checked when it is set, not at the next STT_CreateStream)
  if (aBeamWidth < 1 || aBeamWidth > STT_MAX_BEAM) return STT_ERR_INVALID_SHAPE;
  aCtx->beam_width_ = aBeamWidth;
  return STT_ERR_OK;
}
int STT_GetModelSampleRate(const ModelState* aCtx) { return aCtx->g.sample_rate; }
void STT_FreeModel(ModelState* ctx) {
  if (!ctx) return;
  { std::lock_guard<std::mutex> lk(g_models_mu); g_models.erase(ctx); }
  delete ctx;
}

static int enable_scorer(ModelState* aCtx, const char* data, size_t len, const char* path) {  // stt.cc:414-449
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(aCtx->device));
    auto sc = std::make_shared<ScorerDev>();
    const int err = path ? sc->LoadFile(path, aCtx->alphabet_) : sc->LoadBuffer(data, len, aCtx->alphabet_);
    if (err != STT_ERR_OK) return (int)STT_ERR_INVALID_SCORER;
    aCtx->scorer_ = sc;
    return (int)STT_ERR_OK;
  }, STT_ERR_INVALID_SCORER);
}
int STT_EnableExternalScorer(ModelState* aCtx, const char* aScorerPath) { return enable_scorer(aCtx, nullptr, 0, aScorerPath); }
int STT_EnableExternalScorerFromBuffer(ModelState* aCtx, const char* aScorerBuffer, unsigned int aBufferSize) {
  return enable_scorer(aCtx, aScorerBuffer, aBufferSize, nullptr);
}
int STT_AddHotWord(ModelState* aCtx, const char* word, float boost) {  // stt.cc:451-466
  if (!aCtx->scorer_) return STT_ERR_SCORER_NOT_ENABLED;
  const size_t before = aCtx->hot_words_.size();
  aCtx->hot_words_.insert(std::pair<std::string, float>(word, boost));
  return aCtx->hot_words_.size() == before ? STT_ERR_FAIL_INSERT_HOTWORD : STT_ERR_OK;
}
int STT_EraseHotWord(ModelState* aCtx, const char* word) {  // stt.cc:468-483
  if (!aCtx->scorer_) return STT_ERR_SCORER_NOT_ENABLED;
  const size_t before = aCtx->hot_words_.size();
  aCtx->hot_words_.erase(word);
  return aCtx->hot_words_.size() == before ? STT_ERR_FAIL_ERASE_HOTWORD : STT_ERR_OK;
}
int STT_ClearHotWords(ModelState* aCtx) {  // stt.cc:485-497
  if (!aCtx->scorer_) return STT_ERR_SCORER_NOT_ENABLED;
  aCtx->hot_words_.clear();
  return STT_ERR_OK;
}
int STT_DisableExternalScorer(ModelState* aCtx) {  // stt.cc:499-506
  if (!aCtx->scorer_) return STT_ERR_SCORER_NOT_ENABLED;
  aCtx->scorer_.reset();
  return STT_ERR_OK;
}
int STT_SetScorerAlphaBeta(ModelState* aCtx, float aAlpha, float aBeta) {  // stt.cc:508-517
  if (!aCtx->scorer_) return STT_ERR_SCORER_NOT_ENABLED;
  aCtx->scorer_->reset_params(aAlpha, aBeta);
  return STT_ERR_OK;
}

int STT_CreateStream(ModelState* aCtx, StreamingState** retval) { return create_stream(aCtx, retval, false); }

void STT_FeedAudioContent(StreamingState* aSctx, const short* aBuffer, unsigned int aBufferSize) {
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); aSctx->feedAudioContent(aBuffer, aBufferSize); return 0; }, 0);
}
char* STT_IntermediateDecode(const StreamingState* aSctx) {
  char* r = nullptr;
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); r = decode_string(aSctx); return 0; }, 0);
  return r;
}
Metadata* STT_IntermediateDecodeWithMetadata(const StreamingState* aSctx, unsigned int aNumResults) {
  Metadata* r = nullptr;
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); r = decode_metadata(aSctx, aNumResults); return 0; }, 0);
  return r;
}
char* STT_IntermediateDecodeFlushBuffers(StreamingState* aSctx) {
  char* r = nullptr;
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); aSctx->flushBuffers(false); r = decode_string(aSctx); return 0; }, 0);
  return r;
}
Metadata* STT_IntermediateDecodeWithMetadataFlushBuffers(StreamingState* aSctx, unsigned int aNumResults) {
  Metadata* r = nullptr;
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); aSctx->flushBuffers(false); r = decode_metadata(aSctx, aNumResults); return 0; }, 0);
  return r;
}
void STT_FreeStream(StreamingState* aSctx) {
  if (!aSctx) return;
  ModelState* m = aSctx->model_;
  { std::lock_guard<std::mutex> lk(g_models_mu); if (!g_models.count(m)) m = nullptr; }
  if (m) {  // park the stream (and its HBM buffers) for the next STT_CreateStream; anything in flight on it is ordered before
            // the next user's work, which runs on the same HIP stream
    aSctx->recycle();
    std::lock_guard<std::mutex> lk(m->stream_pool_mu_);
    if (m->stream_pool_.size() < 1024) { m->stream_pool_.push_back(aSctx); return; }
  }
  delete aSctx;
}
// ---- many streams per call (stt_amd.h)
static char** strings_of(const ModelState* m, const std::vector<std::vector<Output>>& outs) {
  char** r = (char**)malloc(sizeof(char*) * std::max<size_t>(1, outs.size()));
  for (size_t i = 0; i < outs.size(); ++i) {
    const std::string t = outs[i].empty() ? std::string() : m->alphabet_.Decode(outs[i][0].tokens.data(), (int)outs[i][0].tokens.size());
    r[i] = strdup(t.c_str());
  }
  return r;
}
void STTX_FeedAudioContentBatch(StreamingState* const* aStreams, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aCount) {
  guarded([&]() {
    if (!aCount) return 0;
    std::vector<StreamingState*> ss(aStreams, aStreams + aCount);
    HIP_CHECK(hipSetDevice(ss[0]->model_->device));
    if (streams_batchable(ss)) streams_feed_batch(ss, aBuffers, aBufferSizes);
    else for (unsigned i = 0; i < aCount; ++i) ss[i]->feedAudioContent(aBuffers[i], aBufferSizes[i]);
    return 0;
  }, 0);
}
void STTX_FeedAudioContentBatchEx(StreamingState* const* aStreams, const short* const* aBuffers, const unsigned int* aBufferSizes, const unsigned char* aLast, unsigned int aCount) {
  guarded([&]() {
    if (!aCount) return 0;
    std::vector<StreamingState*> ss(aStreams, aStreams + aCount);
    HIP_CHECK(hipSetDevice(ss[0]->model_->device));
    if (streams_batchable(ss)) streams_feed_batch(ss, aBuffers, aBufferSizes, aLast);
    else for (unsigned i = 0; i < aCount; ++i) { ss[i]->feedAudioContent(aBuffers[i], aBufferSizes[i]); if (aLast && aLast[i]) ss[i]->flushBuffers(true); }
    return 0;
  }, 0);
}
char** STTX_IntermediateDecodeBatch(StreamingState* const* aStreams, unsigned int aCount) {
  char** r = nullptr;
  guarded([&]() {
    if (!aCount) return 0;
    std::vector<StreamingState*> ss(aStreams, aStreams + aCount);
    HIP_CHECK(hipSetDevice(ss[0]->model_->device));
    if (streams_batchable(ss)) r = strings_of(ss[0]->model_, streams_decode_batch(ss, 1));
    else r = strings_each(ss, [](StreamingState* s) { return decode_string(s); });
    return 0;
  }, 0);
  return r;
}
char** STTX_DecodeStreamsBatch(StreamingState* const* aStreams, const unsigned char* aFinish, unsigned int aCount) {
  char** r = nullptr;
  guarded([&]() {
    if (!aCount) return 0;
    std::vector<StreamingState*> ss(aStreams, aStreams + aCount), fin;
    for (unsigned i = 0; i < aCount; ++i) if (aFinish && aFinish[i]) fin.push_back(ss[i]);
    HIP_CHECK(hipSetDevice(ss[0]->model_->device));
    if (streams_batchable(ss)) {
      if (!fin.empty()) streams_flush_batch(fin, true);        // (nothing left to do for streams whose last audio carried the flush)
      r = strings_of(ss[0]->model_, streams_decode_batch(ss, 1));   // ONE ranking + back-tracking launch: the hop's intermediate results and the finishes'
    } else {
      unsigned i = 0;
      r = strings_each(ss, [&](StreamingState* s) { if (aFinish && aFinish[i++]) s->flushBuffers(true); return decode_string(s); });
    }
    return 0;
  }, 0);
  for (unsigned i = 0; i < aCount; ++i) if (aFinish && aFinish[i]) STT_FreeStream(aStreams[i]);   // like STT_FinishStream: also when the decode failed
  return r;
}
char** STTX_FinishStreamBatch(StreamingState* const* aStreams, unsigned int aCount) {
  char** r = nullptr;
  guarded([&]() {
    if (!aCount) return 0;
    std::vector<StreamingState*> ss(aStreams, aStreams + aCount);
    HIP_CHECK(hipSetDevice(ss[0]->model_->device));
    if (streams_batchable(ss)) { streams_flush_batch(ss, true); r = strings_of(ss[0]->model_, streams_decode_batch(ss, 1)); }
    else r = strings_each(ss, [](StreamingState* s) { s->flushBuffers(true); return decode_string(s); });
    return 0;
  }, 0);
  for (unsigned i = 0; i < aCount; ++i) STT_FreeStream(aStreams[i]);
  return r;
}


char* STT_FinishStream(StreamingState* aSctx) {
  char* r = nullptr;
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); aSctx->flushBuffers(true); r = decode_string(aSctx); return 0; }, 0);
  STT_FreeStream(aSctx);
  return r;
}
Metadata* STT_FinishStreamWithMetadata(StreamingState* aSctx, unsigned int aNumResults) {
  Metadata* r = nullptr;
  guarded([&]() { HIP_CHECK(hipSetDevice(aSctx->model_->device)); aSctx->flushBuffers(true); r = decode_metadata(aSctx, aNumResults); return 0; }, 0);
  STT_FreeStream(aSctx);
  return r;
}

// One-shot calls are "create stream, feed everything, finish" in the reference (stt.cc:641-688); the result only
// depends on the whole utterance, so they run as a batch of one through the time-parallel path (same kernels, same
// per-row arithmetic as the chunked streaming path).
char* STT_SpeechToText(ModelState* aCtx, const short* aBuffer, unsigned int aBufferSize) {
  if (aCtx->async_any()) {  // the group slots belong to the batches in flight: the reference's own form of this call (stt.cc:641-662)
    StreamingState* ctx;
    if (create_stream(aCtx, &ctx, false) != STT_ERR_OK) return nullptr;
    STT_FeedAudioContent(ctx, aBuffer, aBufferSize);
    return STT_FinishStream(ctx);
  }
  char** r = STTX_SpeechToTextBatch(aCtx, &aBuffer, &aBufferSize, 1);
  if (!r) return nullptr;
  char* s = r[0];
  free(r);
  return s;
}
Metadata* STT_SpeechToTextWithMetadata(ModelState* aCtx, const short* aBuffer, unsigned int aBufferSize, unsigned int aNumResults) {
  if (aCtx->async_any()) {  // (stt.cc:664-672)
    StreamingState* ctx;
    if (create_stream(aCtx, &ctx, false) != STT_ERR_OK) return nullptr;
    STT_FeedAudioContent(ctx, aBuffer, aBufferSize);
    return STT_FinishStreamWithMetadata(ctx, aNumResults);
  }
  Metadata** r = STTX_SpeechToTextBatchWithMetadata(aCtx, &aBuffer, &aBufferSize, 1, aNumResults);
  if (!r) return nullptr;
  Metadata* m = r[0];
  free(r);
  return m;
}
Metadata* STT_SpeechToTextWithEmissions(ModelState* aCtx, const short* aBuffer, unsigned int aBufferSize, unsigned int aNumResults) {
  StreamingState* ctx;  // stt.cc:674-688: needs the stream's last batch of emissions -> streaming path
  if (create_stream(aCtx, &ctx, true) != STT_ERR_OK) return nullptr;
  STT_FeedAudioContent(ctx, aBuffer, aBufferSize);
  return STT_FinishStreamWithMetadata(ctx, aNumResults);
}

void STT_FreeMetadata(Metadata* m) {  // stt.cc:696-726
  if (!m) return;
  for (unsigned i = 0; i < m->num_transcripts; ++i) {
    for (unsigned j = 0; j < m->transcripts[i].num_tokens; ++j) free((void*)m->transcripts[i].tokens[j].text);
    free((void*)m->transcripts[i].tokens);
  }
  free((void*)m->transcripts);
  if (m->emissions) {
    if (m->emissions->symbols) {
      for (int i = 0; i < m->emissions->num_symbols + 1; i++) free((void*)m->emissions->symbols[i]);
      free((void*)m->emissions->symbols);
    }
    if (m->emissions->emissions) free((void*)m->emissions->emissions);
    free((void*)m->emissions);
  }
  free(m);
}
void STT_FreeString(char* str) { free(str); }
char* STT_Version() { return strdup(kVersion); }
char* STT_ErrorCodeToErrorMessage(int aErrorCode) {
#define RETURN_MESSAGE(NAME, VALUE, DESC) case NAME: return strdup(DESC);
  switch (aErrorCode) {
    STT_FOR_EACH_ERROR(RETURN_MESSAGE)
    default: return strdup("Unknown error, please make sure you are using the correct native binary.");
  }
#undef RETURN_MESSAGE
}

// ================================================================ stt_amd.h
int STTX_SetDevice(int aDevice) {
  if (aDevice < 0 || aDevice >= 16) return STT_ERR_FAIL_INIT_SESS;  // per-device launch state is kept in tables of 16 (a node holds 8)
  g_device = aDevice;
  return hipSetDevice(aDevice) == hipSuccess ? STT_ERR_OK : STT_ERR_FAIL_INIT_SESS;
}
int STTX_GetDeviceCount(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : -STT_ERR_FAIL_INIT_SESS; }

char** STTX_SpeechToTextBatchDevice(ModelState* aCtx, const short* aDeviceAudio, unsigned int aStride, const unsigned int* aBufferSizes, unsigned int aBatch) {
  char** res = nullptr;
  guarded([&]() {
    auto outs = batch_run(aCtx, aDeviceAudio, aStride, aBufferSizes, aBatch, 1);
    res = (char**)malloc(sizeof(char*) * std::max(1u, aBatch));
    for (unsigned i = 0; i < aBatch; ++i)
      res[i] = outs[i].empty() ? strdup("") : strdup(aCtx->alphabet_.Decode(outs[i][0].tokens.data(), (int)outs[i][0].tokens.size()).c_str());
    return 0;
  }, 0);
  return res;
}

int STTX_BatchPipelineDepth(void) { return pipeline_depth(nullptr); }
int STTX_BatchPipelineDepthFor(ModelState* aCtx) { return aCtx ? pipeline_depth(aCtx) : STTX_BatchPipelineDepth(); }

int STTX_BatchSubmitDevice(ModelState* aCtx, const short* aDeviceAudio, unsigned int aStride, const unsigned int* aBufferSizes, unsigned int aBatch) {
  int ticket = -STT_ERR_FAIL_RUN_SESS;
  guarded([&]() { ticket = batch_submit(aCtx, aDeviceAudio, aStride, aBufferSizes, aBatch); return 0; }, 0);
  return ticket;
}

int STTX_BatchSubmit(ModelState* aCtx, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aBatch) {
  int ticket = -STT_ERR_FAIL_RUN_SESS;
  guarded([&]() { ticket = batch_submit_host(aCtx, aBuffers, aBufferSizes, aBatch); return 0; }, 0);
  return ticket;
}

char** STTX_BatchCollect(ModelState* aCtx, int aTicket, unsigned int* aCount) {
  char** res = nullptr;
  if (aCount) *aCount = 0;
  guarded([&]() {
    auto outs = batch_collect(aCtx, aTicket);
    res = (char**)malloc(sizeof(char*) * std::max<size_t>(1, outs.size()));
    for (size_t i = 0; i < outs.size(); ++i)
      res[i] = outs[i].empty() ? strdup("") : strdup(aCtx->alphabet_.Decode(outs[i][0].tokens.data(), (int)outs[i][0].tokens.size()).c_str());
    if (aCount) *aCount = (unsigned)outs.size();
    return 0;
  }, 0);
  return res;
}

char** STTX_BatchCollectScored(ModelState* aCtx, int aTicket, unsigned int* aCount, double* aConfidence) {
  char** res = nullptr;
  if (aCount) *aCount = 0;
  guarded([&]() {
    auto outs = batch_collect(aCtx, aTicket);
    res = (char**)malloc(sizeof(char*) * std::max<size_t>(1, outs.size()));
    for (size_t i = 0; i < outs.size(); ++i) {
      res[i] = outs[i].empty() ? strdup("") : strdup(aCtx->alphabet_.Decode(outs[i][0].tokens.data(), (int)outs[i][0].tokens.size()).c_str());
      if (aConfidence) aConfidence[i] = outs[i].empty() ? 0.0 : outs[i][0].confidence;
    }
    if (aCount) *aCount = (unsigned)outs.size();
    return 0;
  }, 0);
  return res;
}

Metadata** STTX_BatchCollectWithMetadata(ModelState* aCtx, int aTicket, unsigned int* aCount) {
  Metadata** res = nullptr;
  if (aCount) *aCount = 0;
  guarded([&]() {
    auto outs = batch_collect(aCtx, aTicket);
    res = (Metadata**)malloc(sizeof(Metadata*) * std::max<size_t>(1, outs.size()));
    for (size_t i = 0; i < outs.size(); ++i) res[i] = make_metadata(aCtx, outs[i]);
    if (aCount) *aCount = (unsigned)outs.size();
    return 0;
  }, 0);
  return res;
}

#ifdef STT_TEST_HOOKS
int STTX_DebugBatchProbs(ModelState* aCtx, int aTicket, float* aProbs, unsigned int aMaxFrames, unsigned int* aNumFrames) {
  return guarded([&]() { batch_probs(aCtx, aTicket, aProbs, aMaxFrames, aNumFrames); return (int)STT_ERR_OK; }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS
int STTX_SetTuning(const char* aName, int aValue) { return tuning_set(aName, aValue) == 0 ? STT_ERR_OK : STT_ERR_INVALID_SHAPE; }
int STTX_GetTuning(const char* aName, int* aValue) { return tuning_get(aName, aValue) == 0 ? STT_ERR_OK : STT_ERR_INVALID_SHAPE; }
void STTX_ConfigureRuntime(void) {
  // The batch path runs on up to eight HIP streams (three acoustic engines, four group slots' searches, the caller's own).  The
  // HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share a queue run
  // one after the other -- a group's output layers behind another group's 0.8 ms search launch.  The runtime reads the flag at
  // the first HIP call of the process, so this must run before it; a value the caller set stays.
  // Sixteen, not eight: a process that also keeps streaming cohorts on model replicas (two more streams each) ran them on shared
  // queues at eight -- the cohorts' passes serialised: 15.8 k x real time and 13 ms hop outliers against 23.2 k x and 4.6 ms at
  // sixteen, the batch path unchanged (profiles/r04_hw_queues.txt).
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

static void upload_batch_audio(ModelState* m, const short* const* bufs, const unsigned* sizes, unsigned B, unsigned& stride) {
  stride = 1;
  for (unsigned i = 0; i < B; ++i) stride = std::max(stride, sizes[i]);
  stride = (stride + 7) & ~7u;
  HIP_CHECK(hipSetDevice(m->device));
  m->ws_audio.reserve((size_t)B * stride * 2);
  HIP_CHECK(hipMemsetAsync(m->ws_audio.p, 0, (size_t)B * stride * 2, m->stream));
  for (unsigned i = 0; i < B; ++i)
    if (sizes[i]) HIP_CHECK(hipMemcpyAsync(m->ws_audio.as<int16_t>() + (size_t)i * stride, bufs[i], (size_t)sizes[i] * 2, hipMemcpyHostToDevice, m->stream));
  HIP_CHECK(hipStreamSynchronize(m->stream));
}

char** STTX_SpeechToTextBatch(ModelState* aCtx, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aBatch) {
  char** res = nullptr;
  guarded([&]() {
    unsigned stride;
    upload_batch_audio(aCtx, aBuffers, aBufferSizes, aBatch, stride);
    res = STTX_SpeechToTextBatchDevice(aCtx, aCtx->ws_audio.as<short>(), stride, aBufferSizes, aBatch);
    return 0;
  }, 0);
  return res;
}
Metadata** STTX_SpeechToTextBatchWithMetadata(ModelState* aCtx, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aBatch, unsigned int aNumResults) {
  Metadata** res = nullptr;
  guarded([&]() {
    unsigned stride;
    upload_batch_audio(aCtx, aBuffers, aBufferSizes, aBatch, stride);
    auto outs = batch_run(aCtx, aCtx->ws_audio.as<int16_t>(), stride, aBufferSizes, aBatch, aNumResults);
    res = (Metadata**)malloc(sizeof(Metadata*) * std::max(1u, aBatch));
    for (unsigned i = 0; i < aBatch; ++i) res[i] = make_metadata(aCtx, outs[i]);
    return 0;
  }, 0);
  return res;
}
void STTX_FreeStrings(char** aStrings, unsigned int aCount) {
  if (!aStrings) return;
  for (unsigned i = 0; i < aCount; ++i) free(aStrings[i]);
  free(aStrings);
}
void STTX_FreeMetadataArray(Metadata** aMetadata, unsigned int aCount) {
  if (!aMetadata) return;
  for (unsigned i = 0; i < aCount; ++i) STT_FreeMetadata(aMetadata[i]);
  free(aMetadata);
}
int STTX_SetProfiling(ModelState* aCtx, int aEnable) {
  Prof& p = prof_of(aCtx);
  p.on = aEnable != 0; p.phase_cycles = aEnable == 2; p.only = aEnable == 3 ? 5 : -1;
  return STT_ERR_OK;
}
int STTX_GetStageTimes(ModelState* aCtx, float* aMs, int aCap) {
  Prof& p = prof_of(aCtx);
  for (int i = 0; i < aCap && i < 8; ++i) aMs[i] = p.ms[i];
  return STT_ERR_OK;
}
int STTX_GetDecoderStats(ModelState* aCtx, unsigned long long* aOut4) {
  Prof& p = prof_of(aCtx);
  for (int i = 0; i < 4; ++i) aOut4[i] = p.dec_stats[i];
  return STT_ERR_OK;
}

int STTX_GetDecoderPhaseCycles(ModelState* aCtx, unsigned long long* aOut8) {
  Prof& p = prof_of(aCtx);
  for (int i = 0; i < 8; ++i) aOut8[i] = p.dec_phase[i];
  return STT_ERR_OK;
}

int STTX_GetDecoderStamps(ModelState* aCtx, unsigned long long* aOut64) {
  Prof& p = prof_of(aCtx);
  for (int i = 0; i < 64; ++i) aOut64[i] = p.dec_stamps[i];
  return STT_ERR_OK;
}

int STTX_GetGeometry(const ModelState* aCtx, int* o) {
  const Geometry& g = aCtx->g;
  o[0] = g.n_input; o[1] = g.n_context; o[2] = g.n_hidden; o[3] = g.n_classes; o[4] = g.n_steps;
  o[5] = g.sample_rate; o[6] = g.win_len; o[7] = g.win_step; o[8] = g.beam_width; o[9] = (int)aCtx->alphabet_.GetSpaceLabel();
  return STT_ERR_OK;
}

int STTX_ComputeMfcc(ModelState* m, const short* aBuffer, unsigned int aNumSamples, float* aOut, unsigned int aCapFrames, unsigned int* aNumFrames) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(m->device));
    const int n = (int)aNumSamples;
    const int T = n_frames_for(m->g, n);
    *aNumFrames = (unsigned)T;
    if ((unsigned)T > aCapFrames) return (int)STT_ERR_INVALID_SHAPE;
    const short* bufs[1] = {aBuffer};
    unsigned stride;
    upload_batch_audio(m, bufs, &aNumSamples, 1, stride);
    std::vector<int> nfr;
    m->run_mfcc(m->ws_audio.as<int16_t>(), &n, 1, (int)stride, T, nfr);
    HIP_CHECK(hipMemcpyAsync(aOut, m->ws_feats.p, (size_t)T * m->g.n_input * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}

int STTX_AcousticProbs(ModelState* m, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aBatch, float* aProbs,
                       unsigned int aMaxFrames, unsigned int* aNumFrames) {
  return guarded([&]() {
    if ((int)aBatch > lstm_max_rows(m->g.n_hidden)) return (int)STT_ERR_INVALID_SHAPE;  // one recurrent launch: 64 rows, 128 with 16 units per workgroup
    unsigned stride;
    upload_batch_audio(m, aBuffers, aBufferSizes, aBatch, stride);
    std::vector<int> hn(aBatch), nfr;
    int t_max = 1;
    for (unsigned b = 0; b < aBatch; ++b) { hn[b] = (int)aBufferSizes[b]; t_max = std::max(t_max, n_frames_for(m->g, hn[b])); }
    if ((unsigned)t_max > aMaxFrames) return (int)STT_ERR_INVALID_SHAPE;
    m->run_mfcc(m->ws_audio.as<int16_t>(), hn.data(), (int)aBatch, (int)stride, t_max, nfr);
    m->run_acoustic(m->ws_feats.as<float>(), m->ws_nframes.as<int>(), (int)aBatch, t_max, nullptr, nullptr, false);
    const int C = m->g.n_classes;
    for (unsigned b = 0; b < aBatch; ++b) {
      aNumFrames[b] = (unsigned)nfr[b];
      HIP_CHECK(hipMemcpyAsync(aProbs + (size_t)b * aMaxFrames * C, m->ws_probs.as<float>() + (size_t)b * t_max * C, (size_t)nfr[b] * C * 4,
                               hipMemcpyDeviceToHost, m->stream));
    }
    HIP_CHECK(hipStreamSynchronize(m->stream));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}

int STTX_InferChunk(ModelState* m, const float* aMfcc, unsigned int aNumFrames, const float* aStateC, const float* aStateH, float* aProbs,
                    float* aNewStateC, float* aNewStateH) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(m->device));
    const Geometry& g = m->g;
    const int T = (int)aNumFrames, kw = g.n_in1(), kp = m->x1_cols(), H = g.n_hidden, C = g.n_classes;
    if (m->i8) {
      std::vector<float> x1((size_t)T * kp, 0.0f);
      for (int t = 0; t < T; ++t) memcpy(&x1[(size_t)t * kp], aMfcc + (size_t)t * kw, (size_t)kw * 4);
      m->ws_x1.upload(x1.data(), x1.size() * 4, m->stream);
    } else {
      std::vector<_Float16> x1((size_t)T * kp, (_Float16)0.0f);
      for (int t = 0; t < T; ++t)
        for (int k = 0; k < kw; ++k) x1[(size_t)t * kp + k] = (_Float16)aMfcc[(size_t)t * kw + k];
      m->ws_x1.upload(x1.data(), x1.size() * 2, m->stream);
    }
    DevBuf c, h;
    c.upload(aStateC, (size_t)H * 4, m->stream); h.upload(aStateH, (size_t)H * 4, m->stream);
    m->ws_probs.reserve((size_t)T * C * 4);
    m->run_acoustic_rows(m->ws_x1.p, 1, T, c.as<float>(), h.as<float>(), true, m->ws_probs.as<float>(), T);
    HIP_CHECK(hipMemcpyAsync(aProbs, m->ws_probs.p, (size_t)T * C * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipMemcpyAsync(aNewStateC, c.p, (size_t)H * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipMemcpyAsync(aNewStateH, h.p, (size_t)H * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}

// ---- decoder on caller-supplied emissions
struct STTX_Decoder {
  ModelState* m;
  std::shared_ptr<ScorerDev> scorer;
  std::map<std::string, float> hot;
  DecoderBatch db;
  DecParams p;
  DevBuf probs, fbegin, fcount, wide, stamps;
  // A decoder runs on one of the model's decoder streams (ModelState::decoder_stream: a pool of eight, dealt round-robin) and result blocks
  // of its OWN: several decoders of one model may be driven side by side from several host threads (one decoder = one workgroup per
  // stream: 64 streams are a quarter of the chip) -- round 6, bench.py's decoder-stage workloads.
  hipStream_t st = nullptr;   // (the model's: not destroyed here)
  DevBuf ws_out;
  PinnedBuf h_out;
  HotTables ht;
  int prof = 0;            // STTX_DecoderSetProfiling
  float search_ms = 0;     // HIP-event time of the search launches since profiling was switched on
};
int STTX_DecoderCreate(ModelState* m, unsigned int aNumStreams, unsigned int aBeamWidth, double aCutoffProb, unsigned int aCutoffTopN, STTX_Decoder** retval) {
  *retval = nullptr;
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(m->device));
    std::unique_ptr<STTX_Decoder> d(new STTX_Decoder());
    d->m = m; d->scorer = m->scorer_; d->hot = m->hot_words_;
    d->st = m->decoder_stream();
    m->decoder_create(d->db, (int)aNumStreams, (int)aBeamWidth, 256, d->scorer);  // arenas for 256 frames up front, like a stream's; longer inputs grow them (decoder_reserve)
    HIP_CHECK(hipStreamSynchronize(m->stream));   // (the table was initialised on the model's stream; everything after runs on the decoder's own)
    d->p = DecParams{};
    d->p.C = m->g.n_classes; d->p.blank = d->p.C - 1; d->p.beam = (int)aBeamWidth; d->p.cutoff_top_n = (int)aCutoffTopN; d->p.cutoff_prob = aCutoffProb;
    *retval = d.release();
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_CREATE_STREAM);
}
int STTX_DecoderNext(STTX_Decoder* d, const float* aProbs, unsigned int aStride, const unsigned int* aNumFrames) {
  return guarded([&]() {
    ModelState* m = d->m;
    HIP_CHECK(hipSetDevice(m->device));
    const int n = d->db.n_streams;
    std::vector<int> more(n), zeros(n, 0);
    for (int i = 0; i < n; ++i) more[i] = (int)aNumFrames[i];
    m->decoder_reserve(d->db, more, d->st);
    d->probs.upload(aProbs, (size_t)n * aStride * d->p.C * 4, d->st);
    d->fbegin.upload(zeros.data(), n * 4, d->st);
    d->fcount.upload(more.data(), n * 4, d->st);
    d->p.t_max = (int)aStride;
    DevScorer ds = m->current_scorer(d->scorer, d->hot, d->ht);
    int max_frames = 1;
    for (int i = 0; i < n; ++i) max_frames = std::max(max_frames, more[i]);
    d->wide.reserve(ctc_rows_ws_bytes(d->p, n, max_frames));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (d->prof) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventRecord(e0, d->st)); }
    launch_ctc_next(d->p, ds, m->dev_alphabet, d->db.table.as<DecStream>(), n, d->probs.as<float>(), d->fbegin.as<int>(), d->fcount.as<int>(), d->st,
                    max_frames, d->wide.p);
    if (d->prof) HIP_CHECK(hipEventRecord(e1, d->st));
    HIP_CHECK(hipStreamSynchronize(d->st));
    if (tune().debug_scribble & 32) usleep(2000);   // (experiment: time for the runtime's asynchronous scratch reclaim)
    if (d->prof) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1)); d->search_ms += ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    HIP_CHECK(hipGetLastError());
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
int STTX_DecoderDecode(const STTX_Decoder* d, unsigned int aNumResults, unsigned int aMaxLen, unsigned int* aTokens, unsigned int* aTimesteps, int* aLens,
                       double* aConfidences, int* aNumResultsOut) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(d->m->device));
    STTX_Decoder* dm = const_cast<STTX_Decoder*>(d);   // (workspaces only)
    auto outs = decode_table(*d->m, d->db.table.as<DecStream>(), d->db.n_streams, d->db.beam, d->db.C, d->scorer, d->hot, dm->ht, aNumResults, (int)aMaxLen, d->st, &dm->ws_out, &dm->h_out);
    for (size_t i = 0; i < outs.size(); ++i) {
      aNumResultsOut[i] = (int)outs[i].size();
      for (size_t r = 0; r < outs[i].size(); ++r) {
        const size_t ob = i * aNumResults + r;
        aLens[ob] = (int)outs[i][r].tokens.size();
        aConfidences[ob] = outs[i][r].confidence;
        for (size_t j = 0; j < outs[i][r].tokens.size(); ++j) {
          aTokens[ob * aMaxLen + j] = outs[i][r].tokens[j];
          if (aTimesteps) aTimesteps[ob * aMaxLen + j] = j < outs[i][r].timesteps.size() ? outs[i][r].timesteps[j] : 0;
        }
      }
    }
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
int STTX_DecoderBeam(const STTX_Decoder* d, unsigned int aStream, float* aScore, float* aPb, float* aPnb, int* aChar, unsigned int aCap) {
  int n = 0;
  guarded([&]() {
    ModelState* m = d->m;
    HIP_CHECK(hipSetDevice(m->device));
    DecStream S;
    HIP_CHECK(hipMemcpy(&S, d->db.table.as<DecStream>() + aStream, sizeof(DecStream), hipMemcpyDeviceToHost));
    n = std::min<int>(S.n, (int)aCap);
    HIP_CHECK(hipMemcpy(aScore, S.score, n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(aPb, S.pb, n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(aPnb, S.pnb, n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(aChar, S.ch, n * 4, hipMemcpyDeviceToHost));
    (void)m;
    return 0;
  }, 0);
  return n;
}
int STTX_DecoderSetProfiling(STTX_Decoder* d, int aLevel) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(d->m->device));
    d->prof = aLevel; d->search_ms = 0;
    d->p.phase_cycles = aLevel >= 2 ? 1 : 0;
    d->p.stamps = nullptr;
    if (aLevel >= 2) {
      d->stamps.reserve((size_t)d->db.n_streams * 64 * 8);
      HIP_CHECK(hipMemset(d->stamps.p, 0, (size_t)d->db.n_streams * 64 * 8));
      d->p.stamps = d->stamps.as<unsigned long long>();
    }
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
int STTX_DecoderGetProfile(const STTX_Decoder* d, unsigned long long* aPhase8, unsigned long long* aStamps64, float* aSearchMs) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(d->m->device));
    std::vector<DecStream> tb(d->db.n_streams);
    HIP_CHECK(hipMemcpy(tb.data(), d->db.table.p, sizeof(DecStream) * tb.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) { aPhase8[k] = 0; for (auto& S : tb) aPhase8[k] += S.phase[k]; }
    for (int k = 0; k < 64; ++k) aStamps64[k] = 0;
    if (d->p.stamps) {
      std::vector<unsigned long long> st((size_t)d->db.n_streams * 64);
      HIP_CHECK(hipMemcpy(st.data(), d->stamps.p, st.size() * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < st.size(); ++i) aStamps64[i & 63] += st[i];
    }
    *aSearchMs = d->search_ms;
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
int STTX_DecoderStats(const STTX_Decoder* d, unsigned long long* aOut4) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(d->m->device));
    std::vector<DecStream> tb(d->db.n_streams);
    HIP_CHECK(hipMemcpy(tb.data(), d->db.table.p, sizeof(DecStream) * tb.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) aOut4[k] = 0;
    for (auto& S : tb) for (int k = 0; k < 4; ++k) aOut4[k] += S.stat[k];
    aOut4[3] |= 0;  // probes
    int err = 0;
    for (auto& S : tb) err |= S.error;
    return err ? (int)STT_ERR_FAIL_RUN_SESS : (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
int STTX_DecoderErrorBits(const STTX_Decoder* d, int* aBits) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(d->m->device));
    std::vector<DecStream> tb(d->db.n_streams);
    HIP_CHECK(hipMemcpy(tb.data(), d->db.table.p, sizeof(DecStream) * tb.size(), hipMemcpyDeviceToHost));
    int err = 0;
    for (auto& S : tb) err |= S.error;
    *aBits = err;
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
void STTX_DecoderFree(STTX_Decoder* d) { delete d; }

// ---- kernel-level hooks
#ifdef STT_TEST_HOOKS
int STTX_TestDense(int M, int N, int K, const float* aX, const float* aW, const float* aBias, float aClip, int aEpilogue, float* aY) {
  return guarded([&]() {
    if (N % 128 || K % 64) return (int)STT_ERR_INVALID_SHAPE;
    HIP_CHECK(hipSetDevice(g_device));
    std::vector<_Float16> x((size_t)M * K), wt((size_t)N * K);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (_Float16)aX[i];
    for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) wt[(size_t)n * K + k] = (_Float16)aW[(size_t)k * N + n];
    DevBuf dx, dw, db, dy;
    dx.upload(x.data(), x.size() * 2); dw.upload(wt.data(), wt.size() * 2); db.upload(aBias, (size_t)N * 4);
    dy.reserve((size_t)M * N * 4);
    DenseArgs d{};
    d.wt = dw.as<_Float16>(); d.x = dx.as<_Float16>(); d.bias = db.as<float>(); d.y = dy.p; d.M = M; d.N = N; d.K = K; d.ldx = K; d.ldy = N; d.relu_clip = aClip;
    if (tune().dense_solo_test >= 0) d.solo = tune().dense_solo_test;  // test hook: the forms that run beside the recurrence
    launch_dense(d, aEpilogue == 0 ? DENSE_EPI_RELU_F16 : DENSE_EPI_BIAS_F32, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipGetLastError());
    if (aEpilogue == 0) {
      std::vector<_Float16> y((size_t)M * N);
      HIP_CHECK(hipMemcpy(y.data(), dy.p, y.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < y.size(); ++i) aY[i] = (float)y[i];
    } else {
      HIP_CHECK(hipMemcpy(aY, dy.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    }
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS

// TFLite's hybrid FULLY_CONNECTED on the int8 MFMA path (kernels.h: launch_quantize_rows + launch_dense_hybrid_i8), as a test hook:
// x f32 [M][K], wq int8 [N][K], wscale [n_scales = 1 or N], bias [N] -> y f32 [M][N] (and, if asked for, the quantised rows and their
// scales); aReps timed repetitions of quantisation + product (HIP events) -> *aElapsedMs per repetition.
int STTX_GetAcousticMode(const ModelState* m) { return m && m->i8 ? 1 : 0; }
#ifdef STT_TEST_HOOKS
int STTX_DebugSlowRows(ModelState* m, unsigned int* aRows) {
  return guarded([&]() {
    if (!m || !aRows) return (int)STT_ERR_INVALID_SHAPE;
    HIP_CHECK(hipSetDevice(m->device));
    HIP_CHECK(hipDeviceSynchronize());
    *aRows = 0;
    if (m->ws_slow.p) HIP_CHECK(hipMemcpy(aRows, m->ws_slow.p, 4, hipMemcpyDeviceToHost));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS

#ifdef STT_TEST_HOOKS
int STTX_TestHybridChain(ModelState* m, const float* aWindows, unsigned int aB, unsigned int aT, const float* aC, const float* aH, float* aL3, int* aAccX, float* aHAll,
                         float* aLogits, float* aProbs, float* aNewC, float* aNewH, unsigned int* aSlowRows, float* aLstmMs) {
  return guarded([&]() {
    if (!m->i8 || !aB || !aT || (int)aB > 128) return (int)STT_ERR_INVALID_SHAPE;
    HIP_CHECK(hipSetDevice(m->device));
    const Geometry& g = m->g;
    const int B = (int)aB, T = (int)aT, M = B * T, kw = g.n_in1(), kp = m->x1_cols(), H = g.n_hidden, C = g.n_classes, CP = g.c_pad8();
    std::vector<float> x1((size_t)M * kp, 0.0f);
    for (int r = 0; r < M; ++r) memcpy(&x1[(size_t)r * kp], aWindows + (size_t)r * kw, (size_t)kw * 4);
    m->ws_x1.upload(x1.data(), x1.size() * 4, m->stream);
    DevBuf c, h;
    std::vector<float> zeros((size_t)B * H, 0.0f);
    c.upload(aC ? aC : zeros.data(), (size_t)B * H * 4, m->stream); h.upload(aH ? aH : zeros.data(), (size_t)B * H * 4, m->stream);
    m->ws_probs.reserve((size_t)M * C * 4);
    unsigned slow0 = 0, slow1 = 0;
    if (m->ws_slow.p) HIP_CHECK(hipMemcpy(&slow0, m->ws_slow.p, 4, hipMemcpyDeviceToHost));
    if (aLstmMs) { HIP_CHECK(hipEventCreate(&m->dbg_ev_[0])); HIP_CHECK(hipEventCreate(&m->dbg_ev_[1])); }
    try { m->run_acoustic_rows(m->ws_x1.p, B, T, c.as<float>(), h.as<float>(), true, m->ws_probs.as<float>(), T); }
    catch (...) { for (auto& e : m->dbg_ev_) if (e) { (void)hipEventDestroy(e); e = nullptr; } throw; }
    HIP_CHECK(hipStreamSynchronize(m->stream));
    if (aLstmMs) {   // prep + aT recurrent steps, HIP events on the stream they run on
      HIP_CHECK(hipEventElapsedTime(aLstmMs, m->dbg_ev_[0], m->dbg_ev_[1]));
      for (auto& e : m->dbg_ev_) { (void)hipEventDestroy(e); e = nullptr; }
    }
    HIP_CHECK(hipMemcpy(&slow1, m->ws_slow.p, 4, hipMemcpyDeviceToHost));
    if (aSlowRows) *aSlowRows = slow1 - slow0;
    if (aL3) HIP_CHECK(hipMemcpy(aL3, m->ws_a.p, (size_t)M * H * 4, hipMemcpyDeviceToHost));
    if (aAccX) HIP_CHECK(hipMemcpy(aAccX, m->ws_xproj.p, (size_t)M * 4 * H * 4, hipMemcpyDeviceToHost));
    if (aHAll) HIP_CHECK(hipMemcpy(aHAll, m->ws_hall.p, (size_t)M * H * 4, hipMemcpyDeviceToHost));
    if (aLogits) HIP_CHECK(hipMemcpy2D(aLogits, (size_t)C * 4, m->ws_logits.p, (size_t)CP * 4, (size_t)C * 4, (size_t)M, hipMemcpyDeviceToHost));
    if (aProbs) HIP_CHECK(hipMemcpy(aProbs, m->ws_probs.p, (size_t)M * C * 4, hipMemcpyDeviceToHost));
    if (aNewC) HIP_CHECK(hipMemcpy(aNewC, c.p, (size_t)B * H * 4, hipMemcpyDeviceToHost));
    if (aNewH) HIP_CHECK(hipMemcpy(aNewH, h.p, (size_t)B * H * 4, hipMemcpyDeviceToHost));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS

#ifdef STT_TEST_HOOKS
int STTX_TestDenseHybrid(const float* aX, unsigned int aM, unsigned int aK, const signed char* aWq, const float* aWScale, unsigned int aNScales, const float* aBias,
                           unsigned int aN, float* aY, signed char* aQ, float* aRowScale, unsigned int aReps, float* aElapsedMs, int aEpi, float aClip) {
  return guarded([&]() {
    const size_t M = aM, K = aK, N = aN;
    if (!M || K % 128 != 0 || (N % 256 != 0 && !(M <= 16 && N % 64 == 0)) || (aNScales != 1 && aNScales != aN)) return (int)STT_ERR_INVALID_SHAPE;
    hipStream_t st = nullptr;
    DevBuf x, q, rs, wq, ws, b, y;
    x.upload(aX, M * K * 4, st); wq.upload(aWq, N * K, st); ws.upload(aWScale, (size_t)aNScales * 4, st); b.upload(aBias, N * 4, st);
    q.reserve(M * K); rs.reserve(M * 4); y.reserve(M * N * 4);
    auto once = [&]() {
      launch_quantize_rows(x.as<float>(), q.as<signed char>(), rs.as<float>(), (int)M, (int)K, st);
      launch_dense_hybrid_i8(q.as<signed char>(), rs.as<float>(), wq.as<signed char>(), ws.as<float>(), (int)aNScales, b.as<float>(), y.as<float>(), (int)M, (int)N, (int)K, st,
                             aEpi == 1 ? DENSE_EPI_I8_RELU_F32 : DENSE_EPI_I8_F32, aClip);
    };
    once();
    HIP_CHECK(hipStreamSynchronize(st));
    if (aReps && aElapsedMs) {
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
      HIP_CHECK(hipEventRecord(e0, st));
      for (unsigned r = 0; r < aReps; ++r) once();
      HIP_CHECK(hipEventRecord(e1, st));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      *aElapsedMs = ms / (float)aReps;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    HIP_CHECK(hipMemcpy(aY, y.p, M * N * 4, hipMemcpyDeviceToHost));
    if (aQ) HIP_CHECK(hipMemcpy(aQ, q.p, M * K, hipMemcpyDeviceToHost));
    if (aRowScale) HIP_CHECK(hipMemcpy(aRowScale, rs.p, M * 4, hipMemcpyDeviceToHost));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS
#ifdef STT_TEST_HOOKS
int STTX_TestLstmSteps(ModelState* m, unsigned int aBatch, unsigned int aSteps, unsigned int aPeriod, int aGraph, const float* aXproj, float* aC, float* aH,
                       unsigned short* aHAll, float* aElapsedMs) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(m->device));
    const int H = m->g.n_hidden, B = (int)aBatch, NT = lstm_nt_for_batch(B);
    if (NT < 0 || B > lstm_max_rows(H) || !aPeriod || !aSteps) return (int)STT_ERR_INVALID_SHAPE;
    DevBuf xp, c, hf, hall, hp0, hp1;
    xp.upload(aXproj, (size_t)aPeriod * B * 4 * H * 4, m->stream);
    const size_t hp_bytes = (size_t)(H / 32) * NT * 64 * 16;
    c.reserve((size_t)B * H * 4); hf.reserve((size_t)B * H * 4); hall.reserve((size_t)aPeriod * B * H * 2); hp0.reserve(hp_bytes); hp1.reserve(hp_bytes);
    HIP_CHECK(hipMemsetAsync(c.p, 0, (size_t)B * H * 4, m->stream));
    HIP_CHECK(hipMemsetAsync(hp0.p, 0, hp_bytes, m->stream));
    HIP_CHECK(hipMemsetAsync(hp1.p, 0, hp_bytes, m->stream));
    LstmArgs l{};
    l.whp = m->whp.as<_Float16>(); l.xproj = xp.as<float>(); l.c = c.as<float>(); l.h_all = hall.as<_Float16>();
    l.n_hidden = H; l.batch = B; l.passes = 0; l.prio = tune().lstm_prio; l.probe = 1;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    DevBuf stamps;
    const int n_wg = H / lstm_units_per_wg(H);
    const size_t stamp_n = (size_t)aSteps * n_wg * 16;
    if (tune().lstm_stamps) { stamps.reserve(stamp_n * 8); HIP_CHECK(hipMemsetAsync(stamps.p, 0, stamp_n * 8, m->stream)); l.stamps = stamps.as<unsigned long long>(); }
    auto steps = [&]() {
      for (unsigned t = 0; t < aSteps; ++t) {
        l.stamp_step = (int)t;
        l.hp_in = (t & 1) ? hp1.as<_Float16>() : hp0.as<_Float16>();
        l.hp_out = (t & 1) ? hp0.as<_Float16>() : hp1.as<_Float16>();
        l.t = (int)(t % aPeriod);
        l.h_f32 = (t + 1 == aSteps) ? hf.as<float>() : nullptr;
        launch_lstm_step(l, NT, m->stream);
      }
    };
    // co-tenant: x-projection GEMMs of a 128-row, 48-frame chunk on a second stream, in the form the batch path runs beside the recurrence
    DevBuf cx, cy;
    hipStream_t cst = nullptr;
    hipEvent_t c0 = nullptr, c1 = nullptr;
    const int n_co = tune().lstm_cotenant;
    auto cotenant = [&]() {
      if (n_co <= 0) return;
      const int M = 6144;
      cx.reserve((size_t)M * H * 2); cy.reserve((size_t)M * 4 * H * 4);
      HIP_CHECK(hipMemsetAsync(cx.p, 0, (size_t)M * H * 2, m->stream));
      HIP_CHECK(hipStreamSynchronize(m->stream));
      HIP_CHECK(hipStreamCreateWithFlags(&cst, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreate(&c0)); HIP_CHECK(hipEventCreate(&c1));
      DenseArgs d{};
      d.wt = m->wxt.as<_Float16>(); d.x = cx.as<_Float16>(); d.bias = m->bl.as<float>(); d.y = cy.p; d.M = M; d.N = 4 * H; d.K = H; d.ldx = H; d.ldy = 4 * H;
      d.relu_clip = 20.0f; d.solo = tune().dense_solo; d.lds_floor = d.solo ? 0 : tune().dense_lds_kb * 1024;
      launch_dense(d, DENSE_EPI_BIAS_F32, cst);  // (first launch: code object, attributes)
      HIP_CHECK(hipStreamSynchronize(cst));
      HIP_CHECK(hipEventRecord(c0, cst));
      for (int i = 0; i < n_co; ++i) launch_dense(d, DENSE_EPI_BIAS_F32, cst);
      HIP_CHECK(hipEventRecord(c1, cst));
    };
    if (aGraph) {
      l.hp_in = hp0.as<_Float16>(); l.hp_out = hp1.as<_Float16>(); l.t = 0; l.h_f32 = nullptr;
      // (first launch of an instantiation sets its function attributes: not inside a capture)
      DevBuf sc, sh0, sh1; sc.reserve((size_t)B * H * 4); sh0.reserve(hp_bytes); sh1.reserve(hp_bytes);
      HIP_CHECK(hipMemsetAsync(sh0.p, 0, hp_bytes, m->stream));
      LstmArgs w = l; w.c = sc.as<float>(); w.hp_in = sh0.as<_Float16>(); w.hp_out = sh1.as<_Float16>();
      HIP_CHECK(hipMemsetAsync(sc.p, 0, (size_t)B * H * 4, m->stream));
      launch_lstm_step(w, NT, m->stream);
      HIP_CHECK(hipStreamSynchronize(m->stream));
      hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
      std::lock_guard<std::recursive_mutex> capturing(hip_capture_mutex());
      HIP_CHECK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeRelaxed));
      try { steps(); } catch (...) { (void)hipStreamEndCapture(m->stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
      HIP_CHECK(hipStreamEndCapture(m->stream, &graph));
      HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      cotenant();
      (void)hipEventRecord(e0, m->stream);
      const hipError_t le = hipGraphLaunch(exec, m->stream);
      (void)hipEventRecord(e1, m->stream);
      const hipError_t se = hipStreamSynchronize(m->stream);
      (void)hipGraphExecDestroy(exec);
      HIP_CHECK(le); HIP_CHECK(se);
    } else {
      {  // (the first launch of an instantiation loads its code object: not inside the timed span)
        DevBuf sc, sh0, sh1; sc.reserve((size_t)B * H * 4); sh0.reserve(hp_bytes); sh1.reserve(hp_bytes);
        HIP_CHECK(hipMemsetAsync(sh0.p, 0, hp_bytes, m->stream)); HIP_CHECK(hipMemsetAsync(sc.p, 0, (size_t)B * H * 4, m->stream));
        LstmArgs w = l; w.c = sc.as<float>(); w.hp_in = sh0.as<_Float16>(); w.hp_out = sh1.as<_Float16>(); w.t = 0; w.h_f32 = nullptr;
        launch_lstm_step(w, NT, m->stream);
        HIP_CHECK(hipStreamSynchronize(m->stream));
      }
      cotenant();
      HIP_CHECK(hipEventRecord(e0, m->stream));
      steps();
      HIP_CHECK(hipEventRecord(e1, m->stream));
    }
    if (cst) {
      HIP_CHECK(hipStreamSynchronize(cst));
      float cms = 0.0f;
      (void)hipEventElapsedTime(&cms, c0, c1);
      fprintf(stderr, "LSTM_COTENANT %d GEMMs (6144 x %d x %d, form %d): %.1f us each = %.3f PF/s\n", n_co, 4 * H, H, tune().dense_solo, 1e3 * cms / n_co,
              2.0 * 6144 * 4.0 * H * H / (1e-3 * cms / n_co) / 1e15);
      (void)hipEventDestroy(c0); (void)hipEventDestroy(c1); (void)hipStreamDestroy(cst);
    }
    HIP_CHECK(hipMemcpyAsync(aC, c.p, (size_t)B * H * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipMemcpyAsync(aH, hf.p, (size_t)B * H * 4, hipMemcpyDeviceToHost, m->stream));
    if (aHAll) HIP_CHECK(hipMemcpyAsync(aHAll, hall.p, (size_t)aPeriod * B * H * 2, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    HIP_CHECK(hipGetLastError());
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (l.stamps) {  // per step: first entry .. last exit over all waves; gaps between steps; where the waves are inside a step (units of 10 ns)
      std::vector<unsigned long long> st(stamp_n);
      HIP_CHECK(hipMemcpy(st.data(), stamps.p, stamp_n * 8, hipMemcpyDeviceToHost));
      double span = 0, gap = 0, loop = 0, bar = 0, epi = 0, skew_in = 0, loop_max = 0; int ns = 0;
      unsigned long long prev_exit = 0;
      for (unsigned t = 0; t < aSteps; ++t) {
        unsigned long long e_min = ~0ull, e_max = 0, x_max = 0; double lp = 0, br = 0, ep = 0, lpm = 0; int nw = 0;
        for (int w = 0; w < n_wg * 4; ++w) {
          const unsigned long long* s4 = &st[((size_t)t * n_wg * 4 + w) * 4];
          if (!s4[0] || !s4[3]) continue;
          e_min = std::min(e_min, s4[0]); e_max = std::max(e_max, s4[0]); x_max = std::max(x_max, s4[3]);
          lp += (double)(s4[1] - s4[0]); br += (double)(s4[2] - s4[1]); ep += (double)(s4[3] - s4[2]); lpm = std::max(lpm, (double)(s4[1] - s4[0])); ++nw;
        }
        if (!nw) continue;
        if (t >= 8) { span += (double)(x_max - e_min); if (prev_exit) gap += (double)e_min - (double)prev_exit; loop += lp / nw; bar += br / nw; epi += ep / nw; skew_in += (double)(e_max - e_min); loop_max += lpm; ++ns; }
        prev_exit = x_max;
      }
      if (ns) fprintf(stderr, "LSTM_STAMPS rows %d probe %d form %d: per step (us) span %.2f gap %.2f | entry skew %.2f k-loop avg %.2f max %.2f barrier-wait %.2f epilogue %.2f\n", B, tune().lstm_probe,
                      tune().lstm_form, span / ns / 100, gap / ns / 100, skew_in / ns / 100, loop / ns / 100, loop_max / ns / 100, bar / ns / 100, epi / ns / 100);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (aElapsedMs) *aElapsedMs = ms;
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS

#ifdef STT_TEST_HOOKS
int STTX_TestMath(int aOp, const float* aA, const float* aB, float* aOut, unsigned int aCount) {
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(g_device));
    DevBuf a, b, o;
    a.upload(aA, (size_t)aCount * 4);
    b.upload(aB ? aB : aA, (size_t)aCount * 4);
    o.reserve((size_t)aCount * 4);
    launch_test_math(aOp, a.as<float>(), b.as<float>(), o.as<float>(), aCount, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(aOut, o.p, (size_t)aCount * 4, hipMemcpyDeviceToHost));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS
int STTX_InspectModel(const char* aModelBuffer, unsigned int aBufferSize, STTX_ModelInfo* aInfo) {
  if (!aModelBuffer || !aInfo) return STT_ERR_FAIL_CREATE_MODEL;
  ModelTensors storage; ModelView v; std::string err;
  int rc = STT_ERR_FAIL_CREATE_MODEL;
  try { rc = parse_model_file(aModelBuffer, aBufferSize, storage, v, err); } catch (const std::exception& e) { err = e.what(); }
  if (rc != STT_ERR_OK) { std::cerr << err << std::endl; return rc; }
  const Geometry& g = v.g;
  *aInfo = STTX_ModelInfo{g.n_input, g.n_context, g.n_hidden, g.n_classes, g.n_steps, g.sample_rate, g.win_len, g.win_step, g.beam_width,
                          g.relu_clip, (unsigned)v.alphabet_bytes, looks_like_tflite(aModelBuffer, aBufferSize) ? 1 : 0,
                          (v.quant && v.quant->all_int8()) ? 1 : 0, storage.asymmetric_inputs ? 1 : 0};
  return STT_ERR_OK;
}

int STTX_ReadModelTensor(const char* aModelBuffer, unsigned int aBufferSize, int aIndex, void* aOut, unsigned long long aCapBytes,
                         unsigned long long* aBytes) {
  if (!aModelBuffer || aIndex < 0 || aIndex > 12) return STT_ERR_FAIL_CREATE_MODEL;
  ModelTensors storage; ModelView v; std::string err;
  int rc = STT_ERR_FAIL_CREATE_MODEL;
  try { rc = parse_model_file(aModelBuffer, aBufferSize, storage, v, err); } catch (const std::exception& e) { err = e.what(); }
  if (rc != STT_ERR_OK) { std::cerr << err << std::endl; return rc; }
  const void* src = aIndex == 12 ? (const void*)v.alphabet : (const void*)v.t[aIndex];
  const unsigned long long n = aIndex == 12 ? v.alphabet_bytes : v.count[aIndex] * 4ull;
  if (aBytes) *aBytes = n;
  if (aOut && src) memcpy(aOut, src, (size_t)std::min(n, aCapBytes));
  return STT_ERR_OK;
}

#ifdef STT_TEST_HOOKS
int STTX_DebugLimitArena(int aFrames) { g_debug_arena_frames = aFrames; return STT_ERR_OK; }
#endif  // STT_TEST_HOOKS

// Host only: walk label sequences through the dictionary tables a scorer package parses into (the repacked automaton, or -- tunable
// dict_tree_mb -- its unfolding into a tree).  aLabels holds aNumSeq sequences of aLen labels each (label ids, -1 = end of sequence);
// aOut[seq * aLen + k] = -1 once a label had no arc, else bit 0 = "a word may end after label k" (the state has a space arc) and bit 1 =
// "the tables are the tree".  What a test compares between the two forms: the language and the word ends are the same.
#ifdef STT_TEST_HOOKS
int STTX_TestDictionaryWalk(const char* aScorer, unsigned int aScorerBytes, int aSpaceLabel, const int* aLabels, unsigned int aNumSeq, unsigned int aLen, int* aOut) {
  return guarded([&]() {
    std::vector<char> copy((size_t)aScorerBytes + 16, 0);
    memcpy(copy.data(), aScorer, aScorerBytes);
    HostScorer hs;
    const int rc = parse_scorer(reinterpret_cast<const uint8_t*>(copy.data()), aScorerBytes, aSpaceLabel, false, hs);
    if (rc != STT_ERR_OK) return rc;
    for (unsigned q = 0; q < aNumSeq; ++q) {
      uint32_t st = (uint32_t)hs.fst_start;
      bool dead = false;
      for (unsigned k = 0; k < aLen; ++k) {
        int& out = aOut[(size_t)q * aLen + k];
        const int lab = aLabels[(size_t)q * aLen + k];
        if (lab < 0) dead = true;
        if (dead) { out = -1; continue; }
        uint32_t next = 0xFFFFFFFFu;
        for (uint32_t a = hs.fst_pos[st]; a < hs.fst_pos[st + 1]; ++a) if (hs.fst_arcs[a].x == (uint32_t)lab + 1u) { next = hs.fst_arcs[a].y; break; }
        if (next == 0xFFFFFFFFu) { dead = true; out = -1; continue; }
        if (hs.fst_tree && lab != aSpaceLabel) {  // the tree's invariant: the child along arc a is node a + 1
          uint32_t a = hs.fst_pos[st]; while (hs.fst_arcs[a].x != (uint32_t)lab + 1u) ++a;
          if (next != a + 1u) return (int)STT_ERR_SCORER_INVALID_TRIE;
        }
        st = next;
        out = (hs.fst_has_space[st] ? 1 : 0) | (hs.fst_tree ? 2 : 0);
      }
    }
    return (int)STT_ERR_OK;
  }, STT_ERR_SCORER_INVALID_TRIE);
}
#endif  // STT_TEST_HOOKS

#ifdef STT_TEST_HOOKS
int STTX_TestLm(const char* aLm, unsigned int aLmBytes, const char* const* aWords, unsigned int aNumWords, int aBos, int aMode, float* aProbs, int* aLens) {
  return guarded([&]() {
    if (aMode == 0 || aMode == 3) {  // host: parse + hashed index (3: + the code-point blocks) + FullScore chain, no GPU
      std::vector<char> copy((size_t)aLmBytes + 16, 0);
      memcpy(copy.data(), aLm, aLmBytes);
      HostScorer hs;
      const int rc = parse_scorer(reinterpret_cast<const uint8_t*>(copy.data()), aLmBytes, -1, true, hs);
      if (rc != STT_ERR_OK) return rc;
      if (!hs.lmi_ok && !(hs.probing && aMode == 0)) return (int)STT_ERR_SCORER_INVALID_LM;   // (a probing binary has no index: its own tables answer)
      KState st[2] = {};
      int cur = 0;
      if (aBos) { st[0].length = 1; st[0].words[0] = hs.bos_index; st[0].backoff[0] = hs.bos_backoff; }
      for (unsigned i = 0; i < aNumWords; ++i) {
        int nl = 0; uint32_t wi = 0;
        if (aMode == 3) {   // every word must be ONE code point of the BMP (UTF-8); the blocks must exist (tunable cp_blocks)
          if (!hs.cpb_ok) return (int)STT_ERR_SCORER_INVALID_LM;
          const unsigned char* w = reinterpret_cast<const unsigned char*>(aWords[i]);
          const size_t wl = strlen(aWords[i]);
          uint32_t cp = 0;
          if (wl == 1 && w[0] < 0x80) cp = w[0];
          else if (wl == 2 && (w[0] & 0xE0) == 0xC0) cp = ((uint32_t)(w[0] & 0x1F) << 6) | (w[1] & 0x3F);
          else if (wl == 3 && (w[0] & 0xF0) == 0xE0) cp = ((uint32_t)(w[0] & 0x0F) << 12) | ((uint32_t)(w[1] & 0x3F) << 6) | (w[2] & 0x3F);
          else return (int)STT_ERR_INVALID_SHAPE;
          aProbs[i] = hs.full_score_blocks(st[cur], cp, st[cur ^ 1], nl, wi);
        } else
        aProbs[i] = hs.probing ? hs.full_score_probing(st[cur], aWords[i], strlen(aWords[i]), st[cur ^ 1], nl, wi)
                               : hs.full_score_indexed(st[cur], aWords[i], strlen(aWords[i]), st[cur ^ 1], nl, wi);
        aLens[i] = nl;
        cur ^= 1;
      }
      return (int)STT_ERR_OK;
    }
    HIP_CHECK(hipSetDevice(g_device));
    ScorerDev sc;
    const int rc = sc.LoadLmOnly(aLm, aLmBytes);
    if (rc != STT_ERR_OK) return rc;
    if (aMode == 2 && (!sc.dev.lmi || sc.dev.order > 5 || !sc.dev.uni_in_vtab)) return (int)STT_ERR_SCORER_INVALID_LM;
    if (aMode == 4 && !sc.dev.cpt) return (int)STT_ERR_SCORER_INVALID_LM;   // (the bigram blocks on the DEVICE: tunable cp_blocks, a code-point model)
    std::vector<uint64_t> hs(aNumWords);
    for (unsigned i = 0; i < aNumWords; ++i) {
      hs[i] = stt_murmur64a(aWords[i], strlen(aWords[i]));
      if (aMode == 4) {   // the unit's bytes, first byte lowest (three at most)
        const size_t wl = strlen(aWords[i]);
        if (wl == 0 || wl > 3) return (int)STT_ERR_INVALID_SHAPE;
        hs[i] = 0;
        for (size_t b = 0; b < wl; ++b) hs[i] |= (uint64_t)(unsigned char)aWords[i][b] << (8 * b);
      }
    }
    DevBuf dh, dp, dl;
    dh.upload(hs.data(), hs.size() * 8); dp.reserve((size_t)aNumWords * 4); dl.reserve((size_t)aNumWords * 4);
    launch_test_lm(sc.dev, dh.as<uint64_t>(), (int)aNumWords, aBos, aMode == 4 ? 2 : (aMode == 2 ? 1 : 0), dp.as<float>(), dl.as<int>(), nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(aProbs, dp.p, (size_t)aNumWords * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(aLens, dl.p, (size_t)aNumWords * 4, hipMemcpyDeviceToHost));
    return (int)STT_ERR_OK;
  }, STT_ERR_FAIL_RUN_SESS);
}
#endif  // STT_TEST_HOOKS

int STTX_PackLstmRecurrent(const float* aKernel, int aHidden, unsigned short* aOut) {
  if (aHidden % 128) return STT_ERR_INVALID_SHAPE;
  pack_lstm_recurrent_host(aKernel, aHidden, reinterpret_cast<_Float16*>(aOut));
  return STT_ERR_OK;
}

}  // extern "C"
