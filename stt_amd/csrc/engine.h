// stt_amd/csrc/engine.h -- host side of the MI355X engine (internal header).
//
// Mirrors the reference's layering so each class can be read next to the file it replaces:
//   Alphabet        native_client/alphabet.{h,cc}
//   ScorerDev       native_client/ctcdecode/scorer.{h,cpp} (loader + constants; queries run on the GPU)
//   ModelState      native_client/modelstate.{h,cc} + tflitemodelstate.{h,cc}
//   StreamingState  native_client/stt.cc:60-334
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "ctc.h"
#include "tuning.h"
#include "kernels.h"

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

// One lock for what may not overlap in a process: a stream capture on one thread and a device-wide synchronisation / free on another
// (HIP answers the latter with "operation not permitted when stream is capturing" and invalidates the capture -- found in round 6 with two
// streaming cohorts on two threads: one captured a hop while the other grew a buffer).  Held around every capture and inside the buffers' reserve().
std::recursive_mutex& hip_capture_mutex();

// Growable device allocation (never shrinks); the engine keeps its workspaces resident in HBM.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) { std::lock_guard<std::recursive_mutex> no_capture(hip_capture_mutex()); (void)hipFree(p); } }
  void reserve(size_t bytes, bool keep = false, hipStream_t st = nullptr);
  void upload(const void* src, size_t bytes, hipStream_t st = nullptr);
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Page-locked host memory (grow-only): source / destination of copies that must not block the host.
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (p) { std::lock_guard<std::recursive_mutex> no_capture(hip_capture_mutex()); (void)hipHostFree(p); } }
  void reserve(size_t bytes);
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
  void* dev() const;  // the same bytes as the GPU addresses them (kernels read / write page-locked host memory over the link)
};
// Small tables and result blocks of the batch path move with a copy KERNEL through mapped page-locked memory, not with
// hipMemcpyAsync: a copy-engine transfer costs a queue hand-over each way (signals between the compute queue and the DMA
// engine), measured at 0.2 ms on some hosts and 1-2 ms on others -- per batch, on the critical stream.  A kernel that moves
// the same 4-130 KB over the link is a few microseconds everywhere.  (STT_AMD_COPY_KERNEL=0: the copy engine, for A/B.)
// A stream of the engine: role 0 = GEMM engine, 1 = recurrence, 2 = output engine, 3 = beam search.  Plain non-blocking streams unless the
// tunables search_cus / am_cus partition the chip (hostutil.cpp).
void create_engine_stream(hipStream_t* st, int role, bool high_priority = false);
unsigned long long layout_generation();   // bumped when any DevBuf / PinnedBuf moves or a tunable changes (hostutil.cpp): what a captured graph may have baked in
void copy_h2d(void* dst_dev, const PinnedBuf& src, size_t bytes, hipStream_t st);
void copy_d2h(PinnedBuf& dst, const void* src_dev, size_t bytes, hipStream_t st, size_t dst_offset = 0);

// ---------------------------------------------------------------------------------------------
class Alphabet {
 public:
  int InitFromFile(const char* path);                     // alphabet.cc:42-68
  int Deserialize(const char* buffer, int buffer_size);   // alphabet.cc:133-169
  std::string Serialize() const;                          // alphabet.cc:102-131
  void InitUTF8();                                        // UTF8Alphabet, alphabet.h:80-91
  size_t GetSize() const { return labels_.size(); }
  int GetSpaceLabel() const { return space_index_; }
  const std::string& DecodeSingle(unsigned idx) const { return labels_[idx]; }
  std::string Decode(const unsigned* idx, int n) const;
  const std::vector<std::string>& labels() const { return labels_; }

 private:
  std::vector<std::string> labels_;
  int space_index_ = -2;
};

// ---------------------------------------------------------------------------------------------
// .scorer package resident in HBM: the KenLM trie blob is uploaded verbatim (the device walks the same
// bit-packed arrays KenLM mmaps), the ConstFst is repacked to {state_pos, final, arcs}.
class ScorerDev {
 public:
  // returns an STT_ERR_* code (scorer.cpp:108-222)
  int LoadFile(const std::string& path, const Alphabet& alphabet);
  int LoadBuffer(const char* data, size_t len, const Alphabet& alphabet);
  int LoadLmOnly(const char* data, size_t len);  // a bare KenLM trie binary (no dictionary): LM test hook
  void reset_params(float a, float b) { dev.alpha = a; dev.beta = b; }
  DevScorer dev{};  // alpha/beta live here (read at every launch, like the reference reads Scorer::alpha)
  bool is_utf8 = false;
  int order = 0;
  uint64_t blob_bytes = 0, lmi_bytes = 0;

 private:
  int Parse(const uint8_t* buf, size_t len, int space_label, bool lm_only);
  DevBuf blob_, fst_pos_, fst_arcs_, fst_space_, fst_rec_, vtab_, hint_, lmi_, memo_, cp_ub_, cpt_, cpb_tab_, cpb_rec_;
};

// ---------------------------------------------------------------------------------------------
struct Geometry {
  int n_input = 26, n_context = 9, n_hidden = 2048, n_classes = 29, n_steps = 16;
  int sample_rate = 16000, win_len = 512, win_step = 320, beam_width = 500;
  float relu_clip = 20.0f;
  int n_in1() const { return n_input * (2 * n_context + 1); }
  int k1_pad() const { return (n_in1() + 63) / 64 * 64; }
  int c_pad() const { return (n_classes + 127) / 128 * 128; }
  int k1_pad8() const { return (n_in1() + 127) / 128 * 128; }   // int8 path: K-tiles of 128 int8
  int c_pad8() const { return (n_classes + 255) / 256 * 256; }  // int8 path: the 128 x 256 tile
  int fft_len() const { int n = 1; while (n < win_len) n <<= 1; return n; }  // TF AudioSpectrogram: NextPowerOfTwo(window_size)
};

// What a model file yields, whichever container it came in (model.cpp: "STTAMDW1"; tflite_reader.cpp: ".tflite"):
// geometry, the serialised alphabet and the twelve f32 tensors, matrices as [inputs][outputs] (checkpoint orientation).
struct ModelTensors {
  Geometry g;
  std::string alphabet;
  std::vector<float> l1w, l1b, l2w, l2b, l3w, l3b, lk, lb, l5w, l5b, l6w, l6b;
  // A dynamic-range quantised `.tflite` (export.py:145-146) also keeps what the file stores for the six matrices (layers 1-3, the cell's
  // kernel, layers 5, 6): int8 [outputs][inputs] as FULLY_CONNECTED holds them, zero point 0, one scale per tensor or per output row.
  // All six present = the reference's CPU path runs them through TFLite's hybrid kernel, and so does the engine (ModelState::i8).
  std::vector<int8_t> wq[6];
  std::vector<float> wq_scale[6];
  bool asymmetric_inputs = false;   // the file asked for FullyConnectedOptions.asymmetric_quantize_inputs: wq dropped, f16 path (tflite_reader.cpp)
  bool all_int8() const { for (int l = 0; l < 6; ++l) if (wq[l].empty()) return false; return true; }
};
bool looks_like_tflite(const char* buf, size_t len);
int read_tflite_model(const char* buf, size_t len, ModelTensors& out, std::string& err);  // STT_ERR_* code
// Host-only parse of either container: pointers into `buf` (STTAMDW1) or into `storage` (.tflite).
struct ModelView {
  Geometry g;
  const char* alphabet = nullptr;
  size_t alphabet_bytes = 0;
  const float* t[12] = {};  // l1w l1b l2w l2b l3w l3b lk lb l5w l5b l6w l6b
  size_t count[12] = {};
  const ModelTensors* quant = nullptr;   // the int8 matrices of a quantised `.tflite` (storage->all_int8()), else null
};
int parse_model_file(const char* buf, size_t len, ModelTensors& storage, ModelView& view, std::string& err);

struct StreamingState;

// Decoder states of a batch of streams (device arrays + host mirror of the pointer table).
struct DecoderBatch {
  int n_streams = 0, beam = 0, C = 0;
  std::vector<DecStream> host;  // pointer table as uploaded
  DevBuf table;                 // DecStream[n_streams]
  DevBuf slab;                  // all per-stream arrays
  std::vector<uint32_t> pa_cap, ta_cap;
  size_t per_stream_fixed = 0;
  // the arena capacities the slab was last laid out / grown for: a batch that is created again on the same slab (a parked stream taken
  // from ModelState::stream_pool_) keeps what it grew to instead of starting at the default again and growing in the same hop as
  // every other stream that started with it
  uint32_t keep_pa = 0, keep_ta = 0, keep_be = 0;
  size_t keep_fixed = 0;
  int keep_n = 0;
  bool decode_cache = false;    // the slab carries the arrays of the incremental back-tracking (DecStream::dpd / dtd / chain)
};

// Hot-word table of a scorer view in HBM, uploaded when the words change, not at every launch (STT_AddHotWord & co.: stt.cc:451-497;
// a stream copies the model's words at creation, stt.cc:547).  `retire`: where the old buffers go when the table changes while
// kernels that read them may still be running (batches in flight); null = nothing can be.
struct HotTables {
  std::unique_ptr<DevBuf> hash, boost;
  std::map<std::string, float> loaded;
  bool valid = false;
};
struct StreamingState;
struct Output {
  double confidence;
  std::vector<unsigned> tokens, timesteps;
};
// Stage timing of the batch path (STTX_SetProfiling / STTX_GetStageTimes): HIP-event marks per stream (api.cpp)
struct Prof {
  bool on = false;
  bool phase_cycles = false;  // level 2: also the search kernel's per-phase cycle counters
  int only = -1;              // level 3: marks of ONE mark list only (5 = the recurrence's stream: what bench.py's roofline needs live); -1 = all of them
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<std::pair<int, hipEvent_t>> marks[7];  // [0] = acoustic stream, [1 + s] = the search stream of group slot s, [5], [6] = recurrence / output-layer streams
  float ms[8] = {};
  unsigned long long dec_stats[4] = {};
  unsigned long long dec_phase[8] = {};
  unsigned long long dec_stamps[64] = {};
  ~Prof() { for (auto e : pool) (void)hipEventDestroy(e); }
};
struct ModelState {
  Geometry g;
  Alphabet alphabet_;
  std::shared_ptr<ScorerDev> scorer_;
  std::map<std::string, float> hot_words_;  // ordered => deterministic upload order
  unsigned beam_width_ = 500;
  hipStream_t stream = nullptr;      // acoustic model, streaming path, decode
  hipStream_t stream_dec = nullptr;  // batch path: the beam search of chunk k runs here while `stream` computes chunk k+1
  // Standalone decoders (STTX_Decoder): by default on the model's own stream; with the tunable decoder_streams = N (2 .. 4) on a POOL of N streams
  // owned by the model, dealt round-robin at creation, so that decoders of one model can be driven side by side from several host threads (two
  // on one stream simply take turns; one decoder is a quarter of the chip, so four is the useful number).  Not a stream per decoder: a process
  // should stay at or below the sixteen hardware queues the engine asks for (streams beyond that share queues).  (The GPU memory fault that the
  // extended fuzz of round 6 ran into with sixteen decoder streams was a race in the code-point search step, fixed: DESIGN.md 10.10.)
  static constexpr int kDecoderStreams = 4, kDecoderStreamsDebug = 16;   // (debug_scribble bit 1: the faulting configuration, for experiments)
  hipStream_t decoder_streams_[kDecoderStreamsDebug] = {};
  unsigned decoder_stream_next_ = 0;
  std::mutex decoder_stream_mu_;
  hipStream_t decoder_stream();
  int device = 0;

  // weights in HBM (f16, transposed / packed; see kernels_am.hip header)
  DevBuf w1t, w2t, w3t, wxt, whp, w5t, w6t;
  DevBuf b1, b2, b3, bl, b5, b6;
  // The released models' own arithmetic (dynamic-range quantised `.tflite`, or tunable am_i8 = 1): TFLite's hybrid FULLY_CONNECTED end to
  // end (engine.cpp: acoustic_rows_i8; kernels.h: launch_dense_hybrid_i8, LstmI8Args).  int8 matrices [N][K] (K zero padded to a multiple of
  // 128, the output layer's N to 256), the cell's kernel split into its x half (a GEMM operand), its h half packed for the recurrent step
  // and the same h half in rows (the step's slow path); scales [1] or [N].
  bool i8 = false;
  hipEvent_t dbg_ev_[2] = {nullptr, nullptr};   // STTX_TestHybridChain: HIP events around the recurrence of the next acoustic_rows_i8 call (null: none)
  DevBuf w1q, w2q, w3q, wxq, whq, whpq, w5q, w6q;
  DevBuf s1, s2, s3, sk, s5, s6, b6q;     // b6q: the output layer's bias padded to c_pad8()
  int sn[6] = {1, 1, 1, 1, 1, 1};         // scales per matrix (1 or N)
  // workspaces of the int8 path (one-stream paths; the three-engine batch path has its own below)
  DevBuf q_x, q_s, q_rng, ws_hq0, ws_hq1, ws_hprev0, ws_pmax, ws_flag, ws_zslow, ws_hlast, ws_slow;
  // feature tables
  DevBuf t_window, t_twiddle, t_melw, t_mel_idx, t_dct;
  // alphabet on device
  DevBuf al_bytes, al_off;
  DevAlphabet dev_alphabet{};
  // workspaces (grown on demand, reused)
  DevBuf ws_audio, ws_nsamp, ws_nframes, ws_feats, ws_x1, ws_a, ws_b, ws_xproj, ws_hall, ws_logits, ws_probs;
  DevBuf ws_c, ws_hp0, ws_hp1, ws_hf32;
  DevBuf ws_wide;  // wide-alphabet row records of the streaming paths
  DevBuf ws_out;  // one DecodeBlock
  HotTables hot_tables_;                                   // of the model's own hot words (batch calls)
  std::vector<std::unique_ptr<DevBuf>> retired_bufs_;      // hot-word tables replaced while batches were in flight; freed when the pipeline has drained
  PinnedBuf h_out;
  // Finished streams are parked here with their HBM buffers (frames, LSTM state, decoder slab: seven allocations) and
  // handed out again by STT_CreateStream: a server that opens a stream per utterance pays hipMalloc/hipFree once per
  // concurrent stream, not once per utterance.
  Prof prof_;
  std::vector<StreamingState*> stream_pool_;
  std::mutex stream_pool_mu_;
  // page-locked staging of the streaming path's audio: a feed returns without waiting for its copy; a slot is reused
  // four feeds later, after its event
  PinnedBuf h_audio[4];
  hipEvent_t ev_audio[4] = {};
  unsigned audio_slot = 0;
  // The batch path keeps several 64-utterance groups in flight (kSlots slots, `pipeline_depth()` of them used): the
  // acoustic models run one after the other on `stream` (two recurrences side by side delay each other's steps), every
  // group's beam search on its slot's own stream.  While the search of group g runs on 64 compute units, the acoustic model
  // of group g+1 has the others; when the recurrence of g+2 leaves a quarter of the chip idle, the search of g+1 is already
  // there to take it; the host unpacks the oldest group meanwhile.  Everything a group owns beyond the acoustic stream's
  // scratch buffers lives in its slot.
  struct GroupSlot {
    DecoderBatch dec;
    DevBuf probs, ints, out;  // out: one DecodeBlock (ctc.h)
    DevBuf stamps;            // profiling level 2: DecParams::stamps
    DevBuf wide;  // per-row class records of the wide-alphabet search path (ctc.h: ctc_is_wide)
    PinnedBuf h_ints, h_table, h_out;
    PinnedBuf h_prof;         // profiling: the group's DecStream table (+ stamps), copied behind the results on the search stream
    DecodeBlock out_layout{};
    hipEvent_t done = nullptr;
    hipStream_t stream_dec = nullptr;  // the group's search stream (slot 0: ModelState::stream_dec, the others their own)
    int Bg = 0, nr = 0, max_len = 0, t_max = 0;
    std::vector<unsigned> idx;  // caller's utterance index of every stream of the group
    // STTX_BatchSubmitDevice: a slot holds one or TWO submitted batches (two 64-utterance batches advanced by one recurrence,
    // 128 rows per recurrent step): streams [part_begin[p], part_begin[p + 1]) belong to ticket part_ticket[p]
    int n_parts = 0, part_begin[3] = {0, 0, 0}, part_ticket[2] = {-1, -1};
    bool part_open[2] = {false, false};          // submitted, not collected yet
    bool prof_enqueued = false;                  // the profiling block rides behind the results of THIS enqueue
    size_t prof_stamp_bytes = 0;
    // what the group was made of (so that it can be decoded again with full-size arenas if one overflowed) and how it was sized
    struct SavedPart { const int16_t* d_audio; unsigned stride; std::vector<unsigned> sizes, idx; };
    std::vector<SavedPart> saved_parts;
    DevScorer saved_ds{};                       // the scorer description the group was enqueued with (its tables stay alive while the
    std::shared_ptr<ScorerDev> saved_scorer;    // group is in flight: retired_bufs_ / this reference): a retry decodes under the SAME scorer
    unsigned saved_num_results = 1;
    bool saved_pipelined = false, optimistic = false;
    bool results_ready = false;
    std::vector<std::vector<Output>> results;  // unpacked once, handed out per part
    bool busy() const { return part_open[0] || part_open[1]; }
  };
  static constexpr int kSlots = 4;
  GroupSlot slots_[kSlots];
  // STTX_BatchSubmitDevice / STTX_BatchCollect: the next ticket, the next slot, and the first half of a pair that waits for its
  // partner (the caller's next submit) -- or for its own collect, which sends it through alone
  struct PendingHalf { bool valid = false; const int16_t* d_audio = nullptr; unsigned stride = 0; std::vector<unsigned> sizes; int ticket = -1; hipEvent_t ready = nullptr; };
  // STTX_BatchSubmit (host audio, the ABI's own contract: const short* buffers, coqui-stt.h:294-297): the rows are gathered into a page-locked
  // staging buffer and copied on a queue of their own; the group's feature kernel waits for the copy's event, nothing else does.  A ring of
  // kStage entries: more than the tickets a caller may hold (kSlots x 2) + the noted first half of a pair, so the entry a submit takes has
  // always been collected.
  static constexpr int kStage = 10;
  struct HostStage { PinnedBuf pin; DevBuf dev; hipEvent_t copied = nullptr; };
  HostStage stage_[kStage];
  hipStream_t stream_h2d = nullptr;
  unsigned long long stage_seq_ = 0;
  PendingHalf pending_;
  int async_next_ = 0;    // next ticket
  int async_groups_ = 0;  // groups enqueued by submits so far (slot = async_groups_ % depth)
  int async_depth_ = 0;   // slots in use while batches are in flight (api.cpp: pipeline_depth)
  bool async_pair_ = false;  // ... and whether they were submitted as pairs
  bool async_any() const { if (pending_.valid) return true; for (const GroupSlot& s : slots_) if (s.busy()) return true; return false; }
  // The acoustic + search pass of a streaming hop (engine.cpp: streams_process) as ONE hipGraph per live-set shape: ~35 launches of a
  // few microseconds each are launch-bound when enqueued one by one.  Keyed by the number of rows, the search configuration and
  // layout_generation(); captured the second time a key comes up, dropped wholesale when the generation moves.
  struct HopGraph { hipGraphExec_t exec = nullptr; };
  std::map<uint64_t, HopGraph> hop_graphs_;
  std::set<uint64_t> hop_seen_;
  unsigned long long hop_generation_ = 0;
  // scratch of the batched streaming calls (STTX_FeedAudioContentBatch & co.)
  DevBuf sb_audio, sb_tab, sb_tab2, sb_tab3, sb_c, sb_h, sb_table;
  PinnedBuf sb_haudio, sb_htab, sb_htab2, sb_htab3;   // (…2: the feature pass's table -- the acoustic pass behind it fills sb_htab while that one is in flight; …3: the arena check's)
  hipEvent_t ev_chunk[2] = {};  // chunk hand-over acoustic stream -> decoder stream (alternating)
  // The acoustic model of the batch path as three engines (STT_AMD_AM_PIPE=0: one stream, as the streaming path runs it):
  // `stream` = features, context windows, layers 1-3 and the x-projection of chunk k+1, k+2; `stream_l` = the recurrence of
  // chunk k -- 250 dependent launches per batch, each leaving half of the compute units idle and the MFMA pipes of the
  // other half mostly so --; `stream_o` = layers 5-6 + softmax of chunk k-1.  The recurrence never waits for a GEMM of its
  // own batch (or of the next one: the x-projections run ahead across batch boundaries), and the GEMMs run at one workgroup
  // per CU (DenseArgs::lds_floor) so that a recurrent-step workgroup always finds registers and LDS beside them.
  // Hand-over through rings of kAmRing chunk buffers, one event pair per slot.
  static constexpr int kAmRing = 3;
  hipStream_t stream_l = nullptr, stream_o = nullptr;
  DevBuf am_xproj[kAmRing], am_hall[kAmRing], ws_o;
  DevBuf am_y3[kAmRing], am_xs[kAmRing], am_qx, am_qs, am_qo, am_qos, am_hq0, am_hq1, am_hprev0, am_pmax, am_flag, am_zslow, am_hlast;   // int8 path
  DevBuf am_c, am_hp0, am_hp1, am_logits;  // the recurrence's own state / the output engine's own logits: stream_l and stream_o never touch
                                           // what the one-stream paths (streaming API, blocking calls) use on `stream`
  hipEvent_t ev_x_ready[kAmRing] = {}, ev_x_free[kAmRing] = {}, ev_h_ready[kAmRing] = {}, ev_h_free[kAmRing] = {};
  unsigned long long am_seq = 0;  // chunks sent through the pipe so far (slot = am_seq % kAmRing)
  // Placement watch (engine.cpp: am_watch_begin / am_watch_end / am_replace_if_slow): one chunk's recurrence at a time is bracketed by two
  // timing events; a pipeline whose steps are picked up late -- a bad placement of the engines' hardware queues -- moves the recurrence and the
  // output engine to fresh streams the next time a group is enqueued.
  hipEvent_t ev_watch[2] = {};
  bool watch_armed = false;
  bool watch_search_bound = false;   // set by the batch path: a setup whose searches fill the chip (code-point scorer, beam > 512) is slow for THAT reason: not watched
  int watch_steps = 0, watch_slow = 0, watch_moves = 0;
  // Placement (engine.cpp): creates stream_l / stream_o on dispatch pipes they share with neither the GEMM engine's stream nor (if it can be
  // helped) the searches' -- by measurement, not by counting streams.  `avoid`: the search streams of the group slots in use.
  void place_engine_streams(hipStream_t* out_l, hipStream_t* out_o);
  void place_batch_streams(hipStream_t* slot_streams, int n_slots, bool spread_searches);
  int placed_search_bound_ = -1;    // the kind of setup the streams were placed for (-1: not placed)   // recurrence, output engine AND the group slots' search streams, each role a pipe class of its own
  std::vector<hipStream_t> placement_avoid_;   // set by the batch path (api.cpp: batch_init_slots): the slots' search streams, most used first
  DevBuf placement_scratch_;
  void am_watch_begin(int T);
  void am_watch_end();
  void am_replace_if_slow();
  // The recurrence of a chunk is T dependent launches whose arguments only depend on (ring slot, T, parity of t0, batch): the
  // second time a combination comes up it is captured into a hipGraph and replayed from then on -- one graph launch instead
  // of 16-48 kernel launches of host time (enqueueing a 64 x 5 s batch: ~300 launches, 1.5 ms on a quiet host, 2.5-5 ms on a
  // busy one, which then starves the GPU).  Keyed by every pointer baked into the nodes.
  struct LstmGraphKey {
    const void *xproj, *hall, *c, *hp0, *hp1, *whp, *nframes;   // (nframes: the frame table the int8 step masks rows with)
    int T, par, B, NT, passes, prio, H, first;                  // (int8 path: `first` carries the row-group size and the probe number, both baked into the launches)
    bool operator<(const LstmGraphKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
  };
  struct LstmGraph { hipGraphExec_t exec = nullptr; };
  std::map<LstmGraphKey, LstmGraph> lstm_graphs_;
  std::set<LstmGraphKey> lstm_seen_;  // combinations seen once, not captured yet
  bool am_pipe_init();            // creates the streams / events on first use; false when switched off
  // chunk [t0, t0+T) of a batch through the three engines; `done` is recorded on stream_o behind the softmax
  void run_acoustic_chunk_piped(const float* d_feats, const int* d_nframes, int B, int t_max, int t0, int T, float* d_probs, hipEvent_t done);
  void run_acoustic_chunk_piped_i8(const float* d_feats, const int* d_nframes, int B, int t_max, int t0, int T, float* d_probs, hipEvent_t done);
  void run_lstm_graph(const LstmGraphKey& key, const std::function<void()>& steps, hipStream_t st = nullptr);   // st: the stream `steps` enqueues on (null: stream_l)

  ModelState() { tuning_model_count(+1); }   // (tuning.h: load-time knobs are frozen while a model is alive)
  ~ModelState();
  int InitFromBuffer(const char* buf, size_t len);  // STT_ERR_* code
  MfccArgs mfcc_args() const;

  // ---- stages (all asynchronous on `stream`) ----
  // audio already in ws_audio ([B][n_max] int16) or at d_audio; fills ws_feats [B][t_max][n_input]
  void run_mfcc(const int16_t* d_audio, const int* h_nsamples, int B, int n_max, int t_max, std::vector<int>& n_frames);
  // feats (device, [B][t_max][n_input]) -> probs (ws_probs [B][t_max][C]); c/h are [B][H] f32 device (in/out, may be null = zero)
  void run_acoustic(const float* d_feats, const int* d_nframes, int B, int t_max, float* d_c, float* d_h, bool carry_in);
  // windows (device f16 [rows][k1_pad], row = t*B+b) -> probs; used by the chunked streaming path and STTX_InferChunk
  // (x1: f16 [rows][k1_pad], or f32 [rows][k1_pad8] for an int8-path model: x1_cols() / x1_bytes())
  void run_acoustic_rows(const void* d_x1, int B, int T, float* d_c, float* d_h, bool carry_in, float* d_probs_out, int probs_t_max);
  int x1_cols() const { return i8 ? g.k1_pad8() : g.k1_pad(); }
  size_t x1_bytes(int rows) const { return (size_t)rows * x1_cols() * (i8 ? 4 : 2); }
  // one time-chunk [t0, t0+T) of a batch: context rows from feats, then the layers; the LSTM state continues from the
  // previous chunk (internal buffers) unless t0 == 0
  void run_acoustic_chunk(const float* d_feats, const int* d_nframes, int B, int t_max, int t0, int T, float* d_probs);

  // ---- decoder ----
  DevScorer current_scorer(std::shared_ptr<ScorerDev> sc, const std::map<std::string, float>& hot, HotTables& ht, bool in_flight = false);
  // `staging`: page-locked room for the stream table; the upload then does not wait for the stream (batch path)
  // `optimistic`: arenas below the never-overflows bound (engine.cpp); the caller must be prepared to decode again on an overflow flag
  // decode_cache: lay out (and clear) the arrays of the incremental back-tracking -- streams that are decoded hop after hop
  void decoder_create(DecoderBatch& db, int n_streams, int beam, int expected_frames, std::shared_ptr<ScorerDev> sc, PinnedBuf* staging = nullptr,
                      bool optimistic = false, bool decode_cache = false);
  void decoder_reserve(DecoderBatch& db, const std::vector<int>& more_frames, hipStream_t st = nullptr);   // (st: the stream the table is read back on; null = the model's)
};


// One resumable utterance: the three buffers of stt.cc:60-71 with the frames and LSTM state kept in HBM.
struct StreamingState {
  ModelState* model_ = nullptr;
  std::shared_ptr<ScorerDev> scorer_;         // captured at creation (stt.cc:542-547)
  std::map<std::string, float> hot_words_;    // copied at creation
  unsigned beam_width_ = 0;
  bool keep_emissions_ = false;
  std::vector<int16_t> audio_buffer_;         // <= win_len samples (kept as int16; scaled on the GPU like stt.cc:113)
  int frames_ = 0;                            // MFCC frames pushed so far (incl. n_context leading zero frames)
  int windows_done_ = 0;                      // context windows already run through the model
  DevBuf d_frames;                            // f32 [frames_cap][n_input]
  int frames_cap = 0;
  DevBuf d_c, d_h;                            // LSTM state [H] f32
  bool state_nonzero = false;
  DecoderBatch dec;                           // one stream
  bool flushed_ = false;                       // the final flush (partial window + trailing context frames) has gone through the model
  uint32_t arena_bound_[3] = {2, 2, 2};       // host-side upper bounds of the fill of the path / time / boundary-entry arenas (each step appends <= beam to each)
  HotTables hot_tables_;
  std::vector<double> probs_;                 // emissions of the last processed batch (keep_emissions_)

  void recycle();  // back to the state of a fresh object, keeping the device buffers
  void feedAudioContent(const short* buffer, unsigned int buffer_size);
  void flushBuffers(bool addZeroMfccVectors);
  void pushFrames(const int16_t* d_audio_span_host, int n_samples_span, int n_new_frames);
  void pushZeroFrames(int n);
  void processReady(bool flush_partial, bool final_flush);
  void reserveArena(int take);
  std::vector<Output> decode(unsigned num_results);
};

std::vector<std::vector<Output>> decode_streams(const ModelState& m, const DecoderBatch& db, std::shared_ptr<ScorerDev> sc,
                                                const std::map<std::string, float>& hot, HotTables& ht, unsigned num_results, int max_len);
// (st / ws / ho: the caller's own stream, device and page-locked result blocks -- a STTX_Decoder's, so that several decoders of one model can run
// side by side from several host threads; null = the model's)
std::vector<std::vector<Output>> decode_table(ModelState& m, const DecStream* d_table, int n, int beam, int C, std::shared_ptr<ScorerDev> sc,
                                              const std::map<std::string, float>& hot, HotTables& ht, unsigned num_results, int max_len,
                                              hipStream_t st = nullptr, DevBuf* ws = nullptr, PinnedBuf* ho = nullptr);
// Many streams of one model at once (same beam width, scorer and hot words; otherwise the callers fall back to one by one):
// what STT_FeedAudioContent / STT_IntermediateDecode / flushBuffers do, with the ready windows of all streams pushed through
// the acoustic model and the beam search as one batch.
bool streams_batchable(const std::vector<StreamingState*>& ss);
void streams_feed_batch(const std::vector<StreamingState*>& ss, const short* const* buffers, const unsigned int* sizes, const unsigned char* last = nullptr);
void streams_flush_batch(const std::vector<StreamingState*>& ss, bool addZeroMfccVectors);
std::vector<std::vector<Output>> streams_decode_batch(const std::vector<StreamingState*>& ss, unsigned num_results);
int n_frames_for(const Geometry& g, int n_samples);
void check_decoder_errors(const int* errors, int n);  // throws when a stream's DecStream::error is set
extern int g_debug_arena_frames;
void stt_prof_mark(ModelState* m, int i);  // HIP-event marks for STTX_GetStageTimes (api.cpp)
void stt_prof_mark_on(ModelState* m, int id, int which, hipStream_t st);
void pack_lstm_recurrent_host(const float* kernel /*[2H][4H]*/, int H, _Float16* out);
void pack_lstm_recurrent_i8_host(const int8_t* kernel_q /*[4H][2H]*/, int H, int8_t* out);
// what the converter's dynamic-range quantisation stores for a [in][out] float matrix (oracle/am_hybrid.py: quantize_weights): int8 [out][in], one scale
void quantize_weights_host(const float* w_in_out, int n_in, int n_out, std::vector<int8_t>& q, std::vector<float>& scale);
