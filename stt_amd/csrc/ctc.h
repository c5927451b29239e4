// stt_amd/csrc/ctc.h -- device-side data model of the CTC prefix beam search (internal header).
//
// The reference keeps a heap trie of PathTrie nodes (path_trie.h:44-113) and walks it with one CPU
// thread per utterance.  Here a stream's search state is a flat struct-of-arrays beam (<= beam_size
// live prefixes, always kept in prefix_compare order) plus two append-only arenas in HBM:
//   path arena  {parent, character}        -> token back-tracking and n-gram reconstruction
//   time arena  {parent, timestep}         -> the TimestepTreeNode tree (path_trie.h:17-37)
// A prefix's identity is a 64-bit path hash (key); "does this child already exist in the beam"
// (get_path_trie's child scan, path_trie.cpp:37-50) becomes an LDS hash probe.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STT_KENLM_MAX_ORDER 6
#define STT_ROOT_CH 0xFFFFFFFFu
#define STT_MAX_BEAM 1024
#define STT_MAX_CLASSES 8192

// KenLM's lm::ngram::State (kenlm/lm/state.hh:15-48) at KENLM_MAX_ORDER = 6
struct KState { uint32_t words[STT_KENLM_MAX_ORDER - 1]; float backoff[STT_KENLM_MAX_ORDER - 1]; int length; };

// Word-mode scorer cache: one entry per scored word boundary ("prefix X, then a space").  The reference rebuilds the
// n-gram strings and re-walks the KenLM trie from a null/BOS context for every such event (scorer.cpp:308-396); the
// score only depends on the last `order` words, so the KenLM state after the word is kept here and the next word's
// query is a single FullScore from it.  Entry 0 is the root (BeginSentence state).
struct __attribute__((aligned(16))) BEntry {
  double raw;         // log_cond_prob + hot_boost of the n-gram ending with this word (before alpha/beta)
  uint32_t prev;      // entry of the previous word boundary (STT_NONE for the root entry)
  uint16_t oov_hist;  // bit k: the word k back (bit 0 = this word) is out of vocabulary
  uint16_t pad;
  float hot_self;     // hot-word boost of this word (0 if none)
  KState st;          // state after this word
};
#define STT_NONE 0xFFFFFFFFu

// One slot of the vocabulary table; carries a copy of the word's unigram record (lm/trie.hh UnigramValue: prob, backoff,
// next) so that a query's first trie level costs no further read.  begin/end are valid when DevScorer::uni_in_vtab.
struct DevVocabSlot { uint64_t hash; uint32_t index; uint32_t used; float prob, backoff; uint32_t begin, end; };

struct DevBitPacked {
  const uint8_t* base;
  uint64_t word_mask, next_mask;
  const uint64_t* off_begin;  // ArrayBhiksha offsets (null for DontBhiksha)
  uint32_t off_count;
  const uint32_t* off_hint;   // off_hint[i >> hint_shift] = upper_bound(offsets, (i >> hint_shift) << hint_shift) - 1 (built at load)
  uint32_t hint_shift;
  uint64_t max_word;          // upper bound of the word field (KenLM's max_vocab), for the interpolation search
  uint8_t word_bits, total_bits, quant_bits, next_bits;
};

// Everything a kernel needs to score with the .scorer package uploaded to HBM (KenLM trie +
// dictionary FST + alphabet label bytes).  Passed by value.
struct DevScorer {
  int enabled;
  int order, quant, utf8;
  double alpha, beta;
  const uint64_t* vocab;       // KenLM SortedVocabulary: sorted MurmurHash64A of the words
  uint64_t vocab_n;
  const DevVocabSlot* vtab;    // open-addressing table over the same hashes (built at load): 1-2 probes instead of log2(n)
  uint32_t vtab_mask;
  int uni_in_vtab;             // the slots' unigram copies are usable (bigram count < 2^32)
  const uint8_t* unigram;
  const float* qprob[STT_KENLM_MAX_ORDER];
  const float* qbackoff[STT_KENLM_MAX_ORDER];
  DevBitPacked middle[STT_KENLM_MAX_ORDER - 2];
  DevBitPacked longest;
  uint32_t prob_mask, backoff_mask;
  uint8_t prob_bits, backoff_bits;
  uint32_t bos_index;
  float bos_backoff;
  // KenLM PROBING / REST_PROBING binaries (model types 0, 1: kenlm/lm/search_hashed.hh): `unigram` = Weights[count + 1] of p_wstride bytes
  // ({prob, backoff[, rest]}; the sign bit of the STORED prob is "independent left", Prob() sets it), one open-addressing table per middle
  // order of p_estride-byte entries {uint64 key, Weights} and one of 12-byte entries {key, prob} for the longest order; key = the chained
  // CombineWordHash of the n-gram's word indices, newest first; bucket = key % buckets, linear probing, key 0 = empty.
  int probing;
  int p_wstride, p_estride;
  const uint8_t* p_mid[STT_KENLM_MAX_ORDER - 2];
  uint64_t p_mid_buckets[STT_KENLM_MAX_ORDER - 2];
  const uint8_t* p_lon;
  uint64_t p_lon_buckets;
  // dictionary FST, repacked: state s -> arcs [state_pos[s], state_pos[s+1]); arc = {ilabel, child state} where
  // child state = Start() if the arc's target is final (path_trie.cpp:79-87), else the target
  int fst_start;
  int fst_tree;                  // the dictionary is a tree in breadth-first order: child along arc k = node k + 1, along a space arc = the root (0)
  const uint32_t* fst_state_pos;
  const uint2* fst_arcs;
  const uint8_t* fst_has_space;  // word mode: state has an out-arc for the space label (a word may end here)
  // per state {index of its first arc with a label >= 1, bitmap of the labels 1..32 on its arcs (bit c = label c + 1)};
  // null when some state's arcs are not strictly ascending by label (the rank of a label's bit is then not its arc)
  const uint2* fst_rec;
  // hashed n-gram index over the trie (lmindex.h); null = not built (vocabulary >= 2^27 words, inconsistent trie)
  const struct LmiEntry* lmi;
  uint32_t lmi_buckets;
  float unk_prob, unk_backoff;  // unigram record of <unk> (word index 0)
  int unk_indep;                // ... and whether it has no children
  // utf8 mode, orders <= 5: lossy direct-mapped cache of FullScore results shared by every stream using this scorer,
  // 32-byte entries { context words[4] | unit hash (u64) | prob (f32) | length : 3, oov : 1, check : 28 } (ctc.hip: lm_memo_*).
  // A hit is verified against the whole key (words, length, hash) -- exact, not a fingerprint -- and against a check word
  // over key and value, which rejects an entry torn between two concurrent writers.  null = none.
  uint32_t* memo;
  uint32_t memo_mask;
  // utf8 mode: cp_ub[u] >= get_log_cond_prob() of ANY n-gram that ends with code point u (U+0000 .. U+FFFF; -1000 = not in the
  // vocabulary), cp_ub_max = the largest entry; null = no table (scorer_dev.cpp: build_unit_bounds)
  const float* cp_ub;
  float cp_ub_max;
  int cp_ub_on;   // cp_ub_max is valid
  // utf8 mode, bigram blocks (scorer_host.h: HostScorer::cpt / cpb_tab / cpb_rec; tunable cp_blocks): FullScore of a code point u after
  // context word w1 through ONE table entry per (w1, u >> 6) -- shared by the 64 sibling code points a prefix's children complete -- and the
  // unigram record cpt[u]; a stored bigram continues in the hashed index from its slot.  null = not built.
  const uint32_t* cpt;       // [65536] x {word index, prob, backoff, flags (1 = in the vocabulary, 2 = no longer n-gram ends with it)}
  const uint32_t* cpb_tab;   // [cpb_mask + 1] x {w1 (0xFFFFFFFF = free), block, offset, count, present (u64), indep (u64)}
  const uint32_t* cpb_rec;   // x {prob, backoff, slot of the bigram in the hashed index}
  uint32_t cpb_mask;
  // hot words (murmur hashes of the words)
  int n_hot;
  const uint64_t* hot_hash;
  const float* hot_boost;
};

struct DevAlphabet {
  int n_labels;              // C - 1
  int space_id;
  const uint8_t* label_bytes;
  const int* label_off;      // [n_labels] end offsets
  int byte_labels;           // 1: label c is the single byte c + 1 for every c (UTF8Alphabet, alphabet.h:156-198)
};

// Per-stream search state; lives in HBM between launches, in LDS during a launch.
struct DecStream {
  int n;                // live prefixes
  int abs_t;            // abs_time_step_
  int start_expanding;  // ctc_beam_search_decoder.cpp:125-132
  int error;            // bit0: path arena full, bit1: time arena full, bit2: candidate workspace full, bit3: scorer cache state lost /
                        // boundary-entry arena full, bit4: an intra-workgroup counter wait of the bitmap step timed out.  Sticky; every decode call
                        // reports it (STT_* return NULL / STT_ERR_FAIL_RUN_SESS instead of a transcript from a damaged beam)
  uint32_t pa_n, ta_n, pa_cap, ta_cap;
  // beam arrays [beam_cap]
  float *score, *pb, *pnb;
  uint32_t *ch, *node, *ts;
  int* fst;
  uint64_t* key;
  uint2* pa;  // {parent, character}; entry 0 = root
  uint2* ta;  // {parent, timestep};  entry 0 = timestep_tree_root_
  uint32_t* bnd;  // beam array: BEntry of the last word boundary at or above the prefix (STT_NONE = unknown -> uncached query)
  uint32_t* pq;   // per path node: BEntry created by scoring "this prefix, then a word boundary" (STT_NONE = not yet)
  BEntry* be;     // boundary-entry arena
  uint32_t be_n, be_cap;
  // per-step candidate workspace [cand_cap]
  float* c_logp;
  uint32_t* c_pi;  // parent beam index | class position << 16 | needs_lm << 31
  int* c_fst;
  uint64_t* c_key;
  uint64_t* sel_keys;  // [beam_cap + cand_cap]
  uint32_t cand_cap;
  // Incremental back-tracking (streams only; null = none): a decode with one result walks the best prefix's path back only to where
  // it meets the path the PREVIOUS decode of this stream walked (ctc_decode_kernel).  dpd[node] / dtd[time node] = depth of a node that
  // was on a decoded chain (0 = unknown; the depth of a node never changes), chain = [token count | timestep count | tokens of the last
  // decoded best prefix [chain_cap] | their path nodes | its timesteps | their time nodes | scratch 4 x chain_cap]
  uint32_t *dpd, *dtd, *chain;
  uint32_t chain_cap, pad_;
  // statistics (DESIGN.md roofline accounting): steps, candidates, lm queries, lm memory probes
  unsigned long long stat[4];
  // shader cycles (s_memtime, thread 0; profiling level 2) per phase: [0] setup, [1] expand: events, [2] expand: items, [3] LM
  // queue + early merges, [4] merge + keys, [5] select, [6] rank + write, [7] the LM wave's own time (runs beside [1]+[2])
  unsigned long long phase[8];
};

struct DecParams {
  int C, blank, beam, cutoff_top_n;
  double cutoff_prob;
  int t_max;  // row stride of probs in frames
  int phase_cycles;  // accumulate shader cycles per phase into DecStream::phase (profiling level 2)
  int all_begin, all_count;  // frame range of every stream when launch_ctc_next gets no frame_begin / frame_count tables
  // wide alphabets (ctc_is_wide): per-row records prepared by ctc_wide_rows_kernel; filled in by launch_ctc_next
  const unsigned char* wide_rows;
  unsigned long long wide_stride;
  int wide_max_frames;
  int lds_kb;       // LDS budget of the search kernel's layout in KiB (filled in by launch_ctc_next; host and device carve the same layout)
  int item_cap;    // bitmap step, test hook: items the expand table holds per pass (0 = all it has room for; filled in by launch_ctc_next)
  int lm_prio;     // bitmap step: s_setprio of the language-model waves while they run their queries (filled in by launch_ctc_next)
  int wait_spins;  // bitmap step: polls a counter wait may take before it gives up with error bit 0x10 (filled in by launch_ctc_next)
  unsigned long long key_mask;  // test hook (tunable debug_key_bits): path keys truncated, so that collisions happen and the guard (error bit 0x20) can be seen to fire; ~0 otherwise (filled in by launch_ctc_next)
  int n_lm_waves;  // bitmap step: waves of the workgroup that only run language-model queries (0 = by beam width; filled in by launch_ctc_next)
  // profiling level 2: [n_streams][64] shader cycles, summed over the steps.  Slots: [w] wave w reaches the end of the expand phase
  // (since the step began; wave 0: since its last phase tick), [16 + w] its wait there, [32 + w] (bitmap step) arrival at the end of
  // the score phase; bitmap step, wave 9: [50] item table complete, [51] taking a chunk, [52] chunks, [53] chunk bodies; write phase:
  // [54] live / [55] new entries, wave 0 [56..59] and the last wave [60..63]: ranking | barrier | own list | tail
  unsigned long long* stamps;
};

struct DecodeOut {
  uint32_t* tokens;     // [n_streams][num_results][max_len]
  uint32_t* timesteps;  // same shape
  int* lens;            // [n_streams][num_results]
  double* confidence;   // [n_streams][num_results]
  int* n_results;       // [n_streams]
  int* errors;          // [n_streams] DecStream::error of the stream (0 = its search state is intact)
  int num_results, max_len;
};

// `max_frames` = upper bound of frame_count[]; `wide_ws` = ctc_rows_ws_bytes(...) bytes of device memory: required when
// ctc_is_wide(p.beam, p.C); optional otherwise (with it, the per-row class sort of C > cutoff_top_n alphabets -- byte mode --
// runs row-parallel ahead of the search instead of inside every sequential step).
void launch_debug_scribble(hipStream_t st, int mode);   // test hook (tunable debug_scribble; ctc.hip)
void launch_ctc_next(const DecParams& p, const DevScorer& s, const DevAlphabet& al, DecStream* streams, int n_streams,
                     const float* probs, const int* frame_begin, const int* frame_count, hipStream_t st,
                     int max_frames = 0, void* wide_ws = nullptr);
bool ctc_is_wide(int beam, int C, bool utf8 = false);
size_t ctc_wide_row_bytes(int C);
inline bool ctc_sorts_classes(const DecParams& p) { return p.cutoff_prob < 1.0 || p.cutoff_top_n < p.C; }  // :337
inline size_t ctc_rows_ws_bytes(const DecParams& p, int n_streams, int max_frames) {
  return (ctc_is_wide(p.beam, p.C) || ctc_sorts_classes(p)) ? (size_t)n_streams * (size_t)max_frames * ctc_wide_row_bytes(p.C) : 0;
}
// All outputs of a decode launch in ONE block (so they come back with one copy): [n_results | errors | lens | confidence | tokens |
// timesteps]; `view` points a DecodeOut into a block at `base` (device or host).
struct DecodeBlock {
  size_t off_n, off_err, off_len, off_conf, off_tok, off_ts, bytes;
  static DecodeBlock layout(int n_streams, int num_results, int max_len) {
    DecodeBlock b{};
    const size_t nr = (size_t)n_streams * num_results;
    b.off_n = 0; b.off_err = (size_t)n_streams * 4; b.off_len = (size_t)n_streams * 8; b.off_conf = (b.off_len + nr * 4 + 7) & ~(size_t)7;
    b.off_tok = b.off_conf + nr * 8; b.off_ts = b.off_tok + nr * max_len * 4; b.bytes = b.off_ts + nr * max_len * 4;
    return b;
  }
  DecodeOut view(void* base, int num_results, int max_len) const {
    unsigned char* p = (unsigned char*)base;
    DecodeOut o{};
    o.n_results = (int*)(p + off_n); o.errors = (int*)(p + off_err); o.lens = (int*)(p + off_len); o.confidence = (double*)(p + off_conf);
    o.tokens = (uint32_t*)(p + off_tok); o.timesteps = (uint32_t*)(p + off_ts); o.num_results = num_results; o.max_len = max_len;
    return o;
  }
};
void launch_ctc_decode(const DecParams& p, const DevScorer& s, const DevAlphabet& al, const DecStream* streams, int n_streams,
                       const DecodeOut& out, hipStream_t st);
size_t ctc_next_lds_bytes(int beam, int C, bool utf8 = false);
void launch_ctc_init(DecStream* streams, int n_streams, const DevScorer* scorer_or_null, hipStream_t st);
// batched streaming: every stream owns a one-entry table; a launch over many of them works on a gathered copy
void launch_gather_streams(const DecStream* const* src, DecStream* dst, int n, hipStream_t st);
void launch_scatter_streams(DecStream* const* dst, const DecStream* src, int n, hipStream_t st);
