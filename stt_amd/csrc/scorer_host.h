// stt_amd/csrc/scorer_host.h -- host-side view of a parsed .scorer package / KenLM trie binary (internal header).
// parse_scorer() needs no GPU: layout of the KenLM blob (kenlm/lm/binary_format.cc, search_trie.cc), the repacked
// dictionary, the vocabulary table and the hashed n-gram index of lmindex.h.  ScorerDev::Parse uploads it.
#pragma once
#include <stdint.h>

#include <vector>

#include "ctc.h"
#include "lmindex.h"

struct HostBitPacked {
  uint64_t base_off = 0, off_begin_off = 0, entries = 0;
  uint32_t off_count = 0;
  uint8_t word_bits = 0, total_bits = 0, quant_bits = 0, next_bits = 0;
};

struct HostScorer {
  const uint8_t* buf = nullptr;  // caller's buffer (must stay alive; >= 8 readable bytes past lm_end)
  int order = 0, model_type = 0;
  bool quant = false, utf8 = false;
  double alpha = 0.0, beta = 0.0;
  uint64_t counts[STT_KENLM_MAX_ORDER] = {};
  uint64_t vocab_n = 0, vocab_off = 0, unigram_off = 0, lm_end = 0;
  uint8_t prob_bits = 0, backoff_bits = 0;
  uint64_t qprob_off[STT_KENLM_MAX_ORDER] = {}, qback_off[STT_KENLM_MAX_ORDER] = {};
  HostBitPacked mid[STT_KENLM_MAX_ORDER - 2], lon;
  uint32_t bos_index = 0;
  float bos_backoff = 0.0f;
  // PROBING / REST_PROBING binaries (model types 0, 1): byte offsets of the open-addressing tables and their bucket counts
  bool probing = false;
  int p_wstride = 0, p_estride = 0;
  uint64_t p_vocab_tab_off = 0, p_vocab_buckets = 0;
  uint64_t p_mid_off[STT_KENLM_MAX_ORDER - 2] = {}, p_mid_buckets[STT_KENLM_MAX_ORDER - 2] = {};
  uint64_t p_lon_off = 0, p_lon_buckets = 0;
  // dictionary (ctc.h: DevScorer::fst_*)
  int fst_start = 0;
  uint64_t n_states = 0;
  std::vector<uint32_t> fst_pos;
  std::vector<uint2> fst_arcs, fst_rec;
  std::vector<uint8_t> fst_has_space;
  bool fst_bitmap_ok = false;
  bool fst_tree = false;   // the tables hold the dictionary unfolded into a tree (node k + 1 = target of arc k, root 0)
  // vocabulary table, Bhiksha hints
  std::vector<DevVocabSlot> vtab;
  bool uni_ok = false;
  std::vector<uint32_t> hints;
  size_t hint_off[STT_KENLM_MAX_ORDER - 2] = {};
  uint32_t hint_shift[STT_KENLM_MAX_ORDER - 2] = {};
  // hashed n-gram index
  bool lmi_ok = false;
  std::vector<LmiEntry> lmi;
  uint32_t lmi_buckets = 0;
  float unk_prob = 0.0f, unk_backoff = 0.0f;
  bool unk_indep = true;
  // code-point scorers: an upper bound of get_log_cond_prob() (natural log, rounded up to float) of any n-gram that ENDS with the
  // code point, indexed by code point (U+0000 .. U+FFFF); empty = none (scorer_dev.cpp: build_unit_bounds)
  std::vector<float> cp_ub;
  float cp_ub_max = 0.0f;

  // code-point scorers (tunable cp_blocks; host side and its tests in round 4, the search step's use comes next -- DESIGN.md 7.1):
  // the bigrams (u | w1) regrouped by CONTEXT and by blocks of 64 consecutive code points u, so that the 64 children of a prefix that
  // complete a code point -- one context, 64 consecutive u -- find their records through ONE table entry and one contiguous slice
  // instead of 64 scattered probes.  cpt: unigram record by code point; cpb_tab: open addressing over (w1, u >> 6);
  // cpb_rec[offset + popcount(present below u & 63)] = {prob, backoff, slot of the bigram in the hashed index (its orders >= 3 continue there)}
  struct CptEntry { uint32_t wi; float prob, backoff; uint32_t flags; };                        // flags: 1 = in the vocabulary, 2 = no longer n-gram ends with it
  struct CpbEntry { uint32_t w1, block, offset, count; uint64_t present, indep; };              // w1 == 0xFFFFFFFF: free
  struct CpbRec { float prob, backoff; uint32_t slot; };
  std::vector<CptEntry> cpt;
  std::vector<CpbEntry> cpb_tab;
  std::vector<CpbRec> cpb_rec;
  uint32_t cpb_mask = 0;
  bool cpb_ok = false;
  // FullScore of code point `cp` (U+0001 .. U+FFFF) from state `in` through cpt / cpb_tab (order 2) and the hashed index (orders >= 3)
  float full_score_blocks(const KState& in, uint32_t cp, KState& out, int& ngram_length, uint32_t& word_index) const;

  float middle_prob(int om2, uint64_t at) const;
  float middle_backoff(int om2, uint64_t at) const;
  float longest_prob(uint64_t at) const;
  uint32_t vocab_index(uint64_t murmur_hash) const;
  // GenericModel::FullScore through the index (the arithmetic of the search kernel's LM waves, on the host)
  float full_score_indexed(const KState& in, const char* word, size_t word_len, KState& out, int& ngram_length, uint32_t& word_index) const;
  // ... and through a PROBING / REST_PROBING binary's own hash tables (the arithmetic of ctc.hip: kenlm_full_score_probing, on the host)
  float full_score_probing(const KState& in, const char* word, size_t word_len, KState& out, int& ngram_length, uint32_t& word_index) const;
};

// STT_ERR_* code.  lm_only: a bare KenLM binary without the 'TRIE' trailer (test hook).
int parse_scorer(const uint8_t* buf, size_t len, int space_label, bool lm_only, HostScorer& out);
