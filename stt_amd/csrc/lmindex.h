// stt_amd/csrc/lmindex.h -- hashed n-gram index over a KenLM trie model (internal header, host + device).
//
// KenLM answers FullScore (kenlm/lm/model.cc:170-176,285-338) by walking its reverse trie: unigram record of the new
// word, then one bit-packed interpolation search per history word (lm/trie.cc:32-99, lm/bhiksha.hh:76-95), every search
// bounded by the [begin, end) range the previous level handed out -- a chain of 10-20 dependent HBM reads per query on
// the GPU, and the search kernel's critical path.  The n-grams of a model are a fixed set, so at scorer load time every
// record of order >= 2 is also entered into an open-addressing table keyed by a hash of
//     (vocabulary hash of the newest word, index of history word 1, ..., index of history word k-1)
// -- everything the key needs is known *before* any trie level has been read, so the lookups of all orders go out
// together and a query is two round trips (cached state of the previous word boundary, then all levels at once) whatever
// the order.  An entry carries exactly what the trie walk would have decoded (probability, backoff, "has no children"),
// plus (order, word, parent entry) so that a hit is verified exactly and not by fingerprint: the entry of order k must
// name the entry matched at order k-1 as its parent.  Values are the model's own floats, the control flow of FullScore is
// restated in lmi_combine(), so results are bit-identical to the trie walk (tests/test_lm_index.py on the host,
// tests/test_gpu_lm.py on the device, against answers of the real KenLM).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define LMI_HD __host__ __device__ __forceinline__
#else
#define LMI_HD inline
#endif

#define LMI_MAX_HIST 5                    // STT_KENLM_MAX_ORDER - 1
#define LMI_EMPTY 0xFFFFFFFFu             // LmiEntry::wl of a free slot
#define LMI_WORD_MASK 0x07FFFFFFu         // word index of the entry's oldest word (vocabularies of < 2^27 words)
#define LMI_LEVEL_SHIFT 27                // 3 bits: order of the n-gram (2..6)
#define LMI_INDEP_BIT 0x40000000u         // the trie record has no children (independent_left, model.cc:300-305)
#define LMI_BUCKET 4                      // entries per bucket: one 64-byte line
#define LMI_NOT_FOUND 0xFFFFFFFFu
#define LMI_UNK_H 0x756e6b3e3c000001ULL   // key seed of <unk> (index 0 has no string hash in KenLM's vocabulary)

struct LmiEntry { uint32_t wl, parent; float prob, backoff; };  // 16 bytes; parent = word index (order 2) or slot id (order > 2)

// key of the order-(k+1) n-gram from the key of its order-k suffix-less prefix (newest word first, KenLM's reverse order)
LMI_HD uint64_t lmi_step(uint64_t acc, uint32_t hist_word) {
  acc = (acc ^ (uint64_t)(hist_word + 1u)) * 0x9E3779B97F4A7C15ULL;
  return acc ^ (acc >> 32);
}
LMI_HD uint32_t lmi_bucket(uint64_t acc, uint32_t n_buckets) {
  const uint32_t x = (uint32_t)(acc ^ (acc >> 29));
  return (uint32_t)(((uint64_t)x * (uint64_t)n_buckets) >> 32);
}
LMI_HD uint32_t lmi_wl(uint32_t word, int level, bool indep) {
  return (word & LMI_WORD_MASK) | ((uint32_t)level << LMI_LEVEL_SHIFT) | (indep ? LMI_INDEP_BIT : 0u);
}
LMI_HD bool lmi_is(uint32_t wl, uint32_t word, int level) {
  return (wl & ~LMI_INDEP_BIT) == ((word & LMI_WORD_MASK) | ((uint32_t)level << LMI_LEVEL_SHIFT));
}

// slot hash of the bigram blocks' table (scorer_host.h: cpb_tab; key = context word, block of 64 consecutive code points)
LMI_HD uint32_t cpb_hash(uint32_t w1, uint32_t block) {
  uint64_t a = ((uint64_t)w1 << 32 | block) * 0x9E3779B97F4A7C15ULL;
  a ^= a >> 29;
  return (uint32_t)(a * 0xD6E8FEB86659FD93ULL >> 32);
}

// What one trie level contributes to FullScore: found = the n-gram of that order exists (and every shorter one did).
struct LmiLevel { int found; float prob, backoff; int indep; };

LMI_HD bool lmi_has_extension(float backoff) {  // kNoExtensionBackoff = -0.0f (lm/blank.hh:20-36)
  union { float f; uint32_t u; } c; c.f = backoff;
  return c.u != 0x80000000u;
}

// GenericModel::FullScore (model.cc:170-176) = ScoreExceptBackoff (:285-310) + ResumeScore (:312-338) on the results of
// the per-order lookups lv[hi] (history index hi <-> order hi + 2).  Same statements in the same order as
// kenlm_full_score() in ctc.hip, which performs the lookups one after the other instead.  KS = KState (ctc.h).
template <class KS>
LMI_HD float lmi_combine(int order, const KS& in, uint32_t new_word, float uni_prob, float uni_backoff, bool uni_indep, const LmiLevel* lv,
                         KS& out, int& ngram_length) {
  float prob = uni_prob;
  out.backoff[0] = uni_backoff;
  bool independent_left = uni_indep;
  int nl = 1;
  int out_len = lmi_has_extension(uni_backoff) ? 1 : 0;
  out.words[0] = new_word;
  bool go = in.length != 0;
#if defined(__HIPCC__) || defined(__HIP__)
#pragma unroll
#endif
  for (int om2 = 0; om2 < LMI_MAX_HIST; ++om2) {
    if (om2 + 1 < LMI_MAX_HIST) { out.words[om2 + 1] = in.words[om2]; out.backoff[om2 + 1] = 0.0f; }
    if (go) {
      if (om2 == in.length || independent_left) go = false;
      else if (om2 == order - 2) { go = false; if (lv[om2].found) { prob = lv[om2].prob; nl = order; } }
      else if (om2 < LMI_MAX_HIST - 1) {
        if (!lv[om2].found) go = false;
        else {
          out.backoff[om2 + 1] = lv[om2].backoff; prob = lv[om2].prob; nl = om2 + 2; independent_left = lv[om2].indep != 0;
          if (lmi_has_extension(lv[om2].backoff)) out_len = nl;
        }
      }
    }
  }
  out.length = out_len;
#if defined(__HIPCC__) || defined(__HIP__)
#pragma unroll
#endif
  for (int i = 0; i < LMI_MAX_HIST; ++i)
    if (i >= nl - 1 && i < in.length) prob = prob + in.backoff[i];
  ngram_length = nl;
  return prob;
}

// Sequential lookup (host build checks, device slow path): scans buckets from `b` on for the order-`level` entry of
// (`word`, `parent`).  A bucket with a free slot ends the search: the builder never skips a free slot.
LMI_HD uint32_t lmi_probe(const LmiEntry* tab, uint32_t n_buckets, uint32_t b, int level, uint32_t word, uint32_t parent, LmiEntry& hit) {
  for (uint32_t tries = 0; tries < n_buckets; ++tries) {
    bool has_free = false;
    for (int j = 0; j < LMI_BUCKET; ++j) {
      const LmiEntry e = tab[(uint64_t)b * LMI_BUCKET + j];
      if (e.wl == LMI_EMPTY) { has_free = true; continue; }
      if (lmi_is(e.wl, word, level) && e.parent == parent) { hit = e; return b * LMI_BUCKET + (uint32_t)j; }
    }
    if (has_free) return LMI_NOT_FOUND;
    b = b + 1 == n_buckets ? 0u : b + 1;
  }
  return LMI_NOT_FOUND;
}
