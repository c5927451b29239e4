/* oracle/stt_port.c -- TEST INFRASTRUCTURE ONLY ("port" oracle).
 *
 * Plain-C restatement of the decoder half of the hot path (SURVEY.md section 8a rows
 * a7-a13): scorer package reader, KenLM trie query, dictionary FST match, CTC prefix beam
 * search.  It is a *restatement*, not a copy: the reference's pointer trie is replaced by
 * the flat struct-of-arrays beam the HIP kernel uses, but every arithmetic step, rounding
 * point and visiting order follows the reference lines cited at each function.
 *
 * PINNED against the real reference (oracle/_ref/libctcdecode_ref.so, built from
 * /root/reference by oracle/Makefile) by tests/test_oracle_port.py: KenLM known answers
 * restated from native_client/kenlm/lm/model_test.cc, get_log_cond_prob on the shipped
 * smoke-test scorers, and whole-beam equality (tokens, timesteps, float scores bit for
 * bit) on seeded emissions.  The product never links this file.
 *
 * Documented deviations (all concern orders the reference itself leaves to libstdc++):
 *  - ties of (score, character) in std::partial_sort / std::nth_element
 *    (ctc_beam_search_decoder.cpp:138,264,305) are broken by `origin`: live prefixes (by beam
 *    index) before prefixes created this step (by parent beam index); see step().
 *  - std::sort ties between equal class probabilities (ctc_beam_search_decoder.cpp:338)
 *    are broken by class index.
 *  - KenLM probing-hash models (model types 0/1) are not supported, trie types 2-5 are.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glibc_flt.h"

#define NEG_INF (-FLT_MAX) /* -NUM_FLT_INF, decoder_utils.h:11 */
#define KENLM_MAX_ORDER 6
#define OOV_SCORE (-1000.0) /* scorer.h:16 */

/* ============================================================================================
 * Part A. scorer package  
 * ==========================================================================================*/
typedef struct {
  /* bit-packed array: BitPacked, lm/trie.hh:74-95 */
  const uint8_t* base;
  uint8_t word_bits, total_bits;
  uint64_t word_mask;
  /* middle only */
  uint8_t quant_bits;
  uint8_t next_bits; /* inline bits */
  uint64_t next_mask;
  const uint64_t* offset_begin; /* ArrayBhiksha, lm/bhiksha.hh:66-108; NULL for DontBhiksha */
  const uint64_t* offset_end;
} PBitPacked;

typedef struct {
  const uint8_t* buf;
  size_t len;
  /* KenLM */
  int order, model_type, quant, array;
  uint64_t counts[KENLM_MAX_ORDER];
  const uint64_t* vocab; /* sorted murmur hashes, lm/vocab.hh:72-83 */
  uint64_t vocab_n;
  uint8_t prob_bits, backoff_bits;
  const float* qprob[KENLM_MAX_ORDER];    /* per middle order_minus_2; [order-2] = longest */
  const float* qbackoff[KENLM_MAX_ORDER];
  const uint8_t* unigram; /* {f32 prob, f32 backoff, u64 next}[counts[0]+2], lm/trie.hh:22-26 */
  PBitPacked middle[KENLM_MAX_ORDER];
  PBitPacked longest;
  uint32_t bos_index, eos_index;
  float bos_backoff;
  uint64_t lm_end; /* GetEndOfSearchOffset, lm/model.cc:265-267 */
  /* package header, scorer.cpp:177-222 */
  int utf8;
  double alpha, beta;
  /* ConstFst<StdArc>, const-fst.h:102-110 */
  int64_t fst_start, fst_nstates, fst_narcs;
  const uint8_t* fst_states; /* {f32 weight, u32 pos, u32 narcs, u32 nieps, u32 noeps} */
  const uint8_t* fst_arcs;   /* {i32 ilabel, i32 olabel, f32 weight, i32 nextstate} */
} PortScorer;

static uint8_t required_bits(uint64_t v) { /* util/bit_packing.cc:17-22 */
  if (!v) return 0;
  uint8_t r = 1;
  while (v >>= 1) ++r;
  return r;
}
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float rdf32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }
/* util/bit_packing.hh ReadInt57 / ReadInt25 (little endian) */
static inline uint64_t read_int57(const uint8_t* base, uint64_t bit_off, uint64_t mask) {
  return (rd64(base + (bit_off >> 3)) >> (bit_off & 7)) & mask;
}
static inline uint32_t read_int25(const uint8_t* base, uint64_t bit_off, uint32_t mask) {
  return (rd32(base + (bit_off >> 3)) >> (bit_off & 7)) & mask;
}

/* util/murmur_hash.cc MurmurHash64A, seed 0 == detail::HashForVocab (lm/vocab.cc:23-27) */
uint64_t port_murmur64a(const void* key, size_t len, uint64_t seed) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint64_t h = seed ^ (len * m);
  const uint8_t* data = (const uint8_t*)key;
  const uint8_t* end = data + (len / 8) * 8;
  while (data != end) {
    uint64_t k = rd64(data);
    data += 8;
    k *= m; k ^= k >> r; k *= m;
    h ^= k; h *= m;
  }
  switch (len & 7) {
    case 7: h ^= (uint64_t)data[6] << 48; /* fallthrough */
    case 6: h ^= (uint64_t)data[5] << 40; /* fallthrough */
    case 5: h ^= (uint64_t)data[4] << 32; /* fallthrough */
    case 4: h ^= (uint64_t)data[3] << 24; /* fallthrough */
    case 3: h ^= (uint64_t)data[2] << 16; /* fallthrough */
    case 2: h ^= (uint64_t)data[1] << 8;  /* fallthrough */
    case 1: h ^= (uint64_t)data[0]; h *= m;
  }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

/* SortedVocabulary::Index, lm/vocab.hh:72-83 (any exact search gives the same index) */
static uint32_t vocab_index_hash(const PortScorer* s, uint64_t h) {
  uint64_t lo = 0, hi = s->vocab_n;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    uint64_t v = s->vocab[mid];
    if (v < h) lo = mid + 1; else if (v > h) hi = mid; else return (uint32_t)(mid + 1);
  }
  return 0;
}
uint32_t port_kenlm_index(const PortScorer* s, const char* w, size_t n) { return vocab_index_hash(s, port_murmur64a(w, n, 0)); }

/* lm/bhiksha.cc:35-58 */
static uint8_t chop_bits(uint64_t max_offset, uint64_t max_next, uint8_t cfg_bits) {
  uint8_t required = required_bits(max_next);
  uint8_t best = 0;
  int64_t lowest = INT64_MAX;
  uint8_t lim = required < cfg_bits ? required : cfg_bits;
  for (uint8_t chop = 0; chop <= lim; ++chop) {
    int64_t change = (int64_t)((max_next >> (required - chop)) * 64) - (int64_t)max_offset * (int64_t)chop;
    if (change < lowest) { lowest = change; best = chop; }
  }
  return best;
}

static uint64_t align8(uint64_t x) { return (x + 7) & ~(uint64_t)7; }

void port_scorer_free(PortScorer* s) { free(s); }

/* Parses [KenLM trie binary]['TRIE' header][ConstFst].  Error codes mirror
 * scorer.cpp:108-222 (STT_ERR_SCORER_*), coqui-stt.h:92-124. */
static PortScorer* scorer_load_impl(const uint8_t* buf, size_t len, int lm_only, int* err) {
  static const char kMagic[] = "mmap lm http://kheafield.com/code format version 5\n";
  int e = 0;
  PortScorer* s = (PortScorer*)calloc(1, sizeof(PortScorer));
  s->buf = buf; s->len = len;
  /* lm/binary_format.cc:22-75: Sanity (88 B) + FixedWidthParameters (20 B) + counts */
  if (len < 88 + 20 || memcmp(buf, kMagic, sizeof(kMagic)) != 0) { e = 0x2006; goto fail; }
  {
    const uint8_t* fp = buf + 88;
    s->order = fp[0];
    s->model_type = (int)rd32(fp + 8);
    if (s->order < 2 || s->order > KENLM_MAX_ORDER) { e = 0x2006; goto fail; }
    if (s->model_type < 2 || s->model_type > 5) { e = 0x2006; goto fail; } /* trie family only */
    s->quant = (s->model_type == 3 || s->model_type == 5);
    s->array = (s->model_type == 4 || s->model_type == 5);
    for (int i = 0; i < s->order; ++i) s->counts[i] = rd64(buf + 108 + 8 * i);
  }
  {
    uint64_t off = align8(108 + 8 * (uint64_t)s->order);
    /* vocab: u64 n; u64 hashes[]  -- SortedVocabulary::Size = 8 + 8*counts[0] (lm/vocab.cc:113-116) */
    s->vocab_n = rd64(buf + off);
    s->vocab = (const uint64_t*)(buf + off + 8);
    off += 8 + 8 * s->counts[0];
    /* search: TrieSearch::SetupMemory, lm/search_trie.cc:546-571 */
    if (s->quant) { /* SeparatelyQuantize::SetupMemory, lm/quantize.cc:54-73 */
      s->prob_bits = buf[off + 1];
      s->backoff_bits = buf[off + 2];
      const float* t = (const float*)(buf + off + 8);
      for (int i = 0; i < s->order - 2; ++i) {
        s->qprob[i] = t; t += (1ULL << s->prob_bits);
        s->qbackoff[i] = t; t += (1ULL << s->backoff_bits);
      }
      s->qprob[s->order - 2] = t; t += (1ULL << s->prob_bits);
      off = (uint64_t)((const uint8_t*)t - buf);
    }
    s->unigram = buf + off;
    off += (s->counts[0] + 2) * 16;
    uint8_t cfg_bhiksha_bits = 0;
    if (s->array && s->order > 2) cfg_bhiksha_bits = buf[off + 1]; /* ArrayBhiksha::UpdateConfigFromBinary */
    uint8_t middle_quant_bits = s->quant ? (uint8_t)(s->prob_bits + s->backoff_bits) : 63;
    uint8_t longest_bits = s->quant ? s->prob_bits : 31;
    for (int i = 0; i < s->order - 2; ++i) { /* middle i holds (i+2)-grams */
      PBitPacked* m = &s->middle[i];
      uint64_t entries = s->counts[i + 1], max_vocab = s->counts[0], max_next = s->counts[i + 2];
      uint64_t bh_size = 0;
      uint8_t inline_bits;
      if (s->array) { /* lm/bhiksha.cc:60-84 */
        uint8_t required = required_bits(max_next);
        uint8_t chop = chop_bits(entries + 1, max_next, cfg_bhiksha_bits);
        uint64_t array_count = (max_next >> (required - chop)) + 1;
        bh_size = 8 * (1 + array_count) + 7;
        inline_bits = required - chop;
        uint64_t ab = align8(off);
        m->offset_begin = (const uint64_t*)(buf + ab + 8);
        m->offset_end = m->offset_begin + array_count;
      } else {
        inline_bits = required_bits(max_next);
      }
      m->base = buf + off + bh_size;
      m->word_bits = required_bits(max_vocab);
      m->word_mask = (1ULL << m->word_bits) - 1;
      m->quant_bits = middle_quant_bits;
      m->next_bits = inline_bits;
      m->next_mask = (1ULL << inline_bits) - 1;
      m->total_bits = (uint8_t)(m->word_bits + middle_quant_bits + inline_bits);
      off += bh_size + (((1 + entries) * m->total_bits + 7) / 8 + 8); /* BitPacked::BaseSize, lm/trie.cc:39-46 */
    }
    s->longest.base = buf + off;
    s->longest.word_bits = required_bits(s->counts[0]);
    s->longest.word_mask = (1ULL << s->longest.word_bits) - 1;
    s->longest.total_bits = (uint8_t)(s->longest.word_bits + longest_bits);
    off += ((1 + s->counts[s->order - 1]) * s->longest.total_bits + 7) / 8 + 8;
    s->lm_end = off;
    if (off > len) { e = 0x2006; goto fail; }
  }
  s->bos_index = port_kenlm_index(s, "<s>", 3);
  s->eos_index = port_kenlm_index(s, "</s>", 4);
  s->bos_backoff = rdf32(s->unigram + 16 * (uint64_t)s->bos_index + 4); /* lm/model.cc:115-124 */
  /* ---- package trailer, scorer.cpp:177-222 */
  if (lm_only) { if (err) *err = 0; return s; } /* bare KenLM binary (model_test.cc known answers) */
  if (len <= s->lm_end) { e = 0x2007; goto fail; }
  {
    const uint8_t* p = buf + s->lm_end;
    if (s->lm_end + 25 > len || rd32(p) != 0x54524945u /*'TRIE'*/) { e = 0x2008; goto fail; }
    if ((int)rd32(p + 4) != 6) { e = 0x2009; goto fail; }
    s->utf8 = p[8] != 0;
    double a, b;
    memcpy(&a, p + 9, 8); memcpy(&b, p + 17, 8);
    s->alpha = (double)(float)a; /* Scorer::reset_params(float, float), scorer.cpp:346-351 */
    s->beta = (double)(float)b;
    /* FstHeader::Read, openfst-1.6.7/src/lib/fst.cc:57-84 */
    uint64_t o = s->lm_end + 25;
    if (rd32(buf + o) != 2125659606u) { e = 0x2008; goto fail; }
    o += 4;
    uint32_t l = rd32(buf + o); o += 4 + l; /* fsttype "const" */
    l = rd32(buf + o); o += 4 + l;          /* arctype "standard" */
    o += 4;                                 /* version */
    uint32_t flags = rd32(buf + o); o += 4;
    o += 8;                                 /* properties */
    memcpy(&s->fst_start, buf + o, 8); o += 8;
    memcpy(&s->fst_nstates, buf + o, 8); o += 8;
    memcpy(&s->fst_narcs, buf + o, 8); o += 8;
    if (flags & 3) { e = 0x2008; goto fail; } /* symbol tables never written by the reference */
    if (flags & 4) o = (o + 15) & ~(uint64_t)15; /* AlignInput, lib/util.cc:60-72 (absolute position) */
    s->fst_states = buf + o;
    o += (uint64_t)s->fst_nstates * 20;
    if (flags & 4) o = (o + 15) & ~(uint64_t)15;
    s->fst_arcs = buf + o;
    o += (uint64_t)s->fst_narcs * 16;
    if (o > len) { e = 0x2008; goto fail; }
  }
  if (err) *err = 0;
  return s;
fail:
  if (err) *err = e;
  free(s);
  return NULL;
}

PortScorer* port_scorer_load(const uint8_t* buf, size_t len, int* err) { return scorer_load_impl(buf, len, 0, err); }
PortScorer* port_kenlm_load(const uint8_t* buf, size_t len, int* err) { return scorer_load_impl(buf, len, 1, err); }
int port_scorer_order(const PortScorer* s) { return s->order; }
int port_scorer_utf8(const PortScorer* s) { return s->utf8; }
double port_scorer_alpha(const PortScorer* s) { return s->alpha; }
double port_scorer_beta(const PortScorer* s) { return s->beta; }
void port_scorer_set_alpha_beta(PortScorer* s, float a, float b) { s->alpha = a; s->beta = b; }
uint64_t port_scorer_lm_end(const PortScorer* s) { return s->lm_end; }
int port_scorer_model_type(const PortScorer* s) { return s->model_type; }

/* ---- dictionary FST: SortedMatcher::Find on ConstFst (matcher.h:347-386) + Final() ---- */
static inline int fst_is_final(const PortScorer* s, int64_t st) { /* weight != TropicalWeight::Zero() (+inf) */
  float w = rdf32(s->fst_states + 20 * st);
  return !(w == INFINITY);
}
static int fst_find(const PortScorer* s, int64_t st, int label, int* next) {
  const uint8_t* S = s->fst_states + 20 * st;
  uint32_t pos = rd32(S + 4), narcs = rd32(S + 8);
  uint32_t lo = 0, hi = narcs;
  while (lo < hi) {
    uint32_t mid = (lo + hi) / 2;
    int il = (int)rd32(s->fst_arcs + 16 * (uint64_t)(pos + mid));
    if (il < label) lo = mid + 1; else if (il > label) hi = mid;
    else { *next = (int)rd32(s->fst_arcs + 16 * (uint64_t)(pos + mid) + 12); return 1; }
  }
  return 0;
}
long port_scorer_fst_dump(const PortScorer* s, int* start, int* triples, long cap_arcs, long* n_arcs, uint8_t* finals, long cap_states) {
  long na = 0;
  if (start) *start = (int)s->fst_start;
  for (int64_t st = 0; st < s->fst_nstates; ++st) {
    if (finals && st < cap_states) finals[st] = (uint8_t)fst_is_final(s, st);
    const uint8_t* S = s->fst_states + 20 * st;
    uint32_t pos = rd32(S + 4), narcs = rd32(S + 8);
    for (uint32_t a = 0; a < narcs; ++a, ++na) {
      if (triples && na < cap_arcs) {
        triples[3 * na] = (int)st;
        triples[3 * na + 1] = (int)rd32(s->fst_arcs + 16 * (uint64_t)(pos + a));
        triples[3 * na + 2] = (int)rd32(s->fst_arcs + 16 * (uint64_t)(pos + a) + 12);
      }
    }
  }
  if (n_arcs) *n_arcs = na;
  return (long)s->fst_nstates;
}

/* ---- KenLM trie query ------------------------------------------------------------------ */
typedef struct { uint32_t words[KENLM_MAX_ORDER - 1]; float backoff[KENLM_MAX_ORDER - 1]; uint8_t length; } PState; /* lm/state.hh:15-48 */
typedef struct { uint64_t begin, end; } PNode;

static inline int has_extension(float backoff) { return gf_asuint(backoff) != 0x80000000u; } /* lm/blank.hh:25-33 */

/* FindBitPacked, lm/trie.cc:32-36: exact search of `word` among records [begin,end) */
static int find_bitpacked(const PBitPacked* bp, uint64_t begin, uint64_t end, uint64_t key, uint64_t* at) {
  while (begin < end) {
    uint64_t mid = begin + (end - begin) / 2;
    uint64_t v = read_int57(bp->base, mid * bp->total_bits, bp->word_mask);
    if (v < key) begin = mid + 1; else if (v > key) end = mid; else { *at = mid; return 1; }
  }
  return 0;
}
/* Bhiksha::ReadNext, lm/bhiksha.hh:38-42 (Dont) and :76-95 (Array) */
static void read_next(const PBitPacked* m, uint64_t bit_offset, uint64_t index, PNode* out) {
  if (!m->offset_begin) {
    out->begin = read_int57(m->base, bit_offset, m->next_mask);
    out->end = read_int57(m->base, bit_offset + m->total_bits, m->next_mask);
    return;
  }
  /* upper_bound(offsets, index) - 1 */
  const uint64_t* lo = m->offset_begin; const uint64_t* hi = m->offset_end;
  while (lo < hi) { const uint64_t* mid = lo + (hi - lo) / 2; if (*mid <= index) lo = mid + 1; else hi = mid; }
  const uint64_t* begin_it = lo - 1;
  const uint64_t* end_it;
  for (end_it = begin_it + 1; (end_it < m->offset_end) && (*end_it <= index + 1); ++end_it) {}
  --end_it;
  out->begin = ((uint64_t)(begin_it - m->offset_begin) << m->next_bits) | read_int57(m->base, bit_offset, m->next_mask);
  out->end = ((uint64_t)(end_it - m->offset_begin) << m->next_bits) | read_int57(m->base, bit_offset + m->total_bits, m->next_mask);
}
/* TrieSearch::LookupMiddle, lm/search_trie.hh:80-84 + BitPackedMiddle::Find, lm/trie.cc:88-99 +
 * MiddlePointer::{Prob,Backoff}, lm/quantize.hh:40-62,152-166 */
static int lookup_middle(const PortScorer* s, int order_minus_2, uint32_t word, PNode* node, int* independent_left, float* prob, float* backoff) {
  const PBitPacked* m = &s->middle[order_minus_2];
  uint64_t at;
  if (!find_bitpacked(m, node->begin, node->end, word, &at)) { *independent_left = 1; return 0; }
  uint64_t addr = at * m->total_bits + m->word_bits;
  read_next(m, addr + m->quant_bits, at, node);
  *independent_left = (node->begin == node->end);
  if (s->quant) {
    *backoff = s->qbackoff[order_minus_2][read_int25(m->base, addr, (1u << s->backoff_bits) - 1)];
    *prob = s->qprob[order_minus_2][read_int25(m->base, addr + s->backoff_bits, (1u << s->prob_bits) - 1)];
  } else {
    uint32_t pi = (uint32_t)(rd64(m->base + (addr >> 3)) >> (addr & 7)) | 0x80000000u; /* ReadNonPositiveFloat31 */
    uint32_t bi = (uint32_t)(rd64(m->base + ((addr + 31) >> 3)) >> ((addr + 31) & 7));  /* ReadFloat32 */
    *prob = gf_asfloat(pi);
    *backoff = gf_asfloat(bi);
  }
  return 1;
}
static int lookup_longest(const PortScorer* s, uint32_t word, const PNode* node, float* prob) {
  const PBitPacked* l = &s->longest;
  uint64_t at;
  if (!find_bitpacked(l, node->begin, node->end, word, &at)) return 0;
  uint64_t addr = at * l->total_bits + l->word_bits;
  if (s->quant) *prob = s->qprob[s->order - 2][read_int25(l->base, addr, (1u << s->prob_bits) - 1)];
  else *prob = gf_asfloat((uint32_t)(rd64(l->base + (addr >> 3)) >> (addr & 7)) | 0x80000000u);
  return 1;
}

/* GenericModel::FullScore = ScoreExceptBackoff + ResumeScore + backoff charge,
 * lm/model.cc:170-176, 285-338.  Returns prob; *ngram_length as FullScoreReturn. */
static float kenlm_full_score(const PortScorer* s, const PState* in, uint32_t new_word, PState* out, int* ngram_length) {
  PNode node;
  const uint8_t* u = s->unigram + 16 * (uint64_t)new_word;
  float prob = rdf32(u);
  out->backoff[0] = rdf32(u + 4);
  node.begin = rd64(u + 8);
  node.end = rd64(u + 24);
  int independent_left = (node.begin == node.end);
  int nl = 1;
  out->length = has_extension(out->backoff[0]) ? 1 : 0;
  out->words[0] = new_word;
  if (in->length != 0) {
    const uint32_t* hist = in->words;
    const uint32_t* hend = in->words + in->length;
    float* backoff_out = out->backoff + 1;
    int om2 = 0;
    int broke = 0;
    for (;; ++om2, ++hist, ++backoff_out) { /* ResumeScore, lm/model.cc:312-338 */
      if (hist == hend) break;
      if (independent_left) break;
      if (om2 == s->order - 2) { broke = 1; break; }
      float p, b;
      if (!lookup_middle(s, om2, *hist, &node, &independent_left, &p, &b)) break;
      *backoff_out = b;
      prob = p;
      nl = om2 + 2;
      if (has_extension(b)) out->length = (uint8_t)nl;
    }
    if (broke) {
      float p;
      if (lookup_longest(s, *hist, &node, &p)) { prob = p; nl = s->order; }
    }
    /* CopyRemainingHistory, lm/model.cc:273-277 */
    for (int i = 0; i + 1 < out->length; ++i) out->words[i + 1] = in->words[i];
  }
  for (int i = nl - 1; i < in->length; ++i) prob += in->backoff[i];
  if (ngram_length) *ngram_length = nl;
  return prob;
}

static void state_begin(const PortScorer* s, PState* st, int bos) {
  memset(st, 0, sizeof(*st));
  if (bos) { st->length = 1; st->words[0] = s->bos_index; st->backoff[0] = s->bos_backoff; }
}

/* model_test.cc style sentence scoring (test helper) */
int port_kenlm_score(const PortScorer* s, const char** words, int n, int bos, float* probs, int* ngram_len) {
  PState a, b, *in = &a, *out = &b;
  state_begin(s, in, bos);
  for (int i = 0; i < n; ++i) {
    probs[i] = kenlm_full_score(s, in, port_kenlm_index(s, words[i], strlen(words[i])), out, &ngram_len[i]);
    PState* t = in; in = out; out = t;
  }
  return 0;
}

/* Scorer::get_log_cond_prob over word *hashes*, scorer.cpp:308-344 */
static double log_cond_prob_hashes(const PortScorer* s, const uint64_t* hashes, int n, int bos, int eos) {
  PState a, b, *in = &a, *out = &b;
  state_begin(s, in, bos);
  double cond_prob = 0.0;
  for (int i = 0; i < n; ++i) {
    uint32_t wi = vocab_index_hash(s, hashes[i]);
    if (wi == 0) return OOV_SCORE;
    cond_prob = (double)kenlm_full_score(s, in, wi, out, NULL);
    PState* t = in; in = out; out = t;
  }
  if (eos) cond_prob = (double)kenlm_full_score(s, in, s->eos_index, out, NULL);
  return cond_prob / (double)0.4342944819f; /* NUM_FLT_LOGE is a float constant, decoder_utils.h:13 */
}
double port_scorer_log_cond_prob(const PortScorer* s, const char** words, int n, int bos, int eos) {
  uint64_t h[64];
  if (n > 64) return NAN;
  for (int i = 0; i < n; ++i) h[i] = port_murmur64a(words[i], strlen(words[i]), 0);
  return log_cond_prob_hashes(s, h, n, bos, eos);
}

/* ============================================================================================
 * Part B. CTC prefix beam search  (DecoderState, ctc_beam_search_decoder.cpp:22-358)
 * ==========================================================================================*/
typedef struct {
  int C, blank, space, beam, cutoff_top_n;
  double cutoff_prob;
  PortScorer* sc;
  const uint8_t* label_bytes; /* UTF-8 of every label, concatenated */
  const int* label_off;       /* [C] offsets, label_off[C-1] = total (C-1 labels) */
  int abs_t, start_expanding;
  /* live beam, always kept sorted by (score desc, character asc, origin asc) */
  int n;
  float *score, *pb, *pnb;
  uint32_t* ch; /* 0xFFFFFFFF = root (ROOT_ = -1, path_trie.cpp:20-21) */
  uint32_t *node, *ts;
  int* fst;
  uint64_t* key;
  /* arenas */
  uint32_t *pa_parent, *pa_ch; size_t pa_n, pa_cap; /* node 0 = root */
  uint32_t *ta_parent, *ta_t; size_t ta_n, ta_cap;  /* entry 0 = timestep_tree_root_ */
  int n_hot; uint64_t* hot_hash; float* hot_boost;
  /* statistics (for the roofline accounting in DESIGN.md) */
  uint64_t stat_lm_queries, stat_candidates, stat_steps;
  /* Steps at which the last kept and the first dropped prefix compare equal under prefix_compare (same score, same character): the
   * reference keeps whichever std::nth_element happens to leave in front (libstdc++'s introselect on the trie's iteration order --
   * unspecified by the standard), this restatement and the kernels keep (live before new, beam index).  From such a step on the two
   * may hold different -- equally scored -- prefixes; every other step is determined by the scores alone. */
  uint64_t stat_boundary_ties;
} PortDecoder;

static inline uint64_t child_key(uint64_t parent_key, uint32_t c) {
  uint64_t x = parent_key + 0x9E3779B97F4A7C15ULL * (uint64_t)(c + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
  return x;
}

PortDecoder* port_decoder_new(int C, int space_id, int beam, double cutoff_prob, int cutoff_top_n, PortScorer* sc,
                              const uint8_t* label_bytes, const int* label_off,
                              const char** hot_words, const float* boosts, int n_hot) {
  PortDecoder* d = (PortDecoder*)calloc(1, sizeof(PortDecoder));
  d->C = C; d->blank = C - 1; d->space = space_id; d->beam = beam;
  d->cutoff_prob = cutoff_prob; d->cutoff_top_n = cutoff_top_n; d->sc = sc;
  d->label_bytes = label_bytes; d->label_off = label_off;
  int cap = beam + 1;
  d->score = malloc(sizeof(float) * cap); d->pb = malloc(sizeof(float) * cap); d->pnb = malloc(sizeof(float) * cap);
  d->ch = malloc(sizeof(uint32_t) * cap); d->node = malloc(sizeof(uint32_t) * cap); d->ts = malloc(sizeof(uint32_t) * cap);
  d->fst = malloc(sizeof(int) * cap); d->key = malloc(sizeof(uint64_t) * cap);
  d->pa_cap = 1 << 16; d->pa_parent = malloc(4 * d->pa_cap); d->pa_ch = malloc(4 * d->pa_cap);
  d->ta_cap = 1 << 16; d->ta_parent = malloc(4 * d->ta_cap); d->ta_t = malloc(4 * d->ta_cap);
  d->pa_parent[0] = 0xFFFFFFFFu; d->pa_ch[0] = 0xFFFFFFFFu; d->pa_n = 1;
  d->ta_parent[0] = 0xFFFFFFFFu; d->ta_t[0] = 0; d->ta_n = 1;
  /* root prefix, ctc_beam_search_decoder.cpp:43-56 */
  d->n = 1; d->score[0] = 0.0f; d->pb[0] = 0.0f; d->pnb[0] = NEG_INF; d->ch[0] = 0xFFFFFFFFu;
  d->node[0] = 0; d->ts[0] = 0; d->fst[0] = sc ? (int)sc->fst_start : 0; d->key[0] = 0x5151515151515151ULL;
  d->n_hot = n_hot;
  if (n_hot) {
    d->hot_hash = malloc(8 * n_hot); d->hot_boost = malloc(4 * n_hot);
    for (int i = 0; i < n_hot; ++i) { d->hot_hash[i] = port_murmur64a(hot_words[i], strlen(hot_words[i]), 0); d->hot_boost[i] = boosts[i]; }
  }
  return d;
}
void port_decoder_free(PortDecoder* d) {
  if (!d) return;
  free(d->score); free(d->pb); free(d->pnb); free(d->ch); free(d->node); free(d->ts); free(d->fst); free(d->key);
  free(d->pa_parent); free(d->pa_ch); free(d->ta_parent); free(d->ta_t); free(d->hot_hash); free(d->hot_boost);
  free(d);
}
static uint32_t pa_push(PortDecoder* d, uint32_t parent, uint32_t ch) {
  if (d->pa_n == d->pa_cap) { d->pa_cap *= 2; d->pa_parent = realloc(d->pa_parent, 4 * d->pa_cap); d->pa_ch = realloc(d->pa_ch, 4 * d->pa_cap); }
  d->pa_parent[d->pa_n] = parent; d->pa_ch[d->pa_n] = ch;
  return (uint32_t)d->pa_n++;
}
static uint32_t ta_push(PortDecoder* d, uint32_t parent, uint32_t t) {
  if (d->ta_n == d->ta_cap) { d->ta_cap *= 2; d->ta_parent = realloc(d->ta_parent, 4 * d->ta_cap); d->ta_t = realloc(d->ta_t, 4 * d->ta_cap); }
  d->ta_parent[d->ta_n] = parent; d->ta_t[d->ta_n] = t;
  return (uint32_t)d->ta_n++;
}

/* first byte of a label's UTF-8 string (alphabet_.DecodeSingle(c)[0]) */
static inline uint8_t label_first_byte(const PortDecoder* d, uint32_t c) { return d->label_bytes[c ? d->label_off[c - 1] : 0]; }
static inline int label_len(const PortDecoder* d, uint32_t c) { return d->label_off[c] - (c ? d->label_off[c - 1] : 0); }
static inline const uint8_t* label_ptr(const PortDecoder* d, uint32_t c) { return d->label_bytes + (c ? d->label_off[c - 1] : 0); }

/* Scorer::is_scoring_boundary, scorer.cpp:272-299; `node`/`ch` describe the prefix (ch = its last character). */
static int is_scoring_boundary(const PortDecoder* d, uint32_t node, uint32_t ch, uint32_t new_label) {
  if (!d->sc->utf8) return (int)new_label == d->space;
  if (ch == 0xFFFFFFFFu) return 0;
  /* distance_to_codepoint_boundary, path_trie.cpp:129-141 */
  int dist = 0; uint8_t first_byte = 0; int found = 0;
  uint32_t cur = node;
  for (;;) {
    uint32_t c = d->pa_ch[cur];
    if ((label_first_byte(d, c) & 0xC0) != 0x80) { first_byte = (uint8_t)((uint8_t)c + 1); dist += 1; found = 1; break; }
    uint32_t par = d->pa_parent[cur];
    if (par != 0xFFFFFFFFu && d->pa_ch[par] != 0xFFFFFFFFu) { dist += 1; cur = par; continue; }
    break; /* assert(false) in the reference: continuation bytes all the way to the root */
  }
  if (!found) return 0;
  int needed;
  if ((first_byte >> 3) == 0x1E) needed = 4;
  else if ((first_byte >> 4) == 0x0E) needed = 3;
  else if ((first_byte >> 5) == 0x06) needed = 2;
  else if ((first_byte >> 7) == 0x00) needed = 1;
  else return 0;
  return dist == needed;
}

/* Scorer::make_ngram (scorer.cpp:370-396) reduced to the murmur hash of every unit.
 * `node` is the path-arena index of the prefix.  Returns the number of units (oldest first). */
static int make_ngram_hashes(const PortDecoder* d, uint32_t node, uint64_t* hashes) {
  const PortScorer* s = d->sc;
  uint64_t tmp[KENLM_MAX_ORDER];
  int n = 0;
  uint32_t cur = node;
  uint8_t wbuf[1024];
  for (int order = 0; order < s->order; ++order) {
    if (cur == 0xFFFFFFFFu || d->pa_ch[cur] == 0xFFFFFFFFu) break;
    /* collect the unit's labels backwards, then emit bytes forwards */
    uint32_t labs[256]; int nl = 0;
    uint32_t stop;
    if (s->utf8) { /* get_prev_grapheme, path_trie.cpp:113-127 */
      uint32_t x = cur;
      for (;;) {
        uint32_t c = d->pa_ch[x];
        if (c == 0xFFFFFFFFu) { stop = x; break; }
        if (nl < 256) labs[nl++] = c;
        if ((label_first_byte(d, c) & 0xC0) != 0x80) { stop = x; break; }
        x = d->pa_parent[x];
      }
    } else { /* get_prev_word, path_trie.cpp:143-157 */
      uint32_t x = cur;
      for (;;) {
        uint32_t c = d->pa_ch[x];
        if (c == (uint32_t)d->space || c == 0xFFFFFFFFu) { stop = x; break; }
        if (nl < 256) labs[nl++] = c;
        uint32_t par = d->pa_parent[x];
        if (par == 0xFFFFFFFFu) { stop = x; break; }
        x = par;
      }
    }
    cur = d->pa_parent[stop];
    size_t wl = 0;
    for (int i = nl - 1; i >= 0; --i) {
      int ll = label_len(d, labs[i]);
      if (wl + ll > sizeof(wbuf)) break;
      memcpy(wbuf + wl, label_ptr(d, labs[i]), ll); wl += ll;
    }
    tmp[n++] = port_murmur64a(wbuf, wl, 0);
  }
  for (int i = 0; i < n; ++i) hashes[i] = tmp[n - 1 - i];
  return n;
}

/* (LM + hot-word) score of ctc_beam_search_decoder.cpp:219-242, returned as the float `score` */
static float lm_score(PortDecoder* d, uint32_t node_to_score, int with_hot) {
  uint64_t h[KENLM_MAX_ORDER];
  int n = make_ngram_hashes(d, node_to_score, h);
  float hot_boost = 0.0f;
  if (with_hot && d->n_hot) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < d->n_hot; ++j)
        if (h[i] == d->hot_hash[j]) hot_boost += d->hot_boost[j];
  }
  int bos = n < d->sc->order;
  d->stat_lm_queries++;
  return (float)((log_cond_prob_hashes(d->sc, h, n, bos, 0) + (double)hot_boost) * d->sc->alpha);
}

typedef struct { uint64_t key; int slot; } HEnt;

/* One timestep of DecoderState::next, ctc_beam_search_decoder.cpp:118-275 */
static void step(PortDecoder* d, const double* prob) {
  const int C = d->C, beam = d->beam;
  if (prob[d->blank] < 0.999) d->start_expanding = 1;
  if (!d->start_expanding) { d->abs_t++; return; }
  const int n = d->n;
  d->stat_steps++;

  float min_cutoff = NEG_INF;
  int full_beam = 0;
  if (d->sc) { /* :136-146 -- the beam is already in prefix_compare order */
    min_cutoff = (float)((double)d->score[n - 1] + log(prob[d->blank]) - fmax(0.0, d->sc->beta));
    full_beam = (n == beam);
  }

  /* get_pruned_emissions, :328-358 */
  int* cls = malloc(sizeof(int) * C);
  float* pf = malloc(sizeof(float) * C);
  float* lp = malloc(sizeof(float) * C);
  int cutoff_len = C;
  for (int i = 0; i < C; ++i) { cls[i] = i; pf[i] = (float)prob[i]; }
  if (d->cutoff_prob < 1.0 || d->cutoff_top_n < cutoff_len) {
    for (int i = 1; i < C; ++i) { /* stable insertion sort, prob desc */
      int c = cls[i]; int j = i - 1;
      while (j >= 0 && pf[cls[j]] < pf[c]) { cls[j + 1] = cls[j]; --j; }
      cls[j + 1] = c;
    }
    if (d->cutoff_prob < 1.0) {
      double cum = 0.0; cutoff_len = 0;
      for (int i = 0; i < C; ++i) { cum += pf[cls[i]]; cutoff_len += 1; if (cum >= d->cutoff_prob || cutoff_len >= d->cutoff_top_n) break; }
    }
  }
  for (int k = 0; k < cutoff_len; ++k) lp[k] = gf_logf(pf[cls[k]] + FLT_MIN);

  /* working set: slots [0,n) = live prefixes, [n, n+m) = prefixes created this step */
  int cap = n + n * cutoff_len + 1;
  float* b_cur = malloc(sizeof(float) * cap);
  float* nb_cur = malloc(sizeof(float) * cap);
  uint8_t* pend = calloc(cap, 1); uint32_t* pend_from = malloc(4 * cap); /* previous_timesteps / new_timestep */
  int* c_parent = malloc(sizeof(int) * cap); uint32_t* c_ch = malloc(4 * cap); int* c_fst = malloc(sizeof(int) * cap);
  uint64_t* c_key = malloc(8 * cap); uint32_t* c_origin = malloc(4 * cap);
  int m = 0;
  for (int i = 0; i < n; ++i) { b_cur[i] = NEG_INF; nb_cur[i] = NEG_INF; c_key[i] = d->key[i]; c_origin[i] = (uint32_t)i; }
  int hcap = 1; while (hcap < 4 * cap) hcap <<= 1;
  HEnt* ht = malloc(sizeof(HEnt) * hcap);
  for (int i = 0; i < hcap; ++i) ht[i].slot = -1;
#define HT_FIND(K, OUT) do { uint64_t hh_ = (K) & (uint64_t)(hcap - 1); OUT = -1; \
    while (ht[hh_].slot >= 0) { if (ht[hh_].key == (K)) { OUT = ht[hh_].slot; break; } hh_ = (hh_ + 1) & (uint64_t)(hcap - 1); } } while (0)
#define HT_PUT(K, SLOT) do { uint64_t hh_ = (K) & (uint64_t)(hcap - 1); while (ht[hh_].slot >= 0) hh_ = (hh_ + 1) & (uint64_t)(hcap - 1); \
    ht[hh_].key = (K); ht[hh_].slot = (SLOT); } while (0)
  for (int i = 0; i < n; ++i) HT_PUT(d->key[i], i);

  for (int k = 0; k < cutoff_len; ++k) {
    const uint32_t c = (uint32_t)cls[k];
    const float log_prob_c = lp[k];
    for (int i = 0; i < n && i < beam; ++i) {
      if (full_beam && log_prob_c + d->score[i] < min_cutoff) break;
      if (d->score[i] == NEG_INF) continue;
      if ((int)c == d->blank) { /* :166-179 */
        float log_p = log_prob_c + d->score[i];
        if (nb_cur[i] < log_p) pend[i] = 0;
        b_cur[i] = gf_log_sum_exp(b_cur[i], log_p);
        continue;
      }
      if (c == d->ch[i]) { /* :182-193 */
        float log_p = log_prob_c + d->pnb[i];
        if (nb_cur[i] < log_p) pend[i] = 0;
        nb_cur[i] = gf_log_sum_exp(nb_cur[i], log_p);
      }
      /* get_path_trie, path_trie.cpp:37-100 */
      uint64_t ck = child_key(d->key[i], c);
      int slot; HT_FIND(ck, slot);
      if (slot < 0) {
        int child_fst = 0;
        if (d->sc) {
          int next;
          if (!fst_find(d->sc, d->fst[i], (int)c + 1, &next)) continue; /* nullptr: word outside dictionary */
          child_fst = fst_is_final(d->sc, next) ? (int)d->sc->fst_start : next;
        }
        slot = n + m++;
        b_cur[slot] = NEG_INF; nb_cur[slot] = NEG_INF; pend[slot] = 0;
        c_parent[slot] = i; c_ch[slot] = c; c_fst[slot] = child_fst; c_key[slot] = ck;
        c_origin[slot] = (uint32_t)(beam + i); /* unique among equal (score, character): one child per (parent, character) */
        HT_PUT(ck, slot);
        d->stat_candidates++;
      }
      float log_p = NEG_INF; /* :199-207 */
      if (c == d->ch[i] && d->pb[i] > NEG_INF) log_p = log_prob_c + d->pb[i];
      else if (c != d->ch[i]) log_p = log_prob_c + d->score[i];
      if (d->sc) { /* :209-243 */
        /* word mode scores the prefix *before* the space; utf8 mode scores the new prefix. */
        int boundary;
        if (d->sc->utf8) {
          /* is_scoring_boundary(prefix_new, c): walk starts at the new character */
          uint8_t fb = label_first_byte(d, c);
          if ((fb & 0xC0) != 0x80) {
            uint8_t first_byte = (uint8_t)((uint8_t)c + 1);
            int needed = ((first_byte >> 3) == 0x1E) ? 4 : ((first_byte >> 4) == 0x0E) ? 3 : ((first_byte >> 5) == 0x06) ? 2 : ((first_byte >> 7) == 0) ? 1 : -1;
            boundary = (needed == 1);
          } else {
            /* distance = 1 + distance_to_codepoint_boundary(parent) when parent is not root */
            boundary = 0;
            if (d->ch[i] != 0xFFFFFFFFu) {
              int dist = 1; uint8_t first_byte = 0; int found = 0; uint32_t cur = d->node[i];
              for (;;) {
                uint32_t cc = d->pa_ch[cur];
                if ((label_first_byte(d, cc) & 0xC0) != 0x80) { first_byte = (uint8_t)((uint8_t)cc + 1); dist += 1; found = 1; break; }
                uint32_t par = d->pa_parent[cur];
                if (par != 0xFFFFFFFFu && d->pa_ch[par] != 0xFFFFFFFFu) { dist += 1; cur = par; continue; }
                break;
              }
              if (found) {
                int needed = ((first_byte >> 3) == 0x1E) ? 4 : ((first_byte >> 4) == 0x0E) ? 3 : ((first_byte >> 5) == 0x06) ? 2 : ((first_byte >> 7) == 0) ? 1 : -1;
                boundary = (dist == needed);
              }
            }
          }
        } else {
          boundary = ((int)c == d->space);
        }
        if (boundary) {
          float score;
          if (d->sc->utf8) {
            /* make_ngram(prefix_new): materialise the new node temporarily */
            uint32_t tmp = pa_push(d, d->node[i], c);
            score = lm_score(d, tmp, 1);
            d->pa_n--; /* pop */
          } else {
            score = lm_score(d, d->node[i], 1);
          }
          log_p += score;
          log_p = (float)((double)log_p + d->sc->beta);
        }
      }
      if (nb_cur[slot] < log_p) { pend[slot] = 1; pend_from[slot] = d->ts[i]; } /* :246-251 */
      nb_cur[slot] = gf_log_sum_exp(nb_cur[slot], log_p);
    }
  }

  /* iterate_to_vec, path_trie.cpp:159-190 */
  const int total = n + m;
  float* nscore = malloc(sizeof(float) * total);
  for (int sidx = 0; sidx < total; ++sidx) nscore[sidx] = gf_log_sum_exp(b_cur[sidx], nb_cur[sidx]);

  /* top beam_size by prefix_compare (decoder_utils.cpp:66-76) + origin; result fully sorted */
  int* ord = malloc(sizeof(int) * total);
  for (int i = 0; i < total; ++i) ord[i] = i;
#define CH_OF(s_) ((s_) < n ? d->ch[(s_)] : c_ch[(s_)])
#define BEFORE(a_, b_) (nscore[a_] != nscore[b_] ? nscore[a_] > nscore[b_] : (CH_OF(a_) != CH_OF(b_) ? CH_OF(a_) < CH_OF(b_) : c_origin[a_] < c_origin[b_]))
  /* heap sort by BEFORE (total order => any algorithm gives the same permutation) */
  for (int start = total / 2 - 1; start >= 0; --start) {
    int root = start;
    for (;;) { int child = 2 * root + 1; if (child >= total) break;
      if (child + 1 < total && BEFORE(ord[child], ord[child + 1])) child++;
      if (BEFORE(ord[root], ord[child])) { int t = ord[root]; ord[root] = ord[child]; ord[child] = t; root = child; } else break; }
  }
  for (int end = total - 1; end > 0; --end) {
    int t = ord[0]; ord[0] = ord[end]; ord[end] = t;
    int root = 0;
    for (;;) { int child = 2 * root + 1; if (child >= end) break;
      if (child + 1 < end && BEFORE(ord[child], ord[child + 1])) child++;
      if (BEFORE(ord[root], ord[child])) { int t2 = ord[root]; ord[root] = ord[child]; ord[child] = t2; root = child; } else break; }
  }
  const int keep = total < beam ? total : beam;
  if (total > beam && nscore[ord[beam - 1]] == nscore[ord[beam]] && CH_OF(ord[beam - 1]) == CH_OF(ord[beam])) d->stat_boundary_ties++;

  /* write the new beam */
  float* o_score = malloc(4 * keep); float* o_pb = malloc(4 * keep); float* o_pnb = malloc(4 * keep);
  uint32_t* o_ch = malloc(4 * keep); uint32_t* o_node = malloc(4 * keep); uint32_t* o_ts = malloc(4 * keep);
  int* o_fst = malloc(sizeof(int) * keep); uint64_t* o_key = malloc(8 * keep);
  for (int r = 0; r < keep; ++r) {
    int sidx = ord[r];
    o_score[r] = nscore[sidx]; o_pb[r] = b_cur[sidx]; o_pnb[r] = nb_cur[sidx];
    uint32_t ts_old;
    if (sidx < n) { o_ch[r] = d->ch[sidx]; o_node[r] = d->node[sidx]; o_fst[r] = d->fst[sidx]; o_key[r] = d->key[sidx]; ts_old = d->ts[sidx]; }
    else { o_ch[r] = c_ch[sidx]; o_node[r] = pa_push(d, d->node[c_parent[sidx]], c_ch[sidx]); o_fst[r] = c_fst[sidx]; o_key[r] = c_key[sidx]; ts_old = 0xFFFFFFFFu; }
    o_ts[r] = pend[sidx] ? ta_push(d, pend_from[sidx], (uint32_t)d->abs_t) : ts_old;
  }
  memcpy(d->score, o_score, 4 * keep); memcpy(d->pb, o_pb, 4 * keep); memcpy(d->pnb, o_pnb, 4 * keep);
  memcpy(d->ch, o_ch, 4 * keep); memcpy(d->node, o_node, 4 * keep); memcpy(d->ts, o_ts, 4 * keep);
  memcpy(d->fst, o_fst, sizeof(int) * keep); memcpy(d->key, o_key, 8 * keep);
  d->n = keep;
  free(o_score); free(o_pb); free(o_pnb); free(o_ch); free(o_node); free(o_ts); free(o_fst); free(o_key);
  free(ord); free(nscore); free(ht); free(c_origin); free(c_key); free(c_fst); free(c_ch); free(c_parent);
  free(pend_from); free(pend); free(nb_cur); free(b_cur); free(lp); free(pf); free(cls);
  d->abs_t++;
}

void port_decoder_next(PortDecoder* d, const double* probs, int T, int C) {
  for (int t = 0; t < T; ++t) step(d, probs + (size_t)t * C);
}

/* DecoderState::decode, ctc_beam_search_decoder.cpp:278-326 */
int port_decoder_decode(PortDecoder* d, int num_results, uint32_t* tokens, uint32_t* timesteps, int* lens, double* confidences, int max_len) {
  const int n = d->n;
  float* scores = malloc(4 * n);
  int* ord = malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) { scores[i] = d->score[i]; ord[i] = i; }
  if (d->sc) {
    for (int i = 0; i < d->beam && i < n; ++i) {
      /* prefix_boundary = utf8 ? prefix : prefix->parent */
      uint32_t bnode; uint32_t bch;
      if (d->sc->utf8) { bnode = d->node[i]; bch = d->ch[i]; }
      else { bnode = d->pa_parent[d->node[i]]; if (bnode == 0xFFFFFFFFu) continue; bch = d->pa_ch[bnode]; }
      if (!is_scoring_boundary(d, bnode, bch, d->ch[i])) {
        float score = lm_score(d, d->node[i], 0); /* no hot-word boost here (:293-298) */
        score = (float)((double)score + d->sc->beta);
        scores[i] += score;
      }
    }
  }
  /* partial_sort by prefix_compare_external; ties by beam order */
  for (int i = 1; i < n; ++i) {
    int x = ord[i]; int j = i - 1;
    while (j >= 0) {
      int y = ord[j];
      int before = scores[x] != scores[y] ? scores[x] > scores[y] : (d->ch[x] != d->ch[y] ? d->ch[x] < d->ch[y] : 0);
      if (!before) break;
      ord[j + 1] = y; --j;
    }
    ord[j + 1] = x;
  }
  int nret = n < num_results ? n : num_results;
  for (int r = 0; r < nret; ++r) {
    int i = ord[r];
    int len = 0;
    for (uint32_t x = d->node[i]; x != 0xFFFFFFFFu && d->pa_ch[x] != 0xFFFFFFFFu; x = d->pa_parent[x]) len++;
    if (len > max_len) { free(scores); free(ord); return -1; }
    lens[r] = len;
    int j = len;
    for (uint32_t x = d->node[i]; x != 0xFFFFFFFFu && d->pa_ch[x] != 0xFFFFFFFFu; x = d->pa_parent[x]) tokens[(size_t)r * max_len + --j] = d->pa_ch[x];
    if (timesteps) {
      int tl = 0;
      for (uint32_t x = d->ts[i]; x != 0xFFFFFFFFu && x != 0; x = d->ta_parent[x]) tl++;
      j = tl < len ? tl : len;
      int skip = tl - j;
      for (uint32_t x = d->ts[i]; x != 0xFFFFFFFFu && x != 0; x = d->ta_parent[x]) { if (skip > 0) { --skip; continue; } timesteps[(size_t)r * max_len + --j] = d->ta_t[x]; }
    }
    confidences[r] = (double)scores[i];
  }
  free(scores); free(ord);
  return nret;
}

int port_decoder_beam(PortDecoder* d, float* score, float* pb, float* pnb, int* last_char, int* path_len, int cap) {
  int n = d->n < cap ? d->n : cap;
  for (int i = 0; i < n; ++i) {
    score[i] = d->score[i]; pb[i] = d->pb[i]; pnb[i] = d->pnb[i]; last_char[i] = (int)d->ch[i];
    int len = 0;
    for (uint32_t x = d->node[i]; x != 0xFFFFFFFFu && d->pa_ch[x] != 0xFFFFFFFFu; x = d->pa_parent[x]) len++;
    path_len[i] = len;
  }
  return n;
}
uint64_t port_decoder_boundary_ties(const PortDecoder* d) { return d->stat_boundary_ties; }
void port_decoder_stats(PortDecoder* d, uint64_t* out3) { out3[0] = d->stat_steps; out3[1] = d->stat_candidates; out3[2] = d->stat_lm_queries; }

/* test hooks for oracle/glibc_flt.h (tests/test_oracle_math.py) */
float port_expf(float x) { return gf_expf(x); }
float port_logf(float x) { return gf_logf(x); }
void port_expf_array(const float* in, float* out, long n) { for (long i = 0; i < n; ++i) out[i] = gf_expf(in[i]); }
void port_logf_array(const float* in, float* out, long n) { for (long i = 0; i < n; ++i) out[i] = gf_logf(in[i]); }

/* ============================================================================================
 * Part D. The same decoder in the REFERENCE'S OWN ORDER (round 5).
 *
 * Where two prefixes compare equal under prefix_compare (same float score, same last character) the reference's result depends on
 * orders the C++ standard leaves open: the order in which PathTrie::iterate_to_vec (path_trie.cpp:159-190) emits the trie -- children
 * before their parent, children in the order get_path_trie (path_trie.cpp:37-100) appended them, positions kept across remove()'s
 * erase (path_trie.cpp:192-209) --, what libstdc++'s std::nth_element (introselect) does to that sequence
 * (ctc_beam_search_decoder.cpp:264), and what std::partial_sort (heap select + sort_heap) does to the survivors at the start of the
 * next step (:138) and in decode() (:305).  The flat decoder above replaces all of that by one documented total order; this one keeps
 * the pointer trie and restates the three libstdc++ algorithms (GCC 11's bits/stl_algo.h / stl_heap.h, from their published
 * description: median-of-three to the front, unguarded Hoare partition, depth limit 2 * lg(n) with heap-select fallback, insertion
 * sort at <= 3 elements; make_heap / adjust_heap / push_heap with the hole technique), so that its output can be compared with the
 * compiled reference INCLUDING the tie cases: tests/test_oracle_port.py, benchmarks/oracle_fuzz_long.py --reference-order.
 * Everything else (scoring, dictionary, timesteps, arithmetic) is shared with the flat decoder.
 * ==========================================================================================*/
typedef struct TNode {
  struct TNode* parent;
  uint32_t ch;                 /* 0xFFFFFFFF = root */
  int exists;
  float b_prev, nb_prev, b_cur, nb_cur, score;
  int fst;
  uint32_t pa;                 /* path-arena node (tokens, n-gram reconstruction: shared helpers) */
  uint32_t ts;                 /* time-arena entry = `timesteps`; 0xFFFFFFFF = nullptr */
  int pend; uint32_t pend_from, new_t;   /* previous_timesteps / new_timestep (pend = previous_timesteps != nullptr) */
  struct TNode** kids; int nk, capk;
  float ext;                   /* decode(): scores[prefix] */
} TNode;

typedef struct {
  PortDecoder* d;              /* scorer, alphabet, arenas, parameters (its own beam stays at the root and is not used) */
  TNode* root;
  TNode** pre; int np, capp;   /* prefixes_ */
} PortTrieDecoder;

static TNode* tnode_new(TNode* parent, uint32_t ch) {
  TNode* t = (TNode*)calloc(1, sizeof(TNode));
  t->parent = parent; t->ch = ch; t->exists = 1;
  t->b_prev = t->nb_prev = t->b_cur = t->nb_cur = t->score = NEG_INF;
  t->ts = 0xFFFFFFFFu;
  return t;
}
static void tnode_free_rec(TNode* t) { for (int i = 0; i < t->nk; ++i) tnode_free_rec(t->kids[i]); free(t->kids); free(t); }

/* prefix_compare, decoder_utils.cpp:66-76 */
static inline int tcmp(const TNode* x, const TNode* y) {
  if (x->score == y->score) { if (x->ch == y->ch) return 0; return x->ch < y->ch; }   /* `character` is unsigned (path_trie.h:93): the root, ROOT_ = -1, sorts last */
  return x->score > y->score;
}
/* prefix_compare_external, decoder_utils.cpp:78-88 */
static inline int tcmp_ext(const TNode* x, const TNode* y) {
  if (x->ext == y->ext) { if (x->ch == y->ch) return 0; return x->ch < y->ch; }
  return x->ext > y->ext;
}
typedef int (*tcmp_fn)(const TNode*, const TNode*);

/* ---- libstdc++ heap primitives on TNode* sequences (stl_heap.h) */
static void ls_push_heap(TNode** first, long hole, long top, TNode* value, tcmp_fn comp) {
  long parent = (hole - 1) / 2;
  while (hole > top && comp(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
  first[hole] = value;
}
static void ls_adjust_heap(TNode** first, long hole, long len, TNode* value, tcmp_fn comp) {
  const long top = hole;
  long second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (comp(first[second], first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    first[hole] = first[second - 1];
    hole = second - 1;
  }
  ls_push_heap(first, hole, top, value, comp);
}
static void ls_make_heap(TNode** first, long len, tcmp_fn comp) {
  if (len < 2) return;
  long parent = (len - 2) / 2;
  for (;;) {
    TNode* v = first[parent];
    ls_adjust_heap(first, parent, len, v, comp);
    if (parent == 0) return;
    parent--;
  }
}
/* __pop_heap(first, last, result): the value at `result` goes through the heap [first, last), the old top lands at `result` */
static void ls_pop_heap(TNode** first, long last, TNode** result, tcmp_fn comp) {
  TNode* v = *result;
  *result = first[0];
  ls_adjust_heap(first, 0, last, v, comp);
}
static void ls_heap_select(TNode** first, long middle, long last, tcmp_fn comp) {
  ls_make_heap(first, middle, comp);
  for (long i = middle; i < last; ++i)
    if (comp(first[i], first[0])) ls_pop_heap(first, middle, &first[i], comp);
}
static void ls_sort_heap(TNode** first, long last, tcmp_fn comp) {
  while (last > 1) { --last; ls_pop_heap(first, last, &first[last], comp); }
}
static void ls_partial_sort(TNode** first, long middle, long last, tcmp_fn comp) {
  if (middle == 0) return;                       /* (std::partial_sort with first == middle does nothing) */
  ls_heap_select(first, middle, last, comp);
  ls_sort_heap(first, middle, comp);
}
/* ---- std::nth_element (stl_algo.h: __introselect) */
static inline void ls_swap(TNode** a, TNode** b) { TNode* t = *a; *a = *b; *b = t; }
static void ls_move_median_to_first(TNode** result, TNode** a, TNode** b, TNode** c, tcmp_fn comp) {
  if (comp(*a, *b)) {
    if (comp(*b, *c)) ls_swap(result, b);
    else if (comp(*a, *c)) ls_swap(result, c);
    else ls_swap(result, a);
  } else if (comp(*a, *c)) ls_swap(result, a);
  else if (comp(*b, *c)) ls_swap(result, c);
  else ls_swap(result, b);
}
static TNode** ls_unguarded_partition(TNode** first, TNode** last, TNode** pivot, tcmp_fn comp) {
  for (;;) {
    while (comp(*first, *pivot)) ++first;
    --last;
    while (comp(*pivot, *last)) --last;
    if (!(first < last)) return first;
    ls_swap(first, last);
    ++first;
  }
}
static void ls_insertion_sort(TNode** first, TNode** last, tcmp_fn comp) {
  if (first == last) return;
  for (TNode** i = first + 1; i != last; ++i) {
    if (comp(*i, *first)) {
      TNode* v = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(TNode*));
      *first = v;
    } else {
      TNode* v = *i; TNode** lastp = i; TNode** next = i - 1;
      while (comp(v, *next)) { *lastp = *next; lastp = next; --next; }
      *lastp = v;
    }
  }
}
static int ls_lg(long n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }
static void ls_nth_element(TNode** base, long nth, long n, tcmp_fn comp) {
  TNode** first = base; TNode** last = base + n; TNode** nthp = base + nth;
  if (first == last || nthp == last) return;
  long depth = 2L * ls_lg(n);
  while (last - first > 3) {
    if (depth == 0) {
      ls_heap_select(first, (nthp + 1) - first, last - first, comp);
      ls_swap(first, nthp);
      return;
    }
    --depth;
    TNode** mid = first + (last - first) / 2;
    ls_move_median_to_first(first, first + 1, mid, last - 1, comp);
    TNode** cut = ls_unguarded_partition(first + 1, last, first, comp);
    if (cut <= nthp) first = cut; else last = cut;
  }
  ls_insertion_sort(first, last, comp);
}
/* ---- std::sort on (class, probability) pairs by probability descending (get_pruned_emissions, :338-339): introsort + final insertion sort */
typedef struct { int cls; float p; } ClsP;
static inline int cp_before(const ClsP* a, const ClsP* b) { return a->p > b->p; }
static void cp_swap(ClsP* a, ClsP* b) { ClsP t = *a; *a = *b; *b = t; }
static void cp_adjust_heap(ClsP* first, long hole, long len, ClsP value) {
  const long top = hole; long second = hole;
  while (second < (len - 1) / 2) { second = 2 * (second + 1); if (cp_before(&first[second], &first[second - 1])) second--; first[hole] = first[second]; hole = second; }
  if ((len & 1) == 0 && second == (len - 2) / 2) { second = 2 * (second + 1); first[hole] = first[second - 1]; hole = second - 1; }
  long parent = (hole - 1) / 2;
  while (hole > top && cp_before(&first[parent], &value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
  first[hole] = value;
}
static void cp_heap_sort_all(ClsP* first, long len) {      /* __partial_sort(first, last, last) = make_heap + sort_heap over the whole range */
  if (len >= 2) for (long parent = (len - 2) / 2;; --parent) { cp_adjust_heap(first, parent, len, first[parent]); if (parent == 0) break; }
  for (long last = len; last > 1;) { --last; ClsP v = first[last]; first[last] = first[0]; cp_adjust_heap(first, 0, last, v); }
}
static void cp_introsort_loop(ClsP* first, ClsP* last, long depth) {
  while (last - first > 16) {
    if (depth == 0) { cp_heap_sort_all(first, last - first); return; }
    --depth;
    ClsP* mid = first + (last - first) / 2;
    ClsP *a = first + 1, *b = mid, *c = last - 1;
    if (cp_before(a, b)) { if (cp_before(b, c)) cp_swap(first, b); else if (cp_before(a, c)) cp_swap(first, c); else cp_swap(first, a); }
    else if (cp_before(a, c)) cp_swap(first, a); else if (cp_before(b, c)) cp_swap(first, c); else cp_swap(first, b);
    ClsP *lo = first + 1, *hi = last;
    for (;;) { while (cp_before(lo, first)) ++lo; --hi; while (cp_before(first, hi)) --hi; if (!(lo < hi)) break; cp_swap(lo, hi); ++lo; }
    cp_introsort_loop(lo, last, depth);
    last = lo;
  }
}
static void cp_insertion(ClsP* first, ClsP* last, int guarded) {
  for (ClsP* i = first + (guarded ? 1 : 0); i < last; ++i) {
    if (guarded && cp_before(i, first)) { ClsP v = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(ClsP)); *first = v; }
    else { ClsP v = *i; ClsP* lastp = i; ClsP* next = i - 1; while (cp_before(&v, next)) { *lastp = *next; lastp = next; --next; } *lastp = v; }
  }
}
static void cp_std_sort(ClsP* first, long n) {
  if (n < 2) return;
  cp_introsort_loop(first, first + n, 2L * ls_lg(n));
  if (n > 16) { cp_insertion(first, first + 16, 1); cp_insertion(first + 16, first + n, 0); }     /* __final_insertion_sort: guarded head, unguarded tail */
  else cp_insertion(first, first + n, 1);
}

PortTrieDecoder* port_tdecoder_new(int C, int space_id, int beam, double cutoff_prob, int cutoff_top_n, PortScorer* sc,
                                   const uint8_t* label_bytes, const int* label_off, const char** hot_words, const float* boosts, int n_hot) {
  PortTrieDecoder* t = (PortTrieDecoder*)calloc(1, sizeof(PortTrieDecoder));
  t->d = port_decoder_new(C, space_id, beam, cutoff_prob, cutoff_top_n, sc, label_bytes, label_off, hot_words, boosts, n_hot);
  t->root = tnode_new(NULL, 0xFFFFFFFFu);
  t->root->score = 0.0f; t->root->b_prev = 0.0f;            /* ctc_beam_search_decoder.cpp:43-46 */
  t->root->pa = 0; t->root->ts = 0;                         /* arena node 0 = root, time entry 0 = timestep_tree_root_ */
  t->root->fst = sc ? (int)sc->fst_start : 0;
  t->capp = 4 * beam + 64; t->pre = (TNode**)malloc(sizeof(TNode*) * t->capp);
  t->pre[0] = t->root; t->np = 1;
  return t;
}
void port_tdecoder_free(PortTrieDecoder* t) {
  if (!t) return;
  tnode_free_rec(t->root); free(t->pre); port_decoder_free(t->d); free(t);
}

/* PathTrie::remove, path_trie.cpp:192-209 */
static void tnode_remove(TNode* x) {
  x->exists = 0;
  if (x->nk == 0) {
    TNode* p = x->parent;
    for (int i = 0; i < p->nk; ++i)
      if (p->kids[i]->ch == x->ch) { memmove(&p->kids[i], &p->kids[i + 1], (size_t)(p->nk - i - 1) * sizeof(TNode*)); p->nk--; break; }
    if (p->nk == 0 && !p->exists) tnode_remove(p);
    free(x->kids); free(x);
  }
}
/* PathTrie::iterate_to_vec, path_trie.cpp:159-190 */
static void tnode_iterate(PortTrieDecoder* t, TNode* x) {
  for (int i = 0; i < x->nk; ++i) tnode_iterate(t, x->kids[i]);
  if (x->exists) {
    x->b_prev = x->b_cur; x->nb_prev = x->nb_cur;
    x->b_cur = NEG_INF; x->nb_cur = NEG_INF;
    x->score = gf_log_sum_exp(x->b_prev, x->nb_prev);
    if (x->pend) x->ts = ta_push(t->d, x->pend_from, x->new_t);     /* (the reference reuses an equal child of the time tree: same history either way) */
    x->pend = 0;
    if (t->np == t->capp) { t->capp *= 2; t->pre = (TNode**)realloc(t->pre, sizeof(TNode*) * t->capp); }
    t->pre[t->np++] = x;
  }
}

static int utf8_boundary_new(const PortDecoder* d, const TNode* prefix, uint32_t c) {   /* is_scoring_boundary(prefix_new, c), utf8 mode (as in step()) */
  uint8_t fb = label_first_byte(d, c);
  if ((fb & 0xC0) != 0x80) {
    uint8_t first_byte = (uint8_t)((uint8_t)c + 1);
    int needed = ((first_byte >> 3) == 0x1E) ? 4 : ((first_byte >> 4) == 0x0E) ? 3 : ((first_byte >> 5) == 0x06) ? 2 : ((first_byte >> 7) == 0) ? 1 : -1;
    return needed == 1;
  }
  if (prefix->ch == 0xFFFFFFFFu) return 0;
  int dist = 1; uint8_t first_byte = 0; int found = 0; uint32_t cur = prefix->pa;
  for (;;) {
    uint32_t cc = d->pa_ch[cur];
    if ((label_first_byte(d, cc) & 0xC0) != 0x80) { first_byte = (uint8_t)((uint8_t)cc + 1); dist += 1; found = 1; break; }
    uint32_t par = d->pa_parent[cur];
    if (par != 0xFFFFFFFFu && d->pa_ch[par] != 0xFFFFFFFFu) { dist += 1; cur = par; continue; }
    break;
  }
  if (!found) return 0;
  int needed = ((first_byte >> 3) == 0x1E) ? 4 : ((first_byte >> 4) == 0x0E) ? 3 : ((first_byte >> 5) == 0x06) ? 2 : ((first_byte >> 7) == 0) ? 1 : -1;
  return dist == needed;
}

/* One timestep of DecoderState::next, ctc_beam_search_decoder.cpp:118-275, on the trie */
static void tstep(PortTrieDecoder* t, const double* prob) {
  PortDecoder* d = t->d;
  const int C = d->C; const long beam = d->beam;
  if (prob[d->blank] < 0.999) d->start_expanding = 1;
  if (!d->start_expanding) { d->abs_t++; return; }
  float min_cutoff = NEG_INF; int full_beam = 0;
  if (d->sc) {
    long num = t->np < beam ? t->np : beam;
    ls_partial_sort(t->pre, num, t->np, tcmp);
    min_cutoff = (float)((double)t->pre[num - 1]->score + log(prob[d->blank]) - fmax(0.0, d->sc->beta));
    full_beam = (num == beam);
  }
  ClsP* pc = (ClsP*)malloc(sizeof(ClsP) * C);
  for (int i = 0; i < C; ++i) { pc[i].cls = i; pc[i].p = (float)prob[i]; }
  int cutoff_len = C;
  if (d->cutoff_prob < 1.0 || d->cutoff_top_n < cutoff_len) {
    cp_std_sort(pc, C);
    if (d->cutoff_prob < 1.0) {
      double cum = 0.0; cutoff_len = 0;
      for (int i = 0; i < C; ++i) { cum += pc[i].p; cutoff_len += 1; if (cum >= d->cutoff_prob || cutoff_len >= d->cutoff_top_n) break; }
    }
  }
  for (int k = 0; k < cutoff_len; ++k) {
    const uint32_t c = (uint32_t)pc[k].cls;
    const float log_prob_c = gf_logf(pc[k].p + FLT_MIN);
    for (long i = 0; i < t->np && i < beam; ++i) {
      TNode* prefix = t->pre[i];
      if (full_beam && log_prob_c + prefix->score < min_cutoff) break;
      if (prefix->score == NEG_INF) continue;
      if ((int)c == d->blank) {
        float log_p = log_prob_c + prefix->score;
        if (prefix->nb_cur < log_p) prefix->pend = 0;
        prefix->b_cur = gf_log_sum_exp(prefix->b_cur, log_p);
        continue;
      }
      if (c == prefix->ch) {
        float log_p = log_prob_c + prefix->nb_prev;
        if (prefix->nb_cur < log_p) prefix->pend = 0;
        prefix->nb_cur = gf_log_sum_exp(prefix->nb_cur, log_p);
      }
      /* get_path_trie, path_trie.cpp:37-100 */
      TNode* pn = NULL;
      for (int q = 0; q < prefix->nk; ++q) if (prefix->kids[q]->ch == c) { pn = prefix->kids[q]; break; }
      if (pn) {
        if (!pn->exists) { pn->exists = 1; pn->b_prev = pn->nb_prev = pn->b_cur = pn->nb_cur = NEG_INF; }
      } else {
        int child_fst = 0;
        if (d->sc) {
          int next;
          if (!fst_find(d->sc, prefix->fst, (int)c + 1, &next)) continue;
          child_fst = fst_is_final(d->sc, next) ? (int)d->sc->fst_start : next;
        }
        pn = tnode_new(prefix, c);
        pn->fst = child_fst;
        pn->pa = pa_push(d, prefix->pa, c);
        if (prefix->nk == prefix->capk) { prefix->capk = prefix->capk ? 2 * prefix->capk : 4; prefix->kids = (TNode**)realloc(prefix->kids, sizeof(TNode*) * prefix->capk); }
        prefix->kids[prefix->nk++] = pn;
        d->stat_candidates++;
      }
      float log_p = NEG_INF;
      if (c == prefix->ch && prefix->b_prev > NEG_INF) log_p = log_prob_c + prefix->b_prev;
      else if (c != prefix->ch) log_p = log_prob_c + prefix->score;
      if (d->sc) {
        const int boundary = d->sc->utf8 ? utf8_boundary_new(d, prefix, c) : ((int)c == d->space);
        if (boundary) {
          const float score = lm_score(d, d->sc->utf8 ? pn->pa : prefix->pa, 1);
          log_p += score;
          log_p = (float)((double)log_p + d->sc->beta);
        }
      }
      if (pn->nb_cur < log_p) { pn->pend = 1; pn->pend_from = prefix->ts; pn->new_t = (uint32_t)d->abs_t; }
      pn->nb_cur = gf_log_sum_exp(pn->nb_cur, log_p);
    }
  }
  free(pc);
  t->np = 0;
  tnode_iterate(t, t->root);
  if (t->np > beam) {
    ls_nth_element(t->pre, beam, t->np, tcmp);
    for (long i = beam; i < t->np; ++i) tnode_remove(t->pre[i]);
    t->np = (int)beam;
  }
  d->stat_steps++;
  d->abs_t++;
}
void port_tdecoder_next(PortTrieDecoder* t, const double* probs, int T, int C) {
  for (int s = 0; s < T; ++s) tstep(t, probs + (size_t)s * C);
}
/* DecoderState::decode, ctc_beam_search_decoder.cpp:278-326 */
int port_tdecoder_decode(PortTrieDecoder* t, int num_results, uint32_t* tokens, uint32_t* timesteps, int* lens, double* confidences, int max_len) {
  PortDecoder* d = t->d;
  const int n = t->np;
  TNode** cp = (TNode**)malloc(sizeof(TNode*) * (n ? n : 1));
  memcpy(cp, t->pre, sizeof(TNode*) * n);
  for (int i = 0; i < n; ++i) cp[i]->ext = cp[i]->score;
  if (d->sc) {
    for (int i = 0; i < d->beam && i < n; ++i) {
      TNode* p = cp[i];
      uint32_t bnode, bch;
      if (d->sc->utf8) { bnode = p->pa; bch = p->ch; }
      else { if (!p->parent) continue; bnode = p->parent->pa; bch = p->parent->ch; }
      if (!is_scoring_boundary(d, bnode, bch, p->ch)) {
        float score = lm_score(d, p->pa, 0);
        score = (float)((double)score + d->sc->beta);
        p->ext += score;
      }
    }
  }
  int nret = n < num_results ? n : num_results;
  ls_partial_sort(cp, nret, n, tcmp_ext);
  for (int r = 0; r < nret; ++r) {
    TNode* p = cp[r];
    int len = 0;
    for (uint32_t x = p->pa; x != 0xFFFFFFFFu && d->pa_ch[x] != 0xFFFFFFFFu; x = d->pa_parent[x]) len++;
    if (len > max_len) { free(cp); return -1; }
    lens[r] = len;
    int j = len;
    for (uint32_t x = p->pa; x != 0xFFFFFFFFu && d->pa_ch[x] != 0xFFFFFFFFu; x = d->pa_parent[x]) tokens[(size_t)r * max_len + --j] = d->pa_ch[x];
    if (timesteps) {
      int tl = 0;
      for (uint32_t x = p->ts; x != 0xFFFFFFFFu && x != 0; x = d->ta_parent[x]) tl++;
      j = tl < len ? tl : len;
      int skip = tl - j;
      for (uint32_t x = p->ts; x != 0xFFFFFFFFu && x != 0; x = d->ta_parent[x]) { if (skip > 0) { --skip; continue; } timesteps[(size_t)r * max_len + --j] = d->ta_t[x]; }
    }
    confidences[r] = (double)p->ext;
  }
  free(cp);
  return nret;
}
/* prefixes_ in the order the vector holds them (test hook: step-by-step comparison with the reference's vector) */
int port_tdecoder_beam(PortTrieDecoder* t, float* score, float* pb, float* pnb, int* last_char, int* path_len, int cap) {
  int n = t->np < cap ? t->np : cap;
  for (int i = 0; i < n; ++i) {
    const TNode* x = t->pre[i];
    score[i] = x->score; pb[i] = x->b_prev; pnb[i] = x->nb_prev; last_char[i] = (int)x->ch;
    int len = 0;
    for (const TNode* y = x; y->parent; y = y->parent) len++;
    path_len[i] = len;
  }
  return n;
}
