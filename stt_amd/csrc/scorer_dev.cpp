// stt_amd/csrc/scorer_dev.cpp -- .scorer package loader: KenLM trie binary + 'TRIE' header + ConstFst.
//
// Replaces Scorer::init_from_filepath / load_lm_* / load_trie_impl (native_client/ctcdecode/scorer.cpp:40-222),
// KenLM's binary-format reader for the trie family (kenlm/lm/binary_format.cc:22-75,193-239;
// lm/search_trie.cc:546-571; lm/quantize.cc:54-73; lm/bhiksha.cc:35-84; lm/vocab.cc:113-124,218-232)
// and ConstFst::Read (openfst-1.6.7/src/include/fst/const-fst.h:195-235, src/lib/fst.cc:57-84).
// The KenLM blob goes to HBM byte for byte; kernels read the same bit-packed records KenLM mmaps.
// Probing-hash models (model types 0/1) are rejected with STT_ERR_SCORER_INVALID_LM.
//
// Two stages: parse_scorer() is host-only (layout, dictionary repacking, vocabulary table, the hashed n-gram index of
// lmindex.h) and testable without a GPU; ScorerDev::Upload() moves the result to HBM.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include "../../include/coqui-stt.h"
#include "engine.h"
#include "tuning.h"
#include "lmindex.h"
#include "scorer_host.h"

namespace {
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline float rdf(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }
inline float bits_f(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
uint8_t required_bits(uint64_t v) { if (!v) return 0; uint8_t r = 1; while (v >>= 1) ++r; return r; }  // util/bit_packing.cc:17-22
uint64_t align8(uint64_t x) { return (x + 7) & ~(uint64_t)7; }
uint8_t chop_bits(uint64_t max_offset, uint64_t max_next, uint8_t cfg_bits) {  // lm/bhiksha.cc:35-50
  const uint8_t required = required_bits(max_next);
  uint8_t best = 0;
  int64_t lowest = INT64_MAX;
  const uint8_t lim = required < cfg_bits ? required : cfg_bits;
  for (uint8_t chop = 0; chop <= lim; ++chop) {
    const int64_t change = (int64_t)((max_next >> (required - chop)) * 64) - (int64_t)max_offset * (int64_t)chop;
    if (change < lowest) { lowest = change; best = chop; }
  }
  return best;
}
uint64_t murmur64a(const void* key, size_t len, uint64_t seed) {  // util/murmur_hash.cc
  const uint64_t m = 0xc6a4a7935bd1e995ULL; const int r = 47;
  uint64_t h = seed ^ (len * m);
  const uint8_t* d = (const uint8_t*)key; const uint8_t* end = d + (len / 8) * 8;
  while (d != end) { uint64_t k = rd64(d); d += 8; k *= m; k ^= k >> r; k *= m; h ^= k; h *= m; }
  uint64_t t = 0; const size_t rem = len & 7;
  for (size_t i = 0; i < rem; ++i) t |= (uint64_t)d[i] << (8 * i);
  if (rem) { h ^= t; h *= m; }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}
inline uint64_t rd57(const uint8_t* base, uint64_t bit_off, uint64_t mask) { return (rd64(base + (bit_off >> 3)) >> (bit_off & 7)) & mask; }

template <class F> void parallel_for(uint64_t n, F&& f) {  // f(begin, end) on contiguous chunks
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 32) nt = 32;
  if (nt < 1 || n < (1u << 16)) nt = 1;
  if (nt == 1) { f((uint64_t)0, n); return; }
  std::vector<std::thread> th;
  const uint64_t per = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const uint64_t b = std::min(n, per * t), e = std::min(n, per * (t + 1));
    if (b < e) th.emplace_back([=, &f]() { f(b, e); });
  }
  for (auto& x : th) x.join();
}
}  // namespace

uint64_t stt_murmur64a(const void* key, size_t len) { return murmur64a(key, len, 0); }

// ------------------------------------------------------------------------------------------- trie records on the host
// (the same decoding as lookup_middle / lookup_longest / read_next in ctc.hip, over a whole level)
float HostScorer::middle_prob(int om2, uint64_t at) const {
  const HostBitPacked& m = mid[om2];
  const uint64_t addr = at * m.total_bits + m.word_bits;
  const uint8_t* base = buf + m.base_off;
  if (quant) { const uint64_t pa = addr + backoff_bits; return rdf(buf + qprob_off[om2] + 4 * (uint64_t)((rd32(base + (pa >> 3)) >> (pa & 7)) & ((1u << prob_bits) - 1))); }
  return bits_f((uint32_t)(rd64(base + (addr >> 3)) >> (addr & 7)) | 0x80000000u);
}
float HostScorer::middle_backoff(int om2, uint64_t at) const {
  const HostBitPacked& m = mid[om2];
  const uint64_t addr = at * m.total_bits + m.word_bits;
  const uint8_t* base = buf + m.base_off;
  if (quant) return rdf(buf + qback_off[om2] + 4 * (uint64_t)((rd32(base + (addr >> 3)) >> (addr & 7)) & ((1u << backoff_bits) - 1)));
  const uint64_t ba = addr + 31;
  return bits_f((uint32_t)(rd64(base + (ba >> 3)) >> (ba & 7)));
}
float HostScorer::longest_prob(uint64_t at) const {
  const uint64_t addr = at * lon.total_bits + lon.word_bits;
  const uint8_t* base = buf + lon.base_off;
  if (quant) return rdf(buf + qprob_off[order - 2] + 4 * (uint64_t)((rd32(base + (addr >> 3)) >> (addr & 7)) & ((1u << prob_bits) - 1)));
  return bits_f((uint32_t)(rd64(base + (addr >> 3)) >> (addr & 7)) | 0x80000000u);
}

// Hashed n-gram index (lmindex.h).  Level by level: `next[c]` of every record of the level (its children start there),
// then every parent hands its key to its children, which enter the table and remember their slot for their own children.
static bool build_lm_index(HostScorer& hs) {
  const int ord = hs.order;
  uint64_t total = 0;
  for (int k = 1; k < ord; ++k) total += hs.counts[k];
  if (hs.counts[0] >= (1ull << 27) || total >= (1ull << 31) || ord < 2) return false;
  for (int k = 1; k < ord; ++k) if (hs.counts[k] >= 0xFFFFFFF0ull) return false;
  // about 1.33 entries per 4-slot bucket: 4.6 % of the buckets are full, 1 % overflow into the next one
  uint64_t nb64 = (total * 3 + 3) / 4 + 16;
  if (nb64 * LMI_BUCKET >= 0xFFFFFFF0ull) return false;
  // byte cap (tunable lm_index_mb): the table costs 48 B per n-gram on the host and again in HBM, on every replica of a fleet;
  // a model beyond the cap keeps the trie walk, which reads the blob the reference mmaps and nothing else
  if (nb64 * LMI_BUCKET * sizeof(LmiEntry) > (uint64_t)std::max(0, tune().lm_index_mb) * (1ull << 20)) return false;
  const uint32_t nb = (uint32_t)nb64;
  std::vector<LmiEntry>& tab = hs.lmi;
  tab.assign((size_t)nb * LMI_BUCKET, LmiEntry{LMI_EMPTY, 0u, 0.0f, 0.0f});
  hs.lmi_buckets = nb;
  const uint8_t* buf = hs.buf;
  // level 1: unigrams.  key seed = the word's vocabulary hash (what a query has before it knows the index), id = index
  const uint64_t n1 = hs.counts[0];
  std::vector<uint64_t> key_prev(n1), key_cur;
  std::vector<uint32_t> id_prev(n1), id_cur;
  std::vector<uint64_t> next_prev(n1 + 1), next_cur;
  for (uint64_t w = 0; w < n1; ++w) {
    key_prev[w] = (w == 0 || w - 1 >= hs.vocab_n) ? LMI_UNK_H : rd64(buf + hs.vocab_off + 8 * (w - 1));
    id_prev[w] = (uint32_t)w;
  }
  for (uint64_t w = 0; w <= n1; ++w) next_prev[w] = rd64(buf + hs.unigram_off + 16 * w + 8);
  for (int level = 2; level <= ord; ++level) {
    const bool is_longest = level == ord;
    const int om2 = level - 2;
    const HostBitPacked& bp = is_longest ? hs.lon : hs.mid[om2];
    const uint64_t cnt = hs.counts[level - 1];
    const uint8_t* base = buf + bp.base_off;
    const uint64_t word_mask = (1ULL << bp.word_bits) - 1, next_mask = bp.next_bits >= 64 ? ~0ULL : (1ULL << bp.next_bits) - 1;
    if (next_prev.back() != cnt) return false;  // the sentinel of the parent level must close the child level exactly
    if (!is_longest) {  // child ranges of this level's records (ArrayBhiksha::ReadNext, lm/bhiksha.hh:76-95)
      next_cur.assign(cnt + 1, 0);
      const uint64_t* offs = bp.off_begin_off ? reinterpret_cast<const uint64_t*>(buf + bp.off_begin_off) : nullptr;
      const uint32_t ocnt = bp.off_count;
      parallel_for(cnt + 1, [&](uint64_t b, uint64_t e) {
        uint64_t bi = 0;
        if (offs) bi = (uint64_t)(std::upper_bound(offs, offs + ocnt, b) - offs) - 1;
        for (uint64_t c = b; c < e; ++c) {
          const uint64_t lo = rd57(base, c * bp.total_bits + bp.word_bits + bp.quant_bits, next_mask);
          if (offs) { while (bi + 1 < ocnt && offs[bi + 1] <= c) ++bi; next_cur[c] = (bi << bp.next_bits) | lo; }
          else next_cur[c] = lo;
        }
      });
    }
    key_cur.assign(cnt, 0); id_cur.assign(cnt, 0);
    const uint64_t n_par = key_prev.size();
    bool bad = false;
    parallel_for(n_par, [&](uint64_t pb, uint64_t pe) {
      for (uint64_t p = pb; p < pe; ++p) {
        const uint64_t cb = next_prev[p], ce = next_prev[p + 1];
        if (cb > ce || ce > cnt) { bad = true; return; }
        for (uint64_t c = cb; c < ce; ++c) {
          const uint32_t word = (uint32_t)rd57(base, c * bp.total_bits, word_mask);
          const uint64_t key = lmi_step(key_prev[p], word);
          LmiEntry e;
          if (is_longest) { e.prob = hs.longest_prob(c); e.backoff = 0.0f; e.wl = lmi_wl(word, level, false); }
          else { e.prob = hs.middle_prob(om2, c); e.backoff = hs.middle_backoff(om2, c); e.wl = lmi_wl(word, level, next_cur[c] == next_cur[c + 1]); }
          e.parent = id_prev[p];
          uint32_t b = lmi_bucket(key, nb), slot = LMI_NOT_FOUND;
          for (uint32_t tries = 0; tries < nb && slot == LMI_NOT_FOUND; ++tries) {
            for (int j = 0; j < LMI_BUCKET; ++j) {
              LmiEntry* t = &tab[(size_t)b * LMI_BUCKET + j];
              uint32_t expect = LMI_EMPTY;
              if (__atomic_load_n(&t->wl, __ATOMIC_RELAXED) == LMI_EMPTY &&
                  __atomic_compare_exchange_n(&t->wl, &expect, e.wl, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                t->parent = e.parent; t->prob = e.prob; t->backoff = e.backoff;
                slot = b * LMI_BUCKET + (uint32_t)j;
                break;
              }
            }
            b = b + 1 == nb ? 0u : b + 1;
          }
          key_cur[c] = key; id_cur[c] = slot;
        }
      }
    });
    if (bad) return false;
    key_prev.swap(key_cur); id_prev.swap(id_cur); next_prev.swap(next_cur);
  }
  // <unk> (word index 0): its unigram record; it only matters as the *new* word of a query when it has children
  hs.unk_prob = rdf(buf + hs.unigram_off); hs.unk_backoff = rdf(buf + hs.unigram_off + 4);
  hs.unk_indep = rd64(buf + hs.unigram_off + 8) == rd64(buf + hs.unigram_off + 24);
  return true;
}

// Code-point scorers: how good can the language-model score of a candidate that completes code point u possibly be?
// GenericModel::FullScore (lm/model.cc:170-176) returns the probability of the LONGEST stored n-gram that ends with u -- an entry of
// u's sub-trie in the reverse trie -- plus the backoffs of the context n-grams that were too long to match (:312-338).  So
//     log10 P(u | any history) <= max over u's sub-trie of prob  +  sum over orders k < order of max(0, largest backoff of order k)
// whatever the history is.  The search kernel compares this bound with a score that 'beam' other prefixes are already known to reach
// and skips the FullScore (three to four dependent reads) of every candidate that cannot make it (ctc.hip, P3 of the code-point step).
// The table is indexed by code point (one read, no vocabulary probe); a code point the model does not know carries OOV_SCORE.
static void build_unit_bounds(HostScorer& hs) {
  const int ord = hs.order;
  const uint64_t n1 = hs.counts[0];
  const uint8_t* buf = hs.buf;
  std::vector<float> ub(n1);
  float bplus = 0.0f;
  { float mb = 0.0f; for (uint64_t w = 0; w < n1; ++w) { ub[w] = rdf(buf + hs.unigram_off + 16 * w); mb = std::max(mb, rdf(buf + hs.unigram_off + 16 * w + 4)); } bplus += mb; }
  std::vector<uint32_t> root_prev(n1), root_cur;
  for (uint64_t w = 0; w < n1; ++w) root_prev[w] = (uint32_t)w;
  std::vector<uint64_t> next_prev(n1 + 1), next_cur;
  for (uint64_t w = 0; w <= n1; ++w) next_prev[w] = rd64(buf + hs.unigram_off + 16 * w + 8);
  for (int level = 2; level <= ord; ++level) {
    const bool is_longest = level == ord;
    const int om2 = level - 2;
    const HostBitPacked& bp = is_longest ? hs.lon : hs.mid[om2];
    const uint64_t cnt = hs.counts[level - 1];
    const uint8_t* base = buf + bp.base_off;
    const uint64_t next_mask = bp.next_bits >= 64 ? ~0ULL : (1ULL << bp.next_bits) - 1;
    if (next_prev.back() != cnt) { hs.cp_ub.clear(); return; }
    if (!is_longest) {
      next_cur.assign(cnt + 1, 0);
      const uint64_t* offs = bp.off_begin_off ? reinterpret_cast<const uint64_t*>(buf + bp.off_begin_off) : nullptr;
      const uint32_t ocnt = bp.off_count;
      uint64_t bi = 0;
      for (uint64_t c = 0; c <= cnt; ++c) {
        const uint64_t lo = rd57(base, c * bp.total_bits + bp.word_bits + bp.quant_bits, next_mask);
        if (offs) { while (bi + 1 < ocnt && offs[bi + 1] <= c) ++bi; next_cur[c] = (bi << bp.next_bits) | lo; }
        else next_cur[c] = lo;
      }
    }
    root_cur.assign(cnt, 0);
    float mb = 0.0f;
    for (uint64_t pnt = 0; pnt < root_prev.size(); ++pnt) {
      const uint64_t cb = next_prev[pnt], ce = next_prev[pnt + 1];
      if (cb > ce || ce > cnt) { hs.cp_ub.clear(); return; }
      for (uint64_t c = cb; c < ce; ++c) {
        root_cur[c] = root_prev[pnt];
        const float pr = is_longest ? hs.longest_prob(c) : hs.middle_prob(om2, c);
        float& u = ub[root_prev[pnt]];
        if (pr > u) u = pr;
        if (!is_longest) mb = std::max(mb, hs.middle_backoff(om2, c));
      }
    }
    bplus += mb;
    root_prev.swap(root_cur); next_prev.swap(next_cur);
  }
  // KenLM adds the probability and the backoffs in FLOAT, one rounding per term; with positive backoffs a real score could end up an ulp or
  // two above a bound that was summed in higher precision and rounded up once.  One more float step up per backoff term covers every such
  // rounding (with all backoffs <= 0 -- the common case -- bplus is 0 and float rounding can only lower the real sum: nothing is added).
  const int slack = bplus > 0.0f ? ord : 0;
  auto up = [slack](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); for (int k = 0; k < slack; ++k) f = std::nextafterf(f, INFINITY); return f; };
  hs.cp_ub.assign(65536, -1000.0f);   // OOV_SCORE (scorer.h:16)
  float mx = -1000.0f;                // (over EVERY unit of the vocabulary, also those beyond U+FFFF that have no table entry)
  for (uint64_t w = 0; w < n1; ++w) mx = std::max(mx, up(((double)ub[w] + (double)bplus) / (double)0.4342944819f));
  for (uint32_t cp = 1; cp < 65536; ++cp) {
    unsigned char u[3]; size_t n;
    if (cp < 0x80) { u[0] = (unsigned char)cp; n = 1; }
    else if (cp < 0x800) { u[0] = (unsigned char)(0xC0 | (cp >> 6)); u[1] = (unsigned char)(0x80 | (cp & 0x3F)); n = 2; }
    else { u[0] = (unsigned char)(0xE0 | (cp >> 12)); u[1] = (unsigned char)(0x80 | ((cp >> 6) & 0x3F)); u[2] = (unsigned char)(0x80 | (cp & 0x3F)); n = 3; }
    const uint32_t w = hs.vocab_index(murmur64a(u, n, 0));
    if (w == 0 || w >= n1) continue;
    // get_log_cond_prob: cond_prob / NUM_FLT_LOGE (scorer.cpp:344), the constant being the float 0.4342944819
    const float v = up(((double)ub[w] + (double)bplus) / (double)0.4342944819f);
    hs.cp_ub[cp] = v;
  }
  hs.cp_ub_max = mx;
}

static size_t utf8_of(uint32_t cp, unsigned char* u) {   // U+0001 .. U+FFFF (surrogates have no encoding: 0)
  if (cp >= 0xD800 && cp < 0xE000) return 0;
  if (cp < 0x80) { u[0] = (unsigned char)cp; return 1; }
  if (cp < 0x800) { u[0] = (unsigned char)(0xC0 | (cp >> 6)); u[1] = (unsigned char)(0x80 | (cp & 0x3F)); return 2; }
  u[0] = (unsigned char)(0xE0 | (cp >> 12)); u[1] = (unsigned char)(0x80 | ((cp >> 6) & 0x3F)); u[2] = (unsigned char)(0x80 | (cp & 0x3F)); return 3;
}

// The bigram blocks of scorer_host.h.  Needs the hashed index (the records carry their slots: orders >= 3 continue there).
static bool build_cp_blocks(HostScorer& hs) {
  hs.cpb_ok = false; hs.cpt.clear(); hs.cpb_tab.clear(); hs.cpb_rec.clear();
  if (!hs.lmi_ok || hs.order < 2) return false;
  const uint8_t* buf = hs.buf;
  const uint64_t n1 = hs.counts[0], n2 = hs.counts[1];
  // code point <-> vocabulary index (a word of the model that is not ONE code point of the BMP has no block: it takes the index alone)
  hs.cpt.assign(65536, HostScorer::CptEntry{0u, 0.0f, 0.0f, 0u});
  std::vector<uint32_t> cp_of(n1, 0xFFFFFFFFu);
  for (uint32_t cp = 1; cp < 65536; ++cp) {
    unsigned char u[3];
    const size_t n = utf8_of(cp, u);
    if (!n) continue;
    const uint32_t w = hs.vocab_index(murmur64a(u, n, 0));
    if (w == 0 || w >= n1) continue;
    const uint8_t* ur = buf + hs.unigram_off + 16 * (uint64_t)w;
    hs.cpt[cp] = HostScorer::CptEntry{w, rdf(ur), rdf(ur + 4), 1u | (rd64(ur + 8) == rd64(ur + 24) ? 2u : 0u)};
    cp_of[w] = cp;
  }
  // every bigram (u | w1) whose u is a code point: {w1, cp, prob, backoff, indep, slot}
  struct Tup { uint32_t w1, cp; float prob, backoff; uint32_t slot; bool indep; };
  std::vector<Tup> tups;
  tups.reserve(n2);
  const HostBitPacked& bp = hs.order == 2 ? hs.lon : hs.mid[0];
  const uint64_t word_mask = (1ULL << bp.word_bits) - 1;
  for (uint64_t u = 1; u < n1; ++u) {
    if (cp_of[u] == 0xFFFFFFFFu) continue;
    const uint64_t cb = rd64(buf + hs.unigram_off + 16 * u + 8), ce = rd64(buf + hs.unigram_off + 16 * (u + 1) + 8);
    if (cb > ce || ce > n2) return false;
    const uint64_t hu = rd64(buf + hs.vocab_off + 8 * (u - 1));
    for (uint64_t c = cb; c < ce; ++c) {
      const uint32_t w1 = (uint32_t)rd57(buf + bp.base_off, c * bp.total_bits, word_mask);
      LmiEntry e;
      const uint64_t key = lmi_step(hu, w1);
      const uint32_t slot = lmi_probe(hs.lmi.data(), hs.lmi_buckets, lmi_bucket(key, hs.lmi_buckets), 2, w1, (uint32_t)u, e);
      if (slot == LMI_NOT_FOUND) return false;   // (the index holds every record of order >= 2)
      tups.push_back(Tup{w1, cp_of[u], e.prob, e.backoff, slot, (e.wl & LMI_INDEP_BIT) != 0});
    }
  }
  std::sort(tups.begin(), tups.end(), [](const Tup& a, const Tup& b) { return a.w1 != b.w1 ? a.w1 < b.w1 : a.cp < b.cp; });
  size_t n_ent = 0;
  for (size_t i = 0; i < tups.size(); ++i) if (i == 0 || tups[i].w1 != tups[i - 1].w1 || (tups[i].cp >> 6) != (tups[i - 1].cp >> 6)) ++n_ent;
  uint32_t cap = 16;
  while ((uint64_t)cap < 4 * (uint64_t)n_ent + 2) cap <<= 1;   // (load <= 1/4: most lookups are for a (context, block) the model has no bigram in -- an unsuccessful search, 1.4 probes at this load, 2.5 at 1/2)
  if (tups.size() >= 0xFFFFFFF0ull) return false;
  hs.cpb_tab.assign(cap, HostScorer::CpbEntry{0xFFFFFFFFu, 0u, 0u, 0u, 0ull, 0ull});
  hs.cpb_mask = cap - 1;
  hs.cpb_rec.resize(tups.size());
  for (size_t i = 0; i < tups.size();) {
    size_t j = i;
    HostScorer::CpbEntry en{tups[i].w1, tups[i].cp >> 6, (uint32_t)i, 0u, 0ull, 0ull};
    while (j < tups.size() && tups[j].w1 == en.w1 && (tups[j].cp >> 6) == en.block) {
      const uint64_t bit = 1ull << (tups[j].cp & 63u);
      if (en.present & bit) return false;   // (one record per (w1, u))
      en.present |= bit;
      if (tups[j].indep) en.indep |= bit;
      hs.cpb_rec[j] = HostScorer::CpbRec{tups[j].prob, tups[j].backoff, tups[j].slot};
      ++j;
    }
    en.count = (uint32_t)(j - i);
    uint32_t h = cpb_hash(en.w1, en.block) & hs.cpb_mask;
    while (hs.cpb_tab[h].w1 != 0xFFFFFFFFu) h = (h + 1) & hs.cpb_mask;
    hs.cpb_tab[h] = en;
    i = j;
  }
  hs.cpb_ok = true;
  return true;
}

float HostScorer::full_score_blocks(const KState& in, uint32_t cp, KState& out, int& ngram_length, uint32_t& word_index) const {
  unsigned char ub[3];
  const size_t un = cp < 65536 ? utf8_of(cp, ub) : 0;
  const CptEntry ct = (un && cp < cpt.size()) ? cpt[cp] : CptEntry{0u, 0.0f, 0.0f, 0u};
  if (!(ct.flags & 1u)) {   // not a word of the model: <unk> -- through the index alone, as every unit without a block
    return full_score_indexed(in, reinterpret_cast<const char*>(ub), un, out, ngram_length, word_index);
  }
  const uint32_t wi = ct.wi;
  word_index = wi;
  const bool uindep = (ct.flags & 2u) != 0;
  LmiLevel lv[LMI_MAX_HIST] = {};
  if (!uindep && in.length > 0 && order >= 2) {
    // order 2: ONE entry for (w1, the block of 64 code points u lies in) -- the same entry for all 64 siblings -- and a slice
    const uint32_t w1 = in.words[0], block = cp >> 6;
    uint32_t h = cpb_hash(w1, block) & cpb_mask;
    const CpbEntry* en = nullptr;
    for (;;) {
      const CpbEntry& t = cpb_tab[h];
      if (t.w1 == 0xFFFFFFFFu) break;
      if (t.w1 == w1 && t.block == block) { en = &t; break; }
      h = (h + 1) & cpb_mask;
    }
    const uint64_t bit = 1ull << (cp & 63u);
    if (en && (en->present & bit)) {
      const CpbRec& r = cpb_rec[en->offset + (uint32_t)__builtin_popcountll(en->present & (bit - 1))];
      lv[0].found = 1; lv[0].prob = r.prob; lv[0].backoff = r.backoff; lv[0].indep = (en->indep & bit) ? 1 : 0;
      // orders >= 3: the hashed index, from the bigram's slot on (full_score_indexed's loop, entered at its second trip)
      uint64_t key = lmi_step(murmur64a(ub, un, 0), w1);
      uint32_t parent = r.slot;
      for (int hi = 1; hi < order - 1 && hi < in.length && hi < LMI_MAX_HIST && !lv[hi - 1].indep; ++hi) {
        key = lmi_step(key, in.words[hi]);
        LmiEntry e;
        const uint32_t slot = lmi_probe(lmi.data(), lmi_buckets, lmi_bucket(key, lmi_buckets), hi + 2, in.words[hi], parent, e);
        if (slot == LMI_NOT_FOUND) break;
        lv[hi].found = 1; lv[hi].prob = e.prob; lv[hi].backoff = e.backoff; lv[hi].indep = (e.wl & LMI_INDEP_BIT) ? 1 : 0;
        parent = slot;
      }
    }
  }
  return lmi_combine(order, in, wi, ct.prob, ct.backoff, uindep, lv, out, ngram_length);
}

// SortedVocabulary::Index (lm/vocab.hh:72-83): binary search over the sorted hashes (host side)
uint32_t HostScorer::vocab_index(uint64_t h) const {
  if (probing) {   // ProbingVocabulary::Index (lm/vocab.hh:162-165): 12-byte entries {hash, index}, linear probing from hash % buckets, hash 0 = empty
    uint64_t i = h % p_vocab_buckets;
    for (uint64_t n = 0; n < p_vocab_buckets; ++n) {
      const uint64_t got = rd64(buf + p_vocab_tab_off + 12 * i);
      if (got == h) return rd32(buf + p_vocab_tab_off + 12 * i + 8);
      if (got == 0) return 0;
      if (++i == p_vocab_buckets) i = 0;
    }
    return 0;
  }
  uint64_t lo = 0, hi = vocab_n;
  while (lo < hi) {
    const uint64_t m2 = lo + (hi - lo) / 2, v = rd64(buf + vocab_off + 8 * m2);
    if (v < h) lo = m2 + 1; else if (v > h) hi = m2; else return (uint32_t)(m2 + 1);
  }
  return 0;
}

// GenericModel<HashedSearch<Value>, ProbingVocabulary>::FullScore (lm/model.cc:170-176,285-338; lm/search_hashed.hh:26-29,93-125) on the host.
float HostScorer::full_score_probing(const KState& in, const char* word, size_t word_len, KState& out, int& ngram_length, uint32_t& word_index) const {
  auto find = [&](uint64_t tab_off, uint64_t buckets, int stride, uint64_t key) -> const uint8_t* {
    uint64_t i = key % buckets;
    for (uint64_t n = 0; n < buckets; ++n) {
      const uint8_t* e = buf + tab_off + i * (uint64_t)stride;
      const uint64_t got = rd64(e);
      if (got == key) return e;
      if (got == 0) return nullptr;
      if (++i == buckets) i = 0;
    }
    return nullptr;
  };
  auto combine = [](uint64_t cur, uint32_t next) { return (cur * 8978948897894561157ULL) ^ ((uint64_t)(1 + next) * 17894857484156487943ULL); };
  auto has_ext = [](float b) { uint32_t u; memcpy(&u, &b, 4); return u != 0x80000000u; };
  auto neg = [](uint32_t raw) { raw |= 0x80000000u; float f; memcpy(&f, &raw, 4); return f; };
  const uint32_t wi = vocab_index(murmur64a(word, word_len, 0));
  word_index = wi;
  const uint8_t* u = buf + unigram_off + (uint64_t)p_wstride * wi;
  const uint32_t praw = rd32(u);
  float prob = neg(praw);
  out = KState{};
  out.backoff[0] = rdf(u + 4);
  bool independent_left = (praw & 0x80000000u) != 0;
  uint64_t node = wi;
  int nl = 1, out_len = has_ext(out.backoff[0]) ? 1 : 0;
  out.words[0] = wi;
  for (int i = 0; i + 1 < STT_KENLM_MAX_ORDER - 1 && i < in.length; ++i) out.words[i + 1] = in.words[i];   // CopyRemainingHistory (entries past length unused)
  for (int om2 = 0; om2 < in.length && om2 < STT_KENLM_MAX_ORDER - 1; ++om2) {
    if (independent_left) break;
    if (om2 == order - 2) {
      const uint8_t* e = find(p_lon_off, p_lon_buckets, 12, combine(node, in.words[om2]));
      if (e) { prob = rdf(e + 8); nl = order; }
      break;
    }
    node = combine(node, in.words[om2]);
    const uint8_t* e = find(p_mid_off[om2], p_mid_buckets[om2], p_estride, node);
    if (!e) break;
    const uint32_t pr = rd32(e + 8);
    const float b = rdf(e + 12);
    independent_left = (pr & 0x80000000u) != 0;
    out.backoff[om2 + 1] = b; prob = neg(pr); nl = om2 + 2;
    if (has_ext(b)) out_len = nl;
  }
  ngram_length = nl;
  out.length = out_len;
  for (int i = nl - 1; i < in.length; ++i) prob += in.backoff[i];
  return prob;
}

// FullScore through the index, on the host: what the search kernel's LM waves do with four lanes per query.
float HostScorer::full_score_indexed(const KState& in, const char* word, size_t word_len, KState& out, int& ngram_length, uint32_t& word_index) const {
  const uint64_t h = murmur64a(word, word_len, 0);
  const uint32_t wi = vocab_index(h);
  word_index = wi;
  const uint8_t* u = buf + unigram_off + 16 * (uint64_t)wi;
  const float uprob = rdf(u), uback = rdf(u + 4);
  const bool uindep = rd64(u + 8) == rd64(u + 24);
  LmiLevel lv[LMI_MAX_HIST] = {};
  uint64_t key = wi ? h : LMI_UNK_H;
  uint32_t parent = wi;
  for (int hi = 0; hi < order - 1 && hi < in.length && hi < LMI_MAX_HIST; ++hi) {
    key = lmi_step(key, in.words[hi]);
    LmiEntry e;
    const uint32_t slot = lmi_probe(lmi.data(), lmi_buckets, lmi_bucket(key, lmi_buckets), hi + 2, in.words[hi], parent, e);
    if (slot == LMI_NOT_FOUND) break;
    lv[hi].found = 1; lv[hi].prob = e.prob; lv[hi].backoff = e.backoff; lv[hi].indep = (e.wl & LMI_INDEP_BIT) ? 1 : 0;
    parent = slot;
  }
  return lmi_combine(order, in, wi, uprob, uback, uindep, lv, out, ngram_length);
}

// ------------------------------------------------------------------------------------------- parse (host only)
int parse_scorer(const uint8_t* buf, size_t len, int space_label, bool lm_only, HostScorer& hs) {
  static const char kMagic[] = "mmap lm http://kheafield.com/code format version 5\n";
  if (len < 88 + 20 || memcmp(buf, kMagic, sizeof(kMagic)) != 0) return STT_ERR_SCORER_INVALID_LM;
  const uint8_t* fp = buf + 88;  // FixedWidthParameters after the 88-byte Sanity block
  const int ord = fp[0];
  const int model_type = (int)rd32(fp + 8);
  if (ord < 2 || ord > STT_KENLM_MAX_ORDER) return STT_ERR_SCORER_INVALID_LM;
  if (model_type < 0 || model_type > 5) return STT_ERR_SCORER_INVALID_LM;
  // PROBING (0: lm/model_type.hh:9): hash tables instead of the bit-packed trie.  REST_PROBING (1) has the same tables with a third float per
  // record (the strides below cover it), but no such binary can be built here to check it against KenLM (build_binary -r wants the lower-order
  // models): refused rather than claimed.
  if (model_type == 1) return STT_ERR_SCORER_INVALID_LM;
  const bool probing = model_type <= 1;
  const bool quant = (model_type == 3 || model_type == 5), array = (model_type == 4 || model_type == 5);
  uint64_t* counts = hs.counts;
  if (len < 108 + 8 * (size_t)ord) return STT_ERR_SCORER_INVALID_LM;
  for (int i = 0; i < ord; ++i) { counts[i] = rd64(buf + 108 + 8 * i); if (counts[i] > (1ull << 40)) return STT_ERR_SCORER_INVALID_LM; }
  if (counts[0] < 1) return STT_ERR_SCORER_INVALID_LM;

  uint64_t off = align8(108 + 8 * (uint64_t)ord);
  if (off + 8 > len) return STT_ERR_SCORER_INVALID_LM;
  if (probing) {
    // GenericModel<HashedSearch<Value>, ProbingVocabulary>::SetupMemory (lm/model.cc:27-35): vocabulary, unigrams, one table per middle order,
    // the longest order's table, back to back.  Table sizes: util/probing_hash_table.hh:108-111 with the file's own probing_multiplier.
    const float mult = rdf(fp + 4);
    if (!(mult > 1.0f) || mult > 1024.0f) return STT_ERR_SCORER_INVALID_LM;
    auto buckets_of = [&](uint64_t entries) { return std::max<uint64_t>(entries + 1, (uint64_t)(mult * (float)entries)); };
    hs.probing = true;
    hs.p_wstride = model_type == 1 ? 12 : 8;      // RestWeights {prob, backoff, rest} / ProbBackoff
    hs.p_estride = 8 + hs.p_wstride;              // {uint64 key, Weights}: 16 bytes, or 20 (#pragma pack(4), lm/value.hh:118-126)
    if (rd32(buf + off) != 0) return STT_ERR_SCORER_INVALID_LM;   // kProbingVocabularyVersion (lm/vocab.cc:257)
    hs.p_vocab_buckets = buckets_of(counts[0]);
    hs.p_vocab_tab_off = off + 8;                 // ALIGN8(sizeof(ProbingVocabularyHeader {version, bound}))
    off += 8 + 12 * hs.p_vocab_buckets;           // ProbingVocabularyEntry {uint64 key, WordIndex value}, #pragma pack(4)
    hs.unigram_off = off;
    off += (counts[0] + 1) * (uint64_t)hs.p_wstride;
    for (int i = 0; i < ord - 2; ++i) {
      hs.p_mid_off[i] = off; hs.p_mid_buckets[i] = buckets_of(counts[i + 1]);
      off += hs.p_mid_buckets[i] * (uint64_t)hs.p_estride;
      if (off > len) return STT_ERR_SCORER_INVALID_LM;
    }
    hs.p_lon_off = off; hs.p_lon_buckets = buckets_of(counts[ord - 1]);
    off += hs.p_lon_buckets * 12;                 // ProbEntry {uint64 key, Prob}, #pragma pack(4) (lm/search_hashed.hh:31-42)
    if (off > len) return STT_ERR_SCORER_INVALID_LM;
    hs.buf = buf; hs.order = ord; hs.model_type = model_type; hs.quant = false;
    hs.vocab_n = counts[0] ? counts[0] - 1 : 0; hs.vocab_off = 0; hs.lm_end = off;
    hs.alpha = 0.0; hs.beta = 0.0; hs.utf8 = false; hs.fst_start = 0;
  }
  const uint64_t vocab_n = probing ? hs.vocab_n : rd64(buf + off);
  const uint64_t vocab_off = off + 8;
  if (vocab_n > counts[0]) return STT_ERR_SCORER_INVALID_LM;
  if (!probing) off += 8 + 8 * counts[0];
  uint8_t prob_bits = 0, backoff_bits = 0;
  uint64_t unigram_off = hs.unigram_off, lm_end = hs.lm_end;
  if (!probing) {
  if (quant) {
    if (off + 8 > len) return STT_ERR_SCORER_INVALID_LM;
    prob_bits = buf[off + 1]; backoff_bits = buf[off + 2];
    if (!prob_bits || !backoff_bits || prob_bits > 25 || backoff_bits > 25) return STT_ERR_SCORER_INVALID_LM;
    uint64_t t = off + 8;
    for (int i = 0; i < ord - 2; ++i) { hs.qprob_off[i] = t; t += 4ULL << prob_bits; hs.qback_off[i] = t; t += 4ULL << backoff_bits; }
    hs.qprob_off[ord - 2] = t; t += 4ULL << prob_bits;
    off = t;
  }
  unigram_off = off;
  off += (counts[0] + 2) * 16;
  if (off > len) return STT_ERR_SCORER_INVALID_LM;
  uint8_t cfg_bhiksha = 0;
  if (array && ord > 2) { if (off + 2 > len) return STT_ERR_SCORER_INVALID_LM; cfg_bhiksha = buf[off + 1]; }
  const uint8_t middle_quant_bits = quant ? (uint8_t)(prob_bits + backoff_bits) : 63;
  const uint8_t longest_bits = quant ? prob_bits : 31;
  for (int i = 0; i < ord - 2; ++i) {
    const uint64_t entries = counts[i + 1], max_vocab = counts[0], max_next = counts[i + 2];
    uint64_t bh_size = 0; uint8_t inline_bits;
    HostBitPacked& m = hs.mid[i];
    if (array) {
      const uint8_t required = required_bits(max_next), chop = chop_bits(entries + 1, max_next, cfg_bhiksha);
      const uint64_t array_count = (max_next >> (required - chop)) + 1;
      bh_size = 8 * (1 + array_count) + 7;
      inline_bits = required - chop;
      m.off_begin_off = align8(off) + 8; m.off_count = (uint32_t)array_count;
    } else {
      inline_bits = required_bits(max_next);
    }
    m.entries = entries;
    m.base_off = off + bh_size;
    m.word_bits = required_bits(max_vocab); m.quant_bits = middle_quant_bits; m.next_bits = inline_bits;
    m.total_bits = (uint8_t)(m.word_bits + middle_quant_bits + inline_bits);
    off += bh_size + (((1 + entries) * m.total_bits + 7) / 8 + 8);
    if (off > len) return STT_ERR_SCORER_INVALID_LM;
  }
  HostBitPacked& lon = hs.lon;
  lon.base_off = off; lon.word_bits = required_bits(counts[0]); lon.total_bits = (uint8_t)(lon.word_bits + longest_bits);
  lon.entries = counts[ord - 1];
  off += ((1 + counts[ord - 1]) * lon.total_bits + 7) / 8 + 8;
  lm_end = off;  // GetEndOfSearchOffset, lm/model.cc:265-267
  if (lm_end > len) return STT_ERR_SCORER_INVALID_LM;
  hs.buf = buf; hs.order = ord; hs.model_type = model_type; hs.quant = quant; hs.prob_bits = prob_bits; hs.backoff_bits = backoff_bits;
  hs.vocab_n = vocab_n; hs.vocab_off = vocab_off; hs.unigram_off = unigram_off; hs.lm_end = lm_end;
  hs.alpha = 0.0; hs.beta = 0.0; hs.utf8 = false; hs.fst_start = 0;
  }   // (!probing: the trie's layout)

  if (!lm_only) {
    if (len <= lm_end) return STT_ERR_SCORER_NO_TRIE;
    // ---- package trailer (scorer.cpp:177-222); every advance of `o` is checked against the buffer before the next read
    const uint8_t* p = buf + lm_end;
    if (lm_end + 25 > len || rd32(p) != 0x54524945u) return STT_ERR_SCORER_INVALID_TRIE;
    if ((int)rd32(p + 4) != 6) return STT_ERR_SCORER_VERSION_MISMATCH;
    hs.utf8 = p[8] != 0;
    double a, b; memcpy(&a, p + 9, 8); memcpy(&b, p + 17, 8);
    hs.alpha = a; hs.beta = b;
    uint64_t o = lm_end + 25;
    auto room = [&](uint64_t n) { return o <= len && n <= len - o; };
    if (!room(4) || rd32(buf + o) != 2125659606u) return STT_ERR_SCORER_INVALID_TRIE;  // FstHeader magic (fst.cc:62-75)
    o += 4;
    for (int k = 0; k < 2; ++k) {  // fsttype, arctype strings
      if (!room(4)) return STT_ERR_SCORER_INVALID_TRIE;
      const uint32_t l = rd32(buf + o); o += 4;
      if (!room(l)) return STT_ERR_SCORER_INVALID_TRIE;
      o += l;
    }
    if (!room(4 + 4 + 8 + 24)) return STT_ERR_SCORER_INVALID_TRIE;
    o += 4;  // version
    const uint32_t flags = rd32(buf + o); o += 4;
    o += 8;  // properties
    int64_t fst_start, nstates, narcs;
    memcpy(&fst_start, buf + o, 8); o += 8; memcpy(&nstates, buf + o, 8); o += 8; memcpy(&narcs, buf + o, 8); o += 8;
    if (flags & 3) return STT_ERR_SCORER_INVALID_TRIE;
    if (nstates <= 0 || narcs < 0 || fst_start < 0 || fst_start >= nstates) return STT_ERR_SCORER_INVALID_TRIE;
    if ((uint64_t)nstates > len / 20 || (uint64_t)narcs > len / 16) return STT_ERR_SCORER_INVALID_TRIE;  // cannot fit: no overflow below
    if (flags & 4) o = (o + 15) & ~(uint64_t)15;
    if (!room((uint64_t)nstates * 20)) return STT_ERR_SCORER_INVALID_TRIE;
    const uint8_t* states = buf + o; o += (uint64_t)nstates * 20;
    if (flags & 4) o = (o + 15) & ~(uint64_t)15;
    if (!room((uint64_t)narcs * 16)) return STT_ERR_SCORER_INVALID_TRIE;
    const uint8_t* arcs = buf + o; o += (uint64_t)narcs * 16;

    // ---- repack the dictionary: arcs sorted by ilabel within a state (SortedMatcher's precondition)
    // The arc's second field is the dictionary state of the child prefix: Start() when the arc's target is final
    // (a completed word, path_trie.cpp:79-87), the target otherwise -- resolved here so the kernel needs one read.
    // Per state also {first arc with a real label, bitmap of the labels 1..32 on its arcs}: with the labels of a
    // state in one word, the search kernel intersects "labels the dictionary allows" with "labels that survive the
    // score cut-off" before it touches a single arc (usable when every state's arcs are strictly ascending by label).
    hs.fst_pos.assign((size_t)nstates + 1, 0);
    std::vector<uint8_t> fin((size_t)nstates);
    hs.fst_arcs.assign((size_t)narcs, make_uint2(0, 0));
    hs.fst_has_space.assign((size_t)nstates + 1, 0);
    hs.fst_rec.assign((size_t)nstates + 1, make_uint2(0, 0));
    hs.fst_bitmap_ok = true;
    for (int64_t s = 0; s < nstates; ++s) fin[s] = !(rdf(states + 20 * s) == INFINITY);  // Final(s) != TropicalWeight::Zero()
    uint32_t w = 0;
    for (int64_t s = 0; s < nstates; ++s) {
      const uint8_t* S = states + 20 * s;
      const uint32_t ap = rd32(S + 4), an = rd32(S + 8);
      if ((uint64_t)ap + an > (uint64_t)narcs || (uint64_t)w + an > (uint64_t)narcs) return STT_ERR_SCORER_INVALID_TRIE;
      hs.fst_pos[s] = w;
      uint32_t mask = 0, first = w + an, prev_label = 0;
      for (uint32_t k = 0; k < an; ++k) {
        const uint8_t* A = arcs + 16 * (uint64_t)(ap + k);
        const uint32_t ilabel = rd32(A), next = rd32(A + 12);
        if (next >= (uint64_t)nstates) return STT_ERR_SCORER_INVALID_TRIE;
        if (k > 0 && ilabel <= prev_label && !(ilabel == 0 && prev_label == 0)) hs.fst_bitmap_ok = false;
        prev_label = ilabel;
        if (ilabel >= 1) { if (first == w + an) first = w; if (ilabel <= 32) mask |= 1u << (ilabel - 1); }
        hs.fst_arcs[w++] = make_uint2(ilabel, fin[next] ? (uint32_t)fst_start : next);
        if (space_label >= 0 && (int64_t)ilabel == (int64_t)space_label + 1) hs.fst_has_space[s] = 1;
      }
      hs.fst_rec[s] = make_uint2(first, mask);
    }
    hs.fst_pos[nstates] = w;
    hs.fst_start = (int)fst_start;
    hs.n_states = (uint64_t)nstates;
    hs.fst_tree = false;
    // ---- word mode: the dictionary unfolded into the tree of its word prefixes (the minimised automaton shares suffixes; the
    // tree has one node per distinct prefix -- at most the number of characters of the vocabulary).  Breadth first, node k + 1 is
    // the target of arc k and the root is node 0, so the child of a node along its r-th labelled arc is `first arc + r + 1`: the
    // search kernel needs no arc read to follow the dictionary (an expand item's arc read was a dependent HBM/L2 miss in the
    // middle of the item).  Same language, same behaviour: the reference only asks "which labels may follow" and "has a word
    // ended" (path_trie.cpp:54-90).  Taken when every arc into a final state carries the space label and the other way round
    // (what generate_scorer_package builds) and the tree stays below the `dict_tree_mb` cap; the repacked automaton otherwise.
    const uint64_t cap_nodes = (uint64_t)std::max(0, tune().dict_tree_mb) * 1024 * 1024 / 21;  // arcs 8 + rec 8 + pos 4 + flag 1 bytes per node
    if (!hs.utf8 && space_label >= 0 && space_label < 32 && hs.fst_bitmap_ok && cap_nodes > 1) {
      const uint32_t NONE = 0xFFFFFFFFu;
      std::vector<uint32_t> node_state(1, (uint32_t)fst_start), t_pos;
      std::vector<uint2> t_arcs, t_rec;
      std::vector<uint8_t> t_space;
      bool ok = true;
      for (size_t v = 0; v < node_state.size() && ok; ++v) {
        const uint32_t st = node_state[v];
        const uint32_t first = (uint32_t)t_arcs.size();
        t_pos.push_back(first);
        uint32_t mask = 0; uint8_t sp = 0;
        if (st != NONE) {
          for (uint32_t k = hs.fst_pos[st]; k < hs.fst_pos[st + 1]; ++k) {
            const uint2 a = hs.fst_arcs[k];
            if (a.x == 0) continue;                                        // epsilon arcs are never matched
            if (a.x > 32) { ok = false; break; }
            const uint32_t raw_next = rd32(arcs + 16 * (uint64_t)(rd32(states + 20 * (uint64_t)st + 4) + (k - hs.fst_pos[st])) + 12);
            const bool to_final = fin[raw_next] != 0, is_space = (int64_t)a.x == (int64_t)space_label + 1;
            if (to_final != is_space) { ok = false; break; }
            mask |= 1u << (a.x - 1);
            if (is_space) sp = 1;
            t_arcs.push_back(make_uint2(a.x, is_space ? 0u : (uint32_t)t_arcs.size() + 1u));
            node_state.push_back(is_space ? NONE : raw_next);               // (a space arc's own node is never entered)
            if (node_state.size() > cap_nodes) { ok = false; break; }
          }
        }
        t_rec.push_back(make_uint2(first, mask));
        t_space.push_back(sp);
      }
      if (ok) {
        t_pos.push_back((uint32_t)t_arcs.size());
        t_rec.push_back(make_uint2((uint32_t)t_arcs.size(), 0)); t_space.push_back(0);
        hs.fst_pos.swap(t_pos); hs.fst_arcs.swap(t_arcs); hs.fst_rec.swap(t_rec); hs.fst_has_space.swap(t_space);
        hs.fst_start = 0; hs.n_states = (uint64_t)node_state.size(); hs.fst_tree = true;
      }
    }
  }

  // ---- vocabulary hash table over KenLM's sorted hash array (index = position + 1, vocab.hh:72-83)
  uint32_t vt_n = 16;
  while ((uint64_t)vt_n < 2 * vocab_n + 2) vt_n <<= 1;
  hs.vtab.assign(vt_n, DevVocabSlot{0, 0, 0, 0.f, 0.f, 0, 0});
  hs.uni_ok = ord >= 2 && counts[1] < 0xFFFFFFFFull;
  if (!probing && vocab_off + 8 * vocab_n > len) return STT_ERR_SCORER_INVALID_LM;
  if (probing) {
    // the engine's own table over ProbingVocabulary's: every occupied bucket {hash, index}; the unigram copies stay unused (uni_ok false: the
    // probing FullScore reads the Weights array itself)
    hs.uni_ok = false;
    for (uint64_t b = 0; b < hs.p_vocab_buckets; ++b) {
      const uint64_t h = rd64(buf + hs.p_vocab_tab_off + 12 * b);
      if (!h) continue;
      const uint32_t wi = rd32(buf + hs.p_vocab_tab_off + 12 * b + 8);
      if (wi == 0 || wi >= counts[0]) return STT_ERR_SCORER_INVALID_LM;
      uint32_t slot = (uint32_t)h & (vt_n - 1);
      while (hs.vtab[slot].used) slot = (slot + 1) & (vt_n - 1);
      hs.vtab[slot] = DevVocabSlot{h, wi, 1, 0.f, 0.f, 0, 0};
    }
  } else
  for (uint64_t i = 0; i < vocab_n; ++i) {
    const uint64_t h = rd64(buf + vocab_off + 8 * i);
    uint32_t slot = (uint32_t)h & (vt_n - 1);
    while (hs.vtab[slot].used) slot = (slot + 1) & (vt_n - 1);
    const uint8_t* u = buf + unigram_off + 16 * (uint64_t)(i + 1);
    hs.vtab[slot] = DevVocabSlot{h, (uint32_t)(i + 1), 1, rdf(u), rdf(u + 4), (uint32_t)rd64(u + 8), (uint32_t)rd64(u + 24)};
  }
  // Bhiksha hint tables: the offsets array is sorted; a coarse direct-index table replaces KenLM's std::upper_bound over it
  // (lm/bhiksha.hh:76-95) by one read plus a short forward scan.  Same result, fewer dependent HBM reads.
  hs.hints.clear();
  for (int i = 0; i < ord - 2; ++i) {
    hs.hint_off[i] = 0; hs.hint_shift[i] = 0;
    if (!hs.mid[i].off_begin_off) continue;
    const uint64_t* offs = reinterpret_cast<const uint64_t*>(buf + hs.mid[i].off_begin_off);
    const uint32_t cnt = hs.mid[i].off_count;
    const uint64_t entries = hs.mid[i].entries + 2;
    uint32_t sh = 0;
    while (((entries >> sh) > 4ull * cnt + 1024) && sh < 40) ++sh;  // about four table slots per offset
    hs.hint_off[i] = hs.hints.size(); hs.hint_shift[i] = sh;
    const uint64_t slots = (entries >> sh) + 2;
    uint32_t pos = 0;  // upper_bound(offs, v) - 1 for increasing v
    for (uint64_t j = 0; j < slots; ++j) {
      const uint64_t v = j << sh;
      while (pos + 1 < cnt && offs[pos + 1] <= v) ++pos;
      hs.hints.push_back(pos);
    }
  }
  if (hs.hints.empty()) hs.hints.push_back(0);
  // <s> index and backoff (lm/model.cc:115-124)
  hs.bos_index = hs.vocab_index(murmur64a("<s>", 3, 0));
  hs.bos_backoff = probing ? rdf(buf + unigram_off + (uint64_t)hs.p_wstride * hs.bos_index + 4) : rdf(buf + unigram_off + 16 * (uint64_t)hs.bos_index + 4);
  // The index is read by the word-mode search step with label bitmaps (ctc.hip: ctc_masked_ok), by the code-point step on a miss of its
  // FullScore memo (tunable cp_index) and by the LM test hooks; order-6 models and word dictionaries without bitmaps never touch it: do
  // not build what cannot be used.
  // (a probing model's tables hold hashes of contexts, not word indices: the n-grams cannot be enumerated, so neither the index nor the
  // code-point tables exist for it -- every query is a FullScore through the file's own hash tables)
  const bool index_usable = !probing && (lm_only || (ord <= 5 && hs.uni_ok && (hs.utf8 ? tune().cp_index != 0 : hs.fst_bitmap_ok)));
  try { hs.lmi_ok = index_usable && build_lm_index(hs); }
  catch (const std::bad_alloc&) { hs.lmi_ok = false; }  // no memory for the table: the scorer still loads (trie walk), as it does in the reference
  if (!hs.lmi_ok) { hs.lmi.clear(); hs.lmi.shrink_to_fit(); hs.lmi_buckets = 0; }
  if (!probing && hs.utf8 && !lm_only && tune().unit_bounds != 0) build_unit_bounds(hs);
  if (!probing && (hs.utf8 || lm_only) && tune().cp_blocks != 0) (void)build_cp_blocks(hs);
  return STT_ERR_OK;
}

// ------------------------------------------------------------------------------------------- upload
int ScorerDev::LoadFile(const std::string& path, const Alphabet& alphabet) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) return STT_ERR_SCORER_UNREADABLE;
  const std::streamsize sz = in.tellg();
  in.seekg(0);
  std::vector<char> data((size_t)sz + 16, 0);  // (+16: the 64-bit bit-packed reads may touch up to 8 bytes past the last record)
  if (sz > 0 && !in.read(data.data(), sz)) return STT_ERR_SCORER_UNREADABLE;
  return Parse(reinterpret_cast<const uint8_t*>(data.data()), (size_t)sz, alphabet.GetSpaceLabel(), false);
}

int ScorerDev::LoadBuffer(const char* data, size_t len, const Alphabet& alphabet) {
  // (a package's trailer follows the KenLM blob: the bit-packed reads past the last record stay inside the buffer)
  return Parse(reinterpret_cast<const uint8_t*>(data), len, alphabet.GetSpaceLabel(), false);
}

int ScorerDev::LoadLmOnly(const char* data, size_t len) {
  std::vector<char> copy(len + 16, 0);
  memcpy(copy.data(), data, len);
  return Parse(reinterpret_cast<const uint8_t*>(copy.data()), len, -1, true);
}

int ScorerDev::Parse(const uint8_t* buf, size_t len, int space_label, bool lm_only) {
  HostScorer hs;
  const int rc = parse_scorer(buf, len, space_label, lm_only, hs);
  if (rc != STT_ERR_OK) return rc;
  const int ord = hs.order;
  blob_.upload(buf, hs.lm_end + 16);
  vtab_.upload(hs.vtab.data(), hs.vtab.size() * sizeof(DevVocabSlot));
  hint_.upload(hs.hints.data(), hs.hints.size() * 4);
  if (!lm_only) {
    fst_pos_.upload(hs.fst_pos.data(), hs.fst_pos.size() * 4);
    fst_arcs_.upload(hs.fst_arcs.empty() ? (const void*)&hs.fst_pos[0] : (const void*)hs.fst_arcs.data(), std::max<size_t>(8, hs.fst_arcs.size() * sizeof(uint2)));
    fst_space_.upload(hs.fst_has_space.data(), hs.fst_has_space.size());
    fst_rec_.upload(hs.fst_rec.data(), hs.fst_rec.size() * sizeof(uint2));
  }
  if (hs.lmi_ok) {
    try { lmi_.upload(hs.lmi.data(), hs.lmi.size() * sizeof(LmiEntry)); }
    catch (const std::exception&) { hs.lmi_ok = false; (void)hipGetLastError(); }  // no HBM for it: trie walk
  }
  const uint8_t* d = blob_.as<uint8_t>();
  DevScorer ds{};
  ds.enabled = 1; ds.order = ord; ds.quant = hs.quant; ds.utf8 = hs.utf8;
  ds.alpha = (double)(float)hs.alpha; ds.beta = (double)(float)hs.beta;  // Scorer::reset_params(float, float)
  ds.vocab = reinterpret_cast<const uint64_t*>(d + hs.vocab_off); ds.vocab_n = hs.vocab_n;
  ds.vtab = vtab_.as<DevVocabSlot>(); ds.vtab_mask = (uint32_t)hs.vtab.size() - 1; ds.uni_in_vtab = hs.uni_ok ? 1 : 0;
  ds.unigram = d + hs.unigram_off;
  for (int i = 0; i < STT_KENLM_MAX_ORDER; ++i) {
    ds.qprob[i] = hs.quant && hs.qprob_off[i] ? reinterpret_cast<const float*>(d + hs.qprob_off[i]) : nullptr;
    ds.qbackoff[i] = hs.quant && hs.qback_off[i] ? reinterpret_cast<const float*>(d + hs.qback_off[i]) : nullptr;
  }
  auto fill = [&](DevBitPacked& o2, const HostBitPacked& b2) {
    o2.base = d + b2.base_off; o2.word_bits = b2.word_bits; o2.total_bits = b2.total_bits; o2.quant_bits = b2.quant_bits; o2.next_bits = b2.next_bits;
    o2.word_mask = (1ULL << b2.word_bits) - 1; o2.next_mask = (1ULL << b2.next_bits) - 1;
    o2.off_begin = b2.off_begin_off ? reinterpret_cast<const uint64_t*>(d + b2.off_begin_off) : nullptr; o2.off_count = b2.off_count;
    o2.off_hint = nullptr; o2.hint_shift = 0; o2.max_word = hs.counts[0];
  };
  for (int i = 0; i < ord - 2; ++i) {
    fill(ds.middle[i], hs.mid[i]);
    if (hs.mid[i].off_begin_off) { ds.middle[i].off_hint = hint_.as<uint32_t>() + hs.hint_off[i]; ds.middle[i].hint_shift = hs.hint_shift[i]; }
  }
  fill(ds.longest, hs.lon);
  ds.prob_bits = hs.prob_bits; ds.backoff_bits = hs.backoff_bits;
  ds.prob_mask = (1u << hs.prob_bits) - 1; ds.backoff_mask = (1u << hs.backoff_bits) - 1;
  ds.bos_index = hs.bos_index; ds.bos_backoff = hs.bos_backoff;
  ds.probing = hs.probing ? 1 : 0; ds.p_wstride = hs.p_wstride; ds.p_estride = hs.p_estride;
  if (hs.probing) {
    for (int i = 0; i < ord - 2; ++i) { ds.p_mid[i] = d + hs.p_mid_off[i]; ds.p_mid_buckets[i] = hs.p_mid_buckets[i]; }
    ds.p_lon = d + hs.p_lon_off; ds.p_lon_buckets = hs.p_lon_buckets;
  }
  ds.fst_start = hs.fst_start; ds.fst_tree = hs.fst_tree ? 1 : 0;
  if (!lm_only) {
    ds.fst_state_pos = fst_pos_.as<uint32_t>(); ds.fst_arcs = fst_arcs_.as<uint2>(); ds.fst_has_space = fst_space_.as<uint8_t>();
    ds.fst_rec = hs.fst_bitmap_ok ? fst_rec_.as<uint2>() : nullptr;
  }
  ds.lmi = hs.lmi_ok ? lmi_.as<LmiEntry>() : nullptr; ds.lmi_buckets = hs.lmi_buckets;
  ds.unk_prob = hs.unk_prob; ds.unk_backoff = hs.unk_backoff; ds.unk_indep = hs.unk_indep ? 1 : 0;
  // (unit_bounds 1: only the largest bound is used -- a comparison, no read; 2: the per-code-point table as well: one more read per
  // candidate, worth it only where the bounds differ from unit to unit; measured a loss on both benchmark scorers)
  if (!hs.cp_ub.empty()) {
    ds.cp_ub_max = hs.cp_ub_max; ds.cp_ub_on = 1;
    if (tune().unit_bounds >= 2) { cp_ub_.upload(hs.cp_ub.data(), hs.cp_ub.size() * 4); ds.cp_ub = cp_ub_.as<float>(); }
  }
  if (hs.cpb_ok && hs.lmi_ok && (hs.utf8 || lm_only)) {   // bigram blocks of the code-point step (ctc.hip: lm_full_score_blocks; a bare LM: the test hook's)
    static_assert(sizeof(HostScorer::CptEntry) == 16 && sizeof(HostScorer::CpbEntry) == 32 && sizeof(HostScorer::CpbRec) == 12, "device layout of the bigram blocks");
    try {
      cpt_.upload(hs.cpt.data(), hs.cpt.size() * sizeof(HostScorer::CptEntry));
      cpb_tab_.upload(hs.cpb_tab.data(), hs.cpb_tab.size() * sizeof(HostScorer::CpbEntry));
      cpb_rec_.upload(hs.cpb_rec.empty() ? (const void*)hs.cpt.data() : (const void*)hs.cpb_rec.data(), std::max<size_t>(12, hs.cpb_rec.size() * sizeof(HostScorer::CpbRec)));
      ds.cpt = cpt_.as<uint32_t>(); ds.cpb_tab = cpb_tab_.as<uint32_t>(); ds.cpb_rec = cpb_rec_.as<uint32_t>(); ds.cpb_mask = hs.cpb_mask;
    } catch (const std::exception&) { ds.cpt = nullptr; ds.cpb_tab = nullptr; ds.cpb_rec = nullptr; (void)hipGetLastError(); }   // no HBM for them: memo + index
  }
  const bool memo_on = tune().lm_memo != 0;  // (0: measure without)
  if (hs.utf8 && ord <= 5 && !lm_only && memo_on) {  // FullScore cache of the code-point search (ctc.h: DevScorer::memo)
    int lg = tune().lm_memo >= 10 && tune().lm_memo <= 26 ? tune().lm_memo : 24;   // (lm_memo 1 = the default 2^24 entries = 512 MB of the 288 GB, 10..26 = log2 of the entry count)
    // The memo is an accelerator, not a requirement (a miss goes through the index or the trie walk): a device that cannot spare the
    // default -- many replicas, a shared GPU -- gets a smaller one, or none, instead of a scorer that fails to load.
    for (; lg >= 14; lg -= 2) {
      const size_t n = (size_t)1 << lg;
      void* p = nullptr;      // exactly the bytes the table takes (DevBuf::reserve would ask for a quarter more and throw where this probe passed)
      if (hipMalloc(&p, n * 32) != hipSuccess) { (void)hipGetLastError(); continue; }
      if (memo_.p) (void)hipFree(memo_.p);
      memo_.p = p; memo_.cap = n * 32;
      HIP_CHECK(hipMemset(memo_.p, 0, n * 32));
      ds.memo = memo_.as<uint32_t>(); ds.memo_mask = (uint32_t)n - 1;
      break;
    }
  }
  dev = ds;
  is_utf8 = hs.utf8; order = ord; blob_bytes = hs.lm_end; lmi_bytes = hs.lmi.size() * sizeof(LmiEntry);
  return STT_ERR_OK;
}
