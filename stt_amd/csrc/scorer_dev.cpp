// stt_amd/csrc/scorer_dev.cpp -- .scorer package loader: KenLM trie binary + 'TRIE' header + ConstFst.
//
// Replaces Scorer::init_from_filepath / load_lm_* / load_trie_impl (native_client/ctcdecode/scorer.cpp:40-222),
// KenLM's binary-format reader for the trie family (kenlm/lm/binary_format.cc:22-75,193-239;
// lm/search_trie.cc:546-571; lm/quantize.cc:54-73; lm/bhiksha.cc:35-84; lm/vocab.cc:113-124,218-232)
// and ConstFst::Read (openfst-1.6.7/src/include/fst/const-fst.h:195-235, src/lib/fst.cc:57-84).
// The KenLM blob goes to HBM byte for byte; kernels read the same bit-packed records KenLM mmaps.
// Probing-hash models (model types 0/1) are rejected with STT_ERR_SCORER_INVALID_LM.
#include <cmath>
#include <cstring>
#include <fstream>
#include <vector>

#include "../../include/coqui-stt.h"
#include "engine.h"

namespace {
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline float rdf(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }
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
}  // namespace

uint64_t stt_murmur64a(const void* key, size_t len) { return murmur64a(key, len, 0); }

int ScorerDev::LoadFile(const std::string& path, const Alphabet& alphabet) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) return STT_ERR_SCORER_UNREADABLE;
  const std::streamsize sz = in.tellg();
  in.seekg(0);
  std::vector<char> data((size_t)sz);
  if (sz > 0 && !in.read(data.data(), sz)) return STT_ERR_SCORER_UNREADABLE;
  return LoadBuffer(data.data(), data.size(), alphabet);
}

int ScorerDev::LoadBuffer(const char* data, size_t len, const Alphabet& alphabet) {
  return Parse(reinterpret_cast<const uint8_t*>(data), len, alphabet.GetSpaceLabel());
}

int ScorerDev::Parse(const uint8_t* buf, size_t len, int space_label) {
  static const char kMagic[] = "mmap lm http://kheafield.com/code format version 5\n";
  if (len < 88 + 20 || memcmp(buf, kMagic, sizeof(kMagic)) != 0) return STT_ERR_SCORER_INVALID_LM;
  const uint8_t* fp = buf + 88;  // FixedWidthParameters after the 88-byte Sanity block
  const int ord = fp[0];
  const int model_type = (int)rd32(fp + 8);
  if (ord < 2 || ord > STT_KENLM_MAX_ORDER) return STT_ERR_SCORER_INVALID_LM;
  if (model_type < 2 || model_type > 5) return STT_ERR_SCORER_INVALID_LM;
  const bool quant = (model_type == 3 || model_type == 5), array = (model_type == 4 || model_type == 5);
  uint64_t counts[STT_KENLM_MAX_ORDER];
  if (len < 108 + 8 * (size_t)ord) return STT_ERR_SCORER_INVALID_LM;
  for (int i = 0; i < ord; ++i) counts[i] = rd64(buf + 108 + 8 * i);

  struct BP { uint64_t base_off; uint8_t word_bits, total_bits, quant_bits, next_bits; uint64_t off_begin_off; uint32_t off_count; uint64_t entries; };
  BP mid[STT_KENLM_MAX_ORDER - 2]{}; BP lon{};
  uint64_t off = align8(108 + 8 * (uint64_t)ord);
  if (off + 8 > len) return STT_ERR_SCORER_INVALID_LM;
  const uint64_t vocab_n = rd64(buf + off);
  const uint64_t vocab_off = off + 8;
  off += 8 + 8 * counts[0];
  uint8_t prob_bits = 0, backoff_bits = 0;
  uint64_t qprob_off[STT_KENLM_MAX_ORDER]{}, qback_off[STT_KENLM_MAX_ORDER]{};
  if (quant) {
    if (off + 8 > len) return STT_ERR_SCORER_INVALID_LM;
    prob_bits = buf[off + 1]; backoff_bits = buf[off + 2];
    if (!prob_bits || !backoff_bits || prob_bits > 25 || backoff_bits > 25) return STT_ERR_SCORER_INVALID_LM;
    uint64_t t = off + 8;
    for (int i = 0; i < ord - 2; ++i) { qprob_off[i] = t; t += 4ULL << prob_bits; qback_off[i] = t; t += 4ULL << backoff_bits; }
    qprob_off[ord - 2] = t; t += 4ULL << prob_bits;
    off = t;
  }
  const uint64_t unigram_off = off;
  off += (counts[0] + 2) * 16;
  if (off > len) return STT_ERR_SCORER_INVALID_LM;
  uint8_t cfg_bhiksha = 0;
  if (array && ord > 2) { if (off + 2 > len) return STT_ERR_SCORER_INVALID_LM; cfg_bhiksha = buf[off + 1]; }
  const uint8_t middle_quant_bits = quant ? (uint8_t)(prob_bits + backoff_bits) : 63;
  const uint8_t longest_bits = quant ? prob_bits : 31;
  for (int i = 0; i < ord - 2; ++i) {
    const uint64_t entries = counts[i + 1], max_vocab = counts[0], max_next = counts[i + 2];
    uint64_t bh_size = 0; uint8_t inline_bits;
    BP& m = mid[i];
    if (array) {
      const uint8_t required = required_bits(max_next), chop = chop_bits(entries + 1, max_next, cfg_bhiksha);
      const uint64_t array_count = (max_next >> (required - chop)) + 1;
      bh_size = 8 * (1 + array_count) + 7;
      inline_bits = required - chop;
      m.off_begin_off = align8(off) + 8; m.off_count = (uint32_t)array_count; m.entries = entries;
    } else {
      inline_bits = required_bits(max_next);
    }
    m.base_off = off + bh_size;
    m.word_bits = required_bits(max_vocab); m.quant_bits = middle_quant_bits; m.next_bits = inline_bits;
    m.total_bits = (uint8_t)(m.word_bits + middle_quant_bits + inline_bits);
    off += bh_size + (((1 + entries) * m.total_bits + 7) / 8 + 8);
  }
  lon.base_off = off; lon.word_bits = required_bits(counts[0]); lon.total_bits = (uint8_t)(lon.word_bits + longest_bits);
  off += ((1 + counts[ord - 1]) * lon.total_bits + 7) / 8 + 8;
  const uint64_t lm_end = off;  // GetEndOfSearchOffset, lm/model.cc:265-267
  if (lm_end > len) return STT_ERR_SCORER_INVALID_LM;
  if (len <= lm_end) return STT_ERR_SCORER_NO_TRIE;

  // ---- package trailer (scorer.cpp:177-222)
  const uint8_t* p = buf + lm_end;
  if (lm_end + 25 > len || rd32(p) != 0x54524945u) return STT_ERR_SCORER_INVALID_TRIE;
  if ((int)rd32(p + 4) != 6) return STT_ERR_SCORER_VERSION_MISMATCH;
  const bool utf8 = p[8] != 0;
  double a, b; memcpy(&a, p + 9, 8); memcpy(&b, p + 17, 8);
  uint64_t o = lm_end + 25;
  if (o + 4 > len || rd32(buf + o) != 2125659606u) return STT_ERR_SCORER_INVALID_TRIE;
  o += 4;
  uint32_t l = rd32(buf + o); o += 4 + l;
  l = rd32(buf + o); o += 4 + l;
  o += 4;
  const uint32_t flags = rd32(buf + o); o += 4;
  o += 8;
  int64_t fst_start, nstates, narcs;
  memcpy(&fst_start, buf + o, 8); o += 8; memcpy(&nstates, buf + o, 8); o += 8; memcpy(&narcs, buf + o, 8); o += 8;
  if (flags & 3) return STT_ERR_SCORER_INVALID_TRIE;
  if (flags & 4) o = (o + 15) & ~(uint64_t)15;
  const uint8_t* states = buf + o; o += (uint64_t)nstates * 20;
  if (flags & 4) o = (o + 15) & ~(uint64_t)15;
  const uint8_t* arcs = buf + o; o += (uint64_t)narcs * 16;
  if (o > len || nstates <= 0) return STT_ERR_SCORER_INVALID_TRIE;

  // ---- repack the dictionary: arcs sorted by ilabel within a state (SortedMatcher's precondition)
  // The arc's second field is the dictionary state of the child prefix: Start() when the arc's target is final
  // (a completed word, path_trie.cpp:79-87), the target otherwise -- resolved here so the kernel needs one read.
  std::vector<uint32_t> pos((size_t)nstates + 1);
  std::vector<uint8_t> fin((size_t)nstates);
  std::vector<uint2> arcv((size_t)narcs);
  std::vector<uint8_t> has_space((size_t)nstates + 1, 0);
  for (int64_t s = 0; s < nstates; ++s) fin[s] = !(rdf(states + 20 * s) == INFINITY);  // Final(s) != TropicalWeight::Zero()
  uint32_t w = 0;
  for (int64_t s = 0; s < nstates; ++s) {
    const uint8_t* S = states + 20 * s;
    const uint32_t ap = rd32(S + 4), an = rd32(S + 8);
    pos[s] = w;
    for (uint32_t k = 0; k < an; ++k) {
      const uint8_t* A = arcs + 16 * (uint64_t)(ap + k);
      if ((uint64_t)ap + k >= (uint64_t)narcs) return STT_ERR_SCORER_INVALID_TRIE;
      const uint32_t next = rd32(A + 12);
      if (next >= (uint64_t)nstates) return STT_ERR_SCORER_INVALID_TRIE;
      arcv[w++] = make_uint2(rd32(A), fin[next] ? (uint32_t)fst_start : next);
      if (space_label >= 0 && (int64_t)rd32(A) == (int64_t)space_label + 1) has_space[s] = 1;
    }
  }
  pos[nstates] = w;

  // ---- vocabulary hash table over KenLM's sorted hash array (index = position + 1, vocab.hh:72-83)
  uint32_t vt_n = 16;
  while ((uint64_t)vt_n < 2 * vocab_n + 2) vt_n <<= 1;
  std::vector<DevVocabSlot> vtab(vt_n, DevVocabSlot{0, 0, 0, 0.f, 0.f, 0, 0});
  const bool uni_ok = ord >= 2 && counts[1] < 0xFFFFFFFFull;
  if (vocab_off + 8 * vocab_n > len) return STT_ERR_SCORER_INVALID_LM;
  for (uint64_t i = 0; i < vocab_n; ++i) {
    const uint64_t h = rd64(buf + vocab_off + 8 * i);
    uint32_t slot = (uint32_t)h & (vt_n - 1);
    while (vtab[slot].used) slot = (slot + 1) & (vt_n - 1);
    vtab[slot] = DevVocabSlot{h, (uint32_t)(i + 1), 1, 0.f, 0.f, 0, 0};
  }

  // ---- upload
  blob_.upload(buf, lm_end + 16);  // +16: the 64-bit bit-packed reads may touch up to 8 bytes past the last record
  fst_pos_.upload(pos.data(), pos.size() * 4);
  for (auto& e : vtab) {
    if (!e.used) continue;
    const uint8_t* u = buf + unigram_off + 16 * (uint64_t)e.index;
    e.prob = rdf(u); e.backoff = rdf(u + 4);
    e.begin = (uint32_t)rd64(u + 8); e.end = (uint32_t)rd64(u + 24);
  }
  vtab_.upload(vtab.data(), vtab.size() * sizeof(DevVocabSlot));
  fst_arcs_.upload(arcv.data(), arcv.size() * sizeof(uint2));
  fst_space_.upload(has_space.data(), has_space.size());
  const uint8_t* d = blob_.as<uint8_t>();
  DevScorer ds{};
  ds.enabled = 1; ds.order = ord; ds.quant = quant; ds.utf8 = utf8;
  ds.alpha = (double)(float)a; ds.beta = (double)(float)b;  // Scorer::reset_params(float, float)
  ds.vocab = reinterpret_cast<const uint64_t*>(d + vocab_off); ds.vocab_n = vocab_n;
  ds.vtab = vtab_.as<DevVocabSlot>(); ds.vtab_mask = vt_n - 1; ds.uni_in_vtab = uni_ok ? 1 : 0;
  ds.unigram = d + unigram_off;
  for (int i = 0; i < STT_KENLM_MAX_ORDER; ++i) {
    ds.qprob[i] = quant && qprob_off[i] ? reinterpret_cast<const float*>(d + qprob_off[i]) : nullptr;
    ds.qbackoff[i] = quant && qback_off[i] ? reinterpret_cast<const float*>(d + qback_off[i]) : nullptr;
  }
  // Bhiksha hint tables: the offsets array is sorted; a coarse direct-index table replaces KenLM's std::upper_bound over it
  // (lm/bhiksha.hh:76-95) by one read plus a short forward scan.  Same result, fewer dependent HBM reads.
  std::vector<uint32_t> hints;
  size_t hint_off[STT_KENLM_MAX_ORDER - 2] = {};
  uint32_t hint_shift[STT_KENLM_MAX_ORDER - 2] = {};
  for (int i = 0; i < ord - 2; ++i) {
    if (!mid[i].off_begin_off) continue;
    const uint64_t* offs = reinterpret_cast<const uint64_t*>(buf + mid[i].off_begin_off);
    const uint32_t cnt = mid[i].off_count;
    const uint64_t entries = mid[i].entries + 2;
    uint32_t sh = 0;
    while (((entries >> sh) > 4ull * cnt + 1024) && sh < 40) ++sh;  // about four table slots per offset
    hint_off[i] = hints.size(); hint_shift[i] = sh;
    const uint64_t slots = (entries >> sh) + 2;
    uint32_t pos = 0;  // upper_bound(offs, v) - 1 for increasing v
    for (uint64_t j = 0; j < slots; ++j) {
      const uint64_t v = j << sh;
      while (pos + 1 < cnt && offs[pos + 1] <= v) ++pos;
      hints.push_back(pos);
    }
  }
  if (hints.empty()) hints.push_back(0);
  hint_.upload(hints.data(), hints.size() * 4);
  auto fill = [&](DevBitPacked& o2, const BP& b2) {
    o2.base = d + b2.base_off; o2.word_bits = b2.word_bits; o2.total_bits = b2.total_bits; o2.quant_bits = b2.quant_bits; o2.next_bits = b2.next_bits;
    o2.word_mask = (1ULL << b2.word_bits) - 1; o2.next_mask = (1ULL << b2.next_bits) - 1;
    o2.off_begin = b2.off_begin_off ? reinterpret_cast<const uint64_t*>(d + b2.off_begin_off) : nullptr; o2.off_count = b2.off_count;
    o2.off_hint = nullptr; o2.hint_shift = 0; o2.max_word = counts[0];
  };
  for (int i = 0; i < ord - 2; ++i) {
    fill(ds.middle[i], mid[i]);
    if (mid[i].off_begin_off) { ds.middle[i].off_hint = hint_.as<uint32_t>() + hint_off[i]; ds.middle[i].hint_shift = hint_shift[i]; }
  }
  fill(ds.longest, lon);
  ds.prob_bits = prob_bits; ds.backoff_bits = backoff_bits;
  ds.prob_mask = (1u << prob_bits) - 1; ds.backoff_mask = (1u << backoff_bits) - 1;
  // <s> index and backoff (lm/model.cc:115-124)
  {
    const uint64_t h = murmur64a("<s>", 3, 0);
    const uint64_t* v = reinterpret_cast<const uint64_t*>(buf + vocab_off);
    uint64_t lo = 0, hi = vocab_n; uint32_t idx = 0;
    while (lo < hi) { const uint64_t m2 = lo + (hi - lo) / 2; if (v[m2] < h) lo = m2 + 1; else if (v[m2] > h) hi = m2; else { idx = (uint32_t)(m2 + 1); break; } }
    ds.bos_index = idx;
    ds.bos_backoff = rdf(buf + unigram_off + 16 * (uint64_t)idx + 4);
  }
  ds.fst_start = (int)fst_start;
  ds.fst_state_pos = fst_pos_.as<uint32_t>(); ds.fst_arcs = fst_arcs_.as<uint2>(); ds.fst_has_space = fst_space_.as<uint8_t>();
  dev = ds;
  is_utf8 = utf8; order = ord; blob_bytes = lm_end;
  return STT_ERR_OK;
}
