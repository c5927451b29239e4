// stt_amd/tools/scorer_tools.cpp -- host-side scorer packaging.
//
//   stt_scorer_tools package  --lm lm.binary --vocab vocab.txt (--alphabet alphabet.txt | --bytes) --package out.scorer
//                             [--default_alpha A --default_beta B]
//       Restates native_client/generate_scorer_package.cpp:18-106: the KenLM binary is copied, then the 'TRIE' trailer
//       (scorer.cpp:224-269) and the vocabulary dictionary as an OpenFst ConstFst<StdArc> are appended.  The reference
//       builds the dictionary with RmEpsilon/Determinize/Minimize (scorer.cpp:398-437); here the words go into a trie
//       that is minimised by merging identical sub-trees, which yields the same minimal deterministic acceptor
//       (state numbering differs; the decoder only follows arcs).
//
//   stt_scorer_tools synth-lm --words N --order K --seed S --out lm.binary --vocab-out vocab.txt [--codepoints 1]
//       (--codepoints: the "words" are N distinct three-byte UTF-8 code points from U+4E00 on -- the units of a code-point level
//        language model for a bytes-output scorer, doc/DECODER.rst:193; package the result with `package --bytes`)
//                             [--avg2 a --avg3 b --avg4 c --avg5 d]
//       Writes a synthetic KenLM language model directly in the binary format KenLM's `build_binary -a 255 -q 8 -v trie`
//       produces (model type QUANT_ARRAY_TRIE, no vocabulary strings): the benchmark needs a huge-vocabulary scorer and
//       neither a corpus nor `lmplz` exists offline.  Layout restated from kenlm/lm/binary_format.cc:22-75,
//       lm/vocab.cc:113-124 (sorted 64-bit MurmurHash64A of the words, index = rank + 1), lm/quantize.cc:54-95,
//       lm/trie.cc:46-139 (bit-packed reverse trie), lm/bhiksha.cc:35-95 (array-compressed next pointers).
//       The n-grams are random but structurally sound: word frequencies are Zipfian, every (n+1)-gram extends a stored
//       n-gram on both sides, and the "has extension" backoff markers (lm/blank.hh:15-23) are exact.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

uint64_t murmur64a(const void* key, size_t len, uint64_t seed = 0) {  // kenlm/util/murmur_hash.cc
  const uint64_t m = 0xc6a4a7935bd1e995ULL; const int r = 47;
  uint64_t h = seed ^ (len * m);
  const uint8_t* d = (const uint8_t*)key; const uint8_t* end = d + (len / 8) * 8;
  while (d != end) { uint64_t k; memcpy(&k, d, 8); d += 8; k *= m; k ^= k >> r; k *= m; h ^= k; h *= m; }
  uint64_t t = 0; const size_t rem = len & 7;
  for (size_t i = 0; i < rem; ++i) t |= (uint64_t)d[i] << (8 * i);
  if (rem) { h ^= t; h *= m; }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

std::map<std::string, std::string> parse_args(int argc, char** argv, int first) {
  std::map<std::string, std::string> a;
  for (int i = first; i < argc; ++i) {
    std::string k = argv[i];
    if (k.rfind("--", 0) != 0) { std::cerr << "unexpected argument " << k << "\n"; exit(2); }
    k = k.substr(2);
    if (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) a[k] = argv[++i]; else a[k] = "1";
  }
  return a;
}

// ------------------------------------------------------------------------------------------------ bit writer
struct BitBuf {
  std::vector<uint8_t> b;
  explicit BitBuf(size_t bytes) : b(bytes, 0) {}
  void put(uint64_t bit_off, unsigned bits, uint64_t v) {  // util::WriteInt57: little endian, <= 57 bits
    if (!bits) return;
    uint64_t cur; memcpy(&cur, &b[bit_off >> 3], 8);
    cur |= v << (bit_off & 7);
    memcpy(&b[bit_off >> 3], &cur, 8);
  }
};
uint8_t required_bits(uint64_t v) { if (!v) return 0; uint8_t r = 1; while (v >>= 1) ++r; return r; }
uint8_t chop_bits(uint64_t max_offset, uint64_t max_next, uint8_t cfg_bits) {  // lm/bhiksha.cc:35-50
  const uint8_t required = required_bits(max_next);
  uint8_t best = 0; int64_t lowest = INT64_MAX;
  const uint8_t lim = required < cfg_bits ? required : cfg_bits;
  for (uint8_t chop = 0; chop <= lim; ++chop) {
    const int64_t change = (int64_t)((max_next >> (required - chop)) * 64) - (int64_t)max_offset * (int64_t)chop;
    if (change < lowest) { lowest = change; best = chop; }
  }
  return best;
}

// ================================================================================================ synth-lm
struct Level {               // one n-gram order in reverse-trie order
  std::vector<uint32_t> word;    // the word this entry adds on the left (the entry's key inside its parent's range)
  std::vector<uint64_t> next;    // first child in the next level
  std::vector<uint32_t> parent;  // entry of the previous level (for order 2: the unigram)
};

std::string make_word(std::mt19937_64& rng) {
  static const char* onset[] = {"b","c","d","f","g","h","j","k","l","m","n","p","r","s","t","v","w","br","ch","cl","cr","dr","fl","fr","gr","pl","pr","sh","sl","sp","st","str","th","tr","wh",""};
  static const char* nucleus[] = {"a","e","i","o","u","ai","ea","ee","io","oo","ou","ie","y"};
  static const char* coda[] = {"","","","n","r","s","t","l","m","d","ng","nd","nt","st","rs","ck","ll","ss","ly","er","ed","es","'s"};
  const int syl = 1 + (int)(rng() % 4);
  std::string w;
  for (int i = 0; i < syl; ++i) {
    w += onset[rng() % (sizeof(onset) / sizeof(*onset))];
    w += nucleus[rng() % (sizeof(nucleus) / sizeof(*nucleus))];
    if (i + 1 == syl || rng() % 3 == 0) w += coda[rng() % (sizeof(coda) / sizeof(*coda))];
  }
  return w;
}

int synth_lm(const std::map<std::string, std::string>& a) {
  auto get = [&](const char* k, const char* d) { auto it = a.find(k); return it == a.end() ? std::string(d) : it->second; };
  const uint64_t n_words = std::stoull(get("words", "100000"));
  const int order = std::stoi(get("order", "5"));
  const uint64_t seed = std::stoull(get("seed", "1"));
  const std::string out = get("out", ""), vocab_out = get("vocab-out", "");
  if (out.empty() || order < 2 || order > 6) { std::cerr << "synth-lm: need --out and 2 <= --order <= 6\n"; return 2; }
  double avg[7] = {0, 0, std::stod(get("avg2", "8")), std::stod(get("avg3", "1.0")), std::stod(get("avg4", "0.6")), std::stod(get("avg5", "0.5")), std::stod(get("avg6", "0.4"))};
  std::mt19937_64 rng(seed);
  const bool codepoints = get("codepoints", "0") != "0";
  if (codepoints && n_words > 20000) { std::cerr << "synth-lm: --codepoints takes at most 20000 units (U+4E00 .. U+9C1F)\n"; return 2; }

  // ---- vocabulary: distinct pronounceable pseudo-words; index = rank of the MurmurHash64A + 1 (lm/vocab.hh:72-83)
  std::unordered_set<std::string> seen = {"<s>", "</s>"};
  std::vector<std::string> words = {"<s>", "</s>"};
  if (codepoints) {   // unit i = U+4E00 + i as UTF-8 (three bytes: E4 B8 80 ...), in a seeded random order so that the Zipfian ranks are spread over the block
    std::vector<uint32_t> cps(n_words);
    for (uint32_t i = 0; i < n_words; ++i) cps[i] = 0x4E00u + i;
    std::shuffle(cps.begin(), cps.end(), rng);
    for (uint32_t cp : cps) { const char u[4] = {(char)(0xE0 | (cp >> 12)), (char)(0x80 | ((cp >> 6) & 0x3F)), (char)(0x80 | (cp & 0x3F)), 0}; words.push_back(std::string(u, 3)); }
  }
  while (words.size() < n_words + 2) { std::string w = make_word(rng); if (w.size() <= 15 && seen.insert(w).second) words.push_back(w); }
  const uint64_t V = words.size() + 1;  // + <unk> (index 0)
  std::vector<std::pair<uint64_t, uint32_t>> hashed(words.size());
  for (uint32_t i = 0; i < words.size(); ++i) hashed[i] = {murmur64a(words[i].data(), words[i].size()), i};
  std::sort(hashed.begin(), hashed.end());
  for (size_t i = 1; i < hashed.size(); ++i) if (hashed[i].first == hashed[i - 1].first) { std::cerr << "hash collision, change --seed\n"; return 1; }
  std::vector<uint32_t> id_of(words.size());  // generation order -> KenLM word index
  for (uint32_t r = 0; r < hashed.size(); ++r) id_of[hashed[r].second] = r + 1;
  const uint32_t bos = id_of[0], eos = id_of[1];
  // Zipfian sampler over the generation order (word 2 is the most frequent); returns a KenLM word index
  std::vector<double> cdf(words.size());
  { double s = 0; for (size_t i = 0; i < words.size(); ++i) { s += 1.0 / std::pow((double)(i < 2 ? 3 : i - 1), 1.05); cdf[i] = s; } for (auto& x : cdf) x /= s; }
  std::uniform_real_distribution<double> U(0.0, 1.0);
  auto zipf = [&]() { const size_t i = std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin(); return id_of[std::min(i, words.size() - 1)]; };

  // ---- n-grams, level by level in reverse-trie order: an entry of level n under parent p adds one word on the left
  std::vector<Level> lv(order + 1);
  // level 2: predecessors of every word
  std::vector<std::vector<uint32_t>> pred(V);  // pred[w] = sorted words v with (v w) stored
  {
    std::poisson_distribution<int> P(avg[2]);
    for (uint32_t w = 1; w < V; ++w) {
      if (w == bos) continue;  // nothing precedes <s>
      int k = 1 + P(rng);
      std::vector<uint32_t>& pv = pred[w];
      for (int j = 0; j < k; ++j) { uint32_t v = (rng() % 7 == 0) ? bos : zipf(); if (v != eos) pv.push_back(v); }
      std::sort(pv.begin(), pv.end()); pv.erase(std::unique(pv.begin(), pv.end()), pv.end());
    }
    for (uint32_t w = 0; w < V; ++w) for (uint32_t v : pred[w]) { lv[2].word.push_back(v); lv[2].parent.push_back(w); }
  }
  std::vector<uint64_t> uni_next(V + 1, 0);
  { uint64_t c = 0; for (uint32_t w = 0; w < V; ++w) { uni_next[w] = c; c += pred[w].size(); } uni_next[V] = c; }
  // entry (level n) -> the entry of level n-1 that spells the same words minus the rightmost one, needed to find the
  // candidates for the next word on the left: children(ctx) where ctx = entry without its last (rightmost) word.
  // Kept implicitly: for level 2 entry (v w) the context entry is unigram v; deeper levels carry `ctx`.
  std::vector<std::vector<uint64_t>> ctx(order + 1);  // ctx[n][e] = entry index in level n-1 (n >= 3) / word (n == 2)
  ctx[2].assign(lv[2].word.begin(), lv[2].word.end());
  auto child_range = [&](int level, uint64_t e, uint64_t& b0, uint64_t& b1) {  // children of entry e of `level` (in level+1)
    if (level == 1) { b0 = uni_next[e]; b1 = uni_next[e + 1]; }
    else { b0 = lv[level].next[e]; b1 = lv[level].next[e + 1]; }
  };
  for (int n = 3; n <= order; ++n) {
    Level& L = lv[n]; const Level& Pp = lv[n - 1];
    const uint64_t np = Pp.word.size();
    lv[n - 1].next.assign(np + 1, 0);
    std::vector<uint64_t> cx;
    for (uint64_t e = 0; e < np; ++e) {
      lv[n - 1].next[e] = L.word.size();
      // entry e spells (x1 .. x_{n-1}); candidates for x0 are the words on the left of the stored (n-1)-gram (x1 .. x_{n-2}),
      // i.e. the children of ctx[n-1][e]
      uint64_t b0, b1;
      child_range(n - 2, ctx[n - 1][e], b0, b1);
      if (b1 <= b0) continue;
      const double want = avg[n];
      uint64_t k = (uint64_t)want; if (U(rng) < want - (double)k) ++k;
      if (Pp.word[e] == bos) k = 0;  // nothing on the left of <s>
      k = std::min<uint64_t>(k, b1 - b0);
      if (!k) continue;
      std::vector<uint64_t> pick;
      if (k * 2 >= b1 - b0) { for (uint64_t c = b0; c < b1 && pick.size() < k; ++c) pick.push_back(c); }
      else { while (pick.size() < k) { uint64_t c = b0 + rng() % (b1 - b0); if (std::find(pick.begin(), pick.end(), c) == pick.end()) pick.push_back(c); } std::sort(pick.begin(), pick.end()); }
      for (uint64_t c : pick) {  // children of a range are sorted by word, so `pick` ascending keeps the new range sorted
        const uint32_t x0 = (n - 1 == 2) ? lv[2].word[c] : lv[n - 1].word[c];
        L.word.push_back(x0); L.parent.push_back((uint32_t)e); cx.push_back(c);
      }
    }
    lv[n - 1].next[np] = L.word.size();
    ctx[n] = std::move(cx);
  }

  // ---- "has extension" markers: an n-gram read left to right is a context iff some stored (n+1)-gram starts with it.
  // In reverse-trie terms entry e of level n+1 with context entry c = ctx[n+1][e] (level n): the (n+1)-gram's first n
  // words are exactly the n-gram stored at c.
  std::vector<std::vector<uint8_t>> ext(order + 1);
  ext[1].assign(V, 0);
  for (int n = 2; n <= order; ++n) ext[n].assign(lv[n].word.size(), 0);
  for (uint32_t v : lv[2].word) ext[1][v] = 1;
  for (int n = 3; n <= order; ++n) for (uint64_t c : ctx[n]) ext[n - 1][c] = 1;

  // ---- file layout (see scorer_dev.cpp, the reader)
  uint64_t counts[7] = {0};
  counts[0] = V;
  for (int n = 2; n <= order; ++n) counts[n - 1] = lv[n].word.size();
  const uint8_t prob_bits = 8, backoff_bits = 8, cfg_bhiksha = 255;
  const uint8_t word_bits = required_bits(V);
  std::vector<uint8_t> file;
  auto append = [&](const void* p, size_t n) { const uint8_t* c = (const uint8_t*)p; file.insert(file.end(), c, c + n); };
  auto pad8 = [&]() { while (file.size() % 8) file.push_back(0); };
  {  // Sanity (88 bytes, lm/binary_format.cc:47-63) + FixedWidthParameters (20 bytes)
    uint8_t s[88] = {0};
    static const char magic[] = "mmap lm http://kheafield.com/code format version 5\n";
    memcpy(s, magic, sizeof(magic));
    const float f3[3] = {0.0f, 1.0f, -0.5f}; memcpy(s + 56, f3, 12);
    const uint32_t w2[2] = {1u, 0xFFFFFFFFu}; memcpy(s + 68, w2, 8);
    const uint64_t one = 1; memcpy(s + 80, &one, 8);
    append(s, 88);
    uint8_t fp[20] = {0};
    fp[0] = (uint8_t)order; const float pm = 1.5f; memcpy(fp + 4, &pm, 4);
    const int32_t mt = 5; memcpy(fp + 8, &mt, 4);  // QUANT_ARRAY_TRIE
    fp[12] = 0;                                     // has_vocabulary = false (build_binary -v)
    const uint32_t sv = 1; memcpy(fp + 16, &sv, 4);
    append(fp, 20);
    append(counts, 8 * (size_t)order);
    pad8();
  }
  {  // SortedVocabulary: count, then the sorted hashes; reserved size is for counts[0] entries
    const uint64_t n = hashed.size(); append(&n, 8);
    for (auto& h : hashed) append(&h.first, 8);
    const uint64_t zero = 0; append(&zero, 8);  // counts[0] = n + 1 slots
  }
  // quantisation tables: [version, prob_bits, backoff_bits, 0...] then per order 2..N-1 prob + backoff bins, order N prob bins
  std::vector<float> pbin(1u << prob_bits), bbin(1u << backoff_bits);
  for (size_t i = 0; i < pbin.size(); ++i) pbin[i] = -7.0f + 6.9f * (float)i / (float)(pbin.size() - 1);  // log10 probs in [-7, -0.1]
  bbin[0] = -0.0f; bbin[1] = 0.0f;  // kNoExtensionBackoff, kExtensionBackoff (lm/quantize.cc:80-81)
  for (size_t i = 2; i < bbin.size(); ++i) bbin[i] = -2.5f + 2.45f * (float)(i - 2) / (float)(bbin.size() - 3);
  {
    uint8_t hdr[8] = {2, prob_bits, backoff_bits, 0, 0, 0, 0, 0};
    append(hdr, 8);
    for (int n = 2; n < order; ++n) { append(pbin.data(), 4 * pbin.size()); append(bbin.data(), 4 * bbin.size()); }
    append(pbin.data(), 4 * pbin.size());
  }
  {  // unigrams: {prob, backoff, next} x (V + 1), plus one spare record (Unigram::Size = (count + 2) * 16)
    for (uint64_t w = 0; w <= V; ++w) {
      float prob = w == 0 ? -6.5f : -1.5f - 4.5f * (float)U(rng);
      if (w == bos) prob = -99.0f;
      float backoff = (w < V && ext[1][w]) ? -0.1f - 1.5f * (float)U(rng) : -0.0f;
      if (w >= V) { prob = 0; backoff = 0; }
      const uint64_t next = uni_next[std::min<uint64_t>(w, V)];
      append(&prob, 4); append(&backoff, 4); append(&next, 8);
    }
    const uint8_t z[16] = {0}; append(z, 16);
  }
  for (int n = 2; n <= order; ++n) {
    const Level& L = lv[n];
    const uint64_t entries = L.word.size();
    if (n < order) {  // middle: [bhiksha offsets][records: word | backoff q | prob q | next low bits]
      const uint64_t max_next = counts[n];
      const uint8_t required = required_bits(max_next), chop = chop_bits(entries + 1, max_next, cfg_bhiksha);
      const uint64_t array_count = (max_next >> (required - chop)) + 1;
      const uint8_t inline_bits = required - chop;
      const uint8_t total_bits = word_bits + prob_bits + backoff_bits + inline_bits;
      const size_t base = file.size();
      const size_t bh_size = 8 * (1 + array_count) + 7;
      file.resize(base + bh_size, 0);
      file[base] = 0; file[base + 1] = cfg_bhiksha;  // kArrayBhikshaVersion, configured bits (lm/bhiksha.cc:88-94)
      const size_t offs_at = ((base + 7) & ~(size_t)7) + 8;
      std::vector<uint64_t> offs(array_count, 0);
      uint64_t write_to = 1;  // offsets[0] = 0
      BitBuf bb(((1 + entries) * total_bits + 7) / 8 + 8);
      for (uint64_t e = 0; e <= entries; ++e) {
        const uint64_t bit = e * total_bits;
        const uint64_t value = L.next.empty() ? 0 : L.next[e];
        if (e < entries) {
          bb.put(bit, word_bits, L.word[e]);
          const uint64_t bq = ext[n][e] ? 2 + rng() % (bbin.size() - 2) : 0;
          const uint64_t pq = rng() % pbin.size();
          bb.put(bit + word_bits, backoff_bits + prob_bits, bq | (pq << backoff_bits));
        }
        const uint64_t encode = value >> inline_bits;  // ArrayBhiksha::WriteNext (lm/bhiksha.hh:97-101)
        for (; write_to <= encode && write_to < array_count; ++write_to) offs[write_to] = e;
        bb.put(bit + word_bits + backoff_bits + prob_bits, inline_bits, value & ((1ULL << inline_bits) - 1));
      }
      for (; write_to < array_count; ++write_to) offs[write_to] = entries + 1;
      memcpy(&file[offs_at], offs.data(), 8 * array_count);
      append(bb.b.data(), bb.b.size());
    } else {  // longest: [records: word | prob q]
      const uint8_t total_bits = word_bits + prob_bits;
      BitBuf bb(((1 + entries) * total_bits + 7) / 8 + 8);
      for (uint64_t e = 0; e < entries; ++e) {
        bb.put(e * total_bits, word_bits, L.word[e]);
        bb.put(e * total_bits + word_bits, prob_bits, rng() % pbin.size());
      }
      append(bb.b.data(), bb.b.size());
    }
  }
  std::ofstream f(out, std::ios::binary);
  f.write((const char*)file.data(), (std::streamsize)file.size());
  if (!f) { std::cerr << "cannot write " << out << "\n"; return 1; }
  if (!vocab_out.empty()) {
    std::ofstream vf(vocab_out);
    for (size_t i = 2; i < words.size(); ++i) vf << words[i] << "\n";
  }
  std::cerr << "synth-lm: " << n_words << " words, order " << order << ", n-grams";
  for (int n = 1; n <= order; ++n) std::cerr << " " << counts[n - 1];
  std::cerr << ", " << file.size() << " bytes\n";
  return 0;
}

// ================================================================================================ package
struct TNode { std::map<uint32_t, uint32_t> next; bool final_ = false; };

int package(const std::map<std::string, std::string>& a) {
  auto get = [&](const char* k, const char* d) { auto it = a.find(k); return it == a.end() ? std::string(d) : it->second; };
  const std::string lm = get("lm", ""), vocab = get("vocab", ""), alphabet = get("alphabet", ""), out = get("package", "");
  const bool bytes_mode = a.count("bytes") != 0;
  const float alpha = std::stof(get("default_alpha", "0.0")), beta = std::stof(get("default_beta", "0.0"));
  if (lm.empty() || vocab.empty() || out.empty() || (!bytes_mode && alphabet.empty())) {
    std::cerr << "package: need --lm --vocab --package and --alphabet or --bytes\n"; return 2;
  }
  // label strings -> label + 1 (scorer.cpp:88-105 setup_char_map; label 0 is reserved for epsilon)
  std::unordered_map<std::string, uint32_t> char_map;
  uint32_t space_label = 0;
  if (bytes_mode) {
    for (uint32_t i = 0; i < 255; ++i) char_map[std::string(1, (char)(i + 1))] = i + 1;  // UTF8Alphabet: label i = byte i+1 (alphabet.h:83-91)
  } else {
    std::ifstream af(alphabet);
    if (!af) { std::cerr << "Invalid alphabet file " << alphabet << "\n"; return 1; }
    std::string line; uint32_t idx = 0;
    while (std::getline(af, line)) {  // alphabet.cc:42-68
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line == "\\#") line = "#";
      else if (!line.empty() && line[0] == '#') continue;
      if (line == " ") space_label = idx + 1;
      char_map[line] = ++idx;
    }
    if (!space_label) { std::cerr << "alphabet has no space label\n"; return 1; }
  }
  std::unordered_set<std::string> words;
  {
    std::ifstream vf(vocab);
    if (!vf) { std::cerr << "Invalid vocabulary file " << vocab << "\n"; return 1; }
    std::string w; while (vf >> w) words.insert(w);
  }
  std::cerr << words.size() << " unique words read from vocabulary file.\n";
  // trie over label sequences (decoder_utils.cpp:90-132: word mode appends the space label, words with characters
  // outside the alphabet are skipped)
  std::vector<TNode> tn(1);
  size_t added = 0;
  for (const std::string& w : words) {
    if (w == "<s>" || w == "</s>" || w == "<unk>") continue;
    std::vector<uint32_t> seq;
    bool ok = true;
    for (size_t i = 0; i < w.size() && ok;) {
      size_t len = 1;
      if (!bytes_mode) { const unsigned char c = (unsigned char)w[i]; len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1; }
      auto it = char_map.find(w.substr(i, len));
      if (it == char_map.end()) ok = false; else seq.push_back(it->second);
      i += len;
    }
    if (!ok || seq.empty()) continue;
    if (!bytes_mode) seq.push_back(space_label);
    uint32_t cur = 0;
    for (uint32_t c : seq) {
      auto it = tn[cur].next.find(c);
      if (it == tn[cur].next.end()) { tn.push_back(TNode()); tn[cur].next[c] = (uint32_t)tn.size() - 1; cur = (uint32_t)tn.size() - 1; }
      else cur = it->second;
    }
    tn[cur].final_ = true;
    ++added;
  }
  // minimise: canonical id per sub-tree signature, children first (nodes were appended parent before child)
  std::vector<uint32_t> canon(tn.size());
  std::unordered_map<std::string, uint32_t> sig;
  std::vector<uint32_t> reps;
  for (size_t i = tn.size(); i-- > 0;) {
    std::string key(1, tn[i].final_ ? 'F' : 'N');
    for (auto& kv : tn[i].next) { const uint32_t pr[2] = {kv.first, canon[kv.second]}; key.append((const char*)pr, 8); }
    auto it = sig.find(key);
    if (it == sig.end()) { sig.emplace(std::move(key), (uint32_t)reps.size()); canon[i] = (uint32_t)reps.size(); reps.push_back((uint32_t)i); }
    else canon[i] = it->second;
  }
  // number states from the start state, breadth first
  std::vector<int64_t> state_of(reps.size(), -1);
  std::vector<uint32_t> order_;
  state_of[canon[0]] = 0; order_.push_back(canon[0]);
  for (size_t q = 0; q < order_.size(); ++q)
    for (auto& kv : tn[reps[order_[q]]].next) { const uint32_t c = canon[kv.second]; if (state_of[c] < 0) { state_of[c] = (int64_t)order_.size(); order_.push_back(c); } }
  const int64_t nstates = (int64_t)order_.size();
  int64_t narcs = 0;
  for (uint32_t c : order_) narcs += (int64_t)tn[reps[c]].next.size();

  // ---- write: LM copy, trailer (scorer.cpp:236-264), ConstFst (const-fst.h:236-290 WriteFst, aligned)
  std::vector<uint8_t> file;
  { std::ifstream lf(lm, std::ios::binary); if (!lf) { std::cerr << "Can't open binary LM file.\n"; return 1; }
    file.assign(std::istreambuf_iterator<char>(lf), std::istreambuf_iterator<char>()); }
  auto append = [&](const void* p, size_t n) { const uint8_t* c = (const uint8_t*)p; file.insert(file.end(), c, c + n); };
  const int32_t MAGIC = 0x54524945, FILE_VERSION = 6;
  append(&MAGIC, 4); append(&FILE_VERSION, 4);
  const uint8_t utf8 = bytes_mode ? 1 : 0; append(&utf8, 1);
  const double da = (double)alpha, db = (double)beta; append(&da, 8); append(&db, 8);
  const int32_t fst_magic = 2125659606; append(&fst_magic, 4);
  auto put_str = [&](const char* s) { const int32_t l = (int32_t)strlen(s); append(&l, 4); append(s, (size_t)l); };
  put_str("const"); put_str("standard");
  const int32_t version = 1, flags = 4;  // kAlignedFileVersion, FstHeader::IS_ALIGNED
  append(&version, 4); append(&flags, 4);
  // property bits as in the reference's own packages (acceptor, deterministic, epsilon-free, label-sorted, ...)
  const uint64_t props = bytes_mode ? 0x0000a56a5a950001ULL : 0x0000a5aa5a950001ULL;
  append(&props, 8);
  const int64_t start = 0; append(&start, 8); append(&nstates, 8); append(&narcs, 8);
  auto pad16 = [&]() { while (file.size() % 16) file.push_back(0); };
  pad16();
  {
    uint32_t pos = 0;
    for (uint32_t c : order_) {
      const TNode& t = tn[reps[c]];
      const float fw = t.final_ ? 0.0f : INFINITY;  // TropicalWeight::One() / Zero()
      const uint32_t na = (uint32_t)t.next.size(), z = 0;
      append(&fw, 4); append(&pos, 4); append(&na, 4); append(&z, 4); append(&z, 4);
      pos += na;
    }
  }
  pad16();
  for (uint32_t c : order_)
    for (auto& kv : tn[reps[c]].next) {
      const int32_t l = (int32_t)kv.first, ns = (int32_t)state_of[canon[kv.second]];
      const float w = 0.0f;
      append(&l, 4); append(&l, 4); append(&w, 4); append(&ns, 4);
    }
  std::ofstream f(out, std::ios::binary);
  f.write((const char*)file.data(), (std::streamsize)file.size());
  if (!f) { std::cerr << "Error when saving package in " << out << ".\n"; return 1; }
  std::cerr << "Package created in " << out << " (" << added << " words, " << nstates << " states, " << narcs << " arcs).\n";
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "usage: stt_scorer_tools {package|synth-lm} --option value ...\n"; return 2; }
  const std::string cmd = argv[1];
  const auto args = parse_args(argc, argv, 2);
  if (cmd == "package") return package(args);
  if (cmd == "synth-lm") return synth_lm(args);
  std::cerr << "unknown command " << cmd << "\n";
  return 2;
}
