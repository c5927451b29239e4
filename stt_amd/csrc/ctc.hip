// stt_amd/csrc/ctc.hip -- CTC prefix beam search with KenLM/FST scorer as gfx950 kernels.
//
// Replaces DecoderState::{next,decode} (native_client/ctcdecode/ctc_beam_search_decoder.cpp:112-326),
// PathTrie (path_trie.cpp:37-209), the Scorer query side (scorer.cpp:272-396) and the KenLM trie
// lookup (kenlm/lm/model.cc:170-176,285-338; lm/trie.cc:32-99; lm/bhiksha.hh:76-95;
// lm/quantize.hh:152-209) plus OpenFst's SortedMatcher::Find on the dictionary (matcher.h:347-386).
//
// One 1024-thread workgroup per stream walks the timesteps of its chunk; a timestep is
//   A/B class log-probs (glibc-exact logf), optional class sort/cut-off        get_pruned_emissions :328-358
//       LDS hash of the live prefixes' path keys
//   P2  expand: blank / repeat events per live prefix, then one work item per (prefix, FST out-arc) found through a
//       workgroup prefix sum                                                    :150-207, path_trie.cpp:37-100
//   P3  language-model scores of boundary extensions (trie walk in HBM); in word mode one KenLM FullScore from the
//       cached state of the previous word boundary                            :209-243, scorer.cpp:308-396
//       (bitmap step, round 6: the LM waves work off a list made when the beam was written and are waited for through a counter;
//        code-point step: FullScore through the bigram blocks -- lm_full_score_blocks)
//   P4  merge the <=3 events of every live prefix in the reference's visiting order :166-193,245-253
//   P5  scores (iterate_to_vec); bucket histogram + prefix sum finds the beam_size-th key, kept keys are ranked inside
//       their bucket segment (rank == position in the new beam)               path_trie.cpp:159-190, :263-274
//   P6  (fused with the ranking) write the new beam, append arena nodes
// Float arithmetic uses sttmath.h (bit-exact glibc expf/logf), so scores are bit-identical to the
// reference; ties the reference leaves to libstdc++ are broken by (live-before-new, beam index).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <stdexcept>
#include <string>

#include "ctc.h"
// The code-point step's cold paths (lm_score: the generic label walk; the trie walk of a scorer without the hashed index) and its FullScore
// wrapper used to be REAL calls: the compiler kept lm_word_query_cached<false, true>, lm_score and is_scoring_boundary out of line and
// kenlm_full_score_call was __noinline__ on purpose ("kept out of the step's registers", round 4).  Measured at the end of round 6
// (benchmarks/r06_nocalls_check.sh): the calling convention cost far more than it kept out -- ctc_next_kernel<2, ...> 122 spilled vector
// registers and 1168 - 1184 B of scratch per lane with the calls, 21 and 480 B with everything inlined (<1, ...>: 53 / 176 B -> 7 - 14 / 32 - 60 B;
// <4, ...>: 1 / 8 B -> none), the bytes workload 43.3 -> 38.2 ms per batch, bit-identical.  -DSTT_DEVICE_CALLS restores the calls.
#ifdef STT_DEVICE_CALLS
#define STT_CALLS_INLINE
#define STT_CALLS_NOINLINE __noinline__
#else
#define STT_CALLS_INLINE __forceinline__
#define STT_CALLS_NOINLINE __forceinline__
#endif
#include "lmindex.h"
#include "tuning.h"
#include "sttmath.h"

using sttm::stt_log_sum_exp;
using sttm::stt_logf;

// Every LDS pointer carries address space 3: the accesses compile to ds_read/ds_write/ds_add (32-bit addresses, LDS
// counter only) instead of flat_* instructions, which resolve the aperture at run time and wait on both counters.
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t lds_add(LDS_AS uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_add(LDS_AS int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_sub(LDS_AS uint32_t* p, uint32_t v) { return __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_or(LDS_AS int* p, int v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_min(LDS_AS uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_max(LDS_AS uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// returns the previous value (0 = the slot was empty and now holds `desired`)
__device__ __forceinline__ uint64_t lds_cas0(LDS_AS uint64_t* p, uint64_t desired) {
  uint64_t expected = 0;
  __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return expected;
}

// The stream's HBM arrays as explicit global-memory (address space 1) pointers: global_load/global_store instead of
// flat_*, so waiting for an LDS result does not also wait for the HBM reads in flight (flat ops count on both counters).
#define GLB_AS __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// The stream's pointers and capacities are NOT copied into registers when the kernel starts (26 scalar registers live from the first
// instruction to the last were the larger half of the kernel's 256 spilled scalars): they are read from the DecStream itself --
// through the constant address space, i.e. with scalar loads -- where they are used.  The kernel never writes these fields.
#define CONST_AS __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ const CONST_AS T* launder(const CONST_AS T* q) { asm volatile("" : "+s"(q)); return q; }
struct GStream {
  const CONST_AS DecStream* g;
  __device__ __forceinline__ GLB_AS uint64_t* pa() const { return (GLB_AS uint64_t*)g->pa; }   // {parent (low word), character / timestep (high word)}
  __device__ __forceinline__ GLB_AS uint64_t* ta() const { return (GLB_AS uint64_t*)g->ta; }
  __device__ __forceinline__ GLB_AS uint32_t* pq() const { return (GLB_AS uint32_t*)g->pq; }
  __device__ __forceinline__ GLB_AS u32x4* be() const { return (GLB_AS u32x4*)g->be; }         // BEntry = 4 x 16 bytes
  __device__ __forceinline__ GLB_AS float* c_logp() const { return (GLB_AS float*)g->c_logp; }
  __device__ __forceinline__ GLB_AS uint32_t* c_pi() const { return (GLB_AS uint32_t*)g->c_pi; }
  __device__ __forceinline__ GLB_AS int* c_fst() const { return (GLB_AS int*)g->c_fst; }
  __device__ __forceinline__ GLB_AS uint64_t* sel_keys() const { return (GLB_AS uint64_t*)g->sel_keys; }
  __device__ __forceinline__ GLB_AS uint32_t* c_elem() const { return (GLB_AS uint32_t*)g->c_key; }   // code-point step: element ids of the candidates that reach the selection
  __device__ __forceinline__ const uint2* pa_generic() const { return g->pa; }                  // for the uncached scorer paths
  __device__ __forceinline__ uint32_t cand_cap() const { return g->cand_cap; }
  __device__ __forceinline__ uint32_t pa_cap() const { return g->pa_cap; }
  __device__ __forceinline__ uint32_t ta_cap() const { return g->ta_cap; }
  __device__ __forceinline__ uint32_t be_cap() const { return g->be_cap; }
};
union BEntryBits { BEntry e; u32x4 q[4]; __device__ BEntryBits() {} };
__device__ __forceinline__ BEntry load_be(const GStream& S, uint32_t idx) {
  BEntryBits b;
#pragma unroll
  for (int i = 0; i < 4; ++i) b.q[i] = S.be()[(size_t)idx * 4 + i];
  return b.e;
}
__device__ __forceinline__ void store_be(const GStream& S, uint32_t idx, const BEntry& e) {
  BEntryBits b; b.e = e;
#pragma unroll
  for (int i = 0; i < 4; ++i) S.be()[(size_t)idx * 4 + i] = b.q[i];
}
__device__ __forceinline__ double load_be_raw(const GStream& S, uint32_t idx) {  // BEntry::raw is the first 8 bytes
  const GLB_AS double* p = (const GLB_AS double*)(S.be() + (size_t)idx * 4);
  return *p;
}
__device__ __forceinline__ uint2 load_node(const GLB_AS uint64_t* a, uint32_t idx) { const uint64_t v = a[idx]; return make_uint2((uint32_t)v, (uint32_t)(v >> 32)); }
__device__ __forceinline__ void store_node(GLB_AS uint64_t* a, uint32_t idx, uint32_t x, uint32_t y) { a[idx] = (uint64_t)x | ((uint64_t)y << 32); }

#define NTHREADS 1024
#define ABSENT_BITS 0xFFFFFFFFu
#define OOV_SCORE_D (-1000.0)  // scorer.h:16

// ------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ uint64_t ld64u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ bool is_absent(float x) { return __float_as_uint(x) == ABSENT_BITS; }
__device__ __forceinline__ float absent() { return __uint_as_float(ABSENT_BITS); }

// Path key of "parent + label": one multiply-xorshift round per label (a 64-bit FNV-1a style rolling hash with a final
// fold; the search kernel is instruction-issue bound and every expand item computes one of these).
// COLLISIONS.  The key is a 63-bit hash of the label sequence, not the sequence: two different prefixes with one key would be taken for
// the same prefix by the LDS hash (a child of one merged into the other).  Chance: a lookup meets a foreign equal key with probability
// <= live prefixes / 2^63; a step makes <= beam x classes lookups -- 500 x 500 x 28 / 2^63 = 7.6e-13 per step, 2e-10 per 5 s utterance.
// Guard (collision_guard below): on every hit the found entry's LAST LABEL is compared with the label that was looked up -- equal for a
// true child by construction; a foreign prefix with the same key has a different last label in C - 1 of C cases -- and a mismatch raises
// error bit 0x20: the stream's results are refused (check_decoder_errors), never silently wrong.  What is left undetected is a
// collision between prefixes that also end in the same label: 1 / C of the figure above.  `kmask` (DecParams::key_mask, tunable
// debug_key_bits) truncates the keys so that a test can watch the guard fire (tests/test_gpu_errors.py).
__device__ __forceinline__ uint64_t child_key(uint64_t parent_key, uint32_t c, uint64_t kmask) {
  uint64_t x = (parent_key ^ (uint64_t)(c + 1)) * 0x9E3779B97F4A7C15ULL;
  x ^= x >> 29;
  return (x & kmask) | 1ULL;  // 0 is the empty marker of the LDS hash
}

// selection key: ascending key order == (score desc, character asc, live before new, beam index asc)
__device__ __forceinline__ uint64_t sel_key(float score, uint32_t ch, uint32_t is_new, uint32_t idx) {
  uint32_t u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
  const uint32_t c16 = ch == STT_ROOT_CH ? 0xFFFFu : (ch & 0xFFFFu);
  return ((uint64_t)(~u) << 32) | ((uint64_t)c16 << 11) | ((uint64_t)is_new << 10) | (uint64_t)idx;
}

// ------------------------------------------------------------------------------------ KenLM trie query
struct KNode { uint64_t begin, end; };

__device__ __forceinline__ uint64_t read_int57(const uint8_t* base, uint64_t bit_off, uint64_t mask) {
  return (ld64u(base + (bit_off >> 3)) >> (bit_off & 7)) & mask;
}
__device__ __forceinline__ bool has_extension(float backoff) { return __float_as_uint(backoff) != 0x80000000u; }

// SortedVocabulary::Index (kenlm/lm/vocab.hh:72-83): position of the hash in the sorted array + 1, 0 = <unk>.  The
// open-addressing table built at load time (scorer_dev.cpp) answers the same question in 1-2 probes.
__device__ __forceinline__ uint32_t vocab_index(const DevScorer& s, uint64_t h, unsigned& probes) {
  uint32_t slot = (uint32_t)h & s.vtab_mask;
  for (;;) {
    const DevVocabSlot e = s.vtab[slot];
    ++probes;
    if (!e.used) return 0;
    if (e.hash == h) return e.index;
    slot = (slot + 1) & s.vtab_mask;
  }
}
// FindBitPacked (lm/trie.cc:32-36) = BoundedSortedUniformFind (util/sorted_uniform.hh:64-84) with the integer pivot of
// Pivot32/Pivot64 (:26-40): interpolation search over the word field, about log log n probes for KenLM's near-uniform
// word ids.  Keys are unique within [begin, end), so any correct search returns the same position.
__device__ __forceinline__ bool find_bitpacked(const DevBitPacked& bp, uint64_t begin, uint64_t end, uint64_t key, uint64_t& at, unsigned& probes) {
  int64_t before_it = (int64_t)begin - 1, after_it = (int64_t)end;
  uint64_t before_v = 0, after_v = bp.max_word;
  while (after_it - before_it > 1) {
    const uint64_t width = (uint64_t)(after_it - before_it - 1);
    const uint64_t off = key - before_v, range = after_v - before_v;
    // word ids are 32-bit (lm::WordIndex) and a range never holds 2^32 entries, so the 64-bit product cannot overflow;
    // the guard only keeps a corrupt file from running away (plain bisection then)
    const uint64_t step = (off <= 0xFFFFFFFFull && width <= 0xFFFFFFFFull) ? (off * width) / (range + 1) : width / 2;
    const int64_t pivot = before_it + 1 + (int64_t)step;
    const uint64_t v = read_int57(bp.base, (uint64_t)pivot * bp.total_bits, bp.word_mask);
    ++probes;
    if (v < key) { before_it = pivot; before_v = v; }
    else if (v > key) { after_it = pivot; after_v = v; }
    else { at = (uint64_t)pivot; return true; }
  }
  return false;
}
__device__ __forceinline__ void read_next(const DevBitPacked& m, uint64_t bit_offset, uint64_t index, KNode& out, unsigned& probes) {
  if (!m.off_begin) {
    out.begin = read_int57(m.base, bit_offset, m.next_mask);
    out.end = read_int57(m.base, bit_offset + m.total_bits, m.next_mask);
    probes += 2;
    return;
  }
  // ArrayBhiksha::ReadNext (lm/bhiksha.hh:76-95): bi = upper_bound(offsets, index) - 1, found from the hint table
  const uint64_t lo = read_int57(m.base, bit_offset, m.next_mask), hi = read_int57(m.base, bit_offset + m.total_bits, m.next_mask);
  uint32_t bi = m.off_hint[index >> m.hint_shift];
  ++probes;
  while (bi + 1 < m.off_count && m.off_begin[bi + 1] <= index) { ++bi; ++probes; }
  uint32_t ei = bi + 1;
  while (ei < m.off_count && m.off_begin[ei] <= index + 1) { ++ei; ++probes; }
  --ei;
  out.begin = ((uint64_t)bi << m.next_bits) | lo;
  out.end = ((uint64_t)ei << m.next_bits) | hi;
  probes += 2;
}
__device__ bool lookup_middle(const DevScorer& s, int om2, uint32_t word, KNode& node, bool& independent_left, float& prob, float& backoff, unsigned& probes) {
  const DevBitPacked& m = s.middle[om2];
  uint64_t at;
  if (!find_bitpacked(m, node.begin, node.end, word, at, probes)) { independent_left = true; return false; }
  const uint64_t addr = at * m.total_bits + m.word_bits;
  read_next(m, addr + m.quant_bits, at, node, probes);
  independent_left = (node.begin == node.end);
  if (s.quant) {
    backoff = s.qbackoff[om2][(ld32u(m.base + (addr >> 3)) >> (addr & 7)) & s.backoff_mask];
    const uint64_t pa = addr + s.backoff_bits;
    prob = s.qprob[om2][(ld32u(m.base + (pa >> 3)) >> (pa & 7)) & s.prob_mask];
  } else {
    prob = __uint_as_float((uint32_t)(ld64u(m.base + (addr >> 3)) >> (addr & 7)) | 0x80000000u);
    const uint64_t ba = addr + 31;
    backoff = __uint_as_float((uint32_t)(ld64u(m.base + (ba >> 3)) >> (ba & 7)));
  }
  probes += 2;
  return true;
}
__device__ bool lookup_longest(const DevScorer& s, uint32_t word, const KNode& node, float& prob, unsigned& probes) {
  const DevBitPacked& l = s.longest;
  uint64_t at;
  if (!find_bitpacked(l, node.begin, node.end, word, at, probes)) return false;
  const uint64_t addr = at * l.total_bits + l.word_bits;
  if (s.quant) prob = s.qprob[s.order - 2][(ld32u(l.base + (addr >> 3)) >> (addr & 7)) & s.prob_mask];
  else prob = __uint_as_float((uint32_t)(ld64u(l.base + (addr >> 3)) >> (addr & 7)) | 0x80000000u);
  ++probes;
  return true;
}
// vocabulary lookup that also hands back the slot (the word's unigram record rides along in it)
__device__ __forceinline__ uint32_t vocab_slot(const DevScorer& s, uint64_t h, DevVocabSlot& out, unsigned& probes) {
  uint32_t slot = (uint32_t)h & s.vtab_mask;
  for (;;) {
    const DevVocabSlot e = s.vtab[slot];
    ++probes;
    if (!e.used) { out = e; return 0; }
    if (e.hash == h) { out = e; return e.index; }
    slot = (slot + 1) & s.vtab_mask;
  }
}

// HashedSearch (kenlm/lm/search_hashed.hh:26-29,107-125): the node of an n-gram is the chained hash of its word indices; a middle / longest
// record is found by linear probing from key % buckets (util/probing_hash_table.hh: DivMod), an empty bucket has key 0.
__device__ __forceinline__ uint64_t combine_word_hash(uint64_t current, uint32_t next) {
  return (current * 8978948897894561157ULL) ^ ((uint64_t)(1 + next) * 17894857484156487943ULL);
}
__device__ __forceinline__ const uint8_t* probing_find(const uint8_t* table, uint64_t buckets, int stride, uint64_t key, unsigned& probes) {
  uint64_t i = key % buckets;
  for (uint64_t n = 0; n < buckets; ++n) {   // (a well-formed table always holds an empty bucket: Size() takes entries + 1 at least)
    const uint8_t* e = table + i * (uint64_t)stride;
    const uint64_t got = ld64u(e);
    ++probes;
    if (got == key) return e;
    if (got == 0) return nullptr;
    if (++i == buckets) i = 0;
  }
  return nullptr;
}
// GenericModel<HashedSearch<...>, ProbingVocabulary>::FullScore: the same ScoreExceptBackoff / ResumeScore (model.cc:285-338) as the trie form
// below, with HashedSearch's lookups.  (Rest costs of REST_PROBING models are never read: FullScore's prob is what the scorer uses.)
__device__ float kenlm_full_score_probing(const DevScorer& s, const KState& in, uint32_t new_word, KState& out, unsigned& probes, int* ngram_length) {
  const uint8_t* u = s.unigram + (uint64_t)s.p_wstride * new_word;
  const uint32_t praw = ld32u(u);
  float prob = __uint_as_float(praw | 0x80000000u);
  out.backoff[0] = __uint_as_float(ld32u(u + 4));
  ++probes;
  bool independent_left = (praw & 0x80000000u) != 0;
  uint64_t node = (uint64_t)new_word;
  int nl = 1;
  int out_len = has_extension(out.backoff[0]) ? 1 : 0;
  out.words[0] = new_word;
  bool go = in.length != 0;
#pragma unroll
  for (int om2 = 0; om2 < STT_KENLM_MAX_ORDER - 1; ++om2) {
    if (om2 + 1 < STT_KENLM_MAX_ORDER - 1) { out.words[om2 + 1] = in.words[om2]; out.backoff[om2 + 1] = 0.0f; }
    if (go) {
      if (om2 == in.length || independent_left) go = false;
      else if (om2 == s.order - 2) {
        go = false;
        const uint8_t* e = probing_find(s.p_lon, s.p_lon_buckets, 12, combine_word_hash(node, in.words[om2]), probes);
        if (e) { prob = __uint_as_float(ld32u(e + 8)); nl = s.order; }
      } else if (om2 < STT_KENLM_MAX_ORDER - 2) {
        node = combine_word_hash(node, in.words[om2]);
        const uint8_t* e = probing_find(s.p_mid[om2], s.p_mid_buckets[om2], s.p_estride, node, probes);
        if (!e) go = false;     // (LookupMiddle sets independent_left; nothing reads it after a miss)
        else {
          const uint32_t pr = ld32u(e + 8);
          const float b = __uint_as_float(ld32u(e + 12));
          independent_left = (pr & 0x80000000u) != 0;
          out.backoff[om2 + 1] = b; prob = __uint_as_float(pr | 0x80000000u); nl = om2 + 2;
          if (has_extension(b)) out_len = nl;
        }
      }
    }
  }
  if (ngram_length) *ngram_length = nl;
  out.length = out_len;
#pragma unroll
  for (int i = 0; i < STT_KENLM_MAX_ORDER - 1; ++i)
    if (i >= nl - 1 && i < in.length) prob = __fadd_rn(prob, in.backoff[i]);
  return prob;
}

// GenericModel::FullScore (model.cc:170-176) = ScoreExceptBackoff (:285-310) + ResumeScore (:312-338).
// Written with compile-time indices only (fully unrolled over KENLM_MAX_ORDER) so that both states stay in registers.
__device__ __forceinline__ float kenlm_full_score(const DevScorer& s, const KState& in, uint32_t new_word, KState& out, unsigned& probes,
                                                  const DevVocabSlot* uni = nullptr, int* ngram_length = nullptr) {
  if (s.probing) return kenlm_full_score_probing(s, in, new_word, out, probes, ngram_length);
  KNode node;
  float prob;
  if (uni) {  // unigram record copied into the vocabulary slot at load time
    prob = uni->prob; out.backoff[0] = uni->backoff; node.begin = uni->begin; node.end = uni->end;
  } else {
    const uint8_t* u = s.unigram + 16 * (uint64_t)new_word;
    const uint64_t pb = ld64u(u);  // {float prob, float backoff, uint64 next}; the three loads are independent
    node.begin = ld64u(u + 8);
    node.end = ld64u(u + 24);
    prob = __uint_as_float((uint32_t)pb);
    out.backoff[0] = __uint_as_float((uint32_t)(pb >> 32));
    probes += 2;
  }
  bool independent_left = (node.begin == node.end);
  int nl = 1;
  int out_len = has_extension(out.backoff[0]) ? 1 : 0;
  out.words[0] = new_word;
  bool go = in.length != 0, at_longest = false;
#pragma unroll
  for (int om2 = 0; om2 < STT_KENLM_MAX_ORDER - 1; ++om2) {  // history word index hi == om2
    if (om2 + 1 < STT_KENLM_MAX_ORDER - 1) { out.words[om2 + 1] = in.words[om2]; out.backoff[om2 + 1] = 0.0f; }  // entries past length are unused
    if (go) {
      if (om2 == in.length || independent_left) go = false;
      else if (om2 == s.order - 2) { at_longest = true; go = false; float p; if (lookup_longest(s, in.words[om2], node, p, probes)) { prob = p; nl = s.order; } }
      else if (om2 < STT_KENLM_MAX_ORDER - 2) {
        float p, b;
        if (!lookup_middle(s, om2, in.words[om2], node, independent_left, p, b, probes)) go = false;
        else { out.backoff[om2 + 1] = b; prob = p; nl = om2 + 2; if (has_extension(b)) out_len = nl; }
      }
    }
  }
  (void)at_longest;
  if (ngram_length) *ngram_length = nl;
  out.length = out_len;
#pragma unroll
  for (int i = 0; i < STT_KENLM_MAX_ORDER - 1; ++i)
    if (i >= nl - 1 && i < in.length) prob = __fadd_rn(prob, in.backoff[i]);
  return prob;
}

// MurmurHash64A(seed 0) of the UTF-8 bytes of labels labs[nl-1], ..., labs[0]  (labels were collected backwards)
__device__ uint64_t hash_labels_reversed(const DevAlphabet& al, const uint32_t* labs, int nl) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  size_t len = 0;
  for (int i = 0; i < nl; ++i) { const uint32_t c = labs[i]; len += al.label_off[c] - (c ? al.label_off[c - 1] : 0); }
  uint64_t h = 0 ^ (len * m);
  uint64_t k = 0;
  int nb = 0;
  for (int i = nl - 1; i >= 0; --i) {
    const uint32_t c = labs[i];
    const int b0 = c ? al.label_off[c - 1] : 0, b1 = al.label_off[c];
    for (int b = b0; b < b1; ++b) {
      k |= (uint64_t)al.label_bytes[b] << (8 * nb);
      if (++nb == 8) {
        k *= m; k ^= k >> r; k *= m;
        h ^= k; h *= m;
        k = 0; nb = 0;
      }
    }
  }
  if (nb) { h ^= k; h *= m; }  // the tail switch of MurmurHash64A xors the remaining bytes little-endian, then multiplies
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

#define MAX_UNIT_LABELS 64  // longest word (in labels) the device reconstructs; dictionary words are far shorter

// Scorer::make_ngram (scorer.cpp:370-396) + get_log_cond_prob (:308-344) + hot words (ctc_beam_search_decoder.cpp:224-236).
// The prefix to score is `first` (a virtual last label, or STT_ROOT_CH for none) on top of path-arena node `node`.
// Returns log_cond_prob + hot_boost as a double; the caller multiplies by alpha and rounds to float exactly like the
// reference's `float score = (get_log_cond_prob(...) + hot_boost) * alpha`.
__device__ STT_CALLS_INLINE double lm_score(const DevScorer& s, const DevAlphabet& al, const uint2* pa, uint32_t node, uint32_t first, bool with_hot, unsigned& probes) {
  uint64_t hashes[STT_KENLM_MAX_ORDER];
  int n = 0;
  uint32_t cur = node;
  uint32_t pending = first;  // virtual top-of-path label
  uint32_t labs[MAX_UNIT_LABELS];
  for (int order = 0; order < s.order; ++order) {
    // current_node == nullptr || character == ROOT
    uint32_t cur_ch;
    if (pending != STT_ROOT_CH) cur_ch = pending;
    else { if (cur == STT_ROOT_CH) break; cur_ch = pa[cur].y; ++probes; }
    if (cur_ch == STT_ROOT_CH) break;
    int nl = 0;
    // walk back to the unit's stop node (space/root in word mode, first byte of the codepoint in utf8 mode)
    for (;;) {
      uint32_t c;
      if (pending != STT_ROOT_CH) c = pending; else { c = pa[cur].y; ++probes; }
      if (s.utf8) {
        if (c == STT_ROOT_CH) break;  // stop = root, nothing pushed
        if (nl < MAX_UNIT_LABELS) labs[nl++] = c;
        const uint8_t fb = al.label_bytes[c ? al.label_off[c - 1] : 0];
        const bool boundary = (fb & 0xC0) != 0x80;
        // move to the parent either way: if boundary, stop = this node and current = stop->parent
        if (pending != STT_ROOT_CH) pending = STT_ROOT_CH; else cur = pa[cur].x;
        if (boundary) break;
      } else {
        if (c == (uint32_t)al.space_id || c == STT_ROOT_CH) {
          // stop = this node; current = stop->parent
          if (pending != STT_ROOT_CH) pending = STT_ROOT_CH; else cur = pa[cur].x;
          break;
        }
        if (nl < MAX_UNIT_LABELS) labs[nl++] = c;
        if (pending != STT_ROOT_CH) pending = STT_ROOT_CH; else cur = pa[cur].x;
        if (cur == STT_ROOT_CH) break;  // unreachable for well-formed arenas (root has character ROOT)
      }
    }
    hashes[n++] = hash_labels_reversed(al, labs, nl);
  }
  // hashes[] is newest-first; the reference reverses to oldest-first
  float hot_boost = 0.0f;
  if (with_hot && s.n_hot) {
    for (int i = n - 1; i >= 0; --i)
      for (int j = 0; j < s.n_hot; ++j)
        if (hashes[i] == s.hot_hash[j]) hot_boost = __fadd_rn(hot_boost, s.hot_boost[j]);
  }
  const bool bos = n < s.order;
  KState a, b;
  KState* in = &a; KState* out = &b;
  in->length = 0;
  if (bos) { in->length = 1; in->words[0] = s.bos_index; in->backoff[0] = s.bos_backoff; }
  double cond_prob = 0.0;
  bool oov = false;
  for (int i = n - 1; i >= 0; --i) {
    const uint32_t wi = vocab_index(s, hashes[i], probes);
    if (wi == 0) { oov = true; break; }
    cond_prob = (double)kenlm_full_score(s, *in, wi, *out, probes);
    KState* t = in; in = out; out = t;
  }
  const double lcp = oov ? OOV_SCORE_D : __ddiv_rn(cond_prob, (double)0.4342944819f);  // / NUM_FLT_LOGE (a float constant)
  return __dadd_rn(lcp, (double)hot_boost);
}

// Scorer::is_scoring_boundary (scorer.cpp:272-299) for the prefix (`first` on top of `node`) and label `new_label`
__device__ STT_CALLS_INLINE bool is_scoring_boundary(const DevScorer& s, const DevAlphabet& al, const uint2* pa, uint32_t node, uint32_t first, uint32_t new_label, unsigned& probes) {
  if (!s.utf8) return (int)new_label == al.space_id;
  uint32_t cur = node, pending = first;
  int dist = 0;
  uint8_t first_byte = 0;
  bool found = false;
  for (;;) {
    uint32_t c;
    if (pending != STT_ROOT_CH) c = pending; else { if (cur == STT_ROOT_CH) break; c = pa[cur].y; ++probes; }
    if (c == STT_ROOT_CH) break;  // prefix->character == -1 -> false / walked past the first label
    const uint8_t fb = al.label_bytes[c ? al.label_off[c - 1] : 0];
    dist += 1;
    if ((fb & 0xC0) != 0x80) { first_byte = (uint8_t)((uint8_t)c + 1); found = true; break; }
    if (pending != STT_ROOT_CH) pending = STT_ROOT_CH; else cur = pa[cur].x;
  }
  if (!found) return false;
  int needed;
  if ((first_byte >> 3) == 0x1E) needed = 4;
  else if ((first_byte >> 4) == 0x0E) needed = 3;
  else if ((first_byte >> 5) == 0x06) needed = 2;
  else if ((first_byte >> 7) == 0x00) needed = 1;
  else return false;
  return dist == needed;
}

// ------------------------------------------------------------------------------------ utf8-mode scorer cache
// Narrow beams keep, per prefix, the code point in progress ("run": the bytes since the prefix's last start byte,
// b0 | b1 << 8 | b2 << 16, their count in bits 24..31 saturating at 255; 0 = no start byte yet), so that
// Scorer::is_scoring_boundary (scorer.cpp:272-299) is a table lookup instead of a walk up the path, and the score of a
// completed code point is ONE FullScore from the boundary entry (BEntry) of the previous code point instead of
// make_ngram + max_order FullScores (scorer.cpp:301-345, 228-270).  Byte sequences that are not well-formed UTF-8 make
// the prefix's boundary entry STT_NONE and take the generic walk.
__device__ __forceinline__ uint32_t utf8_unit_len(uint32_t first_byte) {  // scorer.cpp:283-295
  if ((first_byte >> 3) == 0x1E) return 4;
  if ((first_byte >> 4) == 0x0E) return 3;
  if ((first_byte >> 5) == 0x06) return 2;
  if ((first_byte >> 7) == 0x00) return 1;
  return 0;
}
__device__ __forceinline__ uint32_t utf8_child_run(uint32_t run, uint8_t byte) {
  if ((byte & 0xC0) != 0x80) return (uint32_t)byte | (1u << 24);
  const uint32_t len = run >> 24;
  if (len == 0) return 0;  // a continuation byte with no start byte below it: distance_to_codepoint_boundary finds none
  uint32_t r = run & 0x00FFFFFFu;
  if (len < 3) r |= (uint32_t)byte << (8 * len);
  return r | ((len < 255u ? len + 1u : 255u) << 24);
}
// does `byte` on top of a prefix with this run end a code point?
__device__ __forceinline__ bool utf8_completes(uint32_t run, uint8_t byte) {
  const uint32_t cr = utf8_child_run(run, byte);
  const uint32_t need = utf8_unit_len(cr & 0xFFu);
  return (cr >> 24) != 0 && need != 0 && (cr >> 24) == need;
}
// Is the prefix + byte still a sequence of whole, well-formed code points plus at most one in progress (so that the
// parent's boundary entry describes everything before the code point `byte` belongs to)?  `unit` = the bytes of that
// code point so far, first byte lowest.
__device__ __forceinline__ bool utf8_step_clean(uint32_t run, bool parent_root, uint8_t byte, uint32_t& unit) {
  const uint32_t plen = run >> 24, pneed = utf8_unit_len(run & 0xFFu);
  unit = byte;
  if ((byte & 0xC0) != 0x80) return (parent_root ? plen == 0 : (plen != 0 && plen == pneed)) && utf8_unit_len(byte) != 0;
  unit = (run & 0x00FFFFFFu) | ((uint32_t)byte << (8 * (plen & 3u)));
  return plen >= 1 && plen < pneed;
}

// ------------------------------------------------------------------------------------ word-mode scorer cache
__device__ __forceinline__ int word_nbytes(uint64_t lo, uint64_t hi) {
  if (hi) return 8 + (64 - __clzll((long long)hi) + 7) / 8;
  return lo ? (64 - __clzll((long long)lo) + 7) / 8 : 0;
}
// append one byte to a packed word (all ones = overflow, sticky)
__device__ __forceinline__ void word_push(uint64_t& lo, uint64_t& hi, uint8_t b) {
  if ((lo & hi) == ~0ULL) return;
  const int n = word_nbytes(lo, hi);
  if (n < 8) lo |= (uint64_t)b << (8 * n);
  else if (n < 16) hi |= (uint64_t)b << (8 * (n - 8));
  else { lo = ~0ULL; hi = ~0ULL; }
}
// MurmurHash64A (seed 0) of a word of nbytes <= 16 bytes packed first-byte-lowest in (lo, hi)
__device__ __forceinline__ uint64_t murmur_packed(uint64_t lo, uint64_t hi, int nbytes) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint64_t h = 0 ^ ((uint64_t)nbytes * m);
  if (nbytes >= 8) {
    uint64_t k = lo;
    k *= m; k ^= k >> r; k *= m;
    h ^= k; h *= m;
    if (nbytes == 16) { k = hi; k *= m; k ^= k >> r; k *= m; h ^= k; h *= m; }
    else if (nbytes > 8) { h ^= hi; h *= m; }
  } else if (nbytes > 0) {
    h ^= lo; h *= m;
  }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}
// Walk back from `node` to the previous word boundary collecting the word's bytes (newest label first: shifting each
// byte in from the low end leaves the word first-byte-lowest).  More than 16 bytes -> all ones.
__device__ __forceinline__ void word_walk(const DevAlphabet& al, const GStream& S, const LDS_AS uint8_t* lab1, uint32_t node, uint64_t& lo, uint64_t& hi, unsigned& probes) {
  lo = 0; hi = 0;
  int nbytes = 0;
  for (uint32_t cur = node; cur != STT_ROOT_CH;) {
    const uint2 pn = load_node(S.pa(), cur);
    ++probes;
    if (pn.y == (uint32_t)al.space_id || pn.y == STT_ROOT_CH) break;
    const uint8_t one = lab1 ? lab1[pn.y] : (uint8_t)0;  // LDS copy of single-byte labels (0 / no table = read the label from HBM)
    if (one) {
      hi = (hi << 8) | (lo >> 56);
      lo = (lo << 8) | (uint64_t)one;
      ++nbytes;
    } else {
      const int b0 = pn.y ? al.label_off[pn.y - 1] : 0, b1 = al.label_off[pn.y];
      for (int b = b1 - 1; b >= b0; --b) {
        hi = (hi << 8) | (lo >> 56);
        lo = (lo << 8) | (uint64_t)al.label_bytes[b];
        ++nbytes;
      }
    }
    cur = pn.x;
  }
  if (nbytes > 16) { lo = ~0ULL; hi = ~0ULL; }
}

// MurmurHash64A (seed 0) of the word that ends at path node `node` (back to the previous space / the root), for words the packed
// 16-byte form does not hold.  The path is a backward-linked list and the hash eats 8-byte blocks from the front; instead of a label
// buffer on the stack (256 bytes of scratch memory per lane in a kernel that otherwise has none) every block walks the word again --
// a long word costs len^2 / 8 node reads, and dictionaries hold few of them.
__device__ __forceinline__ uint64_t murmur_path_word(const DevAlphabet& al, const GStream& S, uint32_t node, unsigned& probes) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint32_t len = 0;
  for (uint32_t cur = node; cur != STT_ROOT_CH;) {
    const uint2 pn = load_node(S.pa(), cur);
    ++probes;
    if (pn.y == (uint32_t)al.space_id || pn.y == STT_ROOT_CH) break;
    len += (uint32_t)(al.label_off[pn.y] - (pn.y ? al.label_off[pn.y - 1] : 0));
    cur = pn.x;
  }
  uint64_t h = 0 ^ ((uint64_t)len * m);
  for (uint32_t b0 = 0; b0 < len; b0 += 8) {   // bytes [b0, b0 + 8) of the word, first byte lowest
    uint64_t k = 0;
    uint32_t end = len;                         // the current label's bytes are [end - ll, end)
    for (uint32_t cur = node; cur != STT_ROOT_CH && end > b0;) {
      const uint2 pn = load_node(S.pa(), cur);
      ++probes;
      if (pn.y == (uint32_t)al.space_id || pn.y == STT_ROOT_CH) break;
      const int l0 = pn.y ? al.label_off[pn.y - 1] : 0, l1 = al.label_off[pn.y];
      const uint32_t beg = end - (uint32_t)(l1 - l0);
      for (int q = l0; q < l1; ++q) {
        const uint32_t at = beg + (uint32_t)(q - l0);
        if (at >= b0 && at < b0 + 8) k |= (uint64_t)al.label_bytes[q] << (8 * (at - b0));
      }
      end = beg;
      cur = pn.x;
    }
    if (len - b0 >= 8) { k *= m; k ^= k >> r; k *= m; h ^= k; h *= m; }
    else { h ^= k; h *= m; }                    // the tail: remaining bytes little-endian, then one multiply
  }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

// Score "prefix X, then a word boundary" for a live word-mode prefix X whose last label is neither space nor root,
// given the cached KenLM state of the previous boundary (entry e_prev).  Equivalent to lm_score(): the reference scores
// the last `order` words from a null context (or all words from BeginSentence when there are fewer), and a KenLM state
// holds at most order-1 words, so the state carried from the previous boundary is the state the reference rebuilds.
// Appends a BEntry and records it in S.pq[node]; returns log_cond_prob + hot_boost.
// GenericModel::FullScore through the hashed n-gram index (lmindex.h), one lane per query: one 64-byte bucket per order,
// fetched one after the other (an order-k entry is verified against the order-(k-1) hit, and most queries end at order 2 or
// 3), instead of an interpolation search plus child-range reads per order.  Orders <= 5.  Same floats as the trie walk.
__device__ __forceinline__ float lm_full_score_indexed(const DevScorer& s, const KState& in, uint64_t h, uint32_t wi, const DevVocabSlot& vs, KState& out,
                                                       unsigned& probes, int* ngram_length = nullptr) {
  const float uprob = wi ? vs.prob : s.unk_prob, uback = wi ? vs.backoff : s.unk_backoff;
  const bool uindep = wi ? (vs.begin == vs.end) : (s.unk_indep != 0);
  const GLB_AS u32x4* tab = (const GLB_AS u32x4*)s.lmi;
  const uint32_t nb = s.lmi_buckets;
  LmiLevel lv[LMI_MAX_HIST];
  uint64_t key = wi ? h : LMI_UNK_H;
  uint32_t parent = wi;
  bool going = !uindep;  // (a unigram without children ends ScoreExceptBackoff before any lookup, model.cc:300-305)
#pragma unroll
  for (int q = 0; q < LMI_MAX_HIST; ++q) {
    lv[q].found = 0; lv[q].prob = 0.0f; lv[q].backoff = 0.0f; lv[q].indep = 0;
    if (q < 4) {
      if (going && q < in.length && q + 2 <= s.order) {
        const uint32_t w = in.words[q];
        key = lmi_step(key, w);
        const uint32_t b = lmi_bucket(key, nb);
        const GLB_AS u32x4* bp = tab + (size_t)b * LMI_BUCKET;
        const u32x4 e0 = bp[0], e1 = bp[1], e2 = bp[2], e3 = bp[3];
        ++probes;
        const uint32_t tag = (w & LMI_WORD_MASK) | ((uint32_t)(q + 2) << LMI_LEVEL_SHIFT);
        u32x4 hit = {LMI_EMPTY, 0u, 0u, 0u};
        uint32_t id = LMI_NOT_FOUND;
        if ((e0.x & ~LMI_INDEP_BIT) == tag && e0.y == parent) { hit = e0; id = b * LMI_BUCKET + 0u; }
        else if ((e1.x & ~LMI_INDEP_BIT) == tag && e1.y == parent) { hit = e1; id = b * LMI_BUCKET + 1u; }
        else if ((e2.x & ~LMI_INDEP_BIT) == tag && e2.y == parent) { hit = e2; id = b * LMI_BUCKET + 2u; }
        else if ((e3.x & ~LMI_INDEP_BIT) == tag && e3.y == parent) { hit = e3; id = b * LMI_BUCKET + 3u; }
        else if (e0.x != LMI_EMPTY && e1.x != LMI_EMPTY && e2.x != LMI_EMPTY && e3.x != LMI_EMPTY) {  // full bucket: the entry may have overflowed
          LmiEntry e;
          id = lmi_probe(s.lmi, nb, b + 1 == nb ? 0u : b + 1, q + 2, w, parent, e);
          probes += 2;
          if (id != LMI_NOT_FOUND) { hit.x = e.wl; hit.z = __float_as_uint(e.prob); hit.w = __float_as_uint(e.backoff); }
        }
        if (id == LMI_NOT_FOUND) going = false;
        else {
          lv[q].found = 1; lv[q].prob = __uint_as_float(hit.z); lv[q].backoff = __uint_as_float(hit.w); lv[q].indep = (hit.x & LMI_INDEP_BIT) ? 1 : 0;
          parent = id;
          if (lv[q].indep) going = false;
        }
      } else going = false;
    }
  }
  int nl;
  const float r = lmi_combine(s.order, in, wi, uprob, uback, uindep, lv, out, nl);
  if (ngram_length) *ngram_length = nl;
  return r;
}

// FullScore cache (DevScorer::memo).  The key is (the in-state's context words, their count, the word's hash): backoffs
// in a KenLM state are a function of its words, so equal keys mean equal FullScore results.
struct LmMemoKey { uint32_t w[4]; uint32_t len; uint64_t h; uint32_t slot; uint32_t mix; };
// `unit3`: the unit's bytes when it is a three-byte code point (b0 | b1 << 8 | b2 << 16), else 0.  The 64 code points that share their
// first two bytes -- the children of ONE prefix that complete a code point in a step, scored by 64 neighbouring lanes -- then take 64
// CONSECUTIVE slots (the block is chosen by context + first two bytes, the slot inside it by the last byte): their probes are one 2 KB
// read instead of 64 reads scattered over the table.  An entry still carries its whole key: a hit is exact wherever it sits.
__device__ __forceinline__ LmMemoKey lm_memo_key(const DevScorer& s, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t len, uint64_t h, uint32_t unit3 = 0u) {
  LmMemoKey k;
  k.len = len; k.h = h;
  k.w[0] = len > 0 ? w0 : 0u; k.w[1] = len > 1 ? w1 : 0u; k.w[2] = len > 2 ? w2 : 0u; k.w[3] = len > 3 ? w3 : 0u;
  uint64_t a = unit3 ? (uint64_t)(unit3 & 0xFFFFu) : h;
#pragma unroll
  for (int q = 0; q < 4; ++q) { a = (a ^ k.w[q]) * 0x9E3779B97F4A7C15ULL; a ^= a >> 31; }
  a = (a ^ k.len) * 0xD6E8FEB86659FD93ULL; a ^= a >> 32;
  k.slot = unit3 ? ((((uint32_t)(a >> 8) << 6) | ((unit3 >> 16) & 63u)) & s.memo_mask) : ((uint32_t)(a >> 8) & s.memo_mask);
  k.mix = (uint32_t)a ^ (unit3 * 0x9E3779B1u);
  return k;
}
__device__ __forceinline__ uint32_t lm_memo_meta(const LmMemoKey& k, float prob, bool oov) {
  uint32_t c = (k.mix ^ (__float_as_uint(prob) * 0x85EBCA6Bu)) * 0xC2B2AE35u;
  c ^= c >> 15;
  return k.len | (oov ? 8u : 0u) | ((c | 1u) << 4);  // (an all-zero entry never verifies)
}
__device__ __forceinline__ bool lm_memo_check(const LmMemoKey& k, const u32x4& a, const u32x4& b, float& prob, bool& oov) {
  if (a.x != k.w[0] || a.y != k.w[1] || a.z != k.w[2] || a.w != k.w[3] || b.x != (uint32_t)k.h || b.y != (uint32_t)(k.h >> 32)) return false;
  prob = __uint_as_float(b.z); oov = (b.w & 8u) != 0;
  return b.w == lm_memo_meta(k, prob, oov);
}
__device__ __forceinline__ bool lm_memo_find(const DevScorer& s, const LmMemoKey& k, float& prob, bool& oov, unsigned& probes) {
  const GLB_AS u32x4* m = (const GLB_AS u32x4*)s.memo + (size_t)k.slot * 2;
  const u32x4 a = m[0], b = m[1];
  ++probes;
  return lm_memo_check(k, a, b, prob, oov);
}
__device__ __forceinline__ void lm_memo_store(const DevScorer& s, const LmMemoKey& k, float prob, bool oov) {
  GLB_AS u32x4* m = (GLB_AS u32x4*)s.memo + (size_t)k.slot * 2;
  u32x4 a, b;
  a.x = k.w[0]; a.y = k.w[1]; a.z = k.w[2]; a.w = k.w[3];
  b.x = (uint32_t)k.h; b.y = (uint32_t)(k.h >> 32); b.z = __float_as_uint(prob); b.w = lm_memo_meta(k, prob, oov);
  m[0] = a; m[1] = b;
}

// The code point a unit's packed bytes spell (first byte lowest), if they are its shortest UTF-8 form of one to three bytes -- the form under
// which the vocabulary knows it (scorer_dev.cpp: build_cp_blocks keys cpt by that form); anything else: false.
__device__ __forceinline__ bool utf8_unit_code_point(uint32_t u, uint32_t& cp) {
  const uint32_t b0 = u & 0xFFu, b1 = (u >> 8) & 0xFFu, b2 = (u >> 16) & 0xFFu;
  if (u < 0x80u) { cp = u; return u != 0u; }
  if (u < 0x10000u) { cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu); return (b0 & 0xE0u) == 0xC0u && (b1 & 0xC0u) == 0x80u && cp >= 0x80u; }
  if (u < 0x1000000u) {
    cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
    return (b0 & 0xF0u) == 0xE0u && (b1 & 0xC0u) == 0x80u && (b2 & 0xC0u) == 0x80u && cp >= 0x800u && !(cp >= 0xD800u && cp < 0xE000u);
  }
  return false;
}
// GenericModel::FullScore of code point `cp` (a word of the model: cpt flags bit 0) from state `in` through the bigram blocks: HostScorer::
// full_score_blocks (scorer_dev.cpp) on the device.  Order 2 is ONE table entry for (in.words[0], cp >> 6) -- the same entry for the 64 sibling
// code points a prefix's children complete, whose unigram records are one contiguous 1 KB of cpt -- and a bit of its presence map; a stored
// bigram (rare: 5 % of the bench's queries) continues in the hashed index from the slot its record names.  `unit` = the code point's bytes
// (hashed only on that path).  Same floats as the index and the trie walk (tests/test_cp_blocks.py, tests/test_gpu_lm.py).
__device__ __forceinline__ float lm_full_score_blocks(const DevScorer& s, const KState& in, uint32_t cp, uint32_t unit, const u32x4& ct, KState& out, unsigned& probes,
                                                      int* ngram_length = nullptr) {
  const uint32_t wi = ct.x;
  const float uprob = __uint_as_float(ct.y), uback = __uint_as_float(ct.z);
  const bool uindep = (ct.w & 2u) != 0;
  LmiLevel lv[LMI_MAX_HIST];
#pragma unroll
  for (int q = 0; q < LMI_MAX_HIST; ++q) { lv[q].found = 0; lv[q].prob = 0.0f; lv[q].backoff = 0.0f; lv[q].indep = 0; }
  if (!uindep && in.length > 0 && s.order >= 2) {
    const uint32_t w1 = in.words[0], block = cp >> 6;
    const GLB_AS u32x4* tab = (const GLB_AS u32x4*)s.cpb_tab;
    uint32_t hsl = cpb_hash(w1, block) & s.cpb_mask;
    u32x4 e0, e1;
    bool have = false;
    for (;;) {
      e0 = tab[(size_t)hsl * 2]; e1 = tab[(size_t)hsl * 2 + 1];
      ++probes;
      if (e0.x == 0xFFFFFFFFu) break;
      if (e0.x == w1 && e0.y == block) { have = true; break; }
      hsl = (hsl + 1u) & s.cpb_mask;
    }
    const uint64_t present = (uint64_t)e1.x | ((uint64_t)e1.y << 32), indep = (uint64_t)e1.z | ((uint64_t)e1.w << 32);
    const uint64_t bit = 1ull << (cp & 63u);
    if (have && (present & bit) != 0ull) {
      const GLB_AS uint32_t* r = (const GLB_AS uint32_t*)s.cpb_rec + (size_t)(e0.z + (uint32_t)__popcll(present & (bit - 1ull))) * 3;
      const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
      ++probes;
      lv[0].found = 1; lv[0].prob = __uint_as_float(r0); lv[0].backoff = __uint_as_float(r1); lv[0].indep = (indep & bit) != 0ull ? 1 : 0;
      // orders >= 3: the hashed index, from the bigram's slot on (lm_full_score_indexed's chain, entered at its second level)
      const uint32_t nb = unit < 0x100u ? 1u : (unit < 0x10000u ? 2u : 3u);
      uint64_t key = lmi_step(murmur_packed((uint64_t)unit, 0ull, (int)nb), w1);
      uint32_t parent = r2;
      bool going = !lv[0].indep;
#pragma unroll
      for (int hi = 1; hi < 4; ++hi) {
        if (going && hi < s.order - 1 && hi < in.length) {
          key = lmi_step(key, in.words[hi]);
          LmiEntry e;
          const uint32_t slot = lmi_probe(s.lmi, s.lmi_buckets, lmi_bucket(key, s.lmi_buckets), hi + 2, in.words[hi], parent, e);
          probes += 1;
          if (slot == LMI_NOT_FOUND) going = false;
          else { lv[hi].found = 1; lv[hi].prob = e.prob; lv[hi].backoff = e.backoff; lv[hi].indep = (e.wl & LMI_INDEP_BIT) ? 1 : 0; parent = slot; if (lv[hi].indep) going = false; }
        } else going = false;
      }
    }
  }
  int nl;
  const float r = lmi_combine(s.order, in, wi, uprob, uback, uindep, lv, out, nl);
  if (ngram_length) *ngram_length = nl;
  return r;
}

// IDX: FullScore through the hashed n-gram index (the scorer must have one: orders <= 5), else the trie walk
// the trie walk as a real call: the cold side of the code-point step's FullScore (a scorer without the index), kept out of its registers
__device__ STT_CALLS_NOINLINE float kenlm_full_score_call(const DevScorer& s, const KState* in, uint32_t wi, KState* out, unsigned* probes, const DevVocabSlot* uni) {
  unsigned pr = 0;
  const float r = kenlm_full_score(s, *in, wi, *out, pr, uni);
  *probes += pr;
  return r;
}
// RT_IDX (code-point step): the index if the scorer has one (decided per scorer at load time: a run-time test), else the trie walk
template <bool IDX, bool RT_IDX = false>
__device__ STT_CALLS_INLINE double lm_word_query_cached(const DevScorer& s, const DevAlphabet& al, const GStream& S, const LDS_AS uint8_t* lab1, LDS_AS uint32_t* be_n, uint32_t node, uint32_t e_prev,
                                       bool have_word, uint64_t lo, uint64_t hi, uint32_t& out_entry, unsigned& probes) {
  // The word's bytes come from the beam state (have_word) or from a walk back to the previous boundary; words longer than
  // 16 bytes take the generic label-array path.
  if (!have_word) word_walk(al, S, lab1, node, lo, hi, probes);
  // Code-point step with bigram blocks (DevScorer::cpt): a unit that is one code point of the model's vocabulary needs no hash of its bytes,
  // no vocabulary probe and no memo -- its unigram record is cpt[code point] (the 64 siblings scored by neighbouring lanes read one KB).
  bool via_blocks = false;
  u32x4 ct = {0u, 0u, 0u, 0u};
  uint32_t cp = 0;
  if constexpr (RT_IDX) {
    if (s.cpt != nullptr && have_word && hi == 0ULL && lo < 0x1000000ULL && utf8_unit_code_point((uint32_t)lo, cp)) {
      ct = ((const GLB_AS u32x4*)s.cpt)[cp];
      ++probes;
      via_blocks = (ct.w & 1u) != 0;
    }
  }
  uint64_t h = 0;
  if (!via_blocks || s.n_hot != 0) {   // (the blocks' path hashes the unit only for hot words -- and where a stored bigram continues in the index)
    if ((lo & hi) != ~0ULL) h = murmur_packed(lo, hi, word_nbytes(lo, hi));
    else h = murmur_path_word(al, S, node, probes);   // a word of more than 16 bytes
  }
  const BEntry ep = load_be(S, e_prev);  // issued before the vocabulary probe: the two reads are independent
  ++probes;
  BEntry en;
  float prob;
  bool word_oov;
  // a caller that only wants the value (no new entry, so no out-state) asks the FullScore cache first
  const bool use_memo = !IDX && !via_blocks && be_n == nullptr && s.memo != nullptr && ep.st.length <= 4;
  LmMemoKey mk;
  if (via_blocks) { prob = lm_full_score_blocks(s, ep.st, cp, (uint32_t)lo, ct, en.st, probes); word_oov = false; }
  else {
  if (use_memo) {
    const uint32_t u3 = (have_word && hi == 0 && (lo >> 24) == 0 && ((uint32_t)lo >> 16) != 0 && (((uint32_t)lo >> 4) & 0xFu) == 0xEu) ? (uint32_t)lo : 0u;   // three bytes, the first 1110xxxx
    mk = lm_memo_key(s, ep.st.words[0], ep.st.words[1], ep.st.words[2], ep.st.words[3], (uint32_t)ep.st.length, h, u3);
  }
  if (!(use_memo && lm_memo_find(s, mk, prob, word_oov, probes))) {
    DevVocabSlot vs;
    const uint32_t wi = vocab_slot(s, h, vs, probes);
    if constexpr (IDX) prob = lm_full_score_indexed(s, ep.st, h, wi, vs, en.st, probes);  // (the launcher picks IDX only when the index exists)
    else if (RT_IDX && s.lmi != nullptr && s.uni_in_vtab) prob = lm_full_score_indexed(s, ep.st, h, wi, vs, en.st, probes);
    else if (RT_IDX) prob = kenlm_full_score_call(s, &ep.st, wi, &en.st, &probes, (wi != 0 && s.uni_in_vtab) ? &vs : nullptr);
    else prob = kenlm_full_score(s, ep.st, wi, en.st, probes, (wi != 0 && s.uni_in_vtab) ? &vs : nullptr);
    word_oov = wi == 0;
    if (use_memo) lm_memo_store(s, mk, prob, word_oov);
  }
  }
  en.oov_hist = (uint16_t)((ep.oov_hist << 1) | (word_oov ? 1u : 0u));
  const bool oov = (en.oov_hist & ((1u << s.order) - 1u)) != 0;  // this word + the order-1 before it
  float hot_self = 0.0f, hot_total = 0.0f;
  if (s.n_hot) {
    for (int j = 0; j < s.n_hot; ++j)
      if (h == s.hot_hash[j]) hot_self = __fadd_rn(hot_self, s.hot_boost[j]);
    float hs[STT_KENLM_MAX_ORDER];
    int k = 0;
    uint32_t e = e_prev;
    BEntry cur = ep;
    while (k < s.order - 1 && e != 0 && e != STT_NONE) {  // entry 0 = root: no word
      hs[k++] = cur.hot_self;
      e = cur.prev;
      if (e != 0 && e != STT_NONE) { cur = load_be(S, e); ++probes; }
    }
    for (int i = k - 1; i >= 0; --i) hot_total = __fadd_rn(hot_total, hs[i]);  // oldest word first, like the reference's loop
    hot_total = __fadd_rn(hot_total, hot_self);
  }
  const double lcp = oov ? OOV_SCORE_D : __ddiv_rn((double)prob, (double)0.4342944819f);
  en.raw = __dadd_rn(lcp, (double)hot_total);
  en.prev = e_prev; en.pad = 0; en.hot_self = hot_self;
  out_entry = STT_NONE;
  if (be_n) {  // (null: a read-only caller -- DecoderState::decode -- only wants the value)
    const uint32_t idx = lds_add(be_n, 1u);  // LDS copy of the arena fill (written back when the launch ends)
    if (idx < S.be_cap()) { store_be(S, idx, en); S.pq()[node] = idx; out_entry = idx; }
  }
  return en.raw;
}

__host__ __device__ inline uint32_t pow2_ge(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

// ------------------------------------------------------------------------------------ wide alphabets
// The per-class arrays of a timestep (log-probs, class order, class -> position) live in LDS for ordinary alphabets.  A
// 6000-label alphabet at beam 1024 does not fit (the beam state alone takes ~148 KiB), so for wide alphabets
// a separate, row-parallel kernel prepares one record per emission row in HBM -- get_pruned_emissions
// (ctc_beam_search_decoder.cpp:328-358) for all rows of the chunk at once -- and the search kernel reads the few entries it
// needs (blank, repeated label, the labels on the dictionary arcs) through L2.
//   record = { WideRowHdr | float lpc[C] : log(p + FLT_MIN) by class | u16 pos[C2] : class -> position (0xFFFF = cut off)
//              | u16 cls[C2] : position -> class },  C2 = C rounded up to even
struct WideRowHdr { int cutoff_len; float pblank; double lbl; };
__host__ __device__ inline size_t wide_off_pos(int C) { return sizeof(WideRowHdr) + (size_t)C * 4; }
__host__ __device__ inline size_t wide_off_cls(int C) { return wide_off_pos(C) + (size_t)((C + 1) & ~1) * 2; }
size_t ctc_wide_row_bytes(int C) { return (wide_off_cls(C) + (size_t)((C + 1) & ~1) * 2 + 15) & ~(size_t)15; }
struct WRow { const GLB_AS float* lpc; const GLB_AS uint16_t* pos; const GLB_AS uint16_t* cls; };
__device__ __forceinline__ const WideRowHdr* wide_hdr(const DecParams& p, int stream, int t) {
  return reinterpret_cast<const WideRowHdr*>(p.wide_rows + ((size_t)stream * p.wide_max_frames + t) * p.wide_stride);
}
__device__ __forceinline__ WRow wide_row(const DecParams& p, int stream, int t) {
  const unsigned char* r = p.wide_rows + ((size_t)stream * p.wide_max_frames + t) * p.wide_stride;
  WRow w;
  w.lpc = (const GLB_AS float*)(r + sizeof(WideRowHdr));
  w.pos = (const GLB_AS uint16_t*)(r + wide_off_pos(p.C));
  w.cls = (const GLB_AS uint16_t*)(r + wide_off_cls(p.C));
  return w;
}

#define WIDE_SORT_N 8192  // = STT_MAX_CLASSES: keys of one row in LDS
// One 1024-thread workgroup per (stream, frame of the chunk).
__global__ __launch_bounds__(1024) void ctc_wide_rows_kernel(DecParams p, const float* probs, const int* frame_begin, const int* frame_count) {
  __shared__ uint64_t keys[WIDE_SORT_N];
  __shared__ double log_tab[32];
  __shared__ int s_cut;
  const int s = blockIdx.x / p.wide_max_frames, tt = blockIdx.x - s * p.wide_max_frames;
  if (tt >= (frame_count ? frame_count[s] : p.all_count)) return;
  const int tid = threadIdx.x, C = p.C;
  const float* row = probs + ((size_t)s * p.t_max + (frame_begin ? frame_begin[s] : p.all_begin) + tt) * C;
  unsigned char* rec = const_cast<unsigned char*>(p.wide_rows) + ((size_t)s * p.wide_max_frames + tt) * p.wide_stride;
  float* lpc = reinterpret_cast<float*>(rec + sizeof(WideRowHdr));
  uint16_t* pos = reinterpret_cast<uint16_t*>(rec + wide_off_pos(C));
  uint16_t* cls = reinterpret_cast<uint16_t*>(rec + wide_off_cls(C));
  if (tid < 32) log_tab[tid] = sttm::kLogfTab[tid >> 1][tid & 1];
  __syncthreads();
  const bool sort_classes = (p.cutoff_prob < 1.0) || (p.cutoff_top_n < C);
  const uint32_t n2 = pow2_ge((uint32_t)C);
  for (uint32_t i = tid; i < n2; i += 1024) {
    uint64_t k = ~0ULL;
    if ((int)i < C) {
      const float v = row[i];
      lpc[i] = sttm::stt_logf_t(__fadd_rn(v, STT_FLT_MIN), log_tab);
      // ascending key == (probability descending, class index ascending): std::sort by pair_comp_second_rev, ties by index
      uint32_t u = (v == 0.0f) ? 0u : __float_as_uint(v);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      k = ((uint64_t)(~u) << 32) | (uint64_t)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  if (sort_classes) {
    for (uint32_t k = 2; k <= n2; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = tid; i < n2; i += 1024) {
          const uint32_t ixj = i ^ j;
          if (ixj > i) {
            const uint64_t a = keys[i], b = keys[ixj];
            if ((a > b) == ((i & k) == 0)) { keys[i] = b; keys[ixj] = a; }
          }
        }
        __syncthreads();
      }
  }
  if (tid == 0) {
    int cl = C;
    if (sort_classes && p.cutoff_prob < 1.0) {  // :342-350
      double cum = 0.0; cl = 0;
      for (int i = 0; i < C; ++i) { cum = __dadd_rn(cum, (double)row[(uint32_t)keys[i]]); cl += 1; if (cum >= p.cutoff_prob || cl >= p.cutoff_top_n) break; }
    }
    s_cut = cl;
    WideRowHdr h;
    h.cutoff_len = cl; h.pblank = row[p.blank]; h.lbl = log((double)row[p.blank]);
    *reinterpret_cast<WideRowHdr*>(rec) = h;
  }
  __syncthreads();
  const int cl = s_cut;
  for (int k = tid; k < C; k += 1024) {
    const uint32_t c = (uint32_t)keys[k];
    cls[k] = (uint16_t)c;
    pos[c] = k < cl ? (uint16_t)k : (uint16_t)0xFFFF;
  }
}

// ------------------------------------------------------------------------------------ LDS layout
#define NWAVES (NTHREADS / 64)
#define NBUCKET 1024   // selection histogram bins (one per thread)
#define RCAP 128       // a threshold bucket with more members than this is subdivided instead of ranked pairwise
#define HTN 2048       // LDS hash slots (>= 2 * STT_MAX_BEAM)
#define BLOOM_WORDS 512  // utf8 mode: filter in front of the hash (almost every lookup of the expand phase is a miss)

// A double-buffered LDS array: buffer d starts `blk` bytes after buffer 0.  (An array of two pointers indexed with a
// run-time `cur` would force the whole Lds struct into scratch memory and turn every beam access into a scratch load.)
template <class T> struct DB {
  LDS_AS T* p0; uint32_t blk;
  __device__ __forceinline__ LDS_AS T* operator[](int d) const { return (LDS_AS T*)((LDS_AS unsigned char*)p0 + (uint32_t)d * blk); }
};
struct Lds {
  DB<float> score, pb, pnb;
  DB<uint32_t> ch, node, ts, bnd;
  DB<int> fst;
  DB<uint32_t> a0; DB<uint16_t> an;    // out-arcs of the prefix's dictionary state: first arc, count (read when the beam is written)
  DB<uint32_t> sm;                     // bitmap step (MODE 4): bitmap of the labels on those arcs (same storage as `an`; a0 = first labelled arc)
  DB<uint64_t> key;
  // word mode, narrow beams: the UTF-8 bytes of the prefix's current (unfinished) word, first byte lowest (wlo = bytes
  // 0..7, whi = 8..15; all ones = longer than 16 bytes), and the BEntry of "prefix + boundary" once scored (STT_NONE before)
  DB<uint64_t> wlo, whi; DB<uint32_t> pqe; DB<float> pqs;  // pqs = (float)(raw score * alpha) of entry pqe
  DB<uint32_t> run;  // utf8 mode: the prefix's code point in progress (utf8_child_run)
  LDS_AS float *ev_self, *ev_blank, *ev_ext;  // reused as new pnb / new pb / new score in P4
  LDS_AS uint32_t* ev_exti;                   // parent beam index | needs_lm << 31 ; reused as pending timestep parent
  LDS_AS uint64_t* ht_key; LDS_AS uint16_t* ht_idx;  // path key -> beam index of the live prefixes (rebuilt whenever the beam is written)
  LDS_AS uint32_t* bloom;  // utf8 mode: 16384-bit filter over the live keys (one bit per key); null otherwise
  DB<float> pf, lp; LDS_AS float* lps;        // emissions and their logs (double buffered: the next row is prepared one step ahead); lps = by class position when pruning sorts
  LDS_AS double* lbl;                         // [2] log((double)prob[blank])
  LDS_AS uint16_t *cls, *pos;
  LDS_AS uint8_t* lab1;                       // [C] the byte of every single-byte label (0 otherwise)
  LDS_AS uint32_t *hist, *cumb;
  LDS_AS uint16_t* lmw; uint32_t lmw_cap;    // the LM waves' lists of prefixes to score (bitmap step: one list of lmw_cap entries per beam buffer, see ctc_step)
  LDS_AS uint64_t* exp_tab; LDS_AS double* log_tab;  // sttmath.h tables (32 x u64, 32 x f64)
  LDS_AS uint8_t* own; uint32_t own_cap;     // expand: per-wave table item -> owning lane (aliases skey/sseg/cumb, idle in that phase)
  LDS_AS uint64_t* skey; LDS_AS uint32_t *ssrc, *sseg;
  LDS_AS uint32_t* wtot;
  // the first `mcap` candidates of a step live in LDS, the rest in the stream's HBM workspace
  LDS_AS float* lc_logp; LDS_AS uint32_t* lc_pi; LDS_AS int* lc_fst; uint32_t mcap;
  LDS_AS unsigned long long* acc;  // [0..3] stat counters, [4..11] phase cycles (accumulated in LDS, flushed when the launch ends)
  LDS_AS unsigned long long* stm;  // [64] fine-grained stamps (DecParams::stamps)
  LDS_AS int* sc;  // scalars
};
// phase cycle counters (s_memtime is a scalar memory operation with a wait: only on request, DecParams::phase_cycles)
#define TICK(k) do { if (p.phase_cycles && tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); L.acc[4 + (k)] += now_ - tick_; tick_ = now_; } } while (0)
enum { SC_M = 0, SC_CUTLEN, SC_LMQ, SC_PROBES, SC_ERR, SC_KMIN, SC_KMAX, SC_BT, SC_BTH, SC_BTCUM, SC_PAN, SC_TAN, SC_BEN, SC_NQ, SC_NA, SC_NI, SC_ICUR, SC_FILL, SC_DONE, SC_NS, SC_THB, SC_LMD, SC_NLM0, SC_NLM1, SC_COUNT = 24 };
#define NEG_HI 0xFF7FFFFFu  // high word of the selection key of score == -NUM_FLT_INF


// The layout starts at LDS address LDS_ORIGIN, not at the address of an `extern __shared__` array: the kernel has no static LDS, so its
// dynamic LDS begins at address 0 (checked when the kernel starts), and with literal addresses every array of the layout is a compile-time
// constant folded into the ds_* instructions.  Relative to the array's symbol each base was `symbol + constant` -- a value the compiler
// materialised in a scalar register, hoisted out of the timestep loop and, there being forty of them, spilled: every LDS access in a
// step began with a v_readlane.  (Address 0 itself is skipped: an integer 0 converts to the null pointer of the LDS address space.)
#define LDS_ORIGIN 16
// Layout for a beam capacity CAP (a compile-time constant: every array that only depends on CAP sits at a constant LDS
// address, which the compiler folds into the ds_* instructions instead of keeping ~50 pointers alive in registers).
// The class-count dependent arrays and the candidate staging area come last.
// (host: from the environment once; device: the launch is configured with the host's value, and the kernel recomputes the
// same layout from the same budget passed in DecParams::lds_kb)
__host__ inline int lds_budget_kb_host() { const int k = tune().search_lds_kb; return k < 96 ? 96 : (k > 160 ? 160 : k); }
template <int CAP>
__host__ __device__ __attribute__((always_inline)) inline Lds lds_carve(int C, LDS_AS unsigned char* base, size_t& total, int budget_kb, bool utf8 = false) {
  Lds L{};
  Lds* l = &L;
  constexpr uint32_t cap = CAP;
  constexpr uint32_t sn = cap + RCAP;
  constexpr bool arcs = cap <= 512;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  size_t offs[80]; int k = 0;
  for (int d = 0; d < 2; ++d) {
    offs[k++] = take(cap * 8);                                     // key
    for (int a = 0; a < 8; ++a) offs[k++] = take(cap * 4);         // score pb pnb ch node ts fst bnd
    offs[k++] = take(arcs ? cap * 4 : 0); offs[k++] = take(arcs ? cap * 4 : 0);  // a0, an / sm (wide beams read the FST instead)
    offs[k++] = take(arcs ? cap * 8 : 0); offs[k++] = take(arcs ? cap * 8 : 0); offs[k++] = take(arcs ? cap * 4 : 0);  // wlo, whi, pqe
    offs[k++] = take(arcs ? cap * 4 : 0);                                                                              // pqs
    offs[k++] = take(utf8 ? cap * 4 : 0);                                                                              // run (utf8 mode: code point in progress)
  }
  for (int a = 0; a < 4; ++a) offs[k++] = take(cap * 4);           // events
  offs[k++] = take(HTN * 8); offs[k++] = take(HTN * 2);            // hash
  offs[k++] = take(NBUCKET * 4);                                   // hist
  offs[k++] = take(sn * 4);                                        // ssrc
  const size_t o_own = o;                                          // skey | sseg | cumb double as the expand phase's item-owner tables
  offs[k++] = take(sn * 8); offs[k++] = take(sn * 4); offs[k++] = take((NBUCKET + 1) * 4);
  const uint32_t own_cap = (uint32_t)((o - o_own) / NWAVES) & ~3u;
  offs[k++] = take(64 * 4);
  offs[k++] = take(arcs ? cap * 4 : cap * 2);                      // lmw (narrow beams: one list of cap entries per beam buffer)
  offs[k++] = take(32 * 8); offs[k++] = take(32 * 8);              // exp / log tables
  offs[k++] = take(12 * 8);
  offs[k++] = take(64 * 8);                                        // stm
  offs[k++] = take(SC_COUNT * 4);
  offs[k++] = take(16);                                            // lbl[2]
  offs[k++] = take(utf8 ? BLOOM_WORDS * 4 : 0);                    // bloom
  // ---- class-count dependent from here on
  for (int a = 0; a < 4; ++a) offs[k++] = take((size_t)(C > 0 && C < 32 ? 32 : C) * 4);  // pf[2], lp[2]
  offs[k++] = take((size_t)C * 4);                                                      // lps
  offs[k++] = take((size_t)C * 2); offs[k++] = take((size_t)C * 2);
  offs[k++] = take((size_t)C);
  // Candidate staging: whatever fits into the CU's 160 KiB, at most 2048 (STT_AMD_LDS_KB: budget in KiB).  Round 1 kept
  // 34 KiB free for a co-resident LSTM workgroup; that left room for ~320-576 of the ~1320 candidates a step produces at beam
  // 500, and every candidate beyond went through HBM: written in the expand phase, read back (dependent loads) by the score,
  // key and write phases.  The recurrence's 256 workgroups fit two per CU on the 192 CUs the search does not occupy.
  uint32_t mcap = 0;
  {
    const size_t budget = (size_t)budget_kb * 1024 - LDS_ORIGIN;
    if (o + 256 * 12 <= budget) { mcap = (uint32_t)((budget - o) / 12) & ~63u; if (mcap > 2048) mcap = 2048; }
  }
  const size_t o_lc = o;
  o += (size_t)mcap * 12;
  {
    k = 0;
    const uint32_t blk = (uint32_t)(offs[16] - offs[0]);  // 16 arrays per buffer
    auto db = [&](auto& m, size_t off, bool on) { using P = decltype(m.p0); m.p0 = on ? (P)(base + off) : (P) nullptr; m.blk = blk; };
    db(l->key, offs[0], true);
    db(l->score, offs[1], true); db(l->pb, offs[2], true); db(l->pnb, offs[3], true);
    db(l->ch, offs[4], true); db(l->node, offs[5], true); db(l->ts, offs[6], true);
    db(l->fst, offs[7], true); db(l->bnd, offs[8], true);
    db(l->a0, offs[9], arcs); db(l->an, offs[10], arcs); db(l->sm, offs[10], arcs);
    db(l->wlo, offs[11], arcs); db(l->whi, offs[12], arcs); db(l->pqe, offs[13], arcs); db(l->pqs, offs[14], arcs);
    db(l->run, offs[15], utf8);
    k = 32;
    l->ev_self = (LDS_AS float*)(base + offs[k++]); l->ev_blank = (LDS_AS float*)(base + offs[k++]); l->ev_ext = (LDS_AS float*)(base + offs[k++]);
    l->ev_exti = (LDS_AS uint32_t*)(base + offs[k++]);
    l->ht_key = (LDS_AS uint64_t*)(base + offs[k++]); l->ht_idx = (LDS_AS uint16_t*)(base + offs[k++]);
    l->hist = (LDS_AS uint32_t*)(base + offs[k++]);
    l->ssrc = (LDS_AS uint32_t*)(base + offs[k++]);
    l->skey = (LDS_AS uint64_t*)(base + offs[k++]); l->sseg = (LDS_AS uint32_t*)(base + offs[k++]); l->cumb = (LDS_AS uint32_t*)(base + offs[k++]);
    l->own = (LDS_AS uint8_t*)(base + o_own); l->own_cap = own_cap;
    l->wtot = (LDS_AS uint32_t*)(base + offs[k++]);
    l->lmw = (LDS_AS uint16_t*)(base + offs[k++]); l->lmw_cap = cap;
    l->exp_tab = (LDS_AS uint64_t*)(base + offs[k++]); l->log_tab = (LDS_AS double*)(base + offs[k++]);
    l->acc = (LDS_AS unsigned long long*)(base + offs[k++]);
    l->stm = (LDS_AS unsigned long long*)(base + offs[k++]);
    l->sc = (LDS_AS int*)(base + offs[k++]);
    l->lbl = (LDS_AS double*)(base + offs[k++]);
    { const size_t ob = offs[k++]; l->bloom = utf8 ? (LDS_AS uint32_t*)(base + ob) : (LDS_AS uint32_t*)nullptr; }
    l->pf.p0 = (LDS_AS float*)(base + offs[k]); l->pf.blk = (uint32_t)(offs[k + 1] - offs[k]); k += 2;
    l->lp.p0 = (LDS_AS float*)(base + offs[k]); l->lp.blk = (uint32_t)(offs[k + 1] - offs[k]); k += 2;
    l->lps = (LDS_AS float*)(base + offs[k++]);
    l->cls = (LDS_AS uint16_t*)(base + offs[k++]); l->pos = (LDS_AS uint16_t*)(base + offs[k++]);
    l->lab1 = (LDS_AS uint8_t*)(base + offs[k++]);
    l->mcap = mcap;
    l->lc_logp = (LDS_AS float*)(base + o_lc); l->lc_pi = (LDS_AS uint32_t*)(base + o_lc + (size_t)mcap * 4); l->lc_fst = (LDS_AS int*)(base + o_lc + (size_t)mcap * 8);
  }
  total = o;
  return L;
}
inline int cap_bucket(int beam) { return beam <= 64 ? 64 : beam <= 128 ? 128 : beam <= 256 ? 256 : beam <= 512 ? 512 : 1024; }
// LDS bytes of the search kernel with the class arrays of a C-class alphabet in LDS (C = 0: wide mode, none)
size_t ctc_next_lds_bytes(int beam, int C, bool utf8) {
  size_t t = 0;
  switch (cap_bucket(beam)) {
    case 64: (void)lds_carve<64>(C, nullptr, t, lds_budget_kb_host(), utf8); break;
    case 128: (void)lds_carve<128>(C, nullptr, t, lds_budget_kb_host(), utf8); break;
    case 256: (void)lds_carve<256>(C, nullptr, t, lds_budget_kb_host(), utf8); break;
    case 512: (void)lds_carve<512>(C, nullptr, t, lds_budget_kb_host(), utf8); break;
    default: (void)lds_carve<1024>(C, nullptr, t, lds_budget_kb_host(), utf8); break;
  }
  return t + LDS_ORIGIN;
}

#define STT_LDS_MAX (160 * 1024)
// Wide mode when the class arrays do not fit next to the beam state (or the in-kernel class sort would dominate a step).
bool ctc_is_wide(int beam, int C, bool utf8) { return C > 1024 || ctc_next_lds_bytes(beam, C, utf8) > (size_t)STT_LDS_MAX; }

__device__ __forceinline__ int ht_find(const Lds& L, uint64_t k) {
  // double hashing (odd stride from other key bits): a wave pays for its slowest lane, and at beam 1024 (load 1/2)
  // linear probing's clusters made that 30 probes per lookup round; with key-dependent strides the tail is geometric
  uint32_t h = (uint32_t)(k >> 17) & (HTN - 1);
  const uint32_t step = (uint32_t)(k >> 40) | 1u;
  for (;;) {
    const uint64_t v = L.ht_key[h];
    if (v == k) return (int)L.ht_idx[h];
    if (v == 0) return -1;
    h = (h + step) & (HTN - 1);
  }
}

// Wave-level scans and reductions on the DPP network (row shifts inside the 16-lane rows, then row_bcast:15 / :31 across
// rows): a handful of VALU cycles per step, where __shfl_up / __shfl_xor go through the LDS crossbar (ds_bpermute, one
// LDS round trip per step -- six per scan, on the critical path of a phase).  A lane whose DPP source is outside its row
// (or whose row is masked off) keeps `identity`.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define WAVE_SCAN_STEPS(OP, IDENT)                                                                                        \
  { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, DPP_ROW_SHR(1), 0xF, 0xF, false); v = OP(v, t); } \
  { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, DPP_ROW_SHR(2), 0xF, 0xF, false); v = OP(v, t); } \
  { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, DPP_ROW_SHR(4), 0xF, 0xF, false); v = OP(v, t); } \
  { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, DPP_ROW_SHR(8), 0xF, 0xF, false); v = OP(v, t); } \
  { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false); v = OP(v, t); } \
  { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false); v = OP(v, t); }
#define OP_ADD(a, b) ((a) + (b))
#define OP_MIN(a, b) ((b) < (a) ? (b) : (a))
#define OP_MAX(a, b) ((b) > (a) ? (b) : (a))
#define OP_OR(a, b) ((a) | (b))
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
  WAVE_SCAN_STEPS(OP_ADD, 0u)
  return v;
}
// Exclusive prefix sum of one value per thread over the workgroup (contains one __syncthreads; the caller separates
// two calls by another barrier because `wtot` is reused).
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, LDS_AS uint32_t* wtot, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  const uint32_t t = lane < NWAVES ? wtot[lane] : 0u;
  const uint32_t tinc = wave_incl_scan(t, lane);
  const int wu = __builtin_amdgcn_readfirstlane(w);  // (wave-uniform by construction; now the compiler knows it too)
  const uint32_t wbase = (uint32_t)__builtin_amdgcn_readlane((int)tinc, wu > 0 ? wu - 1 : 0);
  total = (uint32_t)__builtin_amdgcn_readlane((int)tinc, NWAVES - 1);
  return (w > 0 ? wbase : 0u) + inc - v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {  // the same value in every lane
  WAVE_SCAN_STEPS(OP_MIN, 0xFFFFFFFFu)
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  WAVE_SCAN_STEPS(OP_MAX, 0u)
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Everything the search kernel is launched with, as ONE by-value kernel parameter: it sits at offset 0 of the kernel-argument segment,
// and the kernel reads it through __builtin_amdgcn_kernarg_segment_ptr() -- constant address space, scalar loads at the point of use.
// (Named by-value parameters are loaded into scalar registers in the prologue and stay live to the end: DevScorer alone is 76 of them.)
struct NextArgs {
  DecParams p; DevScorer s; DevAlphabet al;
  DecStream* streams; const float* probs; const int* frame_begin; const int* frame_count;
};

#define CAND_PI(x) (((uint32_t)(x) < L.mcap) ? L.lc_pi[x] : S.c_pi()[x])
#define CAND_LOGP(x) (((uint32_t)(x) < L.mcap) ? L.lc_logp[x] : S.c_logp()[x])

// ------------------------------------------------------------------------------------ one timestep
// LDS hash insert of a live prefix key (value = beam index)
__device__ __forceinline__ void ht_insert(const Lds& L, uint64_t k, int idx) {
  uint32_t h = (uint32_t)(k >> 17) & (HTN - 1);
  const uint32_t step = (uint32_t)(k >> 40) | 1u;
  for (;;) {
    if (lds_cas0(&L.ht_key[h], k) == 0ULL) { L.ht_idx[h] = (uint16_t)idx; break; }
    h = (h + step) & (HTN - 1);
  }
  if (L.bloom) { const uint32_t b = (uint32_t)(k >> 28) & (BLOOM_WORDS * 32 - 1); lds_or((LDS_AS int*)&L.bloom[b >> 5], (int)(1u << (b & 31u))); }
}

// Class log-probs of one emission row into buffer `buf` (get_pruned_emissions, :328-358, the part that does not depend
// on the class order): pf = prob, lp = log(prob + NUM_FLT_MIN) in class order, lbl = log((double)prob[blank]) for the
// min_cutoff of :142-143.  `v` is the caller's value for class `tid` (prefetched), classes >= NTHREADS are read here.
__device__ __forceinline__ void prep_row(const DecParams& p, const Lds& L, int buf, const float* row, float v) {
  const int tid = threadIdx.x;
  for (int c = tid; c < p.C; c += NTHREADS) {
    const float x = (c == tid) ? v : row[c];
    L.pf[buf][c] = x;
    L.lp[buf][c] = sttm::stt_logf_t(__fadd_rn(x, STT_FLT_MIN), L.log_tab);
    if (c == p.blank) L.lbl[buf] = log((double)x);
  }
}

// The same for an alphabet of at most 64 classes, by whichever threads hold class c's value (c outside 0..C-1: nothing to do)
__device__ __forceinline__ void prep_row_at(const DecParams& p, const Lds& L, int buf, int c, float x) {
  if (c < 0 || c >= p.C) return;
  L.pf[buf][c] = x;
  L.lp[buf][c] = sttm::stt_logf_t(__fadd_rn(x, STT_FLT_MIN), L.log_tab);
  if (c == p.blank) L.lbl[buf] = log((double)x);
}

// Merge the <= 3 events of live prefix j in the reference's visiting order (class position, then beam index; :166-193,
// :245-253) and leave the results in the event arrays: ev_blank = new log_prob_b, ev_self = new log_prob_nb,
// ev_ext = new score (iterate_to_vec, path_trie.cpp:170), ev_exti = pending timestep parent.  Returns the new score.
template <bool WIDE>
__device__ __forceinline__ float merge_live(const DecParams& p, const Lds& L, const WRow& W, int cur, int j) {
  const float NEG = STT_NEG_INF;
#define LSE(x, y) sttm::stt_log_sum_exp_t((x), (y), L.exp_tab, L.log_tab)
  const float e_self = L.ev_self[j], e_blank = L.ev_blank[j], e_ext = L.ev_ext[j];
  const uint32_t ei = L.ev_exti[j] & 0xFFFFu;  // parent beam index (bit 31: waits for a score)
  const uint32_t NOUP = 0xFFFFFFFEu;  // "no pending update" (previous_timesteps == nullptr)
  const uint32_t chj = L.ch[cur][j];
  const int kblank = WIDE ? (int)W.pos[p.blank] : (int)L.pos[p.blank];
  const int kself = chj == STT_ROOT_CH ? 0xFFFF : (WIDE ? (int)W.pos[chj] : (int)L.pos[chj]);
  const bool blank_first = kblank < kself;
  const bool ext_first = (int)ei < j;
  const uint32_t ts_ext = L.ts[cur][ei];  // (ei == 0 when there is no extension event: a harmless read)
  // The visiting order is [blank if blank_first] [ext if ext_first] self [ext if !ext_first] [blank if !blank_first].
  // log_prob_b collects at most one event and log_prob_nb at most two, and log_sum_exp(-inf, y) returns y itself, so the
  // whole merge needs two real log_sum_exp evaluations: nb = lse(first, second) and score = lse(b, nb).  (Written with one
  // call site each: six inlined copies behind divergent branches made every wave walk through most of them.)
  const float a1 = ext_first ? e_ext : e_self, a2 = ext_first ? e_self : e_ext;
  const uint32_t p1 = ext_first ? ts_ext : NOUP, p2 = ext_first ? NOUP : ts_ext;
  float nb = NEG;
  uint32_t pend = NOUP;  // a blank visited first compares against nb == -inf and leaves pend == NOUP: no effect
  if (!is_absent(a1)) { if (nb < a1) pend = p1; nb = a1; }
  if (!is_absent(a2)) { if (nb < a2) pend = p2; nb = LSE(nb, a2); }
  if (!blank_first && !is_absent(e_blank) && nb < e_blank) pend = NOUP;
  const float bb = is_absent(e_blank) ? NEG : e_blank;
  const float nscore = LSE(bb, nb);
  L.ev_blank[j] = bb; L.ev_self[j] = nb; L.ev_ext[j] = nscore; L.ev_exti[j] = pend;
#undef LSE
  return nscore;
}

// Wait until an LDS counter (bumped once per arriving wave, after a release fence) reaches `want`: a barrier among SOME of the waves
// of the workgroup (s_barrier always takes all of them).  Scalar loop control (see the note on uniform control flow in ctc_step);
// bounded, so that a logic error shows up as error bit 16 instead of a hung GPU.
// Returns false when it gave up: the caller must then NOT consume what the counter stands for (an item table that is not complete
// holds whatever the selection left there -- indices that lead out of every table).
__device__ __forceinline__ bool wait_count(LDS_AS int* ctr, uint32_t want, LDS_AS int* err, int max_spins) {
  int spins = 0;
  bool ok = true;
  while ((uint32_t)__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) {
    __builtin_amdgcn_s_sleep(4);   // (256 cycles: the waiting waves' polls are instructions the working waves cannot issue)
    // (the bound is there so that a logic error ends as an error bit, not as a hung GPU: 4 M polls are half a second -- a correct run
    // under a serialising profiler or starved by priority-3 co-tenants stays far below it)
    if (++spins > max_spins) { lds_or(err, 16); ok = false; break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok;
}
__device__ __forceinline__ void signal_count(LDS_AS int* ctr) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) lds_add(ctr, 1);
}

// `buf` holds this step's prepared emissions; `next_row` (or null) is prepared into buf^1 while the LM phase runs.
// MODE: 0 = no scorer, 1 = word-level scorer, 2 = utf8 (codepoint-level) scorer -- separate instantiations, so the
// hot word-mode kernel does not carry the registers and code of the uncached codepoint paths.
// WIDE: the class arrays of the row come from the HBM record `t_local` of this stream (see "wide alphabets" above).
// MODE_T 4 = word mode (like 1) with the dictionary's label bitmaps (DevScorer::fst_rec): a prefix's work items are the
// labels its dictionary state allows AND that survive the score cut-off of :157-159 -- the arcs that would be rejected are
// never touched (about 5x fewer items on near-uniform emissions) --, two language-model waves, and FullScore through the
// hashed n-gram index.  <= 32 classes, no class pruning, beam capacity <= 512.  Its phases differ from the other modes' (DESIGN.md
// 4.1, "the step of the word-mode kernel as it runs now"): a prefix-per-thread pre-pass fills ONE table of work items which every
// non-LM wave then empties in chunks; the pre-pass and the items end at counters (wait_count), not at barriers, so the LM waves are
// waited for only at the end of the score phase; the key phase adds the LM scores; the write phase runs live and new entries on
// separate waves.  The selection (P5) is the other modes'.
template <int MODE_T, bool WIDE>
__device__ __forceinline__ void ctc_step(const CONST_AS NextArgs* ka, const CONST_AS DecStream* gq, const Lds& L, int& cur, int& n,
                         int& start_expanding, int& abs_t, int buf, const float* next_row, int t_local) {
  // Parameters, scorer description and stream pointers are read where they are used (scalar loads from the kernel-argument segment /
  // the DecStream), not carried in registers from the kernel's first instruction: the pointers are made opaque once per step, so the
  // loads cannot be hoisted out of the timestep loop and kept alive across it (NextArgs, GStream).
  // Inside this function `p`, `s`, `al` and `S` are not variables but views of whatever `ka` / `gq` are NOW (macros, undefined again
  // below the function): STEP_FENCE() at a phase boundary makes the two pointers opaque again, so a field read in one phase is read
  // again in the next instead of being held in a scalar register in between.
#define STEP_FENCE() do { ka = launder(ka); gq = launder(gq); } while (0)
#define p (*(const DecParams*)&ka->p)
#define s (*(const DevScorer*)&ka->s)
#define al (*(const DevAlphabet*)&ka->al)
#define S (GStream{gq})
  STEP_FENCE();
  constexpr int MODE = MODE_T == 4 ? 1 : MODE_T;
  constexpr bool MASKED = MODE_T == 4;
  constexpr bool SC_ON = MODE != 0, SC_UTF8 = MODE == 2;
  WRow W{};
  WideRowHdr wh{};
  if (WIDE) { W = wide_row(p, (int)blockIdx.x, t_local); wh = *wide_hdr(p, (int)blockIdx.x, t_local); }
#define POS_OF(c) (WIDE ? (int)W.pos[c] : (int)L.pos[c])
#define CLS_AT(k) (WIDE ? (uint32_t)W.cls[k] : (uint32_t)L.cls[k])
#define LP_AT(k, c) (WIDE ? W.lpc[c] : lp[k])
  const LDS_AS uint8_t* const lab1 = WIDE ? (const LDS_AS uint8_t*)nullptr : (const LDS_AS uint8_t*)L.lab1;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform by construction; now the compiler knows it too)
  const int C = p.C, beam = p.beam;
  LDS_AS int* sc = L.sc;
  const float NEG = STT_NEG_INF;
  const LDS_AS float* pf = L.pf[buf];
  LDS_AS float* lp = L.lp[buf];

  // wave-uniform by construction (every thread carries the same values); telling the compiler keeps the control flow that
  // depends on them scalar -- and loops around wave-level operations (readfirstlane, ballot) MUST be provably uniform: with a
  // divergent-looking exit the structurizer may send lanes round the loop without the lane the operation relies on
  n = __builtin_amdgcn_readfirstlane(n); cur = __builtin_amdgcn_readfirstlane(cur);
  unsigned long long tick_ = p.phase_cycles ? __builtin_readcyclecounter() : 0ull;
  // Bitmap step: the LM waves -- the youngest waves of the workgroup, last in issue arbitration -- run at their query priority from the first
  // instruction of the step (round 6: they arrived last at the end of the score phase in half of the steps).  Reset after their queries.
  if (MASKED && n > NWAVES && p.lm_prio >= 3) {
    const int nlm0 = n > 256 ? (p.n_lm_waves == 4 ? 4 : (p.n_lm_waves == 1 ? 1 : 2)) : 1;   // (= nlm below)
    if (wave >= NWAVES - nlm0) __builtin_amdgcn_s_setprio(3);
  }
  float pre = 0.0f;
  // (consumed in P3; bitmap form: by wave 8 -- waves 0..7 have a merge per thread there, wave 0 was the last to arrive with this on top)
  const int prep_c = MODE_T == 4 ? tid - 512 : tid;
  if (!WIDE && next_row && prep_c >= 0 && prep_c < C) pre = next_row[prep_c];
  if ((double)(WIDE ? wh.pblank : pf[p.blank]) < 0.999) start_expanding = 1;  // :125-132 (uniform: every thread reads the same value)
  start_expanding = __builtin_amdgcn_readfirstlane(start_expanding);
  if (!start_expanding) {
    if (!WIDE && next_row) { if (MODE_T == 4) prep_row_at(p, L, buf ^ 1, prep_c, pre); else prep_row(p, L, buf ^ 1, next_row, pre); }
    abs_t++;
    __syncthreads();
    return;
  }
  // ---- A: clear the per-prefix events and the selection histogram (the hash of the live prefixes was built when the
  // beam was written); class order / cut-off only when pruning is active
  // (bitmap step: every thread clears the events of its own prefix in the pre-pass below -- no barrier in between)
  if (!MASKED) for (int i = tid; i < n; i += NTHREADS) { L.ev_self[i] = absent(); L.ev_blank[i] = absent(); L.ev_ext[i] = absent(); L.ev_exti[i] = 0; }
  L.hist[tid] = 0;
  if (MASKED && tid == 0) sc[SC_NLM0 + (cur ^ 1)] = 0;   // (the LM list of the beam this step will write: see the LM waves below)
  if (tid == 0) { sc[SC_M] = 0; sc[SC_LMQ] = 0; sc[SC_PROBES] = 0; sc[SC_KMIN] = -1; sc[SC_KMAX] = 0; sc[SC_NQ] = 0; sc[SC_NA] = 0; sc[SC_NS] = 0; sc[SC_THB] = 0; }
  const bool sort_classes = (p.cutoff_prob < 1.0) || (p.cutoff_top_n < C);
  int cutoff_len = WIDE ? wh.cutoff_len : C;
  if (!WIDE && sort_classes && p.wide_rows) {  // class order prepared for all rows of the chunk by ctc_wide_rows_kernel
    const WRow R = wide_row(p, (int)blockIdx.x, t_local);
    cutoff_len = wide_hdr(p, (int)blockIdx.x, t_local)->cutoff_len;
    for (int c = tid; c < C; c += NTHREADS) { L.cls[c] = R.cls[c]; L.pos[c] = R.pos[c]; }
    __syncthreads();
    lp = L.lps;  // log-probs by class *position*
    for (int k = tid; k < cutoff_len; k += NTHREADS) lp[k] = L.lp[buf][L.cls[k]];
  } else if (!WIDE && sort_classes) {  // std::sort by probability, descending (ties: class index)
    for (int c = tid; c < C; c += NTHREADS) {
      const float v = pf[c];
      int rank = 0;
      for (int o = 0; o < C; ++o) { const float w = pf[o]; rank += (w > v) || (w == v && o < c); }
      L.cls[rank] = (uint16_t)c;
      L.pos[c] = 0xFFFF;
    }
    __syncthreads();
    if (tid == 0) {
      int cl = C;
      if (p.cutoff_prob < 1.0) {
        double cum = 0.0; cl = 0;
        for (int i = 0; i < C; ++i) { cum = __dadd_rn(cum, (double)pf[L.cls[i]]); cl += 1; if (cum >= p.cutoff_prob || cl >= p.cutoff_top_n) break; }
      }
      sc[SC_CUTLEN] = cl;
    }
    __syncthreads();
    cutoff_len = sc[SC_CUTLEN];
    lp = L.lps;  // log-probs by class *position*
    for (int k = tid; k < cutoff_len; k += NTHREADS) {
      const int c = L.cls[k];
      L.pos[c] = (uint16_t)k;
      lp[k] = L.lp[buf][c];
    }
  }
  float min_cutoff = NEG;
  bool full_beam = false;
  if (SC_ON) {  // :136-146 (the beam is kept in prefix_compare order, so no partial_sort is needed)
    const double mc = __dadd_rn(__dadd_rn((double)L.score[cur][n - 1], WIDE ? wh.lbl : L.lbl[buf]), -fmax(0.0, s.beta));
    min_cutoff = (float)mc;
    full_beam = (n == beam);
  }
  // Code-point scorer: a score that `beam` prefixes of the next beam are KNOWN to reach before anything is expanded.  Every live prefix
  // i whose blank event is not cut off ends the step with log_sum_exp(b, nb) >= b = lp[blank] + score[i] (log_sum_exp never returns less
  // than its larger operand), the beam is sorted, so with a full beam all `beam` of them reach thr = lp[blank] + score[n - 1].  A NEW
  // prefix whose final log-probability is below thr -- or whose log-probability plus an UPPER BOUND of its language-model score is
  // (DevScorer::cp_ub) -- cannot be among the best `beam` of this step: it is dropped where it is made, or its FullScore is skipped
  // and it is parked at -inf.  The surviving beam is the reference's, entry for entry (bytes goldens, bytes fuzz, test_gpu_configs).
  // (-inf = no such statement this step.  Hot words add to the score and are not in the bound: no pruning with them.)
  float thr = NEG;
  // the same with the bound: log_p + (bound * alpha) + beta, rounded as the reference rounds the score itself (monotone in the bound)
  auto lm_bound = [&](float lp0, float ub) -> float {
    const float lpv = __fadd_rn(lp0, (float)__dmul_rn((double)ub, s.alpha));
    return (float)__dadd_rn((double)lpv, s.beta);
  };
  if (!MASKED) __syncthreads();
  // (AFTER the barrier: with class pruning, `pos[blank]` and `lp[position of blank]` are written by ONE thread of the loops above.  Until
  // the end of round 6 this stood before the barrier -- a wave that ran ahead of the writer read last step's values or "cut off", waves
  // disagreed about thr, and the `thr != NEG` block below holds two barriers: a wrong beam, a launch that never ends or a wild index, in
  // one run of fifty of ONE fuzz case and in every run once something delayed the waves -- DESIGN.md 10.10, profiles/NOTES.md.)
  if (SC_UTF8 && full_beam && s.n_hot == 0 && s.alpha >= 0.0) {
    const int kb = POS_OF(p.blank);
    const float sw = L.score[cur][n - 1];
    if (kb != 0xFFFF && sw != NEG) {
      const float bw = __fadd_rn(LP_AT(kb, p.blank), sw);
      if (!(bw < min_cutoff)) thr = bw;
    }
  }
  TICK(0);
  STEP_FENCE();

  // ---- P2: expand.  Wave w owns prefixes w, w+16, w+32, ...: blank / repeat events per prefix, then one work item per
  // (prefix, candidate label) -- with a dictionary only the out-arcs of the prefix's FST state can succeed
  // (path_trie.cpp:54-64), otherwise every kept class -- dealt to the lanes through a wave-local prefix sum.
  unsigned probes = 0;
  const bool lm_queue = SC_ON && !SC_UTF8;  // word mode: <= 1 scored extension (the space) per prefix per step
  // Word mode, narrow beams: the last wave takes no prefixes.  It finds the prefixes whose "prefix + space" extension will
  // need a language-model score this step (same tests as the expand items: a word may end here, not yet scored, survives
  // the cut-off) and runs those queries *now*, so their dependent HBM reads overlap the expand work of the other waves
  // instead of sitting between two barriers.  P3 then finds the entry in pqe and only reads its score.
  const bool lm_wave = MODE == 1 && L.pqe.p0 != nullptr && n > NWAVES;
  // bitmap mode: 2 (or 4, DecParams::n_lm_waves) language-model waves when the beam needs capacity 512 (lmw holds their lists side by side)
  const int nlm = lm_wave ? ((MASKED && n > 256) ? (p.n_lm_waves == 4 ? 4 : (p.n_lm_waves == 1 ? 1 : 2)) : 1) : 0;
  const int nw_exp = NWAVES - nlm;
  const uint32_t space_u = (uint32_t)al.space_id;
  if (lm_wave && wave >= nw_exp) {
    // (the chain of dependent reads was the critical path of the phase when the phase ended at a barrier: issue it first)
    if (p.lm_prio >= 3) __builtin_amdgcn_s_setprio(3); else if (p.lm_prio == 2) __builtin_amdgcn_s_setprio(2); else if (p.lm_prio == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned long long lmw_t0 = p.phase_cycles ? __builtin_readcyclecounter() : 0ull;
    const int lw = wave - nw_exp;
    unsigned lmq = 0;
    if (MASKED) {
      // Bitmap step, the EAGER list (round 6): every prefix whose word may end and that has no score yet was put on the list of its beam
      // buffer when that beam was written (write phase below; the first step of a launch: the kernel's prologue) -- whether or not its
      // space extension survives this step's cut-off (the entry is a cache of what the reference would compute when it needs it; on the
      // bench's emissions it is the same 13.0 queries per step).  The LM waves no longer scan the beam first: 3.4 k of their 9 k cycles.
      const uint32_t n_list = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_NLM0 + cur]);
      const LDS_AS uint16_t* list = L.lmw + (uint32_t)cur * L.lmw_cap;
      const uint32_t per = (n_list + (uint32_t)nlm - 1u) / (uint32_t)nlm;
      const uint32_t q0 = (uint32_t)lw * per, q1 = q0 + per < n_list ? q0 + per : n_list;
      for (uint32_t q = q0 + (uint32_t)lane; q < q1; q += 64) {
        const int i = (int)list[q];
        if (L.pqe[cur][i] == STT_NONE && L.bnd[cur][i] != STT_NONE && L.score[cur][i] != NEG) {
          uint32_t ne;
          const double raw = lm_word_query_cached<MASKED>(s, al, S, lab1, (LDS_AS uint32_t*)&sc[SC_BEN], L.node[cur][i], L.bnd[cur][i], true, L.wlo[cur][i], L.whi[cur][i], ne, probes);
          L.pqe[cur][i] = ne; L.pqs[cur][i] = (float)__dmul_rn(raw, s.alpha);
          ++lmq;
        }
      }
    } else {
    const int ksp = POS_OF(al.space_id);
    LDS_AS uint16_t* list = L.lmw + lw * (512 / (nlm > 0 ? nlm : 1));
    if (ksp != 0xFFFF) {
      const float lpsp = LP_AT(ksp, al.space_id);
      uint32_t n_need = 0;
      // 64-prefix slices of the (sorted) beam are dealt so that every LM wave gets good and bad ones: the top of the beam
      // survives the cut-off more often and asks for more scores (2 waves: slices 0,3,4,7 | 1,2,5,6; 4 waves: s and 7-s)
      for (int sl = 0; sl * 64 < n; ++sl) {
        const int owner = nlm == 1 ? 0 : nlm == 2 ? (((sl + 1) >> 1) & 1) : ((sl & 7) < 4 ? (sl & 7) : 7 - (sl & 7));
        if (owner != lw) continue;
        const int i = sl * 64 + lane;
        bool need = false;
        if (i < n) {
          const float sci = L.score[cur][i];
          const bool word_may_end = MASKED ? (((L.sm[cur][i] >> space_u) & 1u) != 0) : ((L.an[cur][i] >> 15) != 0);
          need = word_may_end && L.pqe[cur][i] == STT_NONE && L.bnd[cur][i] != STT_NONE && sci != NEG &&
                 !(full_beam && __fadd_rn(lpsp, sci) < min_cutoff);
        }
        const uint64_t mask = __ballot(need);
        if (need) list[n_need + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)i;
        n_need += (uint32_t)__popcll(mask);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (uint32_t q = lane; q < n_need; q += 64) {
        const int i = (int)list[q];
        uint32_t ne;
        const double raw = lm_word_query_cached<MASKED>(s, al, S, lab1, (LDS_AS uint32_t*)&sc[SC_BEN], L.node[cur][i], L.bnd[cur][i], true, L.wlo[cur][i], L.whi[cur][i], ne, probes);
        L.pqe[cur][i] = ne; L.pqs[cur][i] = (float)__dmul_rn(raw, s.alpha);
        ++lmq;
      }
    }
    }
    if (lmq) lds_add(&sc[SC_LMQ], (int)lmq);
    if (p.phase_cycles && lane == 0 && wave == NWAVES - 1) L.acc[4 + 7] += __builtin_readcyclecounter() - lmw_t0;  // phase slot 7: the (last) LM wave's own time
    __builtin_amdgcn_s_setprio(0);
  }
  const bool lmw_ = lm_wave && wave >= nw_exp;             // this wave is a language-model wave (scalar)
  const uint32_t n_cons = (uint32_t)(lm_wave ? nw_exp : NWAVES);  // waves that fill the table and take items
  if (MASKED) {
    // ---- expand with label bitmaps (see the comment above the function), in two halves:
    // (1) pre-pass, thread i = prefix i (the first n threads; the LM waves are in their queries meanwhile): cut-off mask, blank /
    //     repeat events, the prefix's work items (one u16 {prefix, label} each) appended to ONE table for the workgroup;
    // (2) every wave takes chunks of 64 items off the table until it is empty; the LM waves join when their queries are done.
    // Whoever is free takes the next chunk: the phase ends when the work is done, not when the unluckiest wave is.
    const LDS_AS float* lpv = lp;
    const uint32_t lab_mask = ((1u << (C - 1)) - 1u) & ~(1u << p.blank);  // labels 0 .. C-2, never the blank
    LDS_AS uint16_t* own = (LDS_AS uint16_t*)L.own;
    const uint32_t own_room = (L.own_cap * (uint32_t)NWAVES) >> 1;
    const uint32_t own_n = (p.item_cap > 0 && (uint32_t)p.item_cap < own_room) ? (uint32_t)p.item_cap : own_room;  // items the table holds (more: another pass)
    uint32_t em = 0;
    // 64 prefixes per wave: the pre-pass is bound by instruction issue (a CU issues one VALU instruction per cycle over all its
    // waves), so full waves beat more waves -- 36 lanes of 14 waves measured 9.4 k cycles until the table was complete, 64 lanes
    // of 8 waves 6.5 k
    const int i_pre = tid;
    const bool pre_wave = !lmw_ && wave * 64 < n;        // scalar (an LM wave never is one: n <= 512 < 64 * nw_exp)
    if (!lmw_ && i_pre < n) {
      // straight-line (selects, not branches: every lane of the wave has a live prefix almost always, and exec juggling around
      // one or two instructions costs more than they do)
      const int i = i_pre;
      const float sci = L.score[cur][i], pnbi = L.pnb[cur][i];
      const uint32_t chi = L.ch[cur][i], smi = L.sm[cur][i];
      uint32_t pm = 0xFFFFFFFFu;  // classes that survive the cut-off for this prefix (bit c)
      if (full_beam) {   // (scalar)
        pm = 0;
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
          const f32x4v v = reinterpret_cast<const LDS_AS f32x4v*>(lpv)[q4];
          pm |= (!(__fadd_rn(v.x, sci) < min_cutoff) ? 1u : 0u) << (4 * q4);
          pm |= (!(__fadd_rn(v.y, sci) < min_cutoff) ? 2u : 0u) << (4 * q4);
          pm |= (!(__fadd_rn(v.z, sci) < min_cutoff) ? 4u : 0u) << (4 * q4);
          pm |= (!(__fadd_rn(v.w, sci) < min_cutoff) ? 8u : 0u) << (4 * q4);
        }
      }
      if (!(sci != NEG)) pm = 0;  // :160-162
      const bool has_ch = chi != STT_ROOT_CH;
      const float lp_b = lpv[p.blank], lp_s = lpv[has_ch ? chi : 0u];
      const float eb = ((pm >> p.blank) & 1u) ? __fadd_rn(lp_b, sci) : absent();                        // :166-179
      const float es = (has_ch && ((pm >> (chi & 31u)) & 1u)) ? __fadd_rn(lp_s, pnbi) : absent();      // :182-193
      em = smi & pm & lab_mask;
      L.ev_blank[i] = eb; L.ev_self[i] = es; L.ev_ext[i] = absent(); L.ev_exti[i] = 0;
    }
    for (int pass = 0;; ++pass) {
      if (pre_wave) {
        const uint32_t cnt = (uint32_t)__popc(em);
        const uint32_t inc = wave_incl_scan(cnt, lane);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (tot) {  // wave-uniform
          uint32_t b0 = 0;
          if (lane == 63) b0 = lds_add((LDS_AS uint32_t*)&sc[SC_NI], tot);
          b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, 63);
          uint32_t k = b0 + inc - cnt;
          while (em && k < own_n) { const uint32_t c = (uint32_t)__builtin_ctz(em); em &= em - 1u; own[k++] = (uint16_t)((uint32_t)tid | (c << 9)); }
        }
        // "this wave's items are in the table": a counter instead of a barrier, so that the LM waves -- in the middle of their
        // chain of dependent reads -- are not waited for (LDS operations of a wave are performed in order)
        signal_count(&sc[SC_FILL]);
      }
      if (pass == 0) TICK(1);
      const bool filled = wait_count(&sc[SC_FILL], (uint32_t)((n + 63) >> 6), &sc[SC_ERR], p.wait_spins);
      const bool stw_ = p.stamps != nullptr && wave == 9;   // profiling level 2: wave 9 (a pure consumer) stamps its item loop
      unsigned long long ct0_ = 0;
      if (stw_) { ct0_ = __builtin_readcyclecounter(); if (lane == 0) L.stm[50] += ct0_ - tick_; }   // table complete (since the step began)
      const uint32_t total_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((LDS_AS uint32_t*)&sc[SC_NI], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      const uint32_t n_items = !filled ? 0u : (total_items < own_n ? total_items : own_n);   // (a wait that gave up: this wave takes no items)
      // (Two items per lane -- 128-item chunks, both chains of LDS round trips in flight -- measured SLOWER, 3.92 against 3.81 ms per
      // 64 x 250 frames: a chunk then took twice as long.  The phase is bound by instruction issue, not by the latency of the chain.)
      // The loop is controlled by scalars only (the chunk cursor read through lane 0, the item count): a uniform branch.  Lanes past
      // the end of the last chunk skip the body inside an if-region -- no `continue`, no lane-dependent exit: with those the
      // compiler may send lanes round the loop on their own, and the wave-level read would then miss lane 0.
      auto take_chunk = [&]() -> uint32_t {
        uint32_t v = 0;
        if (lane == 0) v = lds_add((LDS_AS uint32_t*)&sc[SC_ICUR], 64u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
      };
#pragma unroll 1
      for (uint32_t xb = lmw_ ? n_items : take_chunk(); xb < n_items; xb = take_chunk()) {   // (the LM waves take no items: they are on their way to the queue of scored extensions)
        if (stw_) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) { L.stm[51] += t_ - ct0_; L.stm[52] += 1; } ct0_ = t_; }   // 51: taking a chunk (+ the previous body's tail), 52: chunks
        const uint32_t x = xb + (uint32_t)lane;
        if (x < n_items) {
          const uint32_t oc = (uint32_t)own[x];
          const int i = (int)(oc & 511u);
          const uint32_t c = oc >> 9;
          const uint32_t smj = L.sm[cur][i];
          const uint32_t aidx = L.a0[cur][i] + (uint32_t)__popc(smj & ((1u << c) - 1u));  // the arc of label c
          uint32_t child_fst;  // the child's dictionary state: arithmetic in the unfolded tree, the arc's second field otherwise
          if (s.fst_tree) child_fst = c == space_u ? 0u : aidx + 1u; else child_fst = s.fst_arcs[aidx].y;
          const float sci = L.score[cur][i];
          const uint32_t chi = L.ch[cur][i];
          const float lpc = lpv[c];
          // :199-207, without branches (the read of pb is unconditional: one more LDS read, no exec juggling around two of them)
          const float pbi = L.pb[cur][i];
          const bool rep = c == chi;
          float log_p = __fadd_rn(lpc, rep ? pbi : sci);
          if (rep && !(pbi > NEG)) log_p = NEG;
          const uint32_t needs_lm = c == space_u ? 1u : 0u;
          const uint64_t ck = child_key(L.key[cur][i], c, p.key_mask);
          const int jj = ht_find(L, ck);
          if (jj >= 0) {  // the child is a live prefix: one extension event per live prefix per step
            if (L.ch[cur][jj] != c) lds_or(&sc[SC_ERR], 0x20);   // collision guard (child_key)
            L.ev_ext[jj] = log_p;
            L.ev_exti[jj] = (uint32_t)i | (needs_lm << 31);
            // (no queue of scored extensions in this form: bit 31 marks them, the key phase adds the score -- score_ext below)
          } else {
            const int slot = lds_add(&sc[SC_M], 1);
            if ((uint32_t)slot < S.cand_cap()) {
              const uint32_t piv = (uint32_t)i | (c << 16) | (needs_lm << 31);  // (class position == class: no pruning in this mode)
              if ((uint32_t)slot < L.mcap) { L.lc_logp[slot] = log_p; L.lc_pi[slot] = piv; L.lc_fst[slot] = (int)child_fst; }
              else { S.c_logp()[slot] = log_p; S.c_pi()[slot] = piv; S.c_fst()[slot] = (int)child_fst; }
            }
          }
        }
        if (stw_) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) L.stm[53] += t_ - ct0_; ct0_ = t_; }   // 53: the chunk's body
      }
      if (total_items <= own_n) break;  // (uniform) everything was in the table
      __syncthreads();
      if (tid == 0) { sc[SC_NI] = 0; sc[SC_ICUR] = 0; sc[SC_FILL] = 0; }
      __syncthreads();
    }
  } else if (!(lm_wave && wave >= nw_exp)) {
    uint32_t ppw = pow2_ge((uint32_t)((n + nw_exp - 1) / nw_exp));  // <= 64 (n <= 64 * 15 when the last wave is set aside: beams <= 512)
    const int i0 = lane * nw_exp + wave;  // interleaved: the beam is sorted by score and good prefixes survive the cut-off for more labels, so every wave gets its share of them
    uint32_t cnt = 0, a0 = 0;
    if (lane < (int)ppw && i0 < n) {
      const int i = i0;
      const float sci = L.score[cur][i];
      if (sci != NEG) {  // :160-162
        const uint32_t chi = L.ch[cur][i];
        {  // blank, :166-179
          const int kb = POS_OF(p.blank);
          if (kb != 0xFFFF) { const float lpc = LP_AT(kb, p.blank); if (!(full_beam && __fadd_rn(lpc, sci) < min_cutoff)) L.ev_blank[i] = __fadd_rn(lpc, sci); }
        }
        if (chi != STT_ROOT_CH) {  // repeated character, :182-193
          const int ks = POS_OF(chi);
          if (ks != 0xFFFF) { const float lpc = LP_AT(ks, chi); if (!(full_beam && __fadd_rn(lpc, sci) < min_cutoff)) L.ev_self[i] = __fadd_rn(lpc, L.pnb[cur][i]); }
        }
        if (SC_ON) {
          if (L.a0.p0) { a0 = L.a0[cur][i]; cnt = L.an[cur][i] & 0x7FFFu; }
          else { const int st = L.fst[cur][i]; a0 = s.fst_state_pos[st]; cnt = s.fst_state_pos[st + 1] - a0; }
        } else cnt = (uint32_t)cutoff_len;
      }
    }
    const uint32_t inc = wave_incl_scan(cnt, lane);
    const uint32_t off = inc - cnt;
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    // Items are taken 64 at a time; the FST arcs of the next two rounds are already in flight while a round is processed,
    // so the HBM/L2 latency of the arc reads overlaps the LDS work of the earlier items.
    // item -> owning lane: a per-wave byte table filled by the owners (one LDS read per item) when the wave's items fit,
    // else a binary search over the lanes' offsets with shuffles
    const bool use_tab = n_items <= L.own_cap;  // wave-uniform
    LDS_AS uint8_t* own = L.own + (uint32_t)wave * L.own_cap;
    if (use_tab) {
      for (uint32_t kk = 0; kk < cnt; ++kk) own[off + kk] = (uint8_t)lane;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#define EXPAND_LOCATE(X, J, KK, V, ARC)                                                                         \
    {                                                                                                             \
      const uint32_t x_ = (X);                                                                                    \
      uint32_t j_ = 0; /* largest lane with off_j <= x (lanes with no items share their successor's offset) */    \
      if (use_tab) j_ = x_ < n_items ? (uint32_t)own[x_] : 0u;                                                    \
      else for (uint32_t step = ppw >> 1; step >= 1; step >>= 1) { const uint32_t v_ = __shfl(off, (int)(j_ + step)); if (v_ <= x_) j_ += step; } \
      const uint32_t offj_ = __shfl(off, (int)j_), a0j_ = __shfl(a0, (int)j_);                                    \
      V = x_ < n_items; J = j_; KK = x_ - offj_;                                                                  \
      ARC = make_uint2(0, 0);                                                                                     \
      if (V && SC_ON) ARC = s.fst_arcs[a0j_ + KK];                                                                \
    }
    TICK(1);
    uint32_t j_0 = 0, kk_0 = 0, j_1 = 0, kk_1 = 0, j_2 = 0, kk_2 = 0;
    uint2 arc_0 = make_uint2(0, 0), arc_1 = arc_0, arc_2 = arc_0;
    bool v_0 = false, v_1 = false, v_2 = false;
    if (n_items > 0) EXPAND_LOCATE(lane, j_0, kk_0, v_0, arc_0)
    if (n_items > 64) EXPAND_LOCATE(64 + lane, j_1, kk_1, v_1, arc_1)
#pragma unroll 1
    for (uint32_t xb = 0; xb < n_items; xb += 64) {
      unsigned long long xt0_ = 0;
      if (p.stamps && wave == 0) xt0_ = __builtin_readcyclecounter();
      if (xb + 128 < n_items) EXPAND_LOCATE(xb + 128 + lane, j_2, kk_2, v_2, arc_2)  // uniform condition
      if (p.stamps && wave == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) L.stm[50] += t_ - xt0_; xt0_ = t_; }
      const uint32_t iju = j_0, kku = kk_0; const uint2 arc_c = arc_0; const bool vu = v_0;
      j_0 = j_1; kk_0 = kk_1; arc_0 = arc_1; v_0 = v_1;
      j_1 = j_2; kk_1 = kk_2; arc_1 = arc_2; v_1 = v_2;
      v_2 = false;
      {
        if (!vu) continue;
        const int i = (int)iju * nw_exp + wave;
        uint32_t c; int k; int child_fst = 0;
        if (SC_ON) {
          const uint2 arc = arc_c;
          if (arc.x == 0 || arc.x > (uint32_t)(C - 1)) continue;  // epsilon / label outside the alphabet: never matched
          c = arc.x - 1;
          child_fst = (int)arc.y;
          k = POS_OF(c);
          if (k == 0xFFFF) continue;
        } else {
          k = (int)kku;
          c = CLS_AT(k);
        }
        if ((int)c == p.blank) continue;
        const float sci = L.score[cur][i];
        const uint32_t chi = L.ch[cur][i];
        const float lpc = LP_AT(k, c);
        if (full_beam && __fadd_rn(lpc, sci) < min_cutoff) continue;  // the `break` of :157-159 (beam is sorted by score)
        float log_p = NEG;  // :199-207
        if (c == chi) { const float pbi = L.pb[cur][i]; if (pbi > NEG) log_p = __fadd_rn(lpc, pbi); }
        else log_p = __fadd_rn(lpc, sci);
        uint32_t needs_lm = 0;
        if (SC_ON) {
          if (SC_UTF8) {
            if (al.byte_labels) needs_lm = utf8_completes(L.run[cur][i], (uint8_t)(c + 1)) ? 1u : 0u;  // (UTF8Alphabet: label c is byte c + 1)
            else needs_lm = is_scoring_boundary(s, al, S.pa_generic(), L.node[cur][i], c, c, probes) ? 1u : 0u;
          } else needs_lm = (int)c == al.space_id ? 1u : 0u;
        }
        const uint64_t ck = child_key(L.key[cur][i], c, p.key_mask);
        if (p.stamps && wave == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) { L.stm[51] += t_ - xt0_; L.stm[54] += 1; } xt0_ = t_; }
        // (code-point scorer, see `thr`: a new prefix that cannot reach the beam even with the best language-model score any code point
        // has is not made; one without a score to come is compared as it is.  Extensions into LIVE prefixes are events of prefixes
        // that stay: they are never dropped -- decided below, once the hash has been asked)
        const bool hopeless = SC_UTF8 && thr != NEG && (needs_lm == 0 || s.cp_ub_on != 0) && ((needs_lm ? lm_bound(log_p, s.cp_ub_max) : log_p) < thr);
        int jj = -1;
        {  // (utf8 mode: a one-bit filter first -- a wave pays for its slowest lane's probe sequence, and ~97 % of the lookups miss)
          const uint32_t fb = (uint32_t)(ck >> 28) & (BLOOM_WORDS * 32 - 1);
          if (!L.bloom || ((L.bloom[fb >> 5] >> (fb & 31u)) & 1u)) jj = ht_find(L, ck);
        }
        if (p.stamps && wave == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) L.stm[52] += t_ - xt0_; xt0_ = t_; }
        if (jj >= 0) {  // the child is a live prefix: one extension event per live prefix per step
          if (L.ch[cur][jj] != c) lds_or(&sc[SC_ERR], 0x20);     // collision guard (child_key)
          L.ev_ext[jj] = log_p;
          L.ev_exti[jj] = (uint32_t)i | (needs_lm << 31);
          if (needs_lm && lm_queue) { const int qi = lds_add(&sc[SC_NQ], 1); L.ssrc[qi] = 0x80000000u | (uint32_t)jj; }
        } else if (!hopeless) {
          const int slot = lds_add(&sc[SC_M], 1);
          if ((uint32_t)slot < S.cand_cap()) {
            const uint32_t piv = (uint32_t)i | ((uint32_t)k << 16) | (needs_lm << 31);
            if ((uint32_t)slot < L.mcap) { L.lc_logp[slot] = log_p; L.lc_pi[slot] = piv; L.lc_fst[slot] = child_fst; }
            else { S.c_logp()[slot] = log_p; S.c_pi()[slot] = piv; S.c_fst()[slot] = child_fst; }
            if (needs_lm && lm_queue) { const int qi = lds_add(&sc[SC_NQ], 1); L.ssrc[qi] = (uint32_t)slot; }
          }
        }
        if (p.stamps && wave == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) L.stm[53] += t_ - xt0_; }
      }
    }
  }
#undef EXPAND_LOCATE
  // profiling level 2: when did each wave reach the end of the expand phase (cycles since the step began; slot = wave), and
  // how long did it then wait (slot 16 + wave)
  // ... and which wave was last (slot 32 + wave counts the steps it was; 48: sum of the last wave's arrival, 49: of the second-last)
  unsigned long long arrive_ = 0;
  if (p.stamps && lane == 0) { arrive_ = __builtin_readcyclecounter(); L.stm[wave] += arrive_ - tick_; if (!MASKED) ((LDS_AS unsigned long long*)L.wtot)[wave] = arrive_; }
  if (MASKED) {
    // The end of the expand phase is a barrier among the item-taking waves only (a counter): the LM waves are still in their
    // queries, and what the later phases need from them -- the scores of this step's "prefix + space" extensions -- is only read
    // in the key phase, after the next real barrier (score_ext below); an LM wave goes straight to that barrier when it is done.
    if (!lmw_) { signal_count(&sc[SC_DONE]); (void)wait_count(&sc[SC_DONE], n_cons, &sc[SC_ERR], p.wait_spins); }
    else {
      // (no barrier after the score phase, see there: an LM wave publishes its scores through a counter, and its threads take elements of
      // the key phase like everybody's -- for that it needs the candidate list complete itself)
      signal_count(&sc[SC_LMD]);
      (void)wait_count(&sc[SC_DONE], n_cons, &sc[SC_ERR], p.wait_spins);
    }
  } else __syncthreads();
  if (p.stamps && lane == 0) L.stm[16 + wave] += __builtin_readcyclecounter() - arrive_;
  if (!MASKED && p.stamps && tid == 0) {
    unsigned long long mx = 0, mx2 = 0, mn = ~0ull; int who = 0;
    for (int w = 0; w < NWAVES; ++w) { const unsigned long long a = ((LDS_AS unsigned long long*)L.wtot)[w]; if (a > mx) { mx2 = mx; mx = a; who = w; } else if (a > mx2) mx2 = a; if (a < mn) mn = a; }
    L.stm[32 + who] += 1; L.stm[48] += mx - mn; L.stm[49] += mx2 - mn;
  }
  int m = __builtin_amdgcn_readfirstlane(sc[SC_M]);
  if ((uint32_t)m > S.cand_cap()) { m = (int)S.cand_cap(); if (tid == 0) sc[SC_ERR] |= 4; }
  TICK(2);
  STEP_FENCE();

  // ---- code-point scorer: where will the cut fall?  (`thr` above is the weakest such statement: the WORST live prefix's blank event.)
  // Every live prefix ends the step at or above max(blank event, repeat event), every new prefix WITHOUT a language-model score to
  // come ends it exactly at its log-probability: a histogram of these values over [thr - 4, best score] and a count from the top
  // give a bin below which `beam` entries are already known to lie above -- theta, its lower edge less one bin for the rounding of
  // the bin arithmetic.  Whatever ends below theta is not in the next beam: new prefixes below it are left out of the selection, and a
  // new prefix that waits for a language-model score is not scored when log-probability + the best score any code point can have
  // (DevScorer::cp_ub / cp_ub_max) + beta stays below it.  With near-uniform emissions that is 95 % of the FullScores of a step.
  float theta = NEG;
  if (SC_UTF8 && thr != NEG && n + m > beam) {   // (uniform)
    const float hi = L.score[cur][0], lo = __fsub_rn(thr, 4.0f);
    const float scale = hi > lo ? 1023.0f / (hi - lo) : 0.0f;
    auto bin_of = [&](float v) -> uint32_t { if (!(v > lo)) return 0u; const float t_ = (v - lo) * scale; return t_ >= 1023.0f ? 1023u : (uint32_t)t_; };
    if (tid < n) {
      const float eb = L.ev_blank[tid], es = L.ev_self[tid];
      float v = NEG;
      if (!is_absent(eb)) v = eb;
      if (!is_absent(es) && es > v) v = es;
      if (v != NEG) lds_add(&L.hist[bin_of(v)], 1u);
    }
    for (int x = tid; x < m; x += NTHREADS) {
      const uint32_t pi = CAND_PI(x);
      if (!(pi >> 31)) lds_add(&L.hist[bin_of(CAND_LOGP(x))], 1u);
    }
    __syncthreads();
    const uint32_t hb = L.hist[1023 - tid];   // thread t: bin 1023 - t, so that the scan counts from the top
    uint32_t all_;
    const uint32_t above = block_excl_scan(hb, L.wtot, all_);
    if (above < (uint32_t)beam && (uint32_t)beam <= above + hb) sc[SC_THB] = 1023 - tid;
    __syncthreads();
    const int tb = __builtin_amdgcn_readfirstlane(sc[SC_THB]);
    if (tb >= 2 && scale > 0.0f) theta = lo + (float)(tb - 1) / scale;
    L.hist[tid] = 0;   // (the selection starts from an empty histogram; a barrier follows before it is used)
  }

  // ---- P3: language model on scoring boundaries (:209-243).  Word mode: the few scored extensions of this step were
  // queued by the expand phase and are taken by the *last* threads of the workgroup, while the first n threads already
  // merge every live prefix that does not wait for a score.  Meanwhile the next row's class log-probs are prepared and
  // the (now dead) hash is cleared for the next beam.
  if (MASKED) { if (!lmw_) for (uint32_t h = tid; h < HTN; h += n_cons * 64u) L.ht_key[h] = 0; }  // (the LM waves come later and go straight to the queue)
  else for (uint32_t h = tid; h < HTN; h += NTHREADS) L.ht_key[h] = 0;
  if (L.bloom) for (uint32_t h = tid; h < BLOOM_WORDS; h += NTHREADS) L.bloom[h] = 0;
  if (!WIDE && next_row) { if (MASKED) prep_row_at(p, L, buf ^ 1, prep_c, pre); else prep_row(p, L, buf ^ 1, next_row, pre); }
  bool merged = false;
  float my_score = NEG;
  // bitmap form: log_p of a "prefix i + space" extension with the language-model score of prefix i added (:209-243) -- the score the
  // LM waves (or, without them, the threads of the phase below) left in pqs
  auto score_ext = [&](float lp0, int i) -> float {
    float lms = 0.0f;
    if (L.pqe[cur][i] != STT_NONE) lms = L.pqs[cur][i]; else lds_or(&sc[SC_ERR], 8);
    const float lpv = __fadd_rn(lp0, lms);                  // log_p += score;
    return (float)__dadd_rn((double)lpv, s.beta);           // log_p += ext_scorer_->beta;
  };
  if (SC_ON) {
    unsigned lmq = 0;
    if (MASKED) {  // no queue in this form; the LM waves (still in their queries, or waiting at the barrier below) have nothing to do here
      if (!lm_wave && tid < n) {
        // no LM waves on the first steps of a stream (n <= 16): every prefix whose word may end gets its score here, whether or not
        // the extension survives the cut-off (the entry is a cache of what the reference would compute when it needs it)
        const int i = tid;
        if (((L.sm[cur][i] >> space_u) & 1u) && L.pqe[cur][i] == STT_NONE && L.bnd[cur][i] != STT_NONE && L.score[cur][i] != NEG) {
          uint32_t ne;
          const double raw = lm_word_query_cached<MASKED>(s, al, S, lab1, (LDS_AS uint32_t*)&sc[SC_BEN], L.node[cur][i], L.bnd[cur][i], true, L.wlo[cur][i], L.whi[cur][i], ne, probes);
          L.pqe[cur][i] = ne; L.pqs[cur][i] = (float)__dmul_rn(raw, s.alpha);
          ++lmq;
        }
      }
      if (lm_wave) {
        // With LM waves: their scores are waited for HERE, through their counter -- since round 6 they work off a list made when the beam was
        // written and are done ~3 k cycles before the items are, so the wait is free in most steps -- and every live prefix is merged in one
        // go.  (Before, a prefix whose extension waited for a score was merged in the key phase: every wave holding one ran the merge twice.)
        (void)wait_count(&sc[SC_LMD], (uint32_t)nlm, &sc[SC_ERR], p.wait_spins);
        if (!lmw_ && tid < n) {
          const uint32_t xi = L.ev_exti[tid];
          if ((xi >> 31) && !is_absent(L.ev_ext[tid])) L.ev_ext[tid] = score_ext(L.ev_ext[tid], (int)(xi & 0xFFFFu));
          my_score = merge_live<WIDE>(p, L, W, cur, tid); merged = true;
        }
      } else if (!lmw_ && tid < n && !((L.ev_exti[tid] >> 31) && !is_absent(L.ev_ext[tid]))) { my_score = merge_live<WIDE>(p, L, W, cur, tid); merged = true; }
    } else if (lm_queue) {
      const int nq = __builtin_amdgcn_readfirstlane(sc[SC_NQ]);
      for (int q = NTHREADS - 1 - tid; q < nq; q += NTHREADS) {
        const uint32_t ent = L.ssrc[q];
        const bool live = (ent >> 31) != 0;
        const int x = (int)(ent & 0x7FFFFFFFu);  // candidate slot, or live prefix index
        uint32_t pi; float lp0;
        if (!live) { pi = CAND_PI(x); lp0 = CAND_LOGP(x); } else { pi = L.ev_exti[x]; lp0 = L.ev_ext[x]; }
        const int i = (int)(pi & 0xFFFFu);
        const uint32_t nodei = L.node[cur][i];
        // word mode scores the prefix *before* the space (:211-216); the score depends only on that prefix, so it is
        // computed once per path node and kept (BEntry); later timesteps that retry "prefix + space" reuse it.
        // The root prefix (no word yet: the reference's n-gram is empty and scores 0) is entry 0, recorded in pq[0] when
        // the stream is created; a prefix that itself ends in a space contributes the empty word, which the backward walk
        // of lm_word_query_cached() produces by itself.  Entries cannot run out before path nodes do (one per node).
        double raw = 0.0;
        float lms_cached = 0.0f;
        const bool in_lds = L.pqe.p0 != nullptr;
        const uint32_t e = in_lds ? L.pqe[cur][i] : S.pq()[nodei];
        const bool cached_f = in_lds && e != STT_NONE;  // scored earlier (or by the LM wave during expand): alpha-scaled score is in LDS
        if (cached_f) lms_cached = L.pqs[cur][i];
        else if (e != STT_NONE) raw = load_be_raw(S, e);
        else {
          const uint32_t bndi = L.bnd[cur][i];
          if (bndi == STT_NONE) { raw = 0.0; lds_or(&sc[SC_ERR], 8); }
          else {
            uint32_t ne;
            raw = lm_word_query_cached<MASKED>(s, al, S, lab1, (LDS_AS uint32_t*)&sc[SC_BEN], nodei, bndi, in_lds, in_lds ? L.wlo[cur][i] : 0ULL,
                                       in_lds ? L.whi[cur][i] : 0ULL, ne, probes);
            if (in_lds) { L.pqe[cur][i] = ne; L.pqs[cur][i] = (float)__dmul_rn(raw, s.alpha); }
            ++lmq;
          }
        }
        const float lms = cached_f ? lms_cached : (float)__dmul_rn(raw, s.alpha);
        float lpv = __fadd_rn(lp0, lms);                       // log_p += score;
        lpv = (float)__dadd_rn((double)lpv, s.beta);           // log_p += ext_scorer_->beta;
        if (!live) { if ((uint32_t)x < L.mcap) L.lc_logp[x] = lpv; else S.c_logp()[x] = lpv; } else L.ev_ext[x] = lpv;
      }
      if (tid < n && !((L.ev_exti[tid] >> 31) && !is_absent(L.ev_ext[tid]))) { my_score = merge_live<WIDE>(p, L, W, cur, tid); merged = true; }
    } else {
      // (a candidate's record is read one trip ahead: beyond the first `mcap` it sits in the stream's HBM workspace, and the read was the
      // first of three dependent round trips of a trip -- record, boundary entry + unigram record, bigram block)
      uint32_t pi_nx = 0; float lp_nx = 0.0f;
      if (tid < m) { pi_nx = CAND_PI(tid); lp_nx = CAND_LOGP(tid); }
      for (int x = tid; x < m + n; x += NTHREADS) {
        uint32_t pi; float lp0;
        const uint32_t pi_c = pi_nx; const float lp_c = lp_nx;
        if (x + NTHREADS < m) { pi_nx = CAND_PI(x + NTHREADS); lp_nx = CAND_LOGP(x + NTHREADS); }
        if (x < m) { pi = pi_c; if (!(pi >> 31)) continue; lp0 = lp_c; }
        else { const int j = x - m; pi = L.ev_exti[j]; if (!(pi >> 31) || is_absent(L.ev_ext[j])) continue; lp0 = L.ev_ext[j]; }
        const int i = (int)(pi & 0xFFFFu);
        const uint32_t first = (x < m) ? CLS_AT((pi >> 16) & 0x7FFFu) : L.ch[cur][x - m];  // utf8 mode scores the *new* prefix
        double raw;
        uint32_t unit = 0;
        const uint32_t bndi = L.bnd[cur][i];
        if (al.byte_labels && bndi != STT_NONE && utf8_step_clean(L.run[cur][i], L.ch[cur][i] == STT_ROOT_CH, (uint8_t)(first + 1), unit)) {
          if (theta != NEG && s.cp_ub_on != 0 && x < m) {   // (a NEW prefix only: an extension into a live prefix is an event of a prefix that stays)
            bool dead = lm_bound(lp0, s.cp_ub_max) < theta;      // no read needed for this one
            if (!dead && s.cp_ub != nullptr) {
              const uint32_t b0 = unit & 0xFFu, nbytes = utf8_unit_len(b0);
              uint32_t cp = 0xFFFFFFFFu;   // the code point these bytes spell (an over-long form lands on the entry of the proper one: a larger bound, still one)
              if (nbytes == 1) cp = b0;
              else if (nbytes == 2) cp = ((b0 & 0x1Fu) << 6) | ((unit >> 8) & 0x3Fu);
              else if (nbytes == 3) cp = ((b0 & 0x0Fu) << 12) | (((unit >> 8) & 0x3Fu) << 6) | ((unit >> 16) & 0x3Fu);
              if (cp < 65536u) dead = lm_bound(lp0, s.cp_ub[cp]) < theta;
            }
            if (dead) {
              if ((uint32_t)x < L.mcap) L.lc_logp[x] = NEG; else S.c_logp()[x] = NEG;
              continue;
            }
          }
          uint32_t ne;  // one FullScore from the state after the previous code point
          raw = lm_word_query_cached<false, true>(s, al, S, lab1, (LDS_AS uint32_t*)nullptr, 0u, bndi, true, (uint64_t)unit, 0ULL, ne, probes);
        } else raw = lm_score(s, al, S.pa_generic(), L.node[cur][i], first, true, probes);
        ++lmq;
        const float lms = (float)__dmul_rn(raw, s.alpha);
        float lpv = __fadd_rn(lp0, lms);
        lpv = (float)__dadd_rn((double)lpv, s.beta);
        if (x < m) { if ((uint32_t)x < L.mcap) L.lc_logp[x] = lpv; else S.c_logp()[x] = lpv; } else L.ev_ext[x - m] = lpv;
      }
    }
    if (lmq) lds_add(&sc[SC_LMQ], (int)lmq);
  }
  if (probes) lds_add(&sc[SC_PROBES], (int)probes);
  if (MASKED && p.stamps && lane == 0) L.stm[32 + wave] += __builtin_readcyclecounter() - tick_;   // profiling level 2: arrival at the end of the score phase (wave 0: since its last TICK)
  // Bitmap step with LM waves: NO barrier between the score phase and the key phase (round 6: the LM waves were the last to arrive in half of
  // the steps, 7.5 k cycles after the first wave).  What the merges and the key phase need from them -- pqe / pqs of this step's "prefix +
  // space" extensions -- was waited for through their counter; everything else the key phase reads was complete when the items were (SC_DONE).
  if (!(MASKED && lm_wave)) __syncthreads();   // (with LM waves: waited for above, before the merges)
  TICK(3);
  STEP_FENCE();

  // ---- P4: merge events of live prefixes in the reference's visiting order (class position, then beam index);
  // selection keys of live prefixes and candidates.  Element e (live prefix e < n, else candidate e - n) belongs to
  // thread e % NTHREADS; its first two keys stay in registers, further ones go to the HBM workspace.
  if (MASKED) {  // (an LM wave passed the end of the expand phase before the items were done: the count is final now)
    m = __builtin_amdgcn_readfirstlane(sc[SC_M]);
    if ((uint32_t)m > S.cand_cap()) m = (int)S.cand_cap();
  }
  int total = n + m;
  uint64_t kreg0 = ~0ULL, kreg1 = ~0ULL;
  uint32_t hmin = 0xFFFFFFFFu, hmax = 0;
  if (SC_UTF8) {
    // code-point step: the live prefix's key stays in a register; the new prefixes that can still reach the beam (log-probability at or
    // above theta) put {key, element} on ONE compact list in the stream's workspace -- with theta in force a few thousand of the ~60 k
    // a step makes -- and the selection below walks that list instead of every candidate
    if (tid < n) {
      const float nscore = merged ? my_score : merge_live<WIDE>(p, L, W, cur, tid);
      kreg0 = sel_key(nscore, L.ch[cur][tid], 0, (uint32_t)tid);
      const uint32_t kh = (uint32_t)(kreg0 >> 32);
      hmin = kh; if (kh != NEG_HI) hmax = kh;
    }
    for (int x = tid; x < m; x += NTHREADS) {
      const float lpx = CAND_LOGP(x);
      if (lpx < theta) continue;   // (also the new prefixes parked at -inf above; theta == -inf: nothing is left out)
      const uint32_t pi = CAND_PI(x);
      const uint64_t k = sel_key(lpx, CLS_AT((pi >> 16) & 0x7FFFu), 1, pi & 0xFFFFu);
      const uint32_t at = lds_add((LDS_AS uint32_t*)&sc[SC_NS], 1u);
      S.sel_keys()[at] = k; S.c_elem()[at] = (uint32_t)(n + x);
      const uint32_t kh = (uint32_t)(k >> 32);
      hmin = kh < hmin ? kh : hmin; if (kh != NEG_HI) hmax = kh > hmax ? kh : hmax;
    }
  } else
  for (int e = tid, r = 0; e < total; e += NTHREADS, ++r) {
    uint64_t k;
    if (e < n) {  // e == tid: either merged during the LM phase, or it waited for a score
      if (MASKED && !merged) {
        const uint32_t xi = L.ev_exti[e];
        if ((xi >> 31) && !is_absent(L.ev_ext[e])) L.ev_ext[e] = score_ext(L.ev_ext[e], (int)(xi & 0xFFFFu));
      }
      const float nscore = merged ? my_score : merge_live<WIDE>(p, L, W, cur, e);
      k = sel_key(nscore, L.ch[cur][e], 0, (uint32_t)e);
    } else {
      const int x = e - n;
      const uint32_t pi = CAND_PI(x);
      float lpx = CAND_LOGP(x);
      if (MASKED && (pi >> 31)) { lpx = score_ext(lpx, (int)(pi & 0xFFFFu)); if ((uint32_t)x < L.mcap) L.lc_logp[x] = lpx; else S.c_logp()[x] = lpx; }
      k = sel_key(lpx, CLS_AT((pi >> 16) & 0x7FFFu), 1, pi & 0xFFFFu);
    }
    if (r == 0) kreg0 = k; else if (r == 1) kreg1 = k; else S.sel_keys()[e] = k;
    const uint32_t kh = (uint32_t)(k >> 32);
    hmin = kh < hmin ? kh : hmin; if (kh != NEG_HI) hmax = kh > hmax ? kh : hmax;
  }
  // element q of the selection: its key / which prefix it is (live prefix e < n, else candidate slot e - n).  Code-point step: q < n is
  // the live prefix q (key in kreg0 of thread q), q >= n entry q - n of the compact list; everywhere else q IS the element.
#define KEY_OF(e, r) (SC_UTF8 ? ((e) < n ? kreg0 : S.sel_keys()[(e) - n]) : ((r) == 0 ? kreg0 : (r) == 1 ? kreg1 : S.sel_keys()[e]))
#define ELEM_OF(e) (SC_UTF8 ? ((e) < n ? (uint32_t)(e) : S.c_elem()[(e) - n]) : (uint32_t)(e))
  hmin = wave_min_u32(hmin); hmax = wave_max_u32(hmax);
  if (lane == 0) { lds_min((LDS_AS uint32_t*)&sc[SC_KMIN], hmin); lds_max((LDS_AS uint32_t*)&sc[SC_KMAX], hmax); }
  __syncthreads();
  TICK(4);
  STEP_FENCE();
  if (SC_UTF8) total = n + (int)__builtin_amdgcn_readfirstlane(sc[SC_NS]);

  // ---- P5: keep the best beam_size (nth_element + resize, :263-274) in sorted order.  Keys are unique, so the rank of
  // a key is its position in the new beam.  Bucket the keys over their live range (NBUCKET bins of 2^sh), prefix-sum the
  // histogram, and find the bucket that holds the keep-th key.  Whole buckets below it are kept; every kept key is
  // scattered to its bucket's segment and ranked inside the segment by pairwise comparison.  A threshold bucket with
  // more than RCAP members is subdivided (next level) instead.
  const int keep = total < beam ? total : beam;
  const int nxt = cur ^ 1;
  {
    // Scores equal to -NUM_FLT_INF (a repeated label on a prefix whose blank probability is still -inf) would stretch
    // the range over the whole float line; they are re-based right after the worst finite score (monotone, injective).
    const uint32_t kmin = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_KMIN]), kmaxf = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_KMAX]);
    const uint32_t neg_to = (kmaxf >= kmin && kmaxf < NEG_HI) ? kmaxf + 1u : NEG_HI;
#define REKEY(k) (((uint32_t)((k) >> 32) == NEG_HI) ? (((uint64_t)neg_to << 32) | ((k) & 0xFFFFFFFFull)) : (k))
    uint64_t base = (uint64_t)kmin << 32;
    const uint64_t range = ((uint64_t)(neg_to - kmin) << 32) | 0xFFFFFFFFull;
    int sh = 64 - __clzll((long long)range) - 10;  // range < 2^(sh+10)
    if (sh < 0) sh = 0;
    int width_sh = 64;  // keys of the current level satisfy (k - base) >> width_sh == 0
    int need = keep, off = 0;
    uint32_t placed = 0;
    for (;;) {
      for (int e = tid, r = 0; e < total; e += NTHREADS, ++r) {
        const uint64_t k0 = KEY_OF(e, r);
        const uint64_t k = REKEY(k0);
        if (k < base) continue;
        const uint64_t d = k - base;
        if (width_sh < 64 && (d >> width_sh) != 0) continue;
        lds_add(&L.hist[(uint32_t)(d >> sh)], 1u);
      }
      __syncthreads();
      const uint32_t h = L.hist[tid];
      uint32_t in_level;
      const uint32_t ex = block_excl_scan(h, L.wtot, in_level);
      L.cumb[tid] = ex;
      if (tid == 0) L.cumb[NBUCKET] = in_level;
      if (ex < (uint32_t)need && (uint32_t)need <= ex + h) { sc[SC_BT] = tid; sc[SC_BTH] = (int)h; sc[SC_BTCUM] = (int)ex; }
      __syncthreads();
      const uint32_t bt = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_BT]), bth = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_BTH]), btcum = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_BTCUM]);
      const bool last = (bth <= RCAP) || sh == 0;
      for (int e = tid, r = 0; e < total; e += NTHREADS, ++r) {
        const uint64_t k0 = KEY_OF(e, r);
        const uint64_t k = REKEY(k0);
        if (k < base) continue;
        const uint64_t d = k - base;
        if (width_sh < 64 && (d >> width_sh) != 0) continue;
        const uint32_t b = (uint32_t)(d >> sh);
        if (b < bt || (last && b == bt)) {
          const uint32_t seg0 = (uint32_t)off + L.cumb[b];
          const uint32_t len = L.cumb[b + 1] - L.cumb[b];
          const uint32_t at = seg0 + (lds_sub(&L.hist[b], 1u) - 1u);
          L.skey[at] = k; L.ssrc[at] = ELEM_OF(e); L.sseg[at] = seg0 | (len << 16);
        }
      }
      if (last) { placed = (uint32_t)off + btcum + bth; break; }
      off += (int)btcum; need -= (int)btcum;
      base += (uint64_t)bt << sh;
      width_sh = sh;
      sh = sh > 10 ? sh - 10 : 0;
      __syncthreads();
      L.hist[tid] = 0;
      __syncthreads();
    }
    __syncthreads();
    TICK(5);
    STEP_FENCE();

    // rank inside the segment -> position r in the new beam; then write the new beam entry (P6) and hash its key.
    // A surviving live prefix is a copy of ~20 fields, a new prefix ~3x the work (arena node, dictionary record, word bytes); mixed in
    // one loop every wave walked through both bodies.  So: (a) rank, and put {source, position} on one of two lists -- live from the
    // front, new from the back of the (idle) histogram; (b) barrier; (c) the first threads take the live list, the last ones the new
    // list: a wave runs one body.
    LDS_AS uint32_t* wl = L.hist;                      // source element (live prefix x < n, else candidate x - n): NBUCKET = 1024 entries >= keep
    LDS_AS uint16_t* wr = (LDS_AS uint16_t*)L.cumb;    // its position in the new beam (the bucket offsets are dead after the scatter)
    unsigned long long w6_ = 0;
#define P6_STAMP(k) do { if (p.stamps && (wave == 0 || wave == NWAVES - 1)) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) L.stm[(wave == 0 ? 56 : 60) + (k)] += t_ - w6_; w6_ = t_; } } while (0)
    if (p.stamps) w6_ = __builtin_readcyclecounter();
    uint2 rec_new = make_uint2(0, 0); uint32_t r_rec = 0xFFFFFFFFu;  // bitmap form: dictionary record of a new prefix, fetched early (below)
    for (uint32_t q0 = 0; q0 < placed; q0 += NTHREADS) {  // (scalar loop control: ballots inside)
      const uint32_t q = q0 + (uint32_t)tid;
      bool is_live = false, is_new = false;
      uint32_t ent = 0, rr = 0;
      if (q < placed) {
        const uint64_t k = L.skey[q];
        const uint32_t seg = L.sseg[q];
        const uint32_t xsrc = L.ssrc[q];   // (read with the other two: one round trip)
        const uint32_t seg0 = seg & 0xFFFFu, len = seg >> 16;
        uint32_t r = seg0;
        // four keys per trip, their reads in flight together: one read per trip made the longest segment of a wave (up to RCAP keys,
        // an LDS round trip each) the length of this phase -- 5 k cycles on the slowest wave, 2 k on an average one.  (The reads past
        // the segment's end stay inside the key / segment arrays.)
        for (uint32_t t = 0; t < len; t += 4) {
          const uint64_t k0 = L.skey[seg0 + t], k1 = L.skey[seg0 + t + 1], k2 = L.skey[seg0 + t + 2], k3 = L.skey[seg0 + t + 3];
          r += (k0 < k ? 1u : 0u) + ((t + 1 < len && k1 < k) ? 1u : 0u) + ((t + 2 < len && k2 < k) ? 1u : 0u) + ((t + 3 < len && k3 < k) ? 1u : 0u);
        }
        if ((int)r < keep) { ent = xsrc; rr = r; is_live = (int)xsrc < n; is_new = !is_live; }
      }
      if (MASKED && is_new) {
        // {first labelled arc, label bitmap} of the new prefix's dictionary state: a read from a table of tens of MB, i.e. a miss --
        // started here, consumed after this thread's share of the write phase (stays in flight across the barrier), so nobody waits
        const int cx = (int)ent - n;
        rec_new = s.fst_rec[((uint32_t)cx < L.mcap) ? L.lc_fst[cx] : S.c_fst()[cx]]; r_rec = rr;
      }
      const uint64_t bl = __ballot(is_live), bn = __ballot(is_new);
      uint32_t base_ln = 0;   // both list lengths in one word (live low, new high): one atomic round trip per wave
      if (lane == 0 && (bl | bn)) base_ln = lds_add((LDS_AS uint32_t*)&sc[SC_NA], (uint32_t)__popcll(bl) | ((uint32_t)__popcll(bn) << 16));
      base_ln = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_ln);
      const uint32_t base_l = base_ln & 0xFFFFu, base_n = base_ln >> 16;
      const uint64_t below = (1ull << lane) - 1ull;
      if (is_live) { const uint32_t at = base_l + (uint32_t)__popcll(bl & below); wl[at] = ent; wr[at] = (uint16_t)rr; }
      if (is_new) { const uint32_t at = NBUCKET - 1 - (base_n + (uint32_t)__popcll(bn & below)); wl[at] = ent; wr[at] = (uint16_t)rr; }
    }
    P6_STAMP(0);   // profiling level 2 (wave 0: slots 56.., last wave: 60..): ranking | wait at the barrier | own list | tail
    __syncthreads();
    P6_STAMP(1);
    STEP_FENCE();
    const uint32_t n_ln = (uint32_t)__builtin_amdgcn_readfirstlane(sc[SC_NA]);
    const uint32_t n_live = n_ln & 0xFFFFu, n_new = n_ln >> 16;
    if (p.stamps && tid == 0) { L.stm[54] += n_live; L.stm[55] += n_new; }
    auto finish_entry = [&](uint32_t r, uint64_t nkey, uint32_t pend, uint32_t ts_new) {
      L.key[nxt][r] = nkey;
      ht_insert(L, nkey, (int)r);
      if (pend != 0xFFFFFFFEu) {  // path_trie.cpp:172-184
        const uint32_t slot = lds_add((LDS_AS uint32_t*)&sc[SC_TAN], 1u);
        if (slot < S.ta_cap()) { store_node(S.ta(), slot, pend, (uint32_t)abs_t); ts_new = slot; }
        else lds_or(&sc[SC_ERR], 2);
      }
      L.ts[nxt][r] = ts_new;
    };
    for (uint32_t t = tid; t < n_live; t += NTHREADS) {
      const uint32_t x = wl[t], r = (uint32_t)wr[t];
      uint32_t ts_new, pend;
      uint64_t nkey;
      {
        // every field READ first, then every field written: the compiler cannot prove that a store to [nxt][r] leaves a later load of
        // [cur][x] alone (same address space, run-time indices), so "a = b; c = d; ..." became twenty dependent LDS round trips (3.0 k cycles of the
        // write phase on the waves that copy live entries, round 6 stamps); with the loads in flight together it is one.
        const float v_score = L.ev_ext[x], v_pb = L.ev_blank[x], v_pnb = L.ev_self[x];
        const uint32_t v_ch = L.ch[cur][x], v_node = L.node[cur][x], v_bnd = L.bnd[cur][x];
        const int v_fst = L.fst[cur][x];
        uint32_t v_a0 = 0, v_sm = 0, v_pqe = 0, v_run = 0; uint16_t v_an = 0; uint64_t v_wlo = 0, v_whi = 0; float v_pqs = 0.0f;
        const bool arcs_plain = !MASKED && L.a0.p0 != nullptr, words = MODE == 1 && L.pqe.p0 != nullptr;
        if (MASKED) { v_a0 = L.a0[cur][x]; v_sm = L.sm[cur][x]; }
        else if (arcs_plain) { v_a0 = L.a0[cur][x]; v_an = L.an[cur][x]; }
        nkey = L.key[cur][x];
        if (words) { v_wlo = L.wlo[cur][x]; v_whi = L.whi[cur][x]; v_pqe = L.pqe[cur][x]; v_pqs = L.pqs[cur][x]; }
        if (MODE == 2) v_run = L.run[cur][x];
        pend = L.ev_exti[x]; ts_new = L.ts[cur][x];
        __builtin_amdgcn_sched_barrier(0);
        L.score[nxt][r] = v_score; L.pb[nxt][r] = v_pb; L.pnb[nxt][r] = v_pnb;
        L.ch[nxt][r] = v_ch; L.node[nxt][r] = v_node; L.fst[nxt][r] = v_fst;
        if (MASKED) { L.a0[nxt][r] = v_a0; L.sm[nxt][r] = v_sm; }
        else if (arcs_plain) { L.a0[nxt][r] = v_a0; L.an[nxt][r] = v_an; }
        L.bnd[nxt][r] = v_bnd;
        if (words) { L.wlo[nxt][r] = v_wlo; L.whi[nxt][r] = v_whi; L.pqe[nxt][r] = v_pqe; L.pqs[nxt][r] = v_pqs; }
        if (MODE == 2) L.run[nxt][r] = v_run;
        // (bitmap step: a word may end here and nobody has scored it yet -> the LM list of the new beam)
        if (MASKED && ((v_sm >> (uint32_t)al.space_id) & 1u) != 0 && v_pqe == STT_NONE) L.lmw[(uint32_t)nxt * L.lmw_cap + (uint32_t)lds_add(&sc[SC_NLM0 + nxt], 1)] = (uint16_t)r;
      }
      finish_entry(r, nkey, pend, ts_new);
    }
    for (uint32_t t = NTHREADS - 1 - tid; t < n_new; t += NTHREADS) {
      const uint32_t x = wl[NBUCKET - 1 - t], r = (uint32_t)wr[NBUCKET - 1 - t];
      uint32_t ts_new, pend;
      uint64_t nkey;
      {
        // loads first, in two waves of independent reads (the candidate's record; then everything of its parent), stores last -- see the copy
        // loop above: written as a sequence of "dst = src" statements every load waited for the store before it
        const uint32_t slot = lds_add((LDS_AS uint32_t*)&sc[SC_PAN], 1u);   // (first: its round trip overlaps the reads below)
        const int cx = (int)x - n;
        const uint32_t pi = CAND_PI(cx);
        const float lpv = CAND_LOGP(cx);
        const int cf = ((uint32_t)cx < L.mcap) ? L.lc_fst[cx] : S.c_fst()[cx];
        const int i = (int)(pi & 0xFFFFu);
        const uint32_t c = CLS_AT((pi >> 16) & 0x7FFFu);
        const bool words = MODE == 1 && L.pqe.p0 != nullptr;
        const uint32_t pnode = L.node[cur][i];
        const uint64_t pkey = L.key[cur][i];
        const uint32_t pbnd = L.bnd[cur][i];
        const uint32_t pts = L.ts[cur][i];
        uint32_t ppqe = STT_NONE, prun = 0, pch = 0;
        uint64_t lo = 0, hi = 0;
        if (words) { ppqe = L.pqe[cur][i]; lo = L.wlo[cur][i]; hi = L.whi[cur][i]; }
        if (MODE == 2) { prun = L.run[cur][i]; pch = L.ch[cur][i]; }
        const uint8_t one = (words && lab1) ? lab1[c] : (uint8_t)0;
        uint32_t f0 = 0, f1 = 0, sp = 0;
        const bool arcs_plain = !MASKED && SC_ON && L.a0.p0 != nullptr;
        if (arcs_plain) {  // arc range of the child's dictionary state; top bit: a word may end here (space arc)
          f0 = s.fst_state_pos[cf]; f1 = s.fst_state_pos[cf + 1];
          sp = MODE == 1 ? (uint32_t)s.fst_has_space[cf] : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
        L.score[nxt][r] = lpv; L.pb[nxt][r] = NEG; L.pnb[nxt][r] = lpv;
        L.ch[nxt][r] = c; L.fst[nxt][r] = cf;
        if (arcs_plain) { L.a0[nxt][r] = f0; L.an[nxt][r] = (uint16_t)((f1 - f0) | (sp << 15)); }   // (bitmap form: a0 / sm are written at the end by the thread that ranked this entry)
        nkey = child_key(pkey, c, p.key_mask);
        uint32_t b = pbnd;
        if (MODE == 1 && (int)c == al.space_id) b = L.pqe.p0 ? ppqe : S.pq()[pnode];  // the boundary entry scored in P3 (or earlier)
        L.bnd[nxt][r] = b;
        if (words) {
          if ((int)c == al.space_id) { lo = 0; hi = 0; }
          else if (one) word_push(lo, hi, one);
          else { const int b0 = c ? al.label_off[c - 1] : 0, b1 = al.label_off[c]; for (int bb = b0; bb < b1; ++bb) word_push(lo, hi, al.label_bytes[bb]); }
          L.wlo[nxt][r] = lo; L.whi[nxt][r] = hi; L.pqe[nxt][r] = STT_NONE;
        }
        if (slot < S.pa_cap()) { store_node(S.pa(), slot, pnode, c); S.pq()[slot] = STT_NONE; L.node[nxt][r] = slot; }
        else { L.node[nxt][r] = 0; lds_or(&sc[SC_ERR], 1); }
        if (MODE == 2) {  // utf8 cache: the child's run, and its boundary entry (a new one when it completes a code point)
          const uint8_t byte = (uint8_t)(c + 1);
          uint32_t unit = 0, nb = STT_NONE;
          unsigned probes6 = 0;
          if (al.byte_labels) {
            L.run[nxt][r] = utf8_child_run(prun, byte);
            if (b != STT_NONE && utf8_step_clean(prun, pch == STT_ROOT_CH, byte, unit)) {
              if (pi >> 31) {
                if (slot < S.pa_cap()) {
                  lm_word_query_cached<false, true>(s, al, S, lab1, (LDS_AS uint32_t*)&sc[SC_BEN], slot, b, true, (uint64_t)unit, 0ULL, nb, probes6);
                  if (nb == STT_NONE) lds_or(&sc[SC_ERR], 8);
                }
              } else nb = b;
            }
          } else L.run[nxt][r] = 0;  // not a UTF8Alphabet: every prefix takes the generic walk (nb stays STT_NONE)
          L.bnd[nxt][r] = nb;
        }
        pend = (NEG < lpv) ? pts : 0xFFFFFFFEu;  // :246-251 with log_prob_nb_cur == -inf
        ts_new = STT_ROOT_CH;                              // timesteps == nullptr
      }
      finish_entry(r, nkey, pend, ts_new);
    }
    P6_STAMP(2);
    if (MASKED && r_rec != 0xFFFFFFFFu) {
      L.a0[nxt][r_rec] = rec_new.x; L.sm[nxt][r_rec] = rec_new.y;
      if (((rec_new.y >> (uint32_t)al.space_id) & 1u) != 0) L.lmw[(uint32_t)nxt * L.lmw_cap + (uint32_t)lds_add(&sc[SC_NLM0 + nxt], 1)] = (uint16_t)r_rec;   // (a new prefix has no score yet)
    }
    __builtin_amdgcn_s_waitcnt(0);
    P6_STAMP(3);
#undef P6_STAMP
  }
#undef REKEY
#undef KEY_OF
#undef ELEM_OF
#undef POS_OF
#undef CLS_AT
#undef LP_AT
  if (tid == 0) {
    if (MASKED) { sc[SC_NI] = 0; sc[SC_ICUR] = 0; sc[SC_FILL] = 0; sc[SC_DONE] = 0; sc[SC_LMD] = 0; }  // the next step's pre-pass starts without a barrier of its own
    L.acc[0] += 1; L.acc[1] += (unsigned long long)m; L.acc[2] += (unsigned long long)sc[SC_LMQ]; L.acc[3] += (unsigned long long)(unsigned)sc[SC_PROBES];
  }
  abs_t++;
  cur = nxt;
  n = keep;
  __syncthreads();
  TICK(6);
  STEP_FENCE();
#undef p
#undef s
#undef al
#undef S
#undef STEP_FENCE
}

// MODE: 0 = no scorer, 1 = word-level scorer, 2 = utf8 (codepoint-level) scorer, 4 = word-level scorer with the dictionary's label
// bitmaps, two language-model waves and FullScore through the hashed n-gram index (<= 32 classes, no class pruning, CAP <= 512)
template <int MODE, int CAP, bool WIDE>
__global__ __launch_bounds__(NTHREADS) void ctc_next_kernel(NextArgs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const CONST_AS NextArgs* ka = (const CONST_AS NextArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const DecParams& p = *(const DecParams*)&ka->p;
  const DevScorer& s = *(const DevScorer*)&ka->s;
  const DevAlphabet& al = *(const DevAlphabet*)&ka->al;
  const float* probs = ka->probs; const int* frame_begin = ka->frame_begin; const int* frame_count = ka->frame_count;
  size_t lds_total;
  // (bitmap step: <= 32 classes -- the layout is carved for 32, so every LDS address but the candidate staging capacity is a constant)
  if ((uint32_t)(uintptr_t)(LDS_AS unsigned char*)smem != 0u) __builtin_trap();   // (no static LDS in this kernel: see LDS_ORIGIN)
  const Lds L = lds_carve<CAP>(WIDE ? 0 : (MODE == 4 ? 32 : p.C), (LDS_AS unsigned char*)(uintptr_t)LDS_ORIGIN, lds_total, p.lds_kb, MODE == 2);
  DecStream& G = ka->streams[blockIdx.x];
  const CONST_AS DecStream* gq = (const CONST_AS DecStream*)&G;
  const int nfr = frame_count ? frame_count[blockIdx.x] : p.all_count;
  if (nfr <= 0) return;
  const GStream GS{gq};
  GLB_AS float* g_score = (GLB_AS float*)G.score; GLB_AS float* g_pb = (GLB_AS float*)G.pb; GLB_AS float* g_pnb = (GLB_AS float*)G.pnb;
  GLB_AS uint32_t* g_ch = (GLB_AS uint32_t*)G.ch; GLB_AS uint32_t* g_node = (GLB_AS uint32_t*)G.node; GLB_AS uint32_t* g_ts = (GLB_AS uint32_t*)G.ts;
  GLB_AS int* g_fst = (GLB_AS int*)G.fst; GLB_AS uint64_t* g_key = (GLB_AS uint64_t*)G.key; GLB_AS uint32_t* g_bnd = (GLB_AS uint32_t*)G.bnd;
  constexpr bool SC_ON = MODE != 0;
  constexpr bool MASKED = MODE == 4, WORDC = MODE == 1 || MODE == 4;
  const int tid = threadIdx.x;
  int n = G.n;
  int cur = 0;
  int start_expanding = G.start_expanding;
  int abs_t = G.abs_t;
  const float* row = probs + ((size_t)blockIdx.x * p.t_max + (frame_begin ? frame_begin[blockIdx.x] : p.all_begin)) * p.C;
  const float v0 = (!WIDE && tid < p.C) ? row[tid] : 0.0f;
  for (int i = tid; i < n; i += NTHREADS) {
    L.score[0][i] = g_score[i]; L.pb[0][i] = g_pb[i]; L.pnb[0][i] = g_pnb[i];
    L.ch[0][i] = g_ch[i]; L.node[0][i] = g_node[i]; L.ts[0][i] = g_ts[i]; const int st = g_fst[i]; L.fst[0][i] = st; L.key[0][i] = g_key[i];
    L.bnd[0][i] = g_bnd[i];
    if (MASKED) {
      const uint2 rec = s.fst_rec[st];
      L.a0[0][i] = rec.x; L.sm[0][i] = rec.y;
    } else if (SC_ON && L.a0.p0) {
      const uint32_t f0 = s.fst_state_pos[st], f1 = s.fst_state_pos[st + 1];
      const uint32_t sp = MODE == 1 ? (uint32_t)s.fst_has_space[st] : 0u;
      L.a0[0][i] = f0; L.an[0][i] = (uint16_t)((f1 - f0) | (sp << 15));
    }
  }
  for (int c = tid; !WIDE && c < p.C; c += NTHREADS) {
    uint8_t one = 0;
    if (c < al.n_labels) { const int b0 = c ? al.label_off[c - 1] : 0; if (al.label_off[c] - b0 == 1) one = al.label_bytes[b0]; }
    L.lab1[c] = one;
    L.cls[c] = (uint16_t)c; L.pos[c] = (uint16_t)c;  // identity class order unless pruning re-sorts it every step
  }
  for (uint32_t h = tid; h < HTN; h += NTHREADS) L.ht_key[h] = 0;
  if (L.bloom) for (uint32_t h = tid; h < BLOOM_WORDS; h += NTHREADS) L.bloom[h] = 0;
  if (tid < 32) { L.exp_tab[tid] = sttm::kExp2Tab[tid]; L.log_tab[tid] = sttm::kLogfTab[tid >> 1][tid & 1]; }
  if (tid == 0) { L.sc[SC_NI] = 0; L.sc[SC_ICUR] = 0; L.sc[SC_FILL] = 0; L.sc[SC_DONE] = 0; L.sc[SC_LMD] = 0; L.sc[SC_NLM0] = 0; L.sc[SC_NLM1] = 0; L.sc[SC_ERR] = 0; L.sc[SC_PAN] = (int)G.pa_n; L.sc[SC_TAN] = (int)G.ta_n; L.sc[SC_BEN] = (int)G.be_n; }
  if (tid < 12) L.acc[tid] = 0;
  if (tid < 64) L.stm[tid] = 0;
  __syncthreads();  // the math tables must be in place before the first row is prepared
  if (!WIDE) prep_row(p, L, 0, row, v0);
  __syncthreads();
  const LDS_AS uint8_t* const lab1 = WIDE ? (const LDS_AS uint8_t*)nullptr : (const LDS_AS uint8_t*)L.lab1;
  if (WORDC && L.pqe.p0) {
    unsigned pr = 0;
    for (int i = tid; i < n; i += NTHREADS) {
      const uint32_t nd = L.node[0][i];
      uint64_t lo, hi;
      word_walk(al, GS, lab1, nd, lo, hi, pr);
      const uint32_t e0 = GS.pq()[nd];
      L.wlo[0][i] = lo; L.whi[0][i] = hi; L.pqe[0][i] = e0;
      L.pqs[0][i] = e0 != STT_NONE ? (float)__dmul_rn(load_be_raw(GS, e0), s.alpha) : 0.0f;
    }
  }
  if (MODE == 2) {  // utf8 cache: rebuild each prefix's code point in progress from its path (the bytes back to the last start byte)
    for (int i = tid; i < n; i += NTHREADS) {
      uint64_t acc = 0; uint32_t len = 0; bool found = false;
      for (uint32_t nd = L.node[0][i]; nd != STT_ROOT_CH;) {
        const uint2 pn = load_node(GS.pa(), nd);
        if (pn.y == STT_ROOT_CH) break;
        const uint8_t byte = (uint8_t)(pn.y + 1);
        acc = (acc << 8) | byte; len = len < 255u ? len + 1u : 255u;  // (newest byte first: the start byte ends up lowest)
        if ((byte & 0xC0) != 0x80) { found = true; break; }
        nd = pn.x;
      }
      L.run[0][i] = (found && al.byte_labels) ? (((uint32_t)acc & 0x00FFFFFFu) | (len << 24)) : 0u;
    }
  }
  for (int i = tid; i < n; i += NTHREADS) ht_insert(L, L.key[0][i], i);
  if (MASKED)   // the LM list of the beam as loaded (ctc_step: the eager list)
    for (int i = tid; i < n; i += NTHREADS)
      if (((L.sm[0][i] >> (uint32_t)al.space_id) & 1u) != 0 && L.pqe[0][i] == STT_NONE) L.lmw[lds_add(&L.sc[SC_NLM0], 1)] = (uint16_t)i;
  __syncthreads();
  for (int t = 0; t < nfr; ++t)
    ctc_step<MODE, WIDE>(ka, gq, L, cur, n, start_expanding, abs_t, t & 1, t + 1 < nfr ? row + (size_t)(t + 1) * p.C : nullptr, t);
  for (int i = tid; i < n; i += NTHREADS) {
    g_score[i] = L.score[cur][i]; g_pb[i] = L.pb[cur][i]; g_pnb[i] = L.pnb[cur][i];
    g_ch[i] = L.ch[cur][i]; g_node[i] = L.node[cur][i]; g_ts[i] = L.ts[cur][i]; g_fst[i] = L.fst[cur][i]; g_key[i] = L.key[cur][i];
    g_bnd[i] = L.bnd[cur][i];
  }
  if (tid == 0) {
    G.n = n; G.start_expanding = start_expanding; G.abs_t = abs_t; G.error |= L.sc[SC_ERR];
    G.pa_n = (uint32_t)L.sc[SC_PAN]; G.ta_n = (uint32_t)L.sc[SC_TAN]; G.be_n = (uint32_t)L.sc[SC_BEN];
#pragma unroll
    for (int k = 0; k < 4; ++k) G.stat[k] += L.acc[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) G.phase[k] += L.acc[4 + k];
  }
  if (p.stamps && tid < 64) p.stamps[(size_t)blockIdx.x * 64 + tid] += L.stm[tid];
}

// ------------------------------------------------------------------------------------ decode
// DecoderState::decode (ctc_beam_search_decoder.cpp:278-326): add the LM score of an unfinished last word,
// rank by prefix_compare_external, back-track tokens and timesteps.  Read-only on the stream.
__global__ __launch_bounds__(NTHREADS) void ctc_decode_kernel(DecParams p, DevScorer s, DevAlphabet al, const DecStream* streams, DecodeOut out) {
  __shared__ uint64_t skey[STT_MAX_BEAM];
  __shared__ uint32_t ssrc[STT_MAX_BEAM];
  __shared__ float sscore[STT_MAX_BEAM];
  __shared__ uint8_t lab1[256];
  const DecStream& S = streams[blockIdx.x];
  const int tid = threadIdx.x;
  const int n = S.n;
  const uint32_t sortn = pow2_ge((uint32_t)(n > 0 ? n : 1));
  unsigned probes = 0;
  const GStream GS{(const CONST_AS DecStream*)&S};
  if (tid < 256) {  // single-byte labels (the word walk reads them); labels >= 256 are looked up in HBM
    uint8_t one = 0;
    if (tid < al.n_labels) { const int b0 = tid ? al.label_off[tid - 1] : 0; if (al.label_off[tid] - b0 == 1) one = al.label_bytes[b0]; }
    lab1[tid] = one;
  }
  __syncthreads();
  for (uint32_t i = tid; i < sortn; i += NTHREADS) {
    if ((int)i >= n) { skey[i] = ~0ULL; continue; }
    float sc = S.score[i];
    if (s.enabled && (int)i < p.beam) {
      const uint32_t node = S.node[i], chi = S.ch[i];
      bool do_score;
      if (s.utf8) {
        do_score = !is_scoring_boundary(s, al, S.pa, node, STT_ROOT_CH, chi, probes);  // prefix_boundary = prefix
      } else {
        const uint32_t par = S.pa[node].x;  // prefix_boundary = prefix->parent (null for the root)
        do_score = (par != STT_ROOT_CH) && !((int)chi == al.space_id);
      }
      if (do_score) {
        // :293-297, no hot-word boost here.  Word mode without hot words: the value is exactly the cached score of
        // "this prefix, then a word boundary" (or one FullScore from the cached state of the previous boundary).
        double lcp;
        const uint32_t bndi = S.bnd[i];
        if (!s.utf8 && s.n_hot == 0 && bndi != STT_NONE && al.n_labels <= 256) {
          const uint32_t e = S.pq[node];
          if (e != STT_NONE) lcp = load_be_raw(GS, e);
          else { uint32_t ne; lcp = lm_word_query_cached<false>(s, al, GS, (const LDS_AS uint8_t*)lab1, (LDS_AS uint32_t*)nullptr, node, bndi, false, 0ULL, 0ULL, ne, probes); }
        } else {
          lcp = lm_score(s, al, S.pa, node, STT_ROOT_CH, false, probes);
        }
        float v = (float)__dmul_rn(lcp, s.alpha);
        v = (float)__dadd_rn((double)v, s.beta);
        sc = __fadd_rn(sc, v);
      }
    }
    sscore[i] = sc;
    skey[i] = sel_key(sc, S.ch[i], 0, i);
  }
  __syncthreads();
  // order by key: keys are unique, so a prefix's rank is the number of smaller keys (one pass over LDS instead of the
  // ~45 barrier-separated stages of a bitonic sort)
  for (int i = tid; i < n; i += NTHREADS) {
    const uint64_t k = skey[i];
    int r = 0;
    for (int j = 0; j < n; ++j) r += skey[j] < k ? 1 : 0;
    ssrc[r] = (uint32_t)i;
  }
  __syncthreads();
  const int nret = n < out.num_results ? n : out.num_results;
  if (tid == 0) { out.n_results[blockIdx.x] = nret; out.errors[blockIdx.x] = S.error; }
  // Incremental back-tracking (a stream that is decoded hop after hop, one result asked for): the new best prefix shares all but its
  // last few nodes with the best prefix of the previous decode, so the walk back through the arenas -- one dependent HBM read per token,
  // 150 us for a 5 s utterance, every hop -- stops where it meets the path walked last time.  A node x is on that path iff the
  // path's entry at x's depth is x (dpd[x] = depth of x, written when a walk first passes it; the depth of a node never changes;
  // chain holds the last path's nodes and tokens by depth); everything below is the cached prefix of the transcript.  Same for the
  // timestep list through the time arena.  Exact: the cache is a cache of walks, not of results.
  if (nret == 1 && out.num_results == 1 && S.dpd != nullptr && S.chain != nullptr) {   // (uniform)
    __shared__ uint32_t s_d[2], s_k[2];
    GLB_AS uint32_t* const chain = (GLB_AS uint32_t*)S.chain;
    const uint32_t cc = S.chain_cap;
    GLB_AS uint32_t* const tokc = chain + 2; GLB_AS uint32_t* const posn = tokc + cc; GLB_AS uint32_t* const tsc = posn + cc; GLB_AS uint32_t* const postn = tsc + cc;
    GLB_AS uint32_t* const tmpv = postn + cc; GLB_AS uint32_t* const tmpn = tmpv + 2 * (size_t)cc;   // [tokens | timesteps][cc]: values / nodes of this walk
    GLB_AS uint32_t* const dpd = (GLB_AS uint32_t*)S.dpd; GLB_AS uint32_t* const dtd = (GLB_AS uint32_t*)S.dtd;
    const uint32_t best = ssrc[0];
    if (tid < 2) {
      const bool tm = tid == 1;   // thread 0: tokens through the path arena, thread 1: timesteps through the time arena
      const GLB_AS uint64_t* arena = tm ? GS.ta() : GS.pa();
      GLB_AS uint32_t* dp = tm ? dtd : dpd;
      const GLB_AS uint32_t* pos = tm ? postn : posn;
      const uint32_t clen = chain[tm ? 1 : 0];
      uint32_t x = tm ? S.ts[best] : S.node[best], k = 0, d = 0;
      for (;;) {
        if (x == STT_ROOT_CH || (tm && x == 0)) break;
        const uint2 pn = load_node(arena, x);
        const uint32_t dd = dp[x];
        if (!tm && pn.y == STT_ROOT_CH) break;
        if (dd != 0 && dd <= clen && pos[dd - 1] == x) { d = dd; break; }
        if (k < cc) { tmpv[(tm ? cc : 0u) + k] = pn.y; tmpn[(tm ? cc : 0u) + k] = x; }
        ++k; x = pn.x;
      }
      s_d[tid] = d; s_k[tid] = k;
    }
    __syncthreads();
    const uint32_t dT = s_d[0], kT = s_k[0], dS = s_d[1], kS = s_k[1];
    const uint32_t len = dT + kT, tlen = dS + kS;
    if (len <= cc && tlen <= cc && len <= (uint32_t)out.max_len && tlen <= (uint32_t)out.max_len) {   // (uniform; else: the plain walk below)
      for (uint32_t j = tid; j < kT; j += NTHREADS) { const uint32_t depth = dT + kT - j, nd = tmpn[j]; tokc[depth - 1] = tmpv[j]; posn[depth - 1] = nd; dpd[nd] = depth; }
      for (uint32_t j = tid; j < kS; j += NTHREADS) { const uint32_t depth = dS + kS - j, nd = tmpn[cc + j]; tsc[depth - 1] = tmpv[cc + j]; postn[depth - 1] = nd; dtd[nd] = depth; }
      if (tid == 0) { chain[0] = len; chain[1] = tlen; }
      __syncthreads();
      // results in the ring order of the plain walk: entry k from the END of the list in slot k (the host turns it round); a prefix
      // with fewer timesteps than tokens (see below) reports zeros for the missing ones
      const size_t ob = (size_t)blockIdx.x * out.num_results;
      for (uint32_t k = tid; k < len; k += NTHREADS) {
        out.tokens[ob * out.max_len + k] = tokc[len - 1 - k];
        out.timesteps[ob * out.max_len + k] = k < tlen ? tsc[tlen - 1 - k] : 0u;
      }
      if (tid == 0) { out.lens[ob] = (int)len; out.confidence[ob] = (double)sscore[best]; }
      return;
    }
  }
  // Back-tracking is a pointer chase through the arenas (one dependent HBM read per token), so each result gets two threads
  // -- tokens and timesteps -- and each chain is walked ONCE: entry k from the end goes to ring slot k % max_len and the
  // host puts the list the right way round (token j of len sits in slot (len-1-j) % max_len).
  for (int q = tid; q < 2 * nret; q += NTHREADS) {
    const int r = q >> 1;
    const uint32_t i = ssrc[r];
    const size_t ob = ((size_t)blockIdx.x * out.num_results + r);
    int k = 0;
    if ((q & 1) == 0) {
      for (uint32_t x = S.node[i]; x != STT_ROOT_CH;) {
        const uint2 pn = load_node(GS.pa(), x);
        if (pn.y == STT_ROOT_CH) break;
        out.tokens[ob * out.max_len + (k % out.max_len)] = pn.y;
        ++k; x = pn.x;
      }
      out.lens[ob] = k;
      out.confidence[ob] = (double)sscore[i];
    } else {
      for (uint32_t x = S.ts[i]; x != STT_ROOT_CH && x != 0;) {
        const uint2 tn = load_node(GS.ta(), x);
        out.timesteps[ob * out.max_len + (k % out.max_len)] = tn.y;
        ++k; x = tn.x;
      }
    }
    // A prefix that never received a finite probability (score == -NUM_FLT_INF; only reachable in an N-best list wider than
    // the set of real hypotheses) has tokens but no timestep list -- the reference dereferences a null TimestepTreeNode
    // there (get_history, path_trie.h:115-136).  Report zeros: the timestep thread clears the slots its token partner
    // (the neighbouring lane) filled beyond its own chain.
    const int k_tok = __shfl_xor(k, 1);
    if (q & 1)
      for (int z = k; z < k_tok && z < k + out.max_len; ++z) out.timesteps[ob * out.max_len + (z % out.max_len)] = 0u;
  }
}

// root prefix of every stream (DecoderState::init, ctc_beam_search_decoder.cpp:43-56)
__global__ void ctc_init_kernel(DecStream* streams, int n_streams, int fst_start, uint32_t bos_index, float bos_backoff) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_streams) return;
  DecStream& S = streams[i];
  S.score[0] = 0.0f; S.pb[0] = 0.0f; S.pnb[0] = STT_NEG_INF; S.ch[0] = STT_ROOT_CH; S.node[0] = 0; S.ts[0] = 0;
  S.fst[0] = fst_start; S.key[0] = 0x5151515151515151ULL; S.bnd[0] = 0;
  S.pa[0] = make_uint2(STT_ROOT_CH, STT_ROOT_CH); S.ta[0] = make_uint2(STT_ROOT_CH, 0);
  S.pq[0] = 0;  // scoring "root, then a word boundary" = entry 0 (raw 0: get_log_cond_prob of an empty n-gram)
  BEntry e{};  // boundary entry 0: the empty prefix, KenLM BeginSentence state (lm/model.cc:115-124)
  e.raw = 0.0; e.prev = STT_NONE; e.oov_hist = 0; e.pad = 0; e.hot_self = 0.0f;
  e.st.words[0] = bos_index; e.st.backoff[0] = bos_backoff; e.st.length = 1;
  S.be[0] = e; S.be_n = 1;
  S.n = 1; S.abs_t = 0; S.start_expanding = 0; S.error = 0; S.pa_n = 1; S.ta_n = 1;
  S.stat[0] = S.stat[1] = S.stat[2] = S.stat[3] = 0;
  for (int k = 0; k < 8; ++k) S.phase[k] = 0;
}
void launch_ctc_init(DecStream* streams, int n_streams, const DevScorer* sc, hipStream_t st) {
  hipLaunchKernelGGL(ctc_init_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, st, streams, n_streams, sc ? sc->fst_start : 0,
                     sc ? sc->bos_index : 0u, sc ? sc->bos_backoff : 0.0f);
}

__global__ void gather_streams_kernel(const DecStream* const* src, DecStream* dst, int n) {
  constexpr int W = sizeof(DecStream) / 4;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * W) return;
  const int i = idx / W, w = idx - i * W;
  reinterpret_cast<uint32_t*>(dst + i)[w] = reinterpret_cast<const uint32_t*>(src[i])[w];
}
__global__ void scatter_streams_kernel(DecStream* const* dst, const DecStream* src, int n) {
  constexpr int W = sizeof(DecStream) / 4;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * W) return;
  const int i = idx / W, w = idx - i * W;
  reinterpret_cast<uint32_t*>(dst[i])[w] = reinterpret_cast<const uint32_t*>(src + i)[w];
}
void launch_gather_streams(const DecStream* const* src, DecStream* dst, int n, hipStream_t st) {
  const int total = n * (int)(sizeof(DecStream) / 4);
  hipLaunchKernelGGL(gather_streams_kernel, dim3((total + 255) / 256), dim3(256), 0, st, src, dst, n);
}
void launch_scatter_streams(DecStream* const* dst, const DecStream* src, int n, hipStream_t st) {
  const int total = n * (int)(sizeof(DecStream) / 4);
  hipLaunchKernelGGL(scatter_streams_kernel, dim3((total + 255) / 256), dim3(256), 0, st, dst, src, n);
}

// ------------------------------------------------------------------------------------ launchers
// Word-mode step selection (tunable search_step): 2 = the step with label bitmaps + two LM waves + indexed FullScore (MODE 4) wherever
// it applies, anything else = the generic step everywhere.  Both give identical beams (tests/test_gpu_decoder.py, test_gpu_fuzz.py).
static bool ctc_masked_ok(const DecParams& p, const DevScorer& s, const DevAlphabet& al) {
  return tune().search_step == 2 && s.enabled && !s.utf8 && p.C <= 32 && p.C >= 2 && p.blank == p.C - 1 && !ctc_sorts_classes(p) && cap_bucket(p.beam) <= 512 &&
         s.fst_rec != nullptr && s.lmi != nullptr && s.order <= 5 && s.uni_in_vtab && al.space_id >= 0 && al.space_id < p.C - 1 && al.n_labels == p.C - 1;
}
static void check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { fprintf(stderr, "stt_amd: %s: %s\n", what, hipGetErrorString(e)); throw std::runtime_error(std::string("kernel launch failed: ") + what); }
}
#ifdef STT_TEST_HOOKS
// Test hook (libstt_test.so only; tunable debug_scribble, bit 0): dirty what a kernel does not own between launches -- the LDS of every compute unit, the
// launch queue's scratch memory (2 KB per lane, more than any instantiation of the search kernel spills), a good part of the vector
// registers.  512 workgroups of 1024 lanes: two passes over the chip's 256 compute units.
template <bool SCRATCH, bool LDSREG>
__global__ __launch_bounds__(1024) void debug_scribble_kernel(unsigned* sink, unsigned pattern, unsigned lds_words, unsigned zero, unsigned w_lo, unsigned w_hi) {
  extern __shared__ unsigned scribble_lds[];
  unsigned acc = 0;
  const unsigned keep = zero ? 0u : 0xFFFFFFFFu;   // (bit 7 of debug_scribble: everything is overwritten with ZEROS -- what fresh memory from the runtime holds)
  if constexpr (LDSREG) {
    for (unsigned i = threadIdx.x; i < lds_words; i += 1024) scribble_lds[i] = (pattern ^ (i * 2654435761u)) & keep;
    unsigned r[64];
#pragma unroll
    for (unsigned k = 0; k < 64; ++k) r[k] = (pattern ^ (k * 0x9E3779B9u) ^ threadIdx.x) & keep;
#pragma unroll
    for (unsigned k = 0; k < 64; ++k) asm volatile("" : "+v"(r[k]));   // (all 64 live in registers at once)
#pragma unroll
    for (unsigned k = 0; k < 64; ++k) acc += r[k];
    __syncthreads();
    acc += scribble_lds[(threadIdx.x * 31u) % lds_words];
  }
  if constexpr (SCRATCH) {
    volatile unsigned priv[512];
    for (unsigned k = (w_lo < 512u ? w_lo : 512u); k < (w_hi < 512u ? w_hi : 512u); ++k) priv[k] = (pattern + k * 0x01010101u) & keep;   // (tunables debug_scribble_lo / _hi: only these words of every lane's scratch)
    for (unsigned k = 0; k < 512; k += 17) acc += priv[(k * 7 + threadIdx.x) & 511];
  }
  if (acc == 0x13572468u && sink) sink[0] = acc;   // (never true in practice: keeps the work alive)
}
// debug_scribble: bit 0 on; bit 2 no scratch; bit 3 no LDS / registers; bit 4 one workgroup instead of 512 (experiments: benchmarks/r06_scribble_fuzz.sh)
void launch_debug_scribble(hipStream_t st, int mode_in) {
  const int mode = mode_in ? mode_in : tune().debug_scribble;
  const bool scratch = !(mode & 4), ldsreg = !(mode & 8);
  const int lds = ldsreg ? 160 * 1024 : 0, grid = (mode & 16) ? 1 : 512;
  static std::once_flag once;
  std::call_once(once, [&]() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(debug_scribble_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(debug_scribble_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  static unsigned seq = 0;
  const unsigned pat = 0xDEAD0000u + (++seq);
  if (scratch && ldsreg) hipLaunchKernelGGL((debug_scribble_kernel<true, true>), dim3(grid), dim3(1024), lds, st, (unsigned*)nullptr, pat, (unsigned)(lds / 4), (unsigned)((mode >> 7) & 1), (unsigned)tune().debug_scribble_lo, (unsigned)tune().debug_scribble_hi);
  else if (scratch) hipLaunchKernelGGL((debug_scribble_kernel<true, false>), dim3(grid), dim3(1024), 0, st, (unsigned*)nullptr, pat, 1u, (unsigned)((mode >> 7) & 1), (unsigned)tune().debug_scribble_lo, (unsigned)tune().debug_scribble_hi);
  else if (ldsreg) hipLaunchKernelGGL((debug_scribble_kernel<false, true>), dim3(grid), dim3(1024), lds, st, (unsigned*)nullptr, pat, (unsigned)(lds / 4), (unsigned)((mode >> 7) & 1), (unsigned)tune().debug_scribble_lo, (unsigned)tune().debug_scribble_hi);
  else hipLaunchKernelGGL((debug_scribble_kernel<false, false>), dim3(grid), dim3(1024), 0, st, (unsigned*)nullptr, pat, 1u, (unsigned)((mode >> 7) & 1), (unsigned)tune().debug_scribble_lo, (unsigned)tune().debug_scribble_hi);
}
#else
void launch_debug_scribble(hipStream_t, int) {}   // (the shipped library carries no scribbler)
#endif

void launch_ctc_next(const DecParams& p_in, const DevScorer& s, const DevAlphabet& al, DecStream* streams, int n_streams,
                     const float* probs, const int* frame_begin, const int* frame_count, hipStream_t st, int max_frames, void* wide_ws) {
  DecParams p = p_in;
  const bool wide = ctc_is_wide(p.beam, p.C, s.enabled && s.utf8);
  p.wide_rows = nullptr; p.wide_stride = 0; p.wide_max_frames = 0; p.n_lm_waves = 0; p.item_cap = 0;
  if (wide && (!wide_ws || max_frames <= 0 || p.C > STT_MAX_CLASSES)) { fprintf(stderr, "stt_amd: launch_ctc_next: wide alphabet without a row workspace\n"); abort(); }
  const int cb = cap_bucket(p.beam);
  if (wide || (ctc_sorts_classes(p) && wide_ws && max_frames > 0 && p.C <= WIDE_SORT_N && !(tune().debug_scribble & 256))) {   // (debug_scribble bit 8: narrow alphabets sort their classes in the search kernel, no row records)
    p.wide_rows = (const unsigned char*)wide_ws; p.wide_stride = ctc_wide_row_bytes(p.C); p.wide_max_frames = max_frames;
    hipLaunchKernelGGL(ctc_wide_rows_kernel, dim3(n_streams * max_frames), dim3(1024), 0, st, p, probs, frame_begin, frame_count);
    check_launch("ctc_wide_rows_kernel");
  }
  if (tune().debug_scribble & 1) { launch_debug_scribble(st, 0); check_launch("debug_scribble_kernel"); }
  p.n_lm_waves = tune().lm_waves; p.item_cap = tune().item_table_cap;
  p.wait_spins = tune().wait_spins > 0 ? tune().wait_spins : (1 << 22);
  p.key_mask = (tune().debug_key_bits >= 4 && tune().debug_key_bits < 63) ? ((1ULL << tune().debug_key_bits) - 1ULL) : ~0ULL;
  p.lm_prio = tune().lm_prio;
  const int mode = !s.enabled ? 0 : (s.utf8 ? 2 : ((!wide && ctc_masked_ok(p, s, al)) ? 4 : 1));
  const size_t lds = ctc_next_lds_bytes(p.beam, wide ? 0 : (mode == 4 ? 32 : p.C), s.enabled && s.utf8);   // (mode 4: the kernel carves its layout for 32 classes)
  p.lds_kb = lds_budget_kb_host();
  NextArgs na;
  na.p = p; na.s = s; na.al = al; na.streams = streams; na.probs = probs; na.frame_begin = frame_begin; na.frame_count = frame_count;
  const int ci = cb == 64 ? 0 : cb == 128 ? 1 : cb == 256 ? 2 : cb == 512 ? 3 : 4;
  // hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the function: remember what was set per device
  static std::mutex mu;
  static size_t configured[16][2][5][5] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int di = dev & 15;
#define STT_CTC_CASE(M, CI, CAPV, W)                                                                                         \
  if (mode == M && ci == CI && wide == W) {                                                                                  \
    {                                                                                                                        \
      std::lock_guard<std::mutex> lk(mu);                                                                                    \
      if (lds > configured[di][W][M][CI]) {                                                                                  \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_next_kernel<M, CAPV, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
          throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the search kernel");          \
        configured[di][W][M][CI] = lds;                                                                                      \
      }                                                                                                                      \
    }                                                                                                                        \
    hipLaunchKernelGGL((ctc_next_kernel<M, CAPV, W>), dim3(n_streams), dim3(NTHREADS), lds, st, na);                          \
    check_launch("ctc_next_kernel");                                                                                         \
    return;                                                                                                                  \
  }
#define STT_CTC_MODE(M, W) STT_CTC_CASE(M, 0, 64, W) STT_CTC_CASE(M, 1, 128, W) STT_CTC_CASE(M, 2, 256, W) STT_CTC_CASE(M, 3, 512, W) STT_CTC_CASE(M, 4, 1024, W)
#ifdef STT_CTC_PROBE   // code-generation experiments (benchmarks/kernel_resources.sh): only the two instantiations the bench's headline and bytes workloads run
  STT_CTC_CASE(4, 3, 512, false) STT_CTC_CASE(2, 4, 1024, false)
#else
  STT_CTC_MODE(0, false) STT_CTC_MODE(1, false) STT_CTC_MODE(2, false)
  STT_CTC_MODE(0, true) STT_CTC_MODE(1, true) STT_CTC_MODE(2, true)
  STT_CTC_CASE(4, 0, 64, false) STT_CTC_CASE(4, 1, 128, false) STT_CTC_CASE(4, 2, 256, false) STT_CTC_CASE(4, 3, 512, false)
#endif
#undef STT_CTC_MODE
#undef STT_CTC_CASE
  throw std::runtime_error("launch_ctc_next: no kernel instance for this configuration");
}
void launch_ctc_decode(const DecParams& p, const DevScorer& s, const DevAlphabet& al, const DecStream* streams, int n_streams,
                       const DecodeOut& out, hipStream_t st) {
  hipLaunchKernelGGL(ctc_decode_kernel, dim3(n_streams), dim3(NTHREADS), 0, st, p, s, al, streams, out);
  check_launch("ctc_decode_kernel");
}

// ------------------------------------------------------------------------------------ sttmath.h test hook
__global__ void test_math_kernel(int op, const float* a, const float* b, float* out, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = op == 0 ? sttm::stt_expf(a[i]) : op == 1 ? sttm::stt_logf(a[i]) : sttm::stt_log_sum_exp(a[i], b[i]);
}
// LM test hook: FullScore over a word sequence (state carried from BeginSentence or the null context), through the trie walk
// (use_index 0) or through the hashed n-gram index (use_index 1), one lane either way -- against answers of the real KenLM.
__global__ void test_lm_kernel(DevScorer s, const uint64_t* hashes, int n, int bos, int use_index, float* probs, int* lens) {
  if (threadIdx.x >= 1) return;
  KState st[2] = {};
  int cur = 0;
  if (bos) { st[0].length = 1; st[0].words[0] = s.bos_index; st[0].backoff[0] = s.bos_backoff; }
  unsigned probes = 0;
  for (int i = 0; i < n; ++i) {
    float prob; int nl = 0;
    if (use_index == 2) {   // the bigram blocks: hashes[i] holds the unit's UTF-8 bytes, first byte lowest (a code point of the vocabulary; else prob = NaN)
      const uint32_t unit = (uint32_t)hashes[i];
      uint32_t cp = 0;
      u32x4 ct = {0u, 0u, 0u, 0u};
      if (s.cpt != nullptr && utf8_unit_code_point(unit, cp)) ct = ((const GLB_AS u32x4*)s.cpt)[cp];
      if (ct.w & 1u) prob = lm_full_score_blocks(s, st[cur], cp, unit, ct, st[cur ^ 1], probes, &nl);
      else { prob = __uint_as_float(0x7FC00000u); st[cur ^ 1] = st[cur]; }
    } else if (use_index) { DevVocabSlot vs; const uint32_t wi = vocab_slot(s, hashes[i], vs, probes); prob = lm_full_score_indexed(s, st[cur], hashes[i], wi, vs, st[cur ^ 1], probes, &nl); }
    else { const uint32_t wi = vocab_index(s, hashes[i], probes); prob = kenlm_full_score(s, st[cur], wi, st[cur ^ 1], probes, nullptr, &nl); }
    if (threadIdx.x == 0) { probs[i] = prob; lens[i] = nl; }
    cur ^= 1;
  }
}
void launch_test_lm(const DevScorer& s, const uint64_t* hashes, int n, int bos, int use_index, float* probs, int* lens, hipStream_t st) {
  hipLaunchKernelGGL(test_lm_kernel, dim3(1), dim3(64), 0, st, s, hashes, n, bos, use_index, probs, lens);
  check_launch("test_lm_kernel");
}
void launch_test_math(int op, const float* a, const float* b, float* out, unsigned n, hipStream_t st) {
  hipLaunchKernelGGL(test_math_kernel, dim3((n + 255) / 256), dim3(256), 0, st, op, a, b, out, n);
}
