// stt_amd/csrc/fleet.cpp -- several GPUs of one node behind the C ABI.
//
// The path shards by utterance: the reference scales the same way with one process per GPU and files dealt from a queue
// (training/coqui_stt_training/transcribe.py:40-56,136-148).  A binding user of libstt.so gets that without Python:
// STTX_FleetCreate() builds one replica of the model (and scorer) per HIP device; STTX_FleetSpeechToTextBatch() sorts the
// utterances by length, deals them longest-processing-time-first to the devices, runs every shard on its own host thread
// through the ordinary batch path (no communication during compute), and gathers the variable-length transcripts with
// RCCL over xGMI: one all-gather of per-rank byte counts, one of the padded byte records.  RCCL is loaded with dlopen at
// the first fleet creation (libstt.so itself links only libamdhip64); a missing librccl is an error, not a fallback.
// The Python twin of the sharding and of the gather is stt_amd/dist.py (tests/test_dist_cpu.py runs both on the same cases).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>

#include "../../include/stt_amd.h"
#ifdef STT_TEST_HOOKS
#include "../../include/stt_amd_test.h"
#endif
#include "engine.h"

namespace {
// the handful of RCCL entry points used, by their NCCL names (rccl.h)
typedef struct ncclComm* ncclComm_t;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclInt32 = 2 };  // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2
struct Rccl {
  void* h = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (h) return true;
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { std::cerr << "stt_amd: cannot load librccl (" << dlerror() << ")" << std::endl; return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
    return CommInitAll && CommDestroy && AllGather;
  }
};
Rccl g_rccl;
}  // namespace

// Longest-processing-time-first assignment (stt_amd/dist.py: shard_utterances): utterances by descending length (stable),
// each to the currently least loaded shard (lowest index on ties).
extern "C" int STTX_ShardUtterances(const unsigned int* aSizes, unsigned int aCount, unsigned int aShards, unsigned int* aShardOf) {
  if (!aShards || (!aSizes && aCount) || (!aShardOf && aCount)) return STT_ERR_INVALID_SHAPE;
  std::vector<unsigned> order(aCount);
  for (unsigned i = 0; i < aCount; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return aSizes[x] > aSizes[y]; });
  std::vector<unsigned long long> load(aShards, 0);
  for (unsigned i : order) {
    unsigned best = 0;
    for (unsigned r = 1; r < aShards; ++r) if (load[r] < load[best]) best = r;
    aShardOf[i] = best;
    load[best] += aSizes[i];
  }
  return STT_ERR_OK;
}

struct STTX_Fleet {
  std::vector<int> devices;
  std::vector<ModelState*> models;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<DevBuf*> d_cnt, d_all_cnt, d_rec, d_all_rec;
  size_t rec_cap = 1 << 20;  // bytes per rank and round of the record exchange (buffers allocated at creation)
  int fail_shard = -1;       // STTX_DebugFleetFailShard
  std::mutex mu;
  ~STTX_Fleet() {
    for (size_t i = 0; i < models.size(); ++i) {
      (void)hipSetDevice(devices[i]);
      if (i < comms.size() && comms[i] && g_rccl.CommDestroy) g_rccl.CommDestroy(comms[i]);
      if (i < streams.size() && streams[i]) (void)hipStreamDestroy(streams[i]);
      for (auto* v : {&d_cnt, &d_all_cnt, &d_rec, &d_all_rec}) if (i < v->size()) delete (*v)[i];
      if (models[i]) STT_FreeModel(models[i]);
    }
  }
};

extern "C" {

int STTX_FleetCreate(const char* aModelPath, const int* aDevices, unsigned int aNumDevices, STTX_Fleet** retval) {
  *retval = nullptr;
  if (!aNumDevices || !aDevices) return STT_ERR_INVALID_SHAPE;
  try {
    {
      static std::mutex load_mu;  // (two threads creating their first fleets at once)
      std::lock_guard<std::mutex> lk(load_mu);
      if (!g_rccl.load()) return STT_ERR_FAIL_INIT_SESS;
    }
    std::unique_ptr<STTX_Fleet> f(new STTX_Fleet());
    f->devices.assign(aDevices, aDevices + aNumDevices);
    for (unsigned i = 0; i < aNumDevices; ++i) {
      if (STTX_SetDevice(aDevices[i]) != STT_ERR_OK) return STT_ERR_FAIL_INIT_SESS;
      ModelState* m = nullptr;
      const int rc = STT_CreateModel(aModelPath, &m);
      if (rc != STT_ERR_OK) return rc;
      f->models.push_back(m);
    }
    f->comms.assign(aNumDevices, nullptr);
    const int rc = g_rccl.CommInitAll(f->comms.data(), (int)aNumDevices, f->devices.data());
    if (rc != ncclSuccess) {
      std::cerr << "stt_amd: ncclCommInitAll failed: " << (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?") << std::endl;
      return STT_ERR_FAIL_INIT_SESS;
    }
    for (unsigned i = 0; i < aNumDevices; ++i) {
      HIP_CHECK(hipSetDevice(aDevices[i]));
      hipStream_t st;
      HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      f->streams.push_back(st);
      f->d_cnt.push_back(new DevBuf()); f->d_all_cnt.push_back(new DevBuf()); f->d_rec.push_back(new DevBuf()); f->d_all_rec.push_back(new DevBuf());
      // everything the exchange touches exists before the first batch: a rank can then only leave a collective through a HIP / RCCL
      // error, never through an allocation
      f->d_cnt.back()->reserve(4); f->d_all_cnt.back()->reserve((size_t)4 * aNumDevices);
      f->d_rec.back()->reserve(f->rec_cap); f->d_all_rec.back()->reserve(f->rec_cap * aNumDevices);
    }
    *retval = f.release();
    return STT_ERR_OK;
  } catch (const std::exception& e) {
    std::cerr << "stt_amd: " << e.what() << std::endl;
    return STT_ERR_FAIL_CREATE_MODEL;
  }
}

unsigned int STTX_FleetSize(const STTX_Fleet* f) { return f ? (unsigned)f->models.size() : 0; }

int STTX_FleetEnableExternalScorer(STTX_Fleet* f, const char* aScorerPath) {
  for (ModelState* m : f->models) { const int rc = STT_EnableExternalScorer(m, aScorerPath); if (rc != STT_ERR_OK) return rc; }
  return STT_ERR_OK;
}
int STTX_FleetSetBeamWidth(STTX_Fleet* f, unsigned int aBeamWidth) {
  for (ModelState* m : f->models) { const int rc = STT_SetModelBeamWidth(m, aBeamWidth); if (rc != STT_ERR_OK) return rc; }
  return STT_ERR_OK;
}

// ---- the records that travel (host side, no GPU): rank r packs [u32 utterance index, u32 byte length, bytes] per transcript
void fleet_pack_record(std::vector<unsigned char>& rec, unsigned id, const char* text) {
  const unsigned len = (unsigned)strlen(text);
  const size_t o = rec.size();
  rec.resize(o + 8 + len);
  memcpy(&rec[o], &id, 4); memcpy(&rec[o + 4], &len, 4); memcpy(&rec[o + 8], text, len);
}
// `gathered` = what an all-gather of the records padded to `cap` bytes leaves on every rank; counts[r] = bytes rank r really sent.
// Fills out[id] (malloc'd) for every record; returns false on a malformed record (index out of range, length past the count).
bool fleet_unpack_records(const unsigned char* gathered, size_t cap, const int* counts, unsigned G, unsigned n_out, char** out) {
  for (unsigned r = 0; r < G; ++r) {
    if (counts[r] < 0 || (size_t)counts[r] > cap) return false;
    const unsigned char* p = gathered + (size_t)r * cap;
    size_t o = 0;
    while (o < (size_t)counts[r]) {
      unsigned id, len;
      if (o + 8 > (size_t)counts[r]) return false;
      memcpy(&id, p + o, 4); memcpy(&len, p + o + 4, 4);
      if (id >= n_out || o + 8 + (size_t)len > (size_t)counts[r] || out[id]) return false;
      out[id] = (char*)malloc((size_t)len + 1);
      memcpy(out[id], p + o + 8, len); out[id][len] = 0;
      o += 8 + len;
    }
  }
  return true;
}
static size_t fleet_cap(const int* counts, unsigned G) {
  int cap = 16;
  for (unsigned q = 0; q < G; ++q) cap = std::max(cap, counts[q]);
  return (size_t)((cap + 15) & ~15);
}

// Transcripts in the caller's order, or NULL if any shard failed (STTX_FreeStrings releases them).
// Every rank ALWAYS enters the first all-gather (a rank whose decode failed announces -1 there), and because every rank then holds
// the same counts, all of them take the same decision about the second one: a failure on one device can never leave the others
// waiting inside a collective.  Nothing is allocated between the two collectives (the record buffers are sized when the fleet is
// created; larger records travel in rounds of that size), so a rank cannot drop out there either.
char** STTX_FleetSpeechToTextBatch(STTX_Fleet* f, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aBatch) {
  std::lock_guard<std::mutex> call_lock(f->mu);  // one batch at a time per fleet (the communicators and record buffers are per fleet)
  const unsigned G = (unsigned)f->models.size();
  std::vector<unsigned> shard_of(aBatch);
  if (STTX_ShardUtterances(aBufferSizes, aBatch, G, shard_of.data()) != STT_ERR_OK) return nullptr;
  std::vector<std::vector<unsigned>> idx(G);
  for (unsigned i = 0; i < aBatch; ++i) idx[shard_of[i]].push_back(i);
  std::vector<std::vector<unsigned char>> rec(G);
  std::vector<int> ok(G, 1);
  std::vector<unsigned char> gathered;                  // (every rank receives everything; rank 0's copy is unpacked)
  std::vector<std::vector<int>> counts(G, std::vector<int>(G, 0));
  const size_t R = f->rec_cap;                          // bytes per rank and round of the record exchange
  auto shard_body = [&](unsigned r) {
    // ---- decode this shard (may fail: the rank still takes part in the exchange below)
    try {
      HIP_CHECK(hipSetDevice(f->devices[r]));
      const unsigned n = (unsigned)idx[r].size();
      std::vector<const short*> bufs(n); std::vector<unsigned> sizes(n);
      for (unsigned k = 0; k < n; ++k) { bufs[k] = aBuffers[idx[r][k]]; sizes[k] = aBufferSizes[idx[r][k]]; }
      if (f->fail_shard == (int)r) throw std::runtime_error("injected failure (STTX_DebugFleetFailShard)");
      char** texts = n ? STTX_SpeechToTextBatch(f->models[r], bufs.data(), sizes.data(), n) : nullptr;
      if (n && !texts) ok[r] = 0;
      for (unsigned k = 0; k < n && texts; ++k) fleet_pack_record(rec[r], idx[r][k], texts[k]);
      if (texts) STTX_FreeStrings(texts, n);
    } catch (const std::exception& e) {
      std::cerr << "stt_amd: fleet shard " << r << ": " << e.what() << std::endl;
      ok[r] = 0;
    }
    // ---- exchange 1: byte counts (a failed shard announces -1)
    try {
      (void)hipSetDevice(f->devices[r]);
      hipStream_t st = f->streams[r];
      const int mine = ok[r] ? (int)rec[r].size() : -1;
      (void)hipMemcpyAsync(f->d_cnt[r]->p, &mine, 4, hipMemcpyHostToDevice, st);   // (pre-allocated buffers: nothing here allocates)
      const int a1 = g_rccl.AllGather(f->d_cnt[r]->p, f->d_all_cnt[r]->p, 1, ncclInt32, f->comms[r], st);
      HIP_CHECK(hipMemcpyAsync(counts[r].data(), f->d_all_cnt[r]->p, 4 * G, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (a1 != ncclSuccess) throw std::runtime_error("ncclAllGather(counts) failed");
      bool all_ok = true;
      for (unsigned q = 0; q < G; ++q) all_ok = all_ok && counts[r][q] >= 0;
      if (!all_ok) { ok[r] = 0; return; }               // every rank sees the same counts: none of them enters exchange 2
      // ---- exchange 2: the records, padded to the largest, in rounds of the pre-allocated size
      const size_t cap = fleet_cap(counts[r].data(), G);
      if (r == 0) gathered.assign(cap * G, 0);
      for (size_t off = 0; off < cap; off += R) {
        const size_t len = std::min(R, cap - off);
        const size_t have = rec[r].size() > off ? std::min(len, rec[r].size() - off) : 0;
        HIP_CHECK(hipMemsetAsync(f->d_rec[r]->p, 0, len, st));
        if (have) HIP_CHECK(hipMemcpyAsync(f->d_rec[r]->p, rec[r].data() + off, have, hipMemcpyHostToDevice, st));
        if (g_rccl.AllGather(f->d_rec[r]->p, f->d_all_rec[r]->p, len, ncclUint8, f->comms[r], st) != ncclSuccess) throw std::runtime_error("ncclAllGather(records) failed");
        if (r == 0)
          for (unsigned q = 0; q < G; ++q)
            HIP_CHECK(hipMemcpyAsync(gathered.data() + (size_t)q * cap + off, (const char*)f->d_all_rec[0]->p + (size_t)q * len, len, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
      }
    } catch (const std::exception& e) {
      std::cerr << "stt_amd: fleet shard " << r << " (exchange): " << e.what() << std::endl;
      ok[r] = 0;
    }
  };
  std::vector<std::thread> th;
  for (unsigned r = 1; r < G; ++r) th.emplace_back(shard_body, r);
  shard_body(0);
  for (auto& t : th) t.join();
  for (unsigned r = 0; r < G; ++r) if (!ok[r] || counts[0][r] < 0) return nullptr;
  char** out = (char**)calloc(std::max(1u, aBatch), sizeof(char*));
  if (!fleet_unpack_records(gathered.data(), fleet_cap(counts[0].data(), G), counts[0].data(), G, aBatch, out)) {
    for (unsigned i = 0; i < aBatch; ++i) free(out[i]);
    free(out);
    return nullptr;
  }
  for (unsigned i = 0; i < aBatch; ++i) if (!out[i]) out[i] = strdup("");
  return out;
}

#ifdef STT_TEST_HOOKS
// Test hooks.  (1) Pack / pad / concatenate / unpack exactly as the two all-gathers do, on the host: aTexts[i] is the transcript of
// utterance i, decoded by shard aShardOf[i] of aShards; returns the strings in the caller's order (STTX_FreeStrings) or NULL.
char** STTX_TestFleetRecords(const char* const* aTexts, const unsigned int* aShardOf, unsigned int aCount, unsigned int aShards) {
  if (!aShards) return nullptr;
  std::vector<std::vector<unsigned char>> rec(aShards);
  for (unsigned i = 0; i < aCount; ++i) { if (aShardOf[i] >= aShards) return nullptr; fleet_pack_record(rec[aShardOf[i]], i, aTexts[i]); }
  std::vector<int> counts(aShards);
  for (unsigned r = 0; r < aShards; ++r) counts[r] = (int)rec[r].size();
  const size_t cap = fleet_cap(counts.data(), aShards);
  std::vector<unsigned char> gathered(cap * aShards, 0);
  for (unsigned r = 0; r < aShards; ++r) if (!rec[r].empty()) memcpy(&gathered[(size_t)r * cap], rec[r].data(), rec[r].size());
  char** out = (char**)calloc(std::max(1u, aCount), sizeof(char*));
  if (!fleet_unpack_records(gathered.data(), cap, counts.data(), aShards, aCount, out)) { for (unsigned i = 0; i < aCount; ++i) free(out[i]); free(out); return nullptr; }
  for (unsigned i = 0; i < aCount; ++i) if (!out[i]) out[i] = strdup("");
  return out;
}
// (2) The next STTX_FleetSpeechToTextBatch call fails on shard aShard before decoding (-1: off): the call must return NULL, not hang.
int STTX_DebugFleetFailShard(STTX_Fleet* f, int aShard) { f->fail_shard = aShard; return STT_ERR_OK; }
#endif  // STT_TEST_HOOKS

void STTX_FleetFree(STTX_Fleet* f) { delete f; }

}  // extern "C"
