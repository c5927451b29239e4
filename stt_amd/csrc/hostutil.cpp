// stt_amd/csrc/hostutil.cpp -- device buffers, Alphabet (native_client/alphabet.{h,cc}).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "engine.h"
#include "tuning.h"

// ---- tunables (tuning.h) ------------------------------------------------------------------------
namespace {
struct TuneEntry { const char* name; int Tuning::*field; };
const TuneEntry kTune[] = {
#define X(name, def, doc) {#name, &Tuning::name},
    STT_TUNING_FIELDS(X)
#undef X
};
}  // namespace
// Bumped whenever a device or page-locked buffer moves or a tunable changes: everything that baked addresses or launch shapes into a
// captured graph (engine.cpp: the streaming hop) compares generations instead of tracking every pointer.
std::atomic<unsigned long long> g_layout_generation{1};
unsigned long long layout_generation() { return g_layout_generation.load(std::memory_order_relaxed); }

static std::atomic<int> g_models_alive{0};
void tuning_model_count(int delta) { g_models_alive += delta; }
int tuning_set(const char* name, int value) {
  if (!name) return -1;
  // lstm_upw decides how the recurrent matrix is PACKED when a model is loaded, and every recurrent launch reads it again: changed
  // under a live model the kernel shape would no longer match the packed matrix (wrong probabilities, silently).  Refused.
  if (!strcmp(name, "lstm_upw") && g_models_alive.load() > 0 && value != tune().lstm_upw) return -1;
  for (const TuneEntry& e : kTune)
    if (!strcmp(e.name, name)) { tune().*(e.field) = value; g_layout_generation.fetch_add(1, std::memory_order_relaxed); return 0; }
  return -1;
}
int tuning_get(const char* name, int* value) {
  if (!name) return -1;
  for (const TuneEntry& e : kTune)
    if (!strcmp(e.name, name)) { if (value) *value = tune().*(e.field); return 0; }
  return -1;
}
Tuning& tune() {
  static Tuning t;
  static const bool seeded = []() {  // STT_AMD_TUNING="name=value,name=value" (A/B scripts); unknown names are reported, not ignored
    const char* e = getenv("STT_AMD_TUNING");
    if (!e) return true;
    std::stringstream ss(e);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
      const size_t eq = tok.find('=');
      if (eq == std::string::npos) continue;
      const std::string name = tok.substr(0, eq);
      bool ok = false;
      for (const TuneEntry& en : kTune)
        if (name == en.name) { t.*(en.field) = atoi(tok.c_str() + eq + 1); ok = true; }
      if (!ok) fprintf(stderr, "stt_amd: STT_AMD_TUNING: no tunable named '%s'\n", name.c_str());
    }
    return true;
  }();
  (void)seeded;
  return t;
}

// The searches take a compute unit's whole register file each (1024 lanes x 128 registers) for most of a batch; the acoustic engines
// live on whatever is left.  Left to the dispatcher, who sits where is decided launch by launch.  With search_cus = N the search streams
// are confined to the CUs of mask bits [0, N) and (am_cus) the chosen acoustic engines to the rest: a fixed partition.  (A masked stream
// cannot also carry a priority: hipExtStreamCreateWithCUMask takes none.)
void create_engine_stream(hipStream_t* st, int role, bool high_priority) {
  const int n = tune().search_cus;
  const bool masked = n > 0 && (role == 3 || (role < 3 && (tune().am_cus >> role & 1)));
  if (masked) {
    int dev = 0;
    hipDeviceProp_t p;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipGetDeviceProperties(&p, dev));
    const int ncu = p.multiProcessorCount;
    std::vector<uint32_t> w((size_t)(ncu + 31) / 32, 0u);
    const int lo = role == 3 ? 0 : std::min(n, ncu), hi = role == 3 ? std::min(n, ncu) : ncu;
    for (int i = lo; i < hi; ++i) w[(size_t)i / 32] |= 1u << (i % 32);
    if (hi > lo) { HIP_CHECK(hipExtStreamCreateWithCUMask(st, (uint32_t)w.size(), w.data())); return; }
  }
  if (high_priority) {
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (hi = numerically lowest = most urgent)
    HIP_CHECK(hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi));
  } else HIP_CHECK(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
}

std::recursive_mutex& hip_capture_mutex() { static std::recursive_mutex* m = new std::recursive_mutex; return *m; }

void DevBuf::reserve(size_t bytes, bool keep, hipStream_t st) {
  if (bytes <= cap && p) return;
  std::lock_guard<std::recursive_mutex> no_capture(hip_capture_mutex());   // (hipDeviceSynchronize / hipFree below; a reserve INSIDE a capture is the capturing thread's own: recursive)
  g_layout_generation.fetch_add(1, std::memory_order_relaxed);
  size_t ncap = bytes + bytes / 4 + 256;
  void* np = nullptr;
  HIP_CHECK(hipMalloc(&np, ncap));
  if (tune().debug_poison) {   // (test hook: see tuning.h; the fill runs on the NULL stream, which the engine's non-blocking streams do not wait for: finish it here)
    HIP_CHECK(hipMemset(np, tune().debug_poison == 2 ? 0xA5 : 0xFF, ncap));
    HIP_CHECK(hipDeviceSynchronize());
  }
  if (p) {
    if (keep) {
      HIP_CHECK(hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    // Growing a live buffer: engines on other streams (the batch path runs three acoustic streams and up to four search streams) may
    // still be reading the old allocation.  hipFree waits for the device in practice; say so explicitly rather than rely on it.
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(p));
  }
  p = np;
  cap = ncap;
}
void PinnedBuf::reserve(size_t bytes) {
  if (bytes <= cap && p) return;
  std::lock_guard<std::recursive_mutex> no_capture(hip_capture_mutex());
  g_layout_generation.fetch_add(1, std::memory_order_relaxed);
  if (p) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipHostFree(p)); }  // (a copy kernel may still be moving the old block)
  p = nullptr;
  cap = bytes + bytes / 4 + 256;
  HIP_CHECK(hipHostMalloc(&p, cap, hipHostMallocDefault));
}
void* PinnedBuf::dev() const {
  void* d = nullptr;
  HIP_CHECK(hipHostGetDevicePointer(&d, p, 0));
  return d;
}
static bool copy_kernel_on() { return tune().copy_kernel != 0; }
void copy_h2d(void* dst_dev, const PinnedBuf& src, size_t bytes, hipStream_t st) {
  if (!bytes) return;
  if (copy_kernel_on()) launch_copy_bytes(dst_dev, src.dev(), bytes, st);
  else HIP_CHECK(hipMemcpyAsync(dst_dev, src.p, bytes, hipMemcpyHostToDevice, st));
}
void copy_d2h(PinnedBuf& dst, const void* src_dev, size_t bytes, hipStream_t st, size_t dst_offset) {
  if (!bytes) return;
  if (copy_kernel_on()) launch_copy_bytes((char*)dst.dev() + dst_offset, src_dev, bytes, st);
  else HIP_CHECK(hipMemcpyAsync((char*)dst.p + dst_offset, src_dev, bytes, hipMemcpyDeviceToHost, st));
}
void DevBuf::upload(const void* src, size_t bytes, hipStream_t st) {
  reserve(bytes ? bytes : 1);
  if (bytes) {
    HIP_CHECK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));  // src may be a temporary
  }
}

// ---- Alphabet --------------------------------------------------------------------------------
// Text format: one label per line; '#' starts a comment, "\#" is the literal '#'; a line holding a
// single space is the space label; \n, \r\n and \r all end a line (alphabet.cc:8-68).
int Alphabet::InitFromFile(const char* path) {
  std::ifstream in(path, std::ios::in | std::ios::binary);
  if (!in) return 1;
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string data = ss.str();
  labels_.clear();
  space_index_ = -2;
  size_t i = 0;
  const size_t n = data.size();
  while (i < n) {
    std::string line;
    while (i < n && data[i] != '\n' && data[i] != '\r') line += data[i++];
    if (i < n) {  // consume the line ending
      if (data[i] == '\r' && i + 1 < n && data[i + 1] == '\n') i += 2; else i += 1;
    }
    if (line.size() == 2 && line[0] == '\\' && line[1] == '#') line = "#";
    else if (!line.empty() && line[0] == '#') continue;
    if (line == " ") space_index_ = (int)labels_.size();
    if (line.empty()) continue;
    labels_.push_back(line);
  }
  return 0;
}
void Alphabet::InitUTF8() {
  labels_.clear();
  for (int idx = 0; idx < 255; ++idx) labels_.push_back(std::string(1, (char)(idx + 1)));
  space_index_ = ' ' - 1;
}
// Binary format: u16 count; count x { u16 key; u16 len; bytes[len] } (alphabet.cc:102-169)
std::string Alphabet::Serialize() const {
  std::string out;
  auto put16 = [&](uint16_t v) { out.append(reinterpret_cast<const char*>(&v), 2); };
  put16((uint16_t)labels_.size());
  for (size_t i = 0; i < labels_.size(); ++i) {
    put16((uint16_t)i);
    put16((uint16_t)labels_[i].size());
    out.append(labels_[i]);
  }
  return out;
}
int Alphabet::Deserialize(const char* buffer, int buffer_size) {
  int offset = 0;
  if (buffer_size - offset < 2) return 1;
  uint16_t size;
  memcpy(&size, buffer + offset, 2); offset += 2;
  labels_.assign(size, std::string());
  space_index_ = -2;
  for (int i = 0; i < size; ++i) {
    if (buffer_size - offset < 4) return 1;
    uint16_t label, len;
    memcpy(&label, buffer + offset, 2); offset += 2;
    memcpy(&len, buffer + offset, 2); offset += 2;
    if (buffer_size - offset < len) return 1;
    std::string val(buffer + offset, len);
    offset += len;
    if (label >= size) return 1;
    if (val == " ") space_index_ = label;
    labels_[label] = val;
  }
  return 0;
}
std::string Alphabet::Decode(const unsigned* idx, int n) const {
  std::string s;
  for (int i = 0; i < n; ++i) if (idx[i] < labels_.size()) s += labels_[idx[i]];
  return s;
}
