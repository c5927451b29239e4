// benchmarks/gather_probe.hip -- how many scattered 32-byte records can ONE compute unit gather per cycle?
//
// The code-point search step (ctc_next_kernel<2,1024,false>) spends its language-model phase on scattered reads: a 32-byte memo entry or
// a 64-byte index bucket per lane, each the input of the next address.  This probe runs exactly that access shape with
// nothing else around it: every lane walks CHAINS independent chains of dependent reads (the record read decides the next index) through
// a table of 2^log2_entries 32-byte records; one workgroup per compute unit, 256 / 512 / 1024 lanes.  Reported per configuration:
// records per cycle and compute unit (shader clock, in-kernel), and the aggregate rate.  "block" = the 64 lanes of a wave read 64
// CONSECUTIVE records (one 2 KB slice per wave and round: the sibling memo blocks), everything else scattered.
//
//   hipcc --offload-arch=gfx950 -O3 benchmarks/gather_probe.hip -o benchmarks/gather_probe && benchmarks/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t a) { a *= 0x9E3779B97F4A7C15ULL; a ^= a >> 29; return a; }

template <int CHAINS, bool BLOCK>
__global__ __launch_bounds__(1024) void gather_kernel(const uint4* __restrict__ tab, uint64_t mask, int rounds, unsigned long long* cyc, uint32_t* sink) {
  const unsigned lane = threadIdx.x & 63u;
  uint64_t s[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s[c] = mix(((uint64_t)blockIdx.x << 32) ^ ((uint64_t)(BLOCK ? threadIdx.x >> 6 : threadIdx.x) << 8) ^ (uint64_t)c ^ 0x1234567ULL);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t acc = 0;
  for (int r = 0; r < rounds; ++r) {
    uint4 a[CHAINS], b[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      uint64_t idx = s[c] & mask;
      if (BLOCK) idx = (idx & ~63ULL) | lane;      // the wave's 64 lanes: 64 consecutive records, chosen by a wave-uniform state
      a[c] = tab[idx * 2]; b[c] = tab[idx * 2 + 1];
    }
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      uint64_t v = ((uint64_t)a[c].x << 32) | b[c].y;
      if (BLOCK) v = __shfl(v, 0);                 // (the next block is chosen by what lane 0 read: still a dependent chain)
      s[c] = mix(s[c] ^ v);
      acc += a[c].w ^ b[c].z;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc;
}

template <int CHAINS, bool BLOCK>
static void run(const uint4* tab, int log2_entries, int wgs, int threads, int rounds, unsigned long long* d_cyc, uint32_t* d_sink, const char* what) {
  const uint64_t mask = (1ULL << log2_entries) - 1;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_kernel<CHAINS, BLOCK>), dim3(wgs), dim3(threads), 0, 0, tab, mask, 4, d_cyc, d_sink);   // warm-up (code, TLB)
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((gather_kernel<CHAINS, BLOCK>), dim3(wgs), dim3(threads), 0, 0, tab, mask, rounds, d_cyc, d_sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(wgs);
  CHECK(hipMemcpy(cyc.data(), d_cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto c : cyc) avg += (double)c; avg /= wgs;
  const double recs = (double)rounds * CHAINS * threads;
  printf("{\"pattern\": \"%s\", \"table_MB\": %.0f, \"workgroups\": %d, \"lanes\": %d, \"chains_per_lane\": %d, \"records_per_cycle_per_cu\": %.4f, \"cycles_per_round\": %.0f, "
         "\"aggregate_Grec_s\": %.2f, \"aggregate_GB_s_at_32B\": %.0f}\n",
         what, (double)(1ULL << log2_entries) * 32 / 1e6, wgs, threads, CHAINS, recs / avg, avg / rounds, recs * wgs / (ms * 1e-3) / 1e9, recs * wgs * 32 / (ms * 1e-3) / 1e9);
  fflush(stdout);
}

int main() {
  const int big = 24, small = 19;   // 512 MB (the memo's size) and 16 MB (L2 / Infinity Cache resident)
  uint4* tab; unsigned long long* d_cyc; uint32_t* d_sink;
  const size_t bytes = (size_t)(1ULL << big) * 32;
  CHECK(hipMalloc(&tab, bytes)); CHECK(hipMalloc(&d_cyc, 4096 * 8)); CHECK(hipMalloc(&d_sink, 64));
  {
    std::vector<uint32_t> h(bytes / 4);
    uint64_t x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
    CHECK(hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice));
  }
  const int R = 200;
  for (int lg : {big, small}) {
    for (int wgs : {64, 256}) {
      run<1, false>(tab, lg, wgs, 1024, R, d_cyc, d_sink, "scattered");
      run<2, false>(tab, lg, wgs, 1024, R, d_cyc, d_sink, "scattered");
      run<4, false>(tab, lg, wgs, 1024, R, d_cyc, d_sink, "scattered");
      run<1, false>(tab, lg, wgs, 256, R, d_cyc, d_sink, "scattered");
      run<4, false>(tab, lg, wgs, 256, R, d_cyc, d_sink, "scattered");
      run<1, true>(tab, lg, wgs, 1024, R, d_cyc, d_sink, "block");
      run<4, true>(tab, lg, wgs, 1024, R, d_cyc, d_sink, "block");
    }
  }
  return 0;
}
