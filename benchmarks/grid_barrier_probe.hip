// benchmarks/grid_barrier_probe.hip -- what would a PERSISTENT recurrent step pay per timestep on this machine?
// A persistent LSTM kernel (weights resident in LDS / registers, one launch per chunk) replaces the kernel boundary between two steps
// by a grid-wide barrier plus a re-read of h from the other CUs.  This probe measures the barrier alone, in the cheapest correct form
// of /opt/skills/guides/MI355X_MICROARCH.md ("barrier-xcd": per-XCC counter, leader release fence -> top counter -> acquire fence ->
// per-XCC generation), for the grids such a kernel would use, and the h re-read (256 KB written by all workgroups, read by all).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp benchmarks/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar { unsigned xcc_count[8 * 16]; unsigned top; unsigned gen[8 * 16]; };  // (counters 64 B apart)

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned n_wg, unsigned& phase, unsigned per_xcc) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned xcc = blockIdx.x & 7;
    ++phase;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned arrived = __hip_atomic_fetch_add(&b->xcc_count[xcc * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == per_xcc * phase) {  // the XCC's last arriver goes up
      const unsigned top = __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (top == 8 * phase) {
        for (int x = 0; x < 8; ++x) __hip_atomic_store(&b->gen[x * 16], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned spins = 0;  // (bounded: a probe must not be able to hang the GPU)
    while (__hip_atomic_load(&b->gen[xcc * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase && ++spins < 4000000u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// steps x { [optional: every workgroup writes its 2 KB slice of h (sc1 stores), barrier, every workgroup reads ALL of h] }
__global__ __launch_bounds__(256) void persistent_probe(Bar* b, uint4* h, unsigned steps, unsigned h_bytes, int with_h, unsigned long long* sink) {
  unsigned phase = 0;
  const unsigned n_wg = gridDim.x, per_xcc = n_wg / 8;
  uint4 acc = {0, 0, 0, 0};
  const unsigned n16 = h_bytes / 16, mine = n16 / n_wg;
  for (unsigned t = 0; t < steps; ++t) {
    if (with_h) {
      for (unsigned i = threadIdx.x; i < mine; i += 256) {
        uint4 v = {t, i, blockIdx.x, acc.x};
        h[(size_t)(t & 1) * n16 + blockIdx.x * mine + i] = v;  // (made visible by the barrier's release fence)
      }
    }
    grid_barrier(b, n_wg, phase, per_xcc);
    if (with_h) {
      const uint4* src = h + (size_t)(t & 1) * n16;
      for (unsigned i = threadIdx.x; i < n16; i += 256) { const uint4 v = src[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
  }
  if (acc.x == 0x12345678u) sink[blockIdx.x] = acc.y + acc.z + acc.w;
}

// Round 6 (the round-5 verdict's item 2c): the same skeleton with what an INT8 recurrence would move -- h as int8 (128 rows x 2048 = 256 KB,
// a 64-row group 128 KB, 32 rows 64 KB) -- published with write-through (sc1) stores and fetched with eight 16-byte sc1 loads in flight per
// lane (the guide's hand-off recipe: no L1 to invalidate, 62 - 122 GB/s per workgroup instead of the 25 a plain loop got in round 3).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void persistent_probe_i8(Bar* b, uint4* h, unsigned steps, unsigned h_bytes, unsigned long long* sink) {
  unsigned phase = 0;
  const unsigned n_wg = gridDim.x, per_xcc = n_wg / 8;
  uint4 acc = {0, 0, 0, 0};
  const unsigned n16 = h_bytes / 16, mine = n16 / n_wg;
  for (unsigned t = 0; t < steps; ++t) {
    for (unsigned i = threadIdx.x; i < mine; i += 256) {
      const u32x4 v = {t, i, blockIdx.x, acc.x};
      uint4* dst = &h[(size_t)(t & 1) * n16 + blockIdx.x * mine + i];
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    grid_barrier(b, n_wg, phase, per_xcc);
    const uint4* src = h + (size_t)(t & 1) * n16;
    for (unsigned i0 = threadIdx.x; i0 < n16; i0 += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned i = i0 + 256u * k;
        const uint4* q = src + (i < n16 ? i : 0);
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"(q) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
    }
  }
  if (acc.x == 0x12345678u) sink[blockIdx.x] = acc.y + acc.z + acc.w;
}

__global__ void trivial(unsigned long long* sink) { if (threadIdx.x == 1025) sink[0] = 1; }

int main() {
  Bar* b; uint4* h; unsigned long long* sink;
  CHECK(hipMalloc(&b, sizeof(Bar))); CHECK(hipMalloc(&h, 4 << 20)); CHECK(hipMalloc(&sink, 8 * 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const unsigned steps = 2000;
  for (int n_wg : {128, 256}) {
    for (int with_h = 0; with_h < 2; ++with_h) {
      for (unsigned hb : {262144u, 524288u}) {
        if (!with_h && hb != 262144u) continue;
        CHECK(hipMemset(b, 0, sizeof(Bar)));
        hipLaunchKernelGGL(persistent_probe, dim3(n_wg), dim3(256), 0, 0, b, h, 10u, hb, with_h, sink);  // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(b, 0, sizeof(Bar)));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(persistent_probe, dim3(n_wg), dim3(256), 0, 0, b, h, steps, hb, with_h, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"probe\": \"persistent step skeleton\", \"workgroups\": %d, \"h_exchange_bytes\": %u, \"us_per_step\": %.3f}\n", n_wg, with_h ? hb : 0u, 1e3 * ms / steps);
      }
    }
  }
  for (int n_wg : {128, 256}) {
    for (unsigned hb : {65536u, 131072u, 262144u}) {
      CHECK(hipMemset(b, 0, sizeof(Bar)));
      hipLaunchKernelGGL(persistent_probe_i8, dim3(n_wg), dim3(256), 0, 0, b, h, 10u, hb, sink);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemset(b, 0, sizeof(Bar)));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(persistent_probe_i8, dim3(n_wg), dim3(256), 0, 0, b, h, steps, hb, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms2; CHECK(hipEventElapsedTime(&ms2, e0, e1));
      printf("{\"probe\": \"persistent step skeleton, int8 h: sc1 stores, 8 x 16 B sc1 loads in flight per lane\", \"workgroups\": %d, \"h_exchange_bytes\": %u, \"us_per_step\": %.3f}\n", n_wg, hb, 1e3 * ms2 / steps);
    }
  }
  // the alternative: a kernel boundary (dependent launches of a trivial 128-workgroup kernel on one stream)
  CHECK(hipEventRecord(e0));
  for (unsigned t = 0; t < steps; ++t) hipLaunchKernelGGL(trivial, dim3(128), dim3(256), 0, 0, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"probe\": \"kernel boundary, trivial 128-workgroup kernels back to back (eager)\", \"us_per_step\": %.3f}\n", 1e3 * ms / steps);
  return 0;
}
