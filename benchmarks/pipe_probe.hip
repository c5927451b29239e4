// benchmarks/pipe_probe.hip -- which HIP streams share a hardware dispatch pipe?
//
// Round 5 found a cliff: the same pipeline runs a batch in 3.0 or 6.1 ms depending on how many streams the process had created before the
// engine's (profiles/r05_queue_placement.json) -- "as if the recurrence's queue were served in turns with another one".  Hypothesis: a
// compute pipe of the command processor works on ONE dispatch at a time, and a dispatch of more workgroups than the chip holds (every GEMM
// of the acoustic engines: ~1500 workgroups, one per CU beside the recurrence) keeps its pipe until its last workgroup has been placed.  A
// queue of short dependent launches (the recurrence: 250 per batch) that sits on the SAME pipe then waits for every GEMM to finish
// dispatching.  Streams map to hardware queues in creation order and queues to pipes round-robin, so who shares with whom is an accident.
//
// The probe: S streams created in order.  For a "hog" stream A -- a launch of many more workgroups than fit (each sleeps ~40 us, LDS sized so
// that wave slots stay free on every CU) -- and every other stream B: a chain of 16 one-wave kernels on B, started right behind the hog,
// timed with events.  A chain that shares the hog's pipe cannot start before the hog's last workgroup is placed.
//
//   hipcc --offload-arch=gfx950 -O3 benchmarks/pipe_probe.hip -o benchmarks/pipe_probe && benchmarks/pipe_probe [streams] [hogs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void hog_kernel(unsigned* sink, int ticks) {
  __shared__ unsigned lds[10240];   // 40 KiB: four workgroups per CU, 16 of its 32 wave slots
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[threadIdx.x] == 0xffffffffu) sink[0] = 1;
}
__global__ void tiny_kernel(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 16, HOGS = argc > 2 ? atoi(argv[2]) : 4, CHAIN = 16;
  std::vector<hipStream_t> st(S);
  unsigned* d = nullptr;
  CHECK(hipMalloc(&d, 4096));
  CHECK(hipMemset(d, 0, 4096));
  for (int i = 0; i < S; ++i) {
    CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st[i], d + i);   // (a stream takes its hardware queue when it is first used)
    CHECK(hipStreamSynchronize(st[i]));
  }
  hipEvent_t e0, e1, h0, h1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&h0)); CHECK(hipEventCreate(&h1));
  const char* env = getenv("GPU_MAX_HW_QUEUES");
  printf("%d streams in creation order, GPU_MAX_HW_QUEUES=%s; chain of %d one-wave launches on stream B beside a hog dispatch (16384 workgroups x 40 us) on stream A: us per chain launch (alone: first row)\n",
         S, env ? env : "(default)", CHAIN);
  printf("%-8s", "A \\ B");
  for (int b = 0; b < S; ++b) printf("%7d", b);
  printf("   hog ms\n");
  for (int a = -1; a < HOGS && a < S; ++a) {
    printf("%-8s", a < 0 ? "no hog" : (std::string("hog ") + std::to_string(a)).c_str());
    float hog_ms = 0.f;
    for (int b = 0; b < S; ++b) {
      if (b == a) { printf("%7s", "-"); continue; }
      if (a >= 0) {
        CHECK(hipEventRecord(h0, st[a]));
        hipLaunchKernelGGL(hog_kernel, dim3(16384), dim3(256), 0, st[a], d + 512, 4000);
        CHECK(hipEventRecord(h1, st[a]));
      }
      CHECK(hipEventRecord(e0, st[b]));
      for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st[b], d + 64 + b);
      CHECK(hipEventRecord(e1, st[b]));
      CHECK(hipDeviceSynchronize());
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, a >= 0 ? h0 : e0, e1));   // (from the hog's start: the chain's own first event may already have waited)
      printf("%7.1f", 1e3f * ms / CHAIN);
      if (a >= 0) CHECK(hipEventElapsedTime(&hog_ms, h0, h1));
    }
    printf("   %.2f\n", hog_ms);
  }
  return 0;
}
