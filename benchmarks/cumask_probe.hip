// benchmarks/cumask_probe.hip -- where do the bits of hipExtStreamCreateWithCUMask land on a MI355X?
//
// The batch pipeline shares the chip between the beam searches (one 1024-lane workgroup per stream: a whole CU's register file each) and
// the three acoustic engines.  Today the hardware dispatcher decides who sits where, launch by launch.  A stream created with a CU mask is
// confined to the CUs whose bits are set -- but which physical CU is bit i?  This probe launches a census kernel (every workgroup records
// the XCC, shader engine and CU it ran on) on streams with a few masks and prints the histograms: per XCC the number of distinct CUs used.
//
//   hipcc --offload-arch=gfx950 -O3 benchmarks/cumask_probe.hip -o benchmarks/cumask_probe && benchmarks/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void census_kernel(uint32_t* out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // hold the CU for a while so that the grid spreads over everything the mask allows
  const unsigned long long t0 = __builtin_readcyclecounter();
  while ((long long)(__builtin_readcyclecounter() - t0) < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}

static void census(const char* what, hipStream_t st, int blocks) {
  uint32_t* d = nullptr;
  CHECK(hipMalloc(&d, blocks * 4));
  hipLaunchKernelGGL(census_kernel, dim3(blocks), dim3(1024), 0, st, d, 200000);
  CHECK(hipStreamSynchronize(st));
  std::vector<uint32_t> h(blocks);
  CHECK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
  CHECK(hipFree(d));
  std::set<uint32_t> cus[16];
  for (uint32_t v : h) {
    const uint32_t xcc = v >> 16, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
    cus[xcc & 15].insert(se << 8 | sh << 4 | cu);
  }
  size_t total = 0;
  printf("%-34s distinct CUs per XCC:", what);
  for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); total += cus[x].size(); }
  printf("  = %zu\n", total);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  printf("%s: %d CUs\n", p.name, ncu);
  hipStream_t plain;
  CHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  census("no mask", plain, 1024);
  struct { const char* name; int lo, hi, stride; } masks[] = {
      {"bits [0,128)", 0, 128, 1}, {"bits [128,256)", 128, 256, 1}, {"bits [0,32)", 0, 32, 1}, {"bits [0,64)", 0, 64, 1},
      {"even bits", 0, 256, 2}, {"bits [0,8)", 0, 8, 1}, {"bits 0,8,16,..,248", 0, 256, 8}};
  for (auto& m : masks) {
    std::vector<uint32_t> w(words, 0);
    for (int i = m.lo; i < m.hi && i < ncu; i += m.stride) w[i / 32] |= 1u << (i % 32);
    hipStream_t st;
    CHECK(hipExtStreamCreateWithCUMask(&st, words, w.data()));
    census(m.name, st, 1024);
    CHECK(hipStreamDestroy(st));
  }
  return 0;
}
