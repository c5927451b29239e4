// stt_amd/csrc/sttmath.h -- float transcendental helpers for the HIP decoder.
//
// The reference decoder's float arithmetic is glibc expf/logf (log_sum_exp<float>,
// native_client/ctcdecode/decoder_utils.h:46-53; get_pruned_emissions,
// ctc_beam_search_decoder.cpp:355).  "Identical beam output" therefore needs the *same*
// roundings on the device, not merely accurate ones: stt_expf/stt_logf below follow the
// published glibc (>= 2.27) algorithm -- double-precision table + polynomial, one rounding
// to float -- and agree with host libm for every one of the 2^32 inputs (measured during
// development; tests/test_gpu_math.py re-checks a dense sample on the device).
// r = fma(InvLn2N, x, -kd) is the contraction glibc's FMA build (the one every AVX2 host
// selects through ifunc) performs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STT_NEG_INF (-3.40282346638528859811704183484516925e+38f)  // -NUM_FLT_INF, decoder_utils.h:11
#define STT_FLT_MIN (1.17549435082228750796873653722224568e-38f)   // NUM_FLT_MIN,  decoder_utils.h:12

namespace sttm {

__device__ __constant__ const uint64_t kExp2Tab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};

__device__ __constant__ const double kLogfTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

// All double arithmetic below is written with explicit __dmul_rn/__dadd_rn/__fma_rn so the
// compiler can neither contract nor reassociate it.
// The table arguments let a kernel keep its own copy of the two tables in LDS (a data-dependent read of __constant__
// memory is a vector load through the cache hierarchy; the search kernel evaluates these on its critical path).
template <class ExpTab>
__device__ __forceinline__ float stt_expf_t(float x, ExpTab exp2_tab) {
  const double N = 32.0;
  const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  const double SHIFT = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+0 * N;
  const uint32_t ux = __float_as_uint(x);
  const uint32_t abstop = (ux >> 20) & 0x7ff;
  if (abstop >= 0x42b) {  // |x| >= 88 or nan
    if (ux == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  const double xd = (double)x;
  const double z = __dmul_rn(InvLn2N, xd);
  double kd = __dadd_rn(z, SHIFT);
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd = __dadd_rn(kd, -SHIFT);
  const double r = __fma_rn(InvLn2N, xd, -kd);
  uint64_t t = exp2_tab[ki & 31];
  t += ki << 47;
  const double s = __longlong_as_double((long long)t);
  const double z2 = __dadd_rn(__dmul_rn(C0, r), C1);
  const double r2 = __dmul_rn(r, r);
  double y = __dadd_rn(__dmul_rn(C2, r), 1.0);
  y = __dadd_rn(__dmul_rn(z2, r2), y);
  y = __dmul_rn(y, s);
  return (float)y;
}

__device__ __forceinline__ float stt_expf(float x) { return stt_expf_t(x, kExp2Tab); }

// log_tab: 16 x {invc, logc} doubles, flat [32]
template <class LogTab>
__device__ __forceinline__ float stt_logf_t(float x, LogTab log_tab) {
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return __uint_as_float(0xff800000u);
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __uint_as_float(0x7fc00000u);
    ix = __float_as_uint(x * 0x1p23f);
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) & 15;
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = log_tab[2 * i], logc = log_tab[2 * i + 1];
  const double z = (double)__uint_as_float(iz);
  const double r = __dadd_rn(__dmul_rn(z, invc), -1.0);
  const double y0 = __dadd_rn(logc, __dmul_rn((double)k, Ln2));
  const double r2 = __dmul_rn(r, r);
  double y = __dadd_rn(__dmul_rn(A1, r), A2);
  y = __dadd_rn(__dmul_rn(A0, r2), y);
  y = __dadd_rn(__dmul_rn(y, r2), __dadd_rn(y0, r));
  return (float)y;
}

__device__ __forceinline__ float stt_logf(float x) { return stt_logf_t(x, &kLogfTab[0][0]); }

// log_sum_exp<float>, decoder_utils.h:46-53: log(exp(x - max) + exp(y - max)) + max.  The larger operand contributes
// exp(+0.0f), which glibc's expf returns as exactly 1.0f, so only the smaller one is evaluated (float addition commutes).
template <class ExpTab, class LogTab>
__device__ __forceinline__ float stt_log_sum_exp_t(float x, float y, ExpTab exp2_tab, LogTab log_tab) {
  if (x <= STT_NEG_INF) return y;
  if (y <= STT_NEG_INF) return x;
  const float xmax = (x < y) ? y : x;  // std::max
  const float xmin = (x < y) ? x : y;
  const float e = stt_expf_t(__fadd_rn(xmin, -xmax), exp2_tab);
  return __fadd_rn(stt_logf_t(__fadd_rn(1.0f, e), log_tab), xmax);
}
__device__ __forceinline__ float stt_log_sum_exp(float x, float y) { return stt_log_sum_exp_t(x, y, kExp2Tab, &kLogfTab[0][0]); }

}  // namespace sttm
