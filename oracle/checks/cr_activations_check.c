/* oracle/checks/cr_activations_check.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The float64 algorithm the int8 recurrent step uses for LOGISTIC and TANH (stt_amd/csrc/kernels_i8.hip: em1_neg_, sigmoid_i8_, tanh_i8_),
 * restated in plain C, against the DEFINITION oracle/am_hybrid.py uses for them: the float64 evaluation of 1 / (1 + exp(-x)) and tanh(x),
 * rounded once to float (and against a long double evaluation: the correctly rounded value).  Swept over float inputs |x| < 200:
 * `n_inputs  sigmoid mismatches vs f64 / vs long double  tanh mismatches ...` -- a handful of half-way cases in 1e8 is the expectation
 * (tests/test_oracle_am.py runs a shorter sweep).  argv[1] = number of inputs.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static const uint64_t TAB[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
static double em1(double y, double* E) {
  const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
  const double hi = 6.93147180369123816490e-01 / 32.0, lo = 1.90821492927058770002e-10 / 32.0;
  double kd = y * InvLn2N + SHIFT;
  uint64_t ki; memcpy(&ki, &kd, 8);
  kd -= SHIFT;
  double r = fma(-kd, hi, y);
  r = fma(-kd, lo, r);
  double p = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0); p = fma(p, r, 1.0 / 24.0); p = fma(p, r, 1.0 / 6.0); p = fma(p, r, 0.5);
  p = fma(p * r, r, r);
  uint64_t tb = TAB[ki & 31] + (ki << 47);
  double T; memcpy(&T, &tb, 8);
  if (E) *E = fma(T, p, T);
  return fma(T, p, T - 1.0);
}
static float sig(float x) { double a = fmin(fabs((double)x), 104.0); double E; em1(-a, &E); double q = 1.0 / (1.0 + E); return (float)(x >= 0 ? q : E * q); }
static float th(float x) { double a = fmin(fabs((double)x), 22.0); double e = em1(-2.0 * a, 0); double t = -e / (2.0 + e); return (float)(x < 0 ? -t : t); }
int main(int argc, char** argv) {
  const uint64_t total = argc > 1 ? strtoull(argv[1], 0, 10) : 200000000ULL;
  // references: float64 evaluation as oracle/am_hybrid.py does, and long double for the true value
  uint64_t n = 0, bad_s64 = 0, bad_t64 = 0, bad_sl = 0, bad_tl = 0;
  uint32_t seed = 1;
  double max_em1 = 0;
  for (uint64_t it = 0; it < total; ++it) {
    uint32_t u;
    if (it < (1u << 26)) u = (uint32_t)(it * 64u + 7u);               // a regular sweep of all exponents ...
    else { seed = seed * 1664525u + 1013904223u; u = seed; }
    float x; memcpy(&x, &u, 4);
    if (!(fabsf(x) < 200.0f)) continue;
    ++n;
    float s = sig(x), t = th(x);
    float s64 = (float)(1.0 / (1.0 + exp(-(double)x))), t64 = (float)tanh((double)x);
    float sl = (float)(1.0L / (1.0L + expl(-(long double)x))), tl = (float)tanhl((long double)x);
    bad_s64 += s != s64; bad_t64 += t != t64; bad_sl += s != sl; bad_tl += t != tl;
  }
  printf("%llu %llu %llu %llu %llu\n", (unsigned long long)n, (unsigned long long)bad_s64, (unsigned long long)bad_sl, (unsigned long long)bad_t64, (unsigned long long)bad_tl);
  return 0;
}
