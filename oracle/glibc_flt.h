/* oracle/glibc_flt.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Bit-exact restatement of the two single-precision libm functions the reference decoder
 * calls (objdump of ctc_beam_search_decoder.o / path_trie.o: `expf`, `logf`; they come from
 * log_sum_exp<float>, native_client/ctcdecode/decoder_utils.h:46-53, and from
 * get_pruned_emissions, ctc_beam_search_decoder.cpp:355).  glibc is a dependency of the
 * reference that is NOT in /root/reference; its expf/logf (glibc >= 2.27,
 * sysdeps/ieee754/flt-32/e_expf.c, e_logf.c) are the ARM "optimized-routines" algorithms
 * (Szabolcs Nagy, MIT licence): double-precision table + polynomial, one final rounding.
 * The algorithm is restated here from its published description; tables: exp2 table
 * T[i] = bits(2^(i/32)) - (i << 47) (regenerated with mpmath), logf table = 16 (1/c, log c)
 * pairs.
 *
 * PINNED: tests/test_oracle_math.py compares these against the host libm (glibc 2.35 here)
 * -- exhaustively over all 2^32 inputs during development: 0 mismatches for logf and for
 * expf *as built for FMA hosts* (glibc's ifunc picks __expf_fma on every AVX2/FMA x86,
 * where the compiler contracts r = InvLn2N*x - kd into one fma; the non-FMA build differs
 * at exactly two inputs, x = -0x1.f8cbb2p+5 and x = 0x1.04845ep+5).
 */
#ifndef ORACLE_GLIBC_FLT_H
#define ORACLE_GLIBC_FLT_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t gf_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float gf_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t gf_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double gf_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

static const uint64_t GF_EXP2_TAB[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};

static inline float gf_expf(float x) {
  const double N = 32.0;
  const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  const double SHIFT = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+0 * N;
  double xd = (double)x;
  uint32_t abstop = (gf_asuint(x) >> 20) & 0x7ff;
  if (abstop >= (gf_asuint(88.0f) >> 20)) {
    if (gf_asuint(x) == gf_asuint(-INFINITY)) return 0.0f;
    if (abstop >= (gf_asuint(INFINITY) >> 20)) return x + x;
    if (x > 0x1.62e42ep6f) return INFINITY;
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  double z = InvLn2N * xd;
  double kd = z + SHIFT;
  uint64_t ki = gf_asuint64(kd);
  kd -= SHIFT;
  double r = fma(InvLn2N, xd, -kd); /* the FMA-host contraction, see header */
  uint64_t t = GF_EXP2_TAB[ki % 32];
  t += ki << 47;
  double s = gf_asdouble(t);
  z = C0 * r + C1;
  double r2 = r * r;
  double y = C2 * r + 1.0;
  y = z * r2 + y;
  y = y * s;
  return (float)y;
}

static const double GF_LOGF_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

static inline float gf_logf(float x) {
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = gf_asuint(x);
  if (ix == 0x3f800000) return 0;
  if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
    if (ix * 2 == 0) return -INFINITY;
    if (ix == 0x7f800000) return x;
    if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return NAN;
    ix = gf_asuint(x * 0x1p23f);
    ix -= 23 << 23;
  }
  uint32_t tmp = ix - 0x3f330000;
  int i = (tmp >> 19) % 16;
  int k = (int32_t)tmp >> 23;
  uint32_t iz = ix - (tmp & 0xff800000);
  double invc = GF_LOGF_TAB[i][0], logc = GF_LOGF_TAB[i][1];
  double z = (double)gf_asfloat(iz);
  double r = z * invc - 1;
  double y0 = logc + (double)k * Ln2;
  double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

/* log_sum_exp<float>, native_client/ctcdecode/decoder_utils.h:46-53 */
static inline float gf_log_sum_exp(float x, float y) {
  const float num_min = -3.40282346638528859811704183484516925e+38f; /* -FLT_MAX */
  if (x <= num_min) return y;
  if (y <= num_min) return x;
  float xmax = x > y ? x : y; /* std::max(x, y) */
  return gf_logf(gf_expf(x - xmax) + gf_expf(y - xmax)) + xmax;
}
#endif
