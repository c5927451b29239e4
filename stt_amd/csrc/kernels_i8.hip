// stt_amd/csrc/kernels_i8.hip -- the recurrent cell in the released models' own arithmetic (gfx950).
//
// Released models are dynamic-range quantised (training/coqui_stt_training/export.py:145-146) and the reference's CPU path runs every
// FULLY_CONNECTED -- the unrolled LSTM cell's concat([x_t, h_(t-1)]) . kernel included -- through TensorFlow Lite's hybrid kernel
// (native_client/tflitemodelstate.cc:200,369-405; restated in oracle/am_hybrid.py): the float input row is quantised to int8 with ONE
// scale, max |row| / 127, int8 x int8 products are summed in int32, the sum is rescaled by (row scale x weight scale) and added to the
// float bias; LOGISTIC / TANH / MUL / ADD stay float32, each operation rounded on its own.
//
// How that maps onto the recurrence (kernels.h: LstmI8Args):
//   * integer sums are exact and associative, so the x half of a row's dot products (a GEMM over all timesteps of a chunk,
//     dense_wide_kernel<DENSE_EPI_I8_RAW>) and the h half (this file, one launch per timestep) may be computed apart and added as int32 --
//     PROVIDED both halves were quantised with the row's joint scale.  The x half is quantised with max |x_t| / 127; that is the joint
//     scale whenever max |x_t| >= max |h_(t-1)|.  |h| < 1 always, layer 3's clipped ReLU over 2048 units is nearly always >= 1.
//   * the step that produces h_(t-1) quantises it for its consumer with 127 / max |x_t| (known before the recurrence starts) and CHECKS
//     the assumption: a workgroup whose units hold a larger |h| flags the row; the consuming step then computes that row again from the
//     f32 x_t and h_(t-1) at the true joint scale (the slow path below: plain int8 dot products over K = 2H for the row).
//   * the cross-wave reduction of the int32 partial sums uses LDS atomics (order does not matter for integers).
//   * a launch covers a 16-unit slice per blockIdx.x and a ROW GROUP per blockIdx.y (launch_lstm_i8_step: 64 rows per workgroup by default,
//     so 128 rows are two workgroups per slice, one on every compute unit): flags, maxima and the slow path are per row, so every
//     dealing of the rows gives the same bits.
// Layouts: recurrent weights int8 packed per (workgroup, k-step of 64, gate tile, lane) -- 16 bytes per lane and MFMA, the same
// fragment rule as the f16 form with twice the k per instruction; h_(t-1) int8 in B-fragment order [H/64][NT][64 lanes][16].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <stdexcept>

#include "kernels.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __constant__ const uint64_t kExp2TabI8[32] = {      // bits(2^(i/32)) - (i << 47), as glibc's expf keeps them
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};

// LOGISTIC and TANH.  TFLite evaluates them with float kernels whose last bits depend on the build (std::exp / std::tanh in the reference
// kernels, Eigen's rational approximations in the optimised ones): each is within an ulp or two of the CORRECTLY ROUNDED float of the
// real function, and that value is what oracle/am_hybrid.py takes (float64 evaluation, one rounding).  The recurrence is unforgiving about
// the difference: a last-bit change of one activation can flip one int8 of the next step's quantised row, and the quantised network
// amplifies that within a few steps (measured: activations 1-3 ulp off -> transcripts of 18 of 64 utterances move).  So the
// device evaluates both functions in float64 as well -- exp(y) - 1 for y <= 0 from a 32-entry table of 2^(i/32) and a degree-7 polynomial
// (absolute error ~3e-16), one IEEE division, one rounding to float -- and agrees with the oracle except where a float64 error of 1e-15
// straddles a float rounding boundary (about one call in 1e7).  Cost: ~35 float64 instructions per activation, full rate on gfx950.
template <bool WANT_EXP>
__device__ __forceinline__ double em1_neg_(double y, const uint64_t* tab) {       // WANT_EXP ? exp(y) : exp(y) - 1, for -104 <= y <= 0
  const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
  const double Ln2N_hi = 6.93147180369123816490e-01 / 32.0, Ln2N_lo = 1.90821492927058770002e-10 / 32.0;   // fdlibm's ln2 split: k * hi is exact
  double kd = __dadd_rn(__dmul_rn(y, InvLn2N), SHIFT);
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd = __dadd_rn(kd, -SHIFT);
  double r = __fma_rn(-kd, Ln2N_hi, y);
  r = __fma_rn(-kd, Ln2N_lo, r);                                                  // |r| <= ln2 / 64
  double p = __fma_rn(r, 1.0 / 5040.0, 1.0 / 720.0);
  p = __fma_rn(p, r, 1.0 / 120.0);
  p = __fma_rn(p, r, 1.0 / 24.0);
  p = __fma_rn(p, r, 1.0 / 6.0);
  p = __fma_rn(p, r, 0.5);
  p = __fma_rn(__dmul_rn(p, r), r, r);                                            // e^r - 1
  const double T = __longlong_as_double((long long)(tab[ki & 31] + (ki << 47)));  // 2^(k/32): the table holds bits(2^(i/32)) - (i << 47)
  return __fma_rn(T, p, WANT_EXP ? T : __dadd_rn(T, -1.0));    // (1 + exp(y) - 1 would cancel for y << 0, exp(y) - 1 computed from exp(y) for y ~ 0)
}
__device__ __forceinline__ float sigmoid_i8_(float x, const uint64_t* tab) {
  const double a = fmin(fabs((double)x), 104.0);           // (exp(-104) is below the smallest float)
  const double E = em1_neg_<true>(-a, tab);                // exp(-|x|)
  const double q = 1.0 / __dadd_rn(1.0, E);                // 1 / (1 + exp(-|x|))
  return (float)(x >= 0.0f ? q : __dmul_rn(E, q));
}
__device__ __forceinline__ float tanh_i8_(float x, const uint64_t* tab) {
  const double a = fmin(fabs((double)x), 22.0);
  const double e = em1_neg_<false>(__dmul_rn(-2.0, a), tab);   // exp(-2|x|) - 1
  const double t = -e / __dadd_rn(2.0, e);                 // (1 - exp(-2|x|)) / (1 + exp(-2|x|))
  return (float)(x < 0.0f ? -t : t);
}
__device__ __forceinline__ signed char quant_i8_(float v, float inv) { return (signed char)fminf(fmaxf(roundf(__fmul_rn(v, inv)), -127.0f), 127.0f); }

template <bool PIN>
__device__ __forceinline__ void mfma_i8_(i32x4& acc, const i32x4& a, const i32x4& b) {
  if constexpr (PIN) asm("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
}

// DBG (timing probes only, wrong results; tunable lstm_probe through STTX_TestHybridChain): bit 0 = activations replaced by a multiply, bit 1 =
// no cross-wave reduction, bit 2 = no cell-update operand loads, bit 3 = every operand load of the k-loop reads the wave's first fragment.
template <int NT, int DBG = 0>
__device__ __forceinline__ void lstm_i8_step_body(const LstmI8Args& a) {
  constexpr bool PIN = NT == 8;                  // 128 accumulator registers pinned in the accumulator file (see kernels_am.hip: lstm_mfma)
  constexpr int NTR = NT * 16;                   // rows a WORKGROUP covers; gridDim.y workgroups share a 16-unit slice, each with its own rows
  constexpr int SLOTS = NT * 64, ITS = (SLOTS + 255) / 256;
  __shared__ int red[4][NT][4][64];              // [gate tile][batch tile][component][lane]: component-major -> conflict-free 4-byte atomics
  __shared__ __attribute__((aligned(16))) signed char sq[2 * 4096];   // slow path: one row's [x_t | h_(t-1)] at the joint scale
  __shared__ float sred[256];
  __shared__ int s_nflag;
  __shared__ unsigned char s_flist[NTR];
  __shared__ uint64_t s_exp2[32];                // 2^(i/32) for the activations (a data-dependent read of __constant__ memory is a vector load through the caches)
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, wg = blockIdx.x;
  if (tid < 32) s_exp2[tid] = kExp2TabI8[tid];   // (visible behind the reduction's barriers, long before the cell update reads it)
  const int H = a.n_hidden, B = a.batch, KS = H / 64, NWG = H / 16;
  const int NTT = NT * (int)gridDim.y, NTRT = NTT * 16;      // batch tiles / rows of the launch (the layouts of hq, flag, pmax, zslow)
  const int jt0 = NT * (int)blockIdx.y, r0 = jt0 * 16;       // this workgroup's first batch tile / row
  const int par = a.t & 1, epoch = a.t + 1;
  // issued now, consumed behind the k-loop: is any row of this step flagged?
  const bool past_end = a.row_frames != nullptr && r0 + tid < B && a.t0 + a.t >= a.row_frames[r0 + tid < B ? r0 + tid : 0];   // (batch path: a row beyond its utterance)
  const int myflag = (tid < NTR && r0 + tid < B && !past_end) ? (a.flag[par * NTRT + r0 + tid] == epoch ? 1 : 0) : 0;

  const i32x4* wp = reinterpret_cast<const i32x4*>(a.whp) + (size_t)wg * KS * 4 * 64 + lane;
  const i32x4* hp = reinterpret_cast<const i32x4*>(a.hq_in) + lane;
  i32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (i32x4){0, 0, 0, 0};

  // wave q takes the k-steps q, q + 4, ...
#define I8_LOAD(W, Hh, s_)                                                                  \
  do {                                                                                      \
    const int ks_ = (s_);                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) W[i_] = wp[(DBG & 8) ? (size_t)0 : (size_t)(ks_ * 4 + i_) * 64];   \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) Hh[j_] = hp[(DBG & 8) ? (size_t)0 : (size_t)(ks_ * NTT + jt0 + j_) * 64]; \
  } while (0)
#define I8_MMA(W, Hh)                                                                       \
  do {                                                                                      \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) {                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) mfma_i8_<PIN>(acc[i_][j_], W[i_], Hh[j_]);  \
    }                                                                                       \
  } while (0)
#define I8_FENCE() __builtin_amdgcn_sched_barrier(0)
  if (KS % 8 == 0) {
    // every wave has an even number (>= 2) of k-steps: two operand groups in flight, the last pair peeled (kernels_am.hip: the same
    // structure and the same reasons -- the asm MFMAs carry no scheduling model, a prefetch condition inside the loop costs a vmcnt(0))
    const int n_my = KS / 4;
    i32x4 wa[4], ha[NT], wb[4], hb[NT];
    I8_LOAD(wa, ha, q);
    int i = 0;
    for (; i + 2 < n_my; i += 2) {
      I8_LOAD(wb, hb, q + 4 * (i + 1));
      I8_FENCE();
      I8_MMA(wa, ha);
      I8_FENCE();
      I8_LOAD(wa, ha, q + 4 * (i + 2));
      I8_FENCE();
      I8_MMA(wb, hb);
      I8_FENCE();
    }
    I8_LOAD(wb, hb, q + 4 * (i + 1));
    I8_FENCE();
    I8_MMA(wa, ha);
    I8_FENCE();
    I8_MMA(wb, hb);
  } else {
    for (int s = q; s < KS; s += 4) {
      i32x4 w[4], hv[NT];
      I8_LOAD(w, hv, s);
      I8_MMA(w, hv);
    }
  }
#undef I8_LOAD
#undef I8_MMA
#undef I8_FENCE
  if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the hazard recogniser does not look inside inline asm

  // The x half of the NEXT step's sums (4 MB per step for 128 rows, written by the GEMM: HBM / Infinity Cache) is touched now, one dword per
  // (row, gate) and lane: workgroup i of every launch runs on XCD i % 8, so the lines wait in the L2 that step t+1's workgroup i reads them
  // from, and their latency passes behind this step's reduction and cell update instead of in front of the next one's.
  int touch[ITS];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    touch[it] = 0;
    if (a.t + 1 < a.T && !(DBG & 4)) {
      const int row = min(r0 + ((tid + 256 * it) >> 6) * 16 + (lane & 15), B - 1);
      touch[it] = a.accx[((size_t)(a.t + 1) * B + row) * (size_t)(4 * H) + (size_t)(lane >> 4) * H + wg * 16];
    }
  }
  // ---- operands of the cell update: in flight while the partial sums meet in LDS
  // slot s = tid + 256 * it: batch tile j = s >> 6, lane-slot ls = s & 63 = (unit group, row of the tile) in the MFMA output layout:
  // this thread ends up with all four gates of units wg*16 + 4*ug .. +3 of batch row j*16 + (ls & 15)
  const int ls = lane, ug = ls >> 4;
  const int unit0 = wg * 16 + ug * 4;
  i32x4 ax[ITS][4];
  float4 cv[ITS];
  float xs_[ITS];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int s = tid + 256 * it, j = s >> 6, row = r0 + j * 16 + (ls & 15);
#pragma unroll
    for (int g = 0; g < 4; ++g) ax[it][g] = (i32x4){0, 0, 0, 0};
    cv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    xs_[it] = 1.0f;
    if (s < SLOTS && row < B && !(DBG & 4)) {
      const size_t rowi = (size_t)a.t * B + row;
#pragma unroll
      for (int g = 0; g < 4; ++g) ax[it][g] = *reinterpret_cast<const i32x4*>(a.accx + rowi * (size_t)(4 * H) + (size_t)g * H + unit0);
      cv[it] = *reinterpret_cast<const float4*>(a.c + (size_t)row * H + unit0);
      xs_[it] = a.xscale[rowi];
    }
  }
  float4 bias4[4], ws4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bias4[g] = *reinterpret_cast<const float4*>(a.bias + (size_t)g * H + unit0);
    if (a.wscale_n > 1) ws4[g] = *reinterpret_cast<const float4*>(a.wscale + (size_t)g * H + unit0);
    else { const float w0 = a.wscale[0]; ws4[g] = make_float4(w0, w0, w0, w0); }
  }

  // ---- cross-wave reduction of the int32 partial sums: wave 0 stores, the others add (integers: any order)
  if (q == 0 || (DBG & 2)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!(DBG & 2) || j == q || j == q + 4) red[i][j][r][lane] = acc[i][j][r];
  }
  const int any = __syncthreads_or(myflag);
  if (q != 0 && q < KS && !(DBG & 2)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&red[i][j][r][lane], acc[i][j][r]);
  }

  // ---- slow path (rare): rows whose max |h_(t-1)| exceeded max |x_t| -- both halves again, at the true joint scale
  if (any) {
    if (tid == 0) s_nflag = 0;
    __syncthreads();
    if (myflag) { const int i = atomicAdd(&s_nflag, 1); s_flist[i] = (unsigned char)tid; }
    __syncthreads();
    const int nf = s_nflag;
    for (int fi = 0; fi < nf; ++fi) {
      const int r = r0 + s_flist[fi];
      float m = 0.0f;
      for (int w = tid; w < NWG; w += 256) m = fmaxf(m, a.pmax[((size_t)par * NTRT + r) * NWG + w]);
      sred[tid] = m;
      __syncthreads();
      for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) sred[tid] = fmaxf(sred[tid], sred[tid + d]);
        __syncthreads();
      }
      const float mh = sred[0];
      const size_t rowi = (size_t)a.t * B + r;
      const float range = fmaxf(a.xrange[rowi], mh);                    // PortableSymmetricQuantizeFloats over concat([x_t, h])
      const float inv = range > 0.0f ? __fdiv_rn(127.0f, range) : 0.0f;
      const float sf = range > 0.0f ? __fdiv_rn(range, 127.0f) : 1.0f;
      const float* xsrc = a.y3 + rowi * H;
      const float* hsrc = a.t == 0 ? a.h_prev0 + (size_t)r * H : a.h_all + ((size_t)(a.t - 1) * B + r) * H;
      for (int k = tid; k < H; k += 256) { sq[k] = quant_i8_(xsrc[k], inv); sq[H + k] = quant_i8_(hsrc[k], inv); }
      __syncthreads();
      const int col = tid >> 2, part = tid & 3;                         // 64 gate columns x four lanes each
      const int n = (col >> 4) * H + wg * 16 + (col & 15);
      const int kq = H / 4;
      const signed char* wxr = a.wxq + (size_t)n * H + part * kq;
      const signed char* whr = a.whq + (size_t)n * H + part * kq;
      const signed char* sx = sq + part * kq;
      const signed char* sh = sq + H + part * kq;
      int dot = 0;
      for (int k = 0; k < kq; ++k) dot += (int)wxr[k] * (int)sx[k] + (int)whr[k] * (int)sh[k];
      dot += __shfl_xor(dot, 1);
      dot += __shfl_xor(dot, 2);
      if (part == 0) {
        float prod;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"((float)dot), "v"(__fmul_rn(sf, a.wscale[a.wscale_n > 1 ? n : 0])));
        a.zslow[((size_t)wg * NTRT + r) * 64 + col] = a.bias[n] + prod;
      }
      __syncthreads();
    }
    if (wg == 0 && tid == 0 && a.slow_count) atomicAdd(a.slow_count, (unsigned)nf);   // (one add per row group)
  }
  __syncthreads();

  // ---- cell update, publish
  const bool last = a.t + 1 >= a.T;
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int s = tid + 256 * it, j = s >> 6, row = r0 + j * 16 + (ls & 15);
    if (s >= SLOTS) break;
    const bool live = row < B;
    const size_t rowi = (size_t)a.t * B + row;
    const bool slow = any && live && a.flag[par * NTRT + row] == epoch && !(a.row_frames != nullptr && a.t0 + a.t >= a.row_frames[row]);
    float hv[4] = {0.f, 0.f, 0.f, 0.f};
    float4 cn4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int sum = ax[it][g][r] + red[g][j][r][ls];
          float prod;   // bias + float(sum) * (row scale * weight scale), each operation rounded on its own (oracle/am_hybrid.py: fully_connected_hybrid)
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"((float)sum), "v"(__fmul_rn(xs_[it], (&ws4[g].x)[r])));
          z[g] = (&bias4[g].x)[r] + prod;
          if (slow) z[g] = a.zslow[((size_t)wg * NTRT + row) * 64 + g * 16 + ug * 4 + r];
        }
        // gate order i, j, f, o (deepspeech_model.py:144-168); MUL, MUL, ADD as separate float ops
        float cn;
        if constexpr (DBG & 1) { cn = __fadd_rn(__fmul_rn(z[2], (&cv[it].x)[r]), __fmul_rn(z[0], z[1])); hv[r] = __fmul_rn(z[3], cn) * 1e-3f; }
        else if constexpr (DBG & 16) {   // experiment (tunable lstm_probe = 16, also honoured by the batch path): the device library's float expf / tanhf -- NOT bit-level
          auto sg = [](float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); };
          cn = __fadd_rn(__fmul_rn(sg(z[2]), (&cv[it].x)[r]), __fmul_rn(sg(z[0]), tanhf(z[1])));
          hv[r] = __fmul_rn(sg(z[3]), tanhf(cn));
        }
        else {
          cn = __fadd_rn(__fmul_rn(sigmoid_i8_(z[2], s_exp2), (&cv[it].x)[r]), __fmul_rn(sigmoid_i8_(z[0], s_exp2), tanh_i8_(z[1], s_exp2)));
          hv[r] = __fmul_rn(sigmoid_i8_(z[3], s_exp2), tanh_i8_(cn, s_exp2));
        }
        (&cn4.x)[r] = cn;
      }
      *reinterpret_cast<float4*>(a.c + (size_t)row * H + unit0) = cn4;
      const float4 h4 = make_float4(hv[0], hv[1], hv[2], hv[3]);
      *reinterpret_cast<float4*>(a.h_all + rowi * H + unit0) = h4;
      if (last) *reinterpret_cast<float4*>(a.h_last + (size_t)row * H + unit0) = h4;
    }
    if (!last) {
      // h_t for step t+1, quantised with 127 / max |x_(t+1)| -- valid when that is the joint range; checked here, per workgroup
      const float rn = live ? a.xrange[rowi + B] : 0.0f;
      const float inv = rn > 0.0f ? __fdiv_rn(127.0f, rn) : 0.0f;
      const unsigned pk = (unsigned)(unsigned char)quant_i8_(hv[0], inv) | ((unsigned)(unsigned char)quant_i8_(hv[1], inv) << 8) |
                          ((unsigned)(unsigned char)quant_i8_(hv[2], inv) << 16) | ((unsigned)(unsigned char)quant_i8_(hv[3], inv) << 24);
      const int ks = wg >> 2, grp = wg & 3;
      reinterpret_cast<unsigned*>(a.hq_out)[(((size_t)ks * NTT + jt0 + j) * 64 + grp * 16 + (ls & 15)) * 4 + ug] = pk;
      float m = fmaxf(fmaxf(fabsf(hv[0]), fabsf(hv[1])), fmaxf(fabsf(hv[2]), fabsf(hv[3])));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      if (ug == 0) {
        a.pmax[((size_t)(par ^ 1) * NTRT + row) * NWG + wg] = m;
        if (live && m > rn && !(a.row_frames != nullptr && a.t0 + a.t + 1 >= a.row_frames[row])) a.flag[(par ^ 1) * NTRT + row] = epoch + 1;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITS; ++it) asm volatile("" ::"v"(touch[it]));   // (the touches are loads whose values nobody needs: keep them from being dropped)
}

template <int NT>
__global__ __launch_bounds__(256, 2) void lstm_i8_step_kernel(LstmI8Args a) { lstm_i8_step_body<NT>(a); }
// 128 rows: 128 accumulator registers + the operand double buffer (kernels_am.hip: lstm_step8_kernel)
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(160))) void lstm_i8_step8_kernel(LstmI8Args a) { lstm_i8_step_body<8>(a); }
#ifdef STT_TEST_HOOKS   // the timing probes (wrong results) exist in libstt_test.so only
template <int DBG>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(160))) void lstm_i8_probe8_kernel(LstmI8Args a) { lstm_i8_step_body<8, DBG>(a); }
#endif

// Before step 0 of a launch sequence: the carried h ([B][H] f32; null = zeros) quantised for step 0 at 127 / max |x_0| -- with the
// TRUE max |h| known here (one workgroup reads the whole row), so the flag of step 0 is exact --, a copy of it for the slow path, and
// the per-step tables cleared.
__global__ __launch_bounds__(256) void lstm_i8_prep_kernel(LstmI8Args a, const float* __restrict__ h_src, int NT) {
  __shared__ float sred[256];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int H = a.n_hidden, B = a.batch, NTR = NT * 16, NWG = H / 16;
  const bool live = row < B && h_src != nullptr;
  float m = 0.0f;
  if (live)
    for (int k = tid; k < H; k += 256) m = fmaxf(m, fabsf(h_src[(size_t)row * H + k]));
  sred[tid] = m;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) sred[tid] = fmaxf(sred[tid], sred[tid + d]);
    __syncthreads();
  }
  const float mh = sred[0];
  const float mx = row < B ? a.xrange[row] : 0.0f;                    // row (t = 0, b = row)
  const float inv = mx > 0.0f ? __fdiv_rn(127.0f, mx) : 0.0f;
  for (int k = tid; k < H; k += 256) {
    const float v = live ? h_src[(size_t)row * H + k] : 0.0f;
    if (row < B) const_cast<float*>(a.h_prev0)[(size_t)row * H + k] = v;
    // byte of (row, k) in the fragment-ordered buffer: [k / 64][row / 16][(k % 64) / 16 * 16 + row % 16][k % 16]
    const_cast<signed char*>(a.hq_in)[((((size_t)(k >> 6) * NT + (row >> 4)) * 64 + ((k & 63) >> 4) * 16 + (row & 15)) << 4) + (k & 15)] = quant_i8_(v, inv);
  }
  for (int w = tid; w < NWG; w += 256) a.pmax[(size_t)row * NWG + w] = w == 0 ? mh : 0.0f;   // pmax[0][row][:]
  if (tid == 0) {
    a.flag[row] = (row < B && mh > mx && !(a.row_frames != nullptr && a.t0 >= a.row_frames[row])) ? 1 : 0;     // flag[0][row]: epoch of step 0 is 1
    a.flag[NTR + row] = 0;
  }
}

}  // namespace

size_t lstm_i8_hq_bytes(int H, int NT) { return (size_t)(H / 64) * NT * 64 * 16; }

void launch_lstm_i8_prep(const LstmI8Args& a, const float* h_src, int NT, hipStream_t st) {
  hipLaunchKernelGGL(lstm_i8_prep_kernel, dim3(NT * 16), dim3(256), 0, st, a, h_src, NT);
}

void launch_lstm_i8_step(const LstmI8Args& a, int NT, hipStream_t st, int rows_per_wg) {
  if (a.n_hidden % 64 != 0 || a.n_hidden > 4096) throw std::runtime_error("lstm int8 step: n_hidden must be a multiple of 64, at most 4096");
  // Row groups: the NT batch tiles of a launch may be dealt to NT / nt workgroups per 16-unit slice (gridDim.y), nt tiles each.  The
  // weights are then read once per row group (from the L2s: 16 MB), but a workgroup's matrix-core work, float64 activations and x-half
  // reads shrink with its rows -- and 128 slices x 2 groups is one workgroup on every compute unit instead of on half of them.
  int nt = NT;
#ifdef STT_TEST_HOOKS
  const bool probing = a.probe != 0;
#else
  const bool probing = false;
#endif
  if (!probing && rows_per_wg >= 16 && rows_per_wg < NT * 16) nt = rows_per_wg / 16;
  if (nt != 1 && nt != 2 && nt != 4 && nt != 8) nt = NT;
  const dim3 grid(a.n_hidden / 16, NT / nt), block(256);
  switch (nt) {
    case 1: hipLaunchKernelGGL(lstm_i8_step_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(lstm_i8_step_kernel<2>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(lstm_i8_step_kernel<4>, grid, block, 0, st, a); break;
    case 8:
#ifndef STT_TEST_HOOKS
      hipLaunchKernelGGL(lstm_i8_step8_kernel, grid, block, 0, st, a);
#else
      switch (a.probe) {      // (STTX_TestHybridChain with the tunable lstm_probe: timing probes, wrong results)
        case 1: hipLaunchKernelGGL(lstm_i8_probe8_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(lstm_i8_probe8_kernel<2>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(lstm_i8_probe8_kernel<4>, grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL(lstm_i8_probe8_kernel<8>, grid, block, 0, st, a); break;
        case 15: hipLaunchKernelGGL(lstm_i8_probe8_kernel<15>, grid, block, 0, st, a); break;
        case 16: hipLaunchKernelGGL(lstm_i8_probe8_kernel<16>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(lstm_i8_step8_kernel, grid, block, 0, st, a); break;
      }
#endif
      break;
    default: throw std::runtime_error("lstm int8 step: batch tiles must be 1, 2, 4 or 8");
  }
}
