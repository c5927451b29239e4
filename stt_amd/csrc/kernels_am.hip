// stt_amd/csrc/kernels_am.hip -- acoustic half of the hot path as hand-written gfx950 kernels.
//
// Reference rows:  a2/a3 feedAudioContent + compute_mfcc (stt.cc:105-128,
// tflitemodelstate.cc:407-436; op definition util/feeding.py:51-73), a4 context windows
// (stt.cc:272-309), a5 infer (tflitemodelstate.cc:369-405; graph deepspeech_model.py:66-89,
// 144-168, 171-263, 357).
//
// Data layout in HBM:
//   audio      int16 [B][n_max]                      (row b = utterance b, zero tail)
//   feats      f32   [B][t_max][26]                  MFCC, the `mfccs` tensor of the reference
//   x1         f16   [t_max*B][512]                  19-frame context rows, time-major (row = t*B+b), K padded 494->512
//   act        f16   [t_max*B][2048]                 dense activations, time-major
//   xproj      f32   [t_max*B][8192]                 x.K[:H] + b  (gate order i,j,f,o)
//   whp        f16   packed per (workgroup, wave, k-step, tile, lane) for the recurrent kernel
//   hp         f16   [H/32][NT][64][8]               h_{t-1} in MFMA B-fragment order (double buffered)
//   probs      f32   [B][t_max][C]
// All dense weights are stored transposed, W^T [N][K] f16, so both MFMA operands are K-contiguous.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <stdexcept>
#include <type_traits>

#include "kernels.h"
#include "tuning.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// =============================================================================================
// MFCC: one workgroup of NFFT / 2 threads per window.  Hann window, NFFT-point FFT (Stockham radix-2, LDS ping-pong; NFFT =
// NextPowerOfTwo(window), as TF's AudioSpectrogram: 512 for 32 ms at 16 kHz, 256 at 8 kHz), |X|^2 -> f32 -> sqrt, 40 triangular
// mel bands, ln, DCT-II -> 26.
// Arithmetic is f64 like the TensorFlow ops it restates (oracle/am_ref.py); the explicit
// __dmul_rn/__dadd_rn keep the accumulation orders of the op (no contraction).
// =============================================================================================
template <int NFFT>
__global__ __launch_bounds__(NFFT / 2) void mfcc_kernel(MfccArgs a) {
  constexpr int HALF = NFFT / 2, NBIN = HALF + 1;  // threads = butterflies per stage; spectrum bins
  // (dynamic LDS: a 2048-point transform -- 32 ms windows at 44.1 / 48 kHz -- takes 2 x 32 KiB of ping-pong + the spectrum: past the 64 KiB a
  // static declaration may have; launch_mfcc asks for it)
  extern __shared__ __attribute__((aligned(16))) unsigned char mfcc_smem[];
  double2 (*buf)[NFFT] = reinterpret_cast<double2 (*)[NFFT]>(mfcc_smem);          // [2][NFFT]
  double* amp = reinterpret_cast<double*>(mfcc_smem + 2 * NFFT * sizeof(double2));  // [NBIN]
  double* lmel = amp + NBIN + 1;                                                    // [40]
  const int frame = blockIdx.x;
  const int b = frame / a.t_max;
  const int f = frame - b * a.t_max;
  const int n = a.n_samples ? a.n_samples[b] : a.all_n_samples;
  const int nf = a.n_frames ? a.n_frames[b] : a.all_n_frames;
  float* out = a.feats_ptrs ? a.feats_ptrs[b] + (size_t)f * a.n_coef : a.feats + ((size_t)b * a.t_max + f) * a.n_coef;
  if (f >= nf) {  // beyond the utterance: defined zeros (never consumed); per-stream outputs end at n_frames
    if (!a.feats_ptrs && threadIdx.x < a.n_coef) out[threadIdx.x] = 0.0f;
    return;
  }
  const int16_t* au = a.audio + (size_t)(a.rows ? a.rows[b] : b) * a.n_max;
  const int tid = threadIdx.x;
  for (int i = tid; i < NFFT; i += HALF) {
    const int s = f * a.win_step + i;
    // stt.cc:113-114: (float)sample * (1.0f / 32768); zero tail = tflitemodelstate.cc:341-355
    const float x = (s < n && i < a.win_len) ? (float)au[s] * (1.0f / 32768.0f) : 0.0f;
    const double w = (i < a.win_len) ? a.window[i] : 0.0;
    buf[0][i] = make_double2(__dmul_rn((double)x, w), 0.0);
  }
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int ns = 1; ns < NFFT; ns <<= 1) {
    const int j = tid;
    const int k = j & (ns - 1);
    const double2 tw = a.twiddle[k * (HALF / ns)];  // (cos, -sin)(2 pi m / NFFT)
    const double2 u = buf[cur][j];
    const double2 v = buf[cur][j + HALF];
    const double vr = __dadd_rn(__dmul_rn(v.x, tw.x), -__dmul_rn(v.y, tw.y));
    const double vi = __dadd_rn(__dmul_rn(v.x, tw.y), __dmul_rn(v.y, tw.x));
    const int j0 = ((j - k) << 1) + k;
    buf[cur ^ 1][j0] = make_double2(__dadd_rn(u.x, vr), __dadd_rn(u.y, vi));
    buf[cur ^ 1][j0 + ns] = make_double2(__dadd_rn(u.x, -vr), __dadd_rn(u.y, -vi));
    cur ^= 1;
    __syncthreads();
  }
  for (int i = tid; i < NBIN; i += HALF) {
    const double2 z = buf[cur][i];
    const float p = (float)__dadd_rn(__dmul_rn(z.x, z.x), __dmul_rn(z.y, z.y));  // spectrogram output is float
    amp[i] = sqrt((double)p);
  }
  __syncthreads();
  if (tid < a.n_mel) {
    // bins whose lower band is tid-1 feed this band with (amp - amp*w); bins whose lower band is tid with amp*w
    double acc = 0.0;
    for (int i = a.mel_lo_begin[tid]; i < a.mel_lo_end[tid]; ++i) {
      const double s = amp[i];
      acc = __dadd_rn(acc, __dadd_rn(s, -__dmul_rn(s, a.mel_w[i])));
    }
    for (int i = a.mel_hi_begin[tid]; i < a.mel_hi_end[tid]; ++i) acc = __dadd_rn(acc, __dmul_rn(amp[i], a.mel_w[i]));
    lmel[tid] = log(acc < 1e-12 ? 1e-12 : acc);
  }
  __syncthreads();
  if (tid < a.n_coef) {
    double acc = 0.0;
    const double* row = a.dct + tid * a.n_mel;
    for (int j = 0; j < a.n_mel; ++j) acc = __dadd_rn(acc, __dmul_rn(row[j], lmel[j]));
    out[tid] = (float)acc;
  }
}

// context rows: x1[t*B+b][k] = feats[b][t - n_context + k/26][k%26] (zero outside [0, T_b)), k < 494; zero pad to 512
__global__ __launch_bounds__(256) void context_kernel(ContextArgs a) {
  const int row = blockIdx.x;  // (t - t0)*B + b
  const int tl = row / a.batch;
  const int b = row - tl * a.batch;
  const int t = a.t0 + tl;
  const int nf = a.n_frames[b];
  const int kw = a.n_coef * (2 * a.n_context + 1);
  for (int k = threadIdx.x; k < a.k_pad; k += blockDim.x) {
    float v = 0.0f;
    if (k < kw && t < nf) {
      const int j = k / a.n_coef;
      const int c = k - j * a.n_coef;
      const int ft = t - a.n_context + j;
      if (ft >= 0 && ft < nf) v = a.feats[((size_t)b * a.t_max + ft) * a.n_coef + c];
    }
    if (a.x1_f32) a.x1_f32[(size_t)row * a.k_pad + k] = v;
    else a.x1[(size_t)row * a.k_pad + k] = (_Float16)v;
  }
}

// =============================================================================================
// Dense layer: Y[M][N] = epi(X[M][K] . W[K][N] + bias) with W stored transposed WT[N][K].
// 128x128x64 tile, 4 waves (2 along N x 2 along M), each wave 4x4 tiles of mfma_f32_16x16x32_f16.
// The MFMA "A" operand is the weight tile (rows = output features), "B" is the activation tile
// (columns = rows of X), so every lane ends up with 4 *consecutive output features* of one X row
// and stores them with one 8-byte (f16) or 16-byte (f32) store.
//
// Staging: both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no staging registers,
// no ds_write pass), double buffered: the DMA of K-tile t+1 is in flight while tile t feeds the
// MFMAs, one barrier per K-tile.  The LDS image of a tile is [128 rows][8 slots of 16 B]; a DMA
// instruction writes wave-base + lane*16, i.e. 8 consecutive rows per wave-instruction, so the
// bank-conflict swizzle sits on the *source* side: slot s of row r holds K-chunk s ^ (r & 7)
// (still one 128-byte line per row), and a fragment read of chunk c goes to slot c ^ (r & 7).
// For the 16-lane groups that a ds_read_b128 is serviced in, the rows r..r+15 at one chunk then
// cover all 64 banks exactly once.
// =============================================================================================
#define GT_BK 64

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) const void gvoid_c;

// T = tile side in units of 64 (2: 128 x 128 tile, 4 waves -- round 1; 4: 256 x 256 tile, 16 waves).  Every wave owns a
// 64 x 64 piece either way (4 x 4 MFMA tiles); the bigger tile puts twice the MFMA work behind each DMA round trip and each
// barrier (the 128-square tile with one tile of prefetch was bound by the landing latency of its global_load_lds stage:
// 0.69 PF/s), and halves the operand bytes per flop.
// NS = LDS stages (K-tiles resident): 2 = one tile of prefetch behind a plain barrier (several workgroups per CU hide each
// other's landing latency); 3 = two tiles in flight with a counted vmcnt and a raw s_barrier (a __syncthreads() would drain
// the DMA queue) -- the shape for ONE workgroup per CU beside a recurrent-step workgroup (engine.cpp): 96 KiB of LDS, and the
// K-tile cadence no longer waits for a full HBM/L2 round trip.  Same k order: bit-identical results.
// W8: the 128-square tile on EIGHT waves (T = 2 only): wave tile 64 features x 32 rows, two waves per SIMD that hide each
// other's LDS and DMA latency where the four-wave solo form has nobody to switch to, <= 128 registers (launch bound) so that
// two of them fit on a SIMD beside a recurrent-step wave (240).  Same k order per output element: bit-identical.
template <int EPI, int T, int NS, bool W8>
__global__ __launch_bounds__(W8 ? 512 : T * T * 64, W8 ? 4 : 1) void dense_kernel(DenseArgs a) {
  constexpr int BM = 64 * T, BN = 64 * T, NTHR = W8 ? 512 : T * T * 64;
  constexpr int MJ = W8 ? 2 : 4;                 // 16-row X sub-tiles per wave
  static_assert(!W8 || T == 2, "the eight-wave form is the 128-square tile");
  constexpr int TILE_BYTES = BM * GT_BK * 2;     // one operand tile (BM == BN): rows of 128 bytes
  constexpr int RPI = NTHR / 8;                  // rows staged per DMA instruction of the whole workgroup
  constexpr int IT = BM / RPI;                   // DMA instructions per thread, operand and stage (4 for both shapes' ... see below)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];  // [buffer][W | X], 4 * TILE_BYTES
  lds_u8* const lds = (lds_u8*)lds_raw;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = W8 ? (wave >> 2) : wave / T, wm = W8 ? (wave & 3) : wave % T;
  const int wm_off = W8 ? wm * 32 : wm * 64;  // this wave's first X row inside the tile
  // XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own 4 MiB L2 that the
  // others cannot see, so an XCD that sweeps an M-band against ALL of N re-streams the whole weight matrix once per
  // M-tile (measured in round 1: 7.8x the algorithmic fetch on the 2048 -> 8192 projection).  Instead the tile grid is
  // cut into xa x xb rectangular blocks, one per XCD (launch_dense picks the cut that minimises operand bytes per XCD:
  // sum over XCDs of block rows + block columns), and inside a block the tiles are walked in strips of 8 N-tiles, M
  // fastest within 8 x 8 sub-blocks: the workgroups resident on an XCD at any time share activation row-tiles and weight
  // row-tiles and advance through K together, so each operand slice is fetched into that L2 about once.
  const int n_tiles_n = a.N / BN;
  const int n_tiles_m = (a.M + BM - 1) / BM;
  int tile_m, tile_n;
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int xm = xcd / a.xb, xn = xcd - xm * a.xb;
    const int Mx = (n_tiles_m + a.xa - 1) / a.xa, Nx = (n_tiles_n + a.xb - 1) / a.xb;
    const int m_lo = xm * Mx, n_lo = xn * Nx;
    const int m_cnt = min(Mx, n_tiles_m - m_lo), n_cnt = min(Nx, n_tiles_n - n_lo);
    if (m_cnt <= 0 || n_cnt <= 0 || idx >= m_cnt * n_cnt) return;  // (uniform per workgroup; before any barrier)
    const int sbn = min(8, n_cnt), per_strip = m_cnt * sbn, full = n_cnt / sbn;
    int strip, rem, w;
    if (idx < full * per_strip) { strip = idx / per_strip; rem = idx - strip * per_strip; w = sbn; }
    else { strip = full; rem = idx - full * per_strip; w = n_cnt - full * sbn; }
    // inside a strip: blocks of 8 M-tiles x w N-tiles, M fastest inside a block
    const int blk = rem / (8 * w), r2 = rem - blk * 8 * w;
    const int mh = min(8, m_cnt - blk * 8);     // rows of this block (the last one may be shorter)
    const int tn_l = r2 / mh, tm_l = r2 - tn_l * mh;
    tile_m = m_lo + blk * 8 + tm_l;
    tile_n = n_lo + strip * sbn + tn_l;
  }
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int K = a.K;

  f32x4 acc[4][MJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DMA sources: chunk q = i*NTHR + tid of a tile is LDS bytes [16q, 16q+16) = row q>>3, slot q&7 = K-chunk (q&7) ^ (row&7)
  const int srow = tid >> 3;                       // rows srow + RPI*i
  const int schunk = (tid & 7) ^ (srow & 7);       // (RPI is a multiple of 8: it does not change row & 7)
  const _Float16* wsrc[IT];
  const _Float16* xsrc[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    wsrc[i] = a.wt + (size_t)(n0 + srow + RPI * i) * K + schunk * 8;
    int mr = m0 + srow + RPI * i;
    mr = mr < a.M ? mr : a.M - 1;
    xsrc[i] = a.x + (size_t)mr * a.ldx + schunk * 8;
  }
  const unsigned wave_off = (unsigned)wave * 64 * 16;  // this wave's 1 KiB piece inside each (NTHR * 16)-byte group
#define STAGE(buf, k0)                                                                                                   \
  do {                                                                                                                   \
    lds_u8* const bw_ = lds + (buf) * 2 * TILE_BYTES + wave_off;                                                         \
    lds_u8* const bx_ = bw_ + TILE_BYTES;                                                                                \
    _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                                                     \
      __builtin_amdgcn_global_load_lds((gvoid_c*)(wsrc[i] + (k0)), (__attribute__((address_space(3))) void*)(bw_ + i * NTHR * 16), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gvoid_c*)(xsrc[i] + (k0)), (__attribute__((address_space(3))) void*)(bx_ + i * NTHR * 16), 16, 0, 0); \
    }                                                                                                                    \
  } while (0)

  // fragment read offsets (bytes inside an operand tile): row * 128 + ((chunk ^ (row & 7)) << 4), chunk = ks*4 + (lane>>4)
  const int frow = lane & 15, fq = lane >> 4;
  unsigned offw[4], offx[MJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rw = wn * 64 + i * 16 + frow;
    offw[i] = (unsigned)rw * 128u + (unsigned)((fq ^ (rw & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int rx = wm_off + j * 16 + frow;
    offx[j] = (unsigned)rx * 128u + (unsigned)((fq ^ (rx & 7)) << 4);
  }
  const int nk = K / GT_BK;
  STAGE(0, 0);
  if (NS == 3 && nk > 1) STAGE(1, GT_BK);
  if (NS == 2) __syncthreads();  // (waits for the DMA: vmcnt(0) + barrier)
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (NS == 2) {
      if (kt + 1 < nk) STAGE(cur ^ 1, (kt + 1) * GT_BK);
    } else {
      // tile kt has landed once at most the 2 * IT DMA instructions of tile kt + 1 are still outstanding (loads retire in
      // order); behind the barrier every wave's part of tile kt is there and every wave is done reading tile kt - 1, whose
      // buffer takes tile kt + 2
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 2 < nk) STAGE(cur >= 1 ? cur - 1 : 2, (kt + 2) * GT_BK);
    }
    const lds_u8* const bw = lds + cur * 2 * TILE_BYTES;
    const lds_u8* const bx = bw + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 fa[4], fb[MJ];
      // chunk ks*4 + fq: the slot index flips bit 2 for ks = 1 -> byte offset ^ 64
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(bw + (offw[i] ^ (unsigned)(ks << 6)));
#pragma unroll
      for (int j = 0; j < MJ; ++j) fb[j] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(bx + (offx[j] ^ (unsigned)(ks << 6)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (NS == 2) {
      __syncthreads();  // tile kt+1 has landed; every wave is done reading tile kt (its buffer is restaged next iteration)
      cur ^= 1;
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of tile kt are complete before it reaches the next barrier
      cur = cur == 2 ? 0 : cur + 1;
    }
  }
#undef STAGE
  // epilogue: lane holds features n = nb + (lane>>4)*4 + 0..3 of X row m = mb + (lane&15)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
    const float4 bias = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int m = m0 + wm_off + j * 16 + (lane & 15);
      if (m >= a.M) continue;
      float v0 = acc[i][j][0] + bias.x, v1 = acc[i][j][1] + bias.y, v2 = acc[i][j][2] + bias.z, v3 = acc[i][j][3] + bias.w;
      if (EPI == DENSE_EPI_RELU_F16) {
        // deepspeech_model.py:82-86: minimum(relu(x), relu_clip)
        v0 = fminf(fmaxf(v0, 0.f), a.relu_clip); v1 = fminf(fmaxf(v1, 0.f), a.relu_clip);
        v2 = fminf(fmaxf(v2, 0.f), a.relu_clip); v3 = fminf(fmaxf(v3, 0.f), a.relu_clip);
        f16x4 o = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.y) + (size_t)m * a.ldy + n) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (size_t)m * a.ldy + n) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

// The 128 (rows) x 256 (features) tile on EIGHT waves, two LDS stages of 48 KiB (96 KiB: one workgroup per CU, like the
// three-stage forms) -- the co-tenant of the 128-row recurrent step.  Every wave owns 64 features x 64 rows (4 x 4 MFMA tiles,
// 64 accumulator registers): per K-tile a SIMD's two waves issue 64 MFMAs behind ONE DMA round trip and one barrier, twice the
// 128-square eight-wave form's, and each operand byte staged feeds 1.33x the flops.  One K-tile of prefetch then covers the
// landing latency by itself (alone: 0.94 PF/s against 0.74; beside the recurrent step 0.62 against 0.51).  110 registers: two
// waves per SIMD fit beside the 128-row step's 288.  Same k order per output element as every other form: bit-identical results.
// NS = 4 (round 6, tunable dense_solo = 4): the same tile and the same 96 KiB as FOUR stages of 24 KiB -- K-tiles of 32 (rows of 64 bytes), three
// of them in flight behind a counted s_waitcnt and a raw s_barrier.  With one workgroup per CU nobody covers a K-tile's DMA round trip but
// the K-tiles before it: one tile of prefetch made a 64-deep K-tile cost max(1.1 k cycles of MFMA, the round trip) + a barrier -- ~2.6 k cycles
// measured alone.  Bank layout of a 64-byte row: slot s of row r holds K-chunk s ^ (((r >> 3) & 1) << 1) -- for the 16-lane groups a
// ds_read_b128 is serviced in ({0-3, 12-15, 20-27}, ...: rows 0-3 and 12-15 at chunk c together with rows 4-11 at chunk c + 1) the sixteen
// 16-byte slots of a 256-byte bank line are then all different (checked by enumeration).  Same k order per output element: bit-identical.
template <int EPI, int NS = 2>
__global__ __launch_bounds__(512, 2) void dense_wide_kernel(DenseArgs a) {
  constexpr int BM = 128, BN = 256, NTHR = 512, MJ = 4;
  constexpr int BK = NS == 4 ? 32 : GT_BK;        // K-tile depth in 2-byte units; a row of a staged tile is BK * 2 bytes
  constexpr int RB = BK * 2, CPR = RB / 16;       // row bytes, 16-byte chunks per row
  constexpr int TILE_W = BN * RB, TILE_X = BM * RB, STAGE_BYTES = TILE_W + TILE_X;  // 32 + 16 KiB (NS = 4: 16 + 8)
  constexpr int RPI = NTHR / CPR;                 // rows staged per DMA instruction of the whole workgroup (64; NS = 4: 128)
  constexpr int ITW = BN / RPI, ITX = BM / RPI;   // 4 + 2 DMA instructions per thread and stage (NS = 4: 2 + 1)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];  // [stage][W | X]
  lds_u8* const lds = (lds_u8*)lds_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wm = wave & 1;        // 4 waves along the features x 2 along the rows
  const int n_tiles_n = a.N / BN;
  const int n_tiles_m = (a.M + BM - 1) / BM;
  int tile_m, tile_n;
  {  // XCD-aware tile order: see dense_kernel
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int xm = xcd / a.xb, xn = xcd - xm * a.xb;
    const int Mx = (n_tiles_m + a.xa - 1) / a.xa, Nx = (n_tiles_n + a.xb - 1) / a.xb;
    const int m_lo = xm * Mx, n_lo = xn * Nx;
    const int m_cnt = min(Mx, n_tiles_m - m_lo), n_cnt = min(Nx, n_tiles_n - n_lo);
    if (m_cnt <= 0 || n_cnt <= 0 || idx >= m_cnt * n_cnt) return;
    const int sbn = min(8, n_cnt), per_strip = m_cnt * sbn, full = n_cnt / sbn;
    int strip, rem, w;
    if (idx < full * per_strip) { strip = idx / per_strip; rem = idx - strip * per_strip; w = sbn; }
    else { strip = full; rem = idx - full * per_strip; w = n_cnt - full * sbn; }
    const int blk = rem / (8 * w), r2 = rem - blk * 8 * w;
    const int mh = min(8, m_cnt - blk * 8);
    const int tn_l = r2 / mh, tm_l = r2 - tn_l * mh;
    tile_m = m_lo + blk * 8 + tm_l;
    tile_n = n_lo + strip * sbn + tn_l;
  }
  const int n0 = tile_n * BN, m0 = tile_m * BM;
  const int K = a.K;
  constexpr bool I8 = EPI >= DENSE_EPI_I8_F32;   // int8 operands: a row of a K-tile is the same 128 bytes (128 k-values instead of 64), K counts 2-byte units
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  typedef typename std::conditional<I8, i32x4_, f32x4>::type acc_t;
  acc_t acc[4][MJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};
  const int srow = tid / CPR;
  const int schunk = NS == 4 ? ((tid & 3) ^ (((srow >> 3) & 1) << 1)) : ((tid & 7) ^ (srow & 7));   // (RPI is a multiple of 16: it changes neither swizzle)
  const _Float16* wsrc[ITW];
  const _Float16* xsrc[ITX];
#pragma unroll
  for (int i = 0; i < ITW; ++i) wsrc[i] = a.wt + (size_t)(n0 + srow + RPI * i) * K + schunk * 8;
#pragma unroll
  for (int i = 0; i < ITX; ++i) {
    int mr = m0 + srow + RPI * i;
    mr = mr < a.M ? mr : a.M - 1;
    xsrc[i] = a.x + (size_t)mr * a.ldx + schunk * 8;
  }
  const unsigned wave_off = (unsigned)wave * 64 * 16;
#define STAGE_W(buf, k0)                                                                                                 \
  do {                                                                                                                   \
    lds_u8* const bw_ = lds + (buf) * STAGE_BYTES + wave_off;                                                            \
    lds_u8* const bx_ = bw_ + TILE_W;                                                                                    \
    _Pragma("unroll") for (int i = 0; i < ITW; ++i)                                                                      \
      __builtin_amdgcn_global_load_lds((gvoid_c*)(wsrc[i] + (k0)), (__attribute__((address_space(3))) void*)(bw_ + i * NTHR * 16), 16, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < ITX; ++i)                                                                      \
      __builtin_amdgcn_global_load_lds((gvoid_c*)(xsrc[i] + (k0)), (__attribute__((address_space(3))) void*)(bx_ + i * NTHR * 16), 16, 0, 0); \
  } while (0)
  const int frow = lane & 15, fq = lane >> 4;
  unsigned offw[4], offx[MJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rw = wn * 64 + i * 16 + frow;
    offw[i] = NS == 4 ? (unsigned)rw * 64u + (unsigned)((fq ^ (((rw >> 3) & 1) << 1)) << 4) : (unsigned)rw * 128u + (unsigned)((fq ^ (rw & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int rx = wm * 64 + j * 16 + frow;
    offx[j] = NS == 4 ? (unsigned)rx * 64u + (unsigned)((fq ^ (((rx >> 3) & 1) << 1)) << 4) : (unsigned)rx * 128u + (unsigned)((fq ^ (rx & 7)) << 4);
  }
  const int nk = K / BK;
  STAGE_W(0, 0);
  if (NS == 4) {
    if (nk > 1) STAGE_W(1, BK);
    if (nk > 2) STAGE_W(2, 2 * BK);
  } else __syncthreads();  // (waits for the DMA: vmcnt(0) + barrier)
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (NS == 4) {
      // tile kt has landed once at most the DMA instructions of the tiles issued after it (kt + 1, kt + 2) are outstanding (loads retire in
      // order); behind the barrier every wave's part of tile kt is there and every wave is done reading tile kt - 1, whose stage takes kt + 3
      const int after = nk - 1 - kt;
      if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (ITW + ITX)) : "memory");
      else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ITW + ITX) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 3 < nk) STAGE_W((cur + 3) & 3, (kt + 3) * BK);
    } else if (kt + 1 < nk) STAGE_W(cur ^ 1, (kt + 1) * BK);
    const lds_u8* const bw = lds + cur * STAGE_BYTES;
    const lds_u8* const bx = bw + TILE_W;
#pragma unroll
    for (int ks = 0; ks < RB / 64; ++ks) {
      f16x8 fa[4], fb[MJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(bw + (offw[i] ^ (unsigned)(ks << 6)));
#pragma unroll
      for (int j = 0; j < MJ; ++j) fb[j] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(bx + (offx[j] ^ (unsigned)(ks << 6)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          if constexpr (I8) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_, fa[i]), __builtin_bit_cast(i32x4_, fb[j]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    if (NS == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of tile kt are complete before it reaches the next barrier
      cur = (cur + 1) & 3;
    } else {
      __syncthreads();  // tile kt+1 has landed; every wave is done reading tile kt
      cur ^= 1;
    }
  }
#undef STAGE_W
  if constexpr (I8) {
    // y = bias + float(acc) * (row scale * weight scale): portable_tensor_utils.cc MatrixBatchVectorMultiplyAccumulate (int8), in its order,
    // every operation rounded on its own as the portable x86-64 build does (no fused multiply-add).  DENSE_EPI_I8_RELU_F32 adds the graph's
    // RELU + MINIMUM (deepspeech_model.py:82-86) on the f32 result; DENSE_EPI_I8_RAW hands out the int32 sums themselves (the x half of the
    // cell's product, rescaled inside the recurrent step together with the h half: lstm_i8_step_kernel).
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
      if constexpr (EPI == DENSE_EPI_I8_RAW) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int m = m0 + wm * 64 + j * 16 + (lane & 15);
          if (m >= a.M) continue;
          *reinterpret_cast<i32x4_*>(reinterpret_cast<int*>(a.y) + (size_t)m * a.ldy + n) = acc[i][j];
        }
      } else {
        const float4 bias = *reinterpret_cast<const float4*>(a.bias + n);
        float cs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[r] = a.col_scale[a.col_scale_n > 1 ? n + r : 0];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int m = m0 + wm * 64 + j * 16 + (lane & 15);
          if (m >= a.M) continue;
          const float rs = a.row_scale[m];
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float prod;   // (an instruction of its own: the compiler contracts a * b + c into v_fma_f32 whatever the pragma says for inlined operators)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"((float)acc[i][j][r]), "v"(__fmul_rn(rs, cs[r])));
            v[r] = (&bias.x)[r] + prod;
            if constexpr (EPI == DENSE_EPI_I8_RELU_F32) v[r] = fminf(fmaxf(v[r], 0.f), a.relu_clip);
          }
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (size_t)m * a.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
    const float4 bias = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int m = m0 + wm * 64 + j * 16 + (lane & 15);
      if (m >= a.M) continue;
      float v0 = acc[i][j][0] + bias.x, v1 = acc[i][j][1] + bias.y, v2 = acc[i][j][2] + bias.z, v3 = acc[i][j][3] + bias.w;
      if (EPI == DENSE_EPI_RELU_F16) {
        v0 = fminf(fmaxf(v0, 0.f), a.relu_clip); v1 = fminf(fmaxf(v1, 0.f), a.relu_clip);
        v2 = fminf(fmaxf(v2, 0.f), a.relu_clip); v3 = fminf(fmaxf(v3, 0.f), a.relu_clip);
        f16x4 o = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.y) + (size_t)m * a.ldy + n) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (size_t)m * a.ldy + n) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

// PortableSymmetricQuantizeFloats per row (tensorflow/lite/kernels/internal/reference/portable_tensor_utils.cc): range = max |x|;
// range == 0 -> zeros, scale 1; else q = clamp(round(x * (127 / range)), -127, 127) (std::round: half away from zero), scale = range / 127.
// One wave per row.
// The row is held in registers between the two passes when it fits (K <= 2048: eight float4 per lane), so the activations are read once.
__device__ __forceinline__ signed char quantize_one_(float v, float inv) { return (signed char)fminf(fmaxf(roundf(__fmul_rn(v, inv)), -127.0f), 127.0f); }
__global__ __launch_bounds__(256) void quantize_rows_kernel(const float* __restrict__ x, signed char* __restrict__ q, float* __restrict__ scale, float* __restrict__ range, int M, int K, int ldx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  signed char* qr = q + (size_t)row * K;
  float mx = 0.0f;
  if (K <= 2048) {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane * 4 + i * 256;
      v[i] = k < K ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float inv = mx > 0.0f ? __fdiv_rn(127.0f, mx) : 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane * 4 + i * 256;
      if (k >= K) break;
      char4 o;
      o.x = quantize_one_(v[i].x, inv); o.y = quantize_one_(v[i].y, inv); o.z = quantize_one_(v[i].z, inv); o.w = quantize_one_(v[i].w, inv);
      *reinterpret_cast<char4*>(qr + k) = o;
    }
  } else {
    for (int k = lane * 4; k < K; k += 256) { const float4 v = *reinterpret_cast<const float4*>(xr + k); mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float inv = mx > 0.0f ? __fdiv_rn(127.0f, mx) : 0.0f;
    for (int k = lane * 4; k < K; k += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + k);
      char4 o;
      o.x = quantize_one_(v.x, inv); o.y = quantize_one_(v.y, inv); o.z = quantize_one_(v.z, inv); o.w = quantize_one_(v.w, inv);
      *reinterpret_cast<char4*>(qr + k) = o;
    }
  }
  if (lane == 0) {
    scale[row] = mx > 0.0f ? __fdiv_rn(mx, 127.0f) : 1.0f;
    if (range) range[row] = mx;
  }
}
void launch_quantize_rows(const float* x, signed char* q, float* scale, int M, int K, hipStream_t st, float* range, int ldx) {
  if (K % 4 != 0) throw std::runtime_error("launch_quantize_rows: K must be a multiple of 4");
  hipLaunchKernelGGL(quantize_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, q, scale, range, M, K, ldx > 0 ? ldx : K);
}

// Skinny form for M <= 16 rows (one stream's 16-frame chunk: STT_FeedAudioContent / STT_SpeechToText): the tiled kernel
// would run N/128 workgroups through 32 barrier-separated K-tiles.  Here one wave owns 16 output features for the full K
// with operands straight from L2 into MFMA fragments (weight rows and x rows are both K-contiguous), eight k-steps of loads
// in flight, no LDS and no barrier; N/64 workgroups.  The accumulation runs over the same 32-deep k-blocks in the same
// order as the tiled kernel, so both give bit-identical results (streaming and batch paths agree exactly).
template <int EPI>
__global__ __launch_bounds__(256) void dense_skinny_kernel(DenseArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = (blockIdx.x * 4 + wave) * 16;
  if (nb >= a.N) return;
  const int K = a.K;
  int row = lane & 15;
  row = row < a.M ? row : a.M - 1;
  const uint4* wp = reinterpret_cast<const uint4*>(a.wt + (size_t)(nb + (lane & 15)) * K + (lane >> 4) * 8);
  const uint4* xp = reinterpret_cast<const uint4*>(a.x + (size_t)row * a.ldx + (lane >> 4) * 8);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int D = 8;  // k-steps in flight
  const int nks = K / 32;
  uint4 wa[D], xa[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nks) { wa[d] = wp[d * 4]; xa[d] = xp[d * 4]; }   // 32 halfs = 4 uint4 per k-step
  for (int s0 = 0; s0 < nks; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s0 + d < nks) {
        const f16x8 fa = *reinterpret_cast<f16x8*>(&wa[d]), fb = *reinterpret_cast<f16x8*>(&xa[d]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
        if (s0 + D + d < nks) { wa[d] = wp[(size_t)(s0 + D + d) * 4]; xa[d] = xp[(size_t)(s0 + D + d) * 4]; }
      }
    }
  }
  const int m = lane & 15;
  if (m >= a.M) return;
  const int n = nb + (lane >> 4) * 4;
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + n);
  float v0 = acc[0] + bias.x, v1 = acc[1] + bias.y, v2 = acc[2] + bias.z, v3 = acc[3] + bias.w;
  if (EPI == DENSE_EPI_RELU_F16) {
    v0 = fminf(fmaxf(v0, 0.f), a.relu_clip); v1 = fminf(fmaxf(v1, 0.f), a.relu_clip);
    v2 = fminf(fmaxf(v2, 0.f), a.relu_clip); v3 = fminf(fmaxf(v3, 0.f), a.relu_clip);
    f16x4 o = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.y) + (size_t)m * a.ldy + n) = o;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (size_t)m * a.ldy + n) = make_float4(v0, v1, v2, v3);
  }
}
template __global__ void dense_skinny_kernel<DENSE_EPI_RELU_F16>(DenseArgs);
template __global__ void dense_skinny_kernel<DENSE_EPI_BIAS_F32>(DenseArgs);

// The skinny form on int8 operands (a stream's 16-frame chunk through the hybrid path): v_mfma_i32_16x16x64_i8, one k-step = 64 int8 =
// the same 16 bytes per lane, so the addressing is the f16 form's (K and ldx count 2-byte units, as in dense_wide_kernel).  Integer
// sums are exact: whatever form computes them, the rescale sees the same int32.
template <int EPI>
__global__ __launch_bounds__(256) void dense_skinny_i8_kernel(DenseArgs a) {
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = (blockIdx.x * 4 + wave) * 16;
  if (nb >= a.N) return;
  const int K = a.K;
  int row = lane & 15;
  row = row < a.M ? row : a.M - 1;
  const uint4* wp = reinterpret_cast<const uint4*>(a.wt + (size_t)(nb + (lane & 15)) * K + (lane >> 4) * 8);
  const uint4* xp = reinterpret_cast<const uint4*>(a.x + (size_t)row * a.ldx + (lane >> 4) * 8);
  i32x4_ acc = (i32x4_){0, 0, 0, 0};
  constexpr int D = 8;
  const int nks = K / 32;   // k-steps of 64 int8 (= 32 two-byte units)
  uint4 wa[D], xa[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nks) { wa[d] = wp[d * 4]; xa[d] = xp[d * 4]; }
  for (int s0 = 0; s0 < nks; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s0 + d < nks) {
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_, wa[d]), __builtin_bit_cast(i32x4_, xa[d]), acc, 0, 0, 0);
        if (s0 + D + d < nks) { wa[d] = wp[(size_t)(s0 + D + d) * 4]; xa[d] = xp[(size_t)(s0 + D + d) * 4]; }
      }
    }
  }
  const int m = lane & 15;
  if (m >= a.M) return;
  const int n = nb + (lane >> 4) * 4;
  if constexpr (EPI == DENSE_EPI_I8_RAW) {
    *reinterpret_cast<i32x4_*>(reinterpret_cast<int*>(a.y) + (size_t)m * a.ldy + n) = acc;
  } else {
    const float4 bias = *reinterpret_cast<const float4*>(a.bias + n);
    const float rs = a.row_scale[m];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float prod;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"((float)acc[r]), "v"(__fmul_rn(rs, a.col_scale[a.col_scale_n > 1 ? n + r : 0])));
      v[r] = (&bias.x)[r] + prod;
      if constexpr (EPI == DENSE_EPI_I8_RELU_F32) v[r] = fminf(fmaxf(v[r], 0.f), a.relu_clip);
    }
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (size_t)m * a.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
}


// =============================================================================================
// LSTM recurrent step (deepspeech_model.py:144-168, tf LSTMCell semantics, forget_bias = 0):
//   z = xproj[t] + h_{t-1} . K[H:2H]        gates i, j, f, o
//   c' = sigmoid(f) * c + sigmoid(i) * tanh(j);   h' = sigmoid(o) * tanh(c')
// One workgroup owns 8 hidden units (32 gate columns = two 16-row MFMA tiles: [i|j] and [f|o]) for all
// batch rows; its 4 waves split K = H into quarters and reduce through LDS.  Weights are pre-packed so
// each wave streams its slice with fully coalesced 1 KiB loads; h is exchanged between steps in
// B-fragment order (hp), so the 256 KiB h read is coalesced too.  HBM/L2-bound: 33.5 MB of f16
// recurrent weights per step for H = 2048, shared by all batch rows.
// =============================================================================================
// (v_rcp_f32, 1 ulp, instead of the IEEE division sequence: the cell update of a 128-row step is ~40 quotients per lane, 1.3 us of
// VALU issue per step as divisions; every kernel form shares these helpers, so all forms still agree bit for bit)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = __fmul_rn(1.0f - e, __builtin_amdgcn_rcpf(1.0f + e));
  return copysignf(t, x);
}

// c' = sigmoid(f) * c + sigmoid(i) * tanh(j) with the roundings pinned: one product rounded, the other fused into the sum.  Left to
// the compiler's contraction the choice of WHICH product is fused follows the schedule of each instantiation, and a row
// decoded alone (one batch tile), in a 17-batch (two) and in the 64-batch (four; owner form) must give the same bits.
__device__ __forceinline__ float lstm_cell_(float zf, float c, float zi, float zj) {
  return __fmaf_rn(sigmoidf_(zf), c, __fmul_rn(sigmoidf_(zi), tanhf_(zj)));
}

// G = k-steps per prefetch group (two groups of weight / h fragments are in flight), MT = 16-row gate tiles per workgroup:
// MT = 2: 8 hidden units per workgroup, 256 workgroups (round 1); MT = 4: 16 units, 128 workgroups, each gate one tile.
// Every workgroup reads ALL of h (2 * H * 64 bytes at 64 batch rows = 256 KiB) besides its slice of the recurrent matrix, so
// with 256 workgroups the h re-reads (64 MiB per step) outweigh the weights (33.5 MiB); 128 workgroups halve them, and a
// workgroup then has a CU to itself (64 KiB reduction buffer, ~230 VGPRs: one wave per SIMD).
// MFMA with the accumulator tile pinned in the accumulator half of the register file ("a" constraint): left to itself the compiler
// keeps part of a 128-register accumulator block in VGPRs and shuttles it through AGPRs around the loads (864 v_accvgpr moves per
// step in the 128-row kernel: 19 us per step with NO operand traffic at all).  Pinned, the VGPR side only holds operands in flight.
// (The hazard recogniser does not look inside inline asm: lstm_acc_settle() pads the distance between the last MFMA and the
// first read of its result.)
template <bool PIN>
__device__ __forceinline__ void lstm_mfma(f32x4& acc, const f16x8& a, const f16x8& b) {
  if constexpr (PIN) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
template <bool PIN>
__device__ __forceinline__ void lstm_acc_settle() {
  if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // 32 wait states >= the 8-pass MFMA's result latency
}
// DBG (timing probes only, wrong results): bit 0 = every h fragment load reads the wave's first one (h served by the L1: what the
// step would cost if h were free), bit 1 = the same for the weight fragments (what it would cost if the weight stream were free),
// bit 2 = no cross-wave reduction (every wave settles its tile from its own partial sums: no parking in LDS, no barrier -- what a form
// that needs no reduction, e.g. waves that split the batch rows instead of K, could save at most).
template <int NT, int G_, int MT, int PHS, int DBG = 0, bool PIN = (NT == 8)>   // PIN: accumulators pinned in the accumulator file (lstm_mfma)
__device__ __forceinline__ void lstm_step_body(const LstmArgs& a) {
  constexpr bool PF = G_ > 0;
  constexpr int UPW = MT * 4;  // hidden units per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lstm_smem[];
  // The cross-wave reduction goes through LDS in PH passes over the batch tiles (two for 64 rows): 32 KiB instead of 64, so
  // that the step's workgroup fits beside a 96 KiB GEMM workgroup on the same CU (engine.cpp, three engines).  The sum of a
  // (row, unit) is red[0] + red[1] + red[2] + red[3] in every form: bit-identical results.
  constexpr int PH = (NT >= 4 && PHS == 2) ? 2 : 1, NTP = NT / PH;  // PHS = 2: the two-pass form (the batch path's three engines)
  typedef float RedT[MT][NTP][64][4];
  RedT* red = reinterpret_cast<RedT*>(lstm_smem);                                        // [4 waves]
  _Float16 (*hout)[UPW] = reinterpret_cast<_Float16 (*)[UPW]>(lstm_smem + 4 * sizeof(RedT));  // [NT * 16][UPW]
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int wg = blockIdx.x;
  // timing probe (STTX_TestLstmSteps with the tunable lstm_stamps): REFCLK (100 MHz) at entry / after the k-loop / behind the
  // reduction barrier / at exit, per (step, workgroup, wave)
  unsigned long long* const stamp = a.stamps ? a.stamps + (((size_t)a.stamp_step * gridDim.x + wg) * 4 + q) * 4 : nullptr;
#define LSTM_STAMP(i) do { if (stamp && lane == 0) stamp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
  LSTM_STAMP(0);
  const int H = a.n_hidden;
  const int ksteps = H / 128;  // 32-deep k-steps per wave
  const uint4* wp = reinterpret_cast<const uint4*>(a.whp) + ((size_t)(wg * 4 + q) * ksteps) * MT * 64 + lane;
  const uint4* hp = reinterpret_cast<const uint4*>(a.hp_in) + ((size_t)(q * ksteps) * NT) * 64 + lane;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // operands of the cell update, fetched now so that their latency hides behind the k-loop
  constexpr int CU_ITEMS = NT * 16 * UPW, CU_ITERS = (CU_ITEMS + 255) / 256;
  const int B = a.batch;
  // PHS = 3, the "owner" form (64 batch rows, 16 units per workgroup): wave q ends up with the complete sums of batch tile q
  // in the MFMA accumulator layout -- lane = (unit / 4) * 16 + row, four consecutive units x all four gates of ONE batch row per
  // lane -- so the cell update runs on registers.  Each wave parks the three tiles it does not own in LDS (48 KiB in all), one
  // barrier, each wave adds the three foreign partials of its own tile in wave order: (((r0 + r1) + r2) + r3), the order of
  // the other forms (bit-identical results), no second pass through LDS, no second barrier, h published from registers.
  // NT = 8 (two 64-utterance batches advanced by ONE step: the 33.5 MB matrix is streamed once per 128 rows): the owner form in
  // two rounds over the batch tiles -- round r settles tiles 4r .. 4r+3, wave q owning tile 4r + q -- through the same 48 KiB.
  constexpr bool OWN = PHS == 3 && (NT == 4 || NT == 8) && MT == 4;
  constexpr int ROUNDS = OWN ? NT / 4 : 1;
  const int ob = q * 16 + (lane & 15), ou = 4 * (lane >> 4);  // owner form: this lane's batch row (of round 0) and first unit
  float4 oxv[4] = {}, ocv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (OWN && !PIN && ob < B) {  // (pinned form: fetched in the tail of the k-loop, when the operand double buffer has registers to spare)
    const float* xp = a.xproj + ((size_t)a.t * B + ob) * (4 * H) + wg * UPW + ou;
#pragma unroll
    for (int g = 0; g < 4; ++g) oxv[g] = *reinterpret_cast<const float4*>(xp + (size_t)g * H);
    ocv = *reinterpret_cast<const float4*>(a.c + (size_t)ob * H + wg * UPW + ou);
  }
  float xv[CU_ITERS][4], cv[CU_ITERS];
#pragma unroll
  for (int it = 0; it < (OWN ? 0 : CU_ITERS); ++it) {
    const int p = tid + it * 256;
    const int b = p / UPW, u = p % UPW;
    xv[it][0] = xv[it][1] = xv[it][2] = xv[it][3] = 0.0f; cv[it] = 0.0f;
    if (p < CU_ITEMS && b < B) {
      const int unit = wg * UPW + u;
      const float* xp = a.xproj + ((size_t)a.t * B + b) * (4 * H) + unit;
      xv[it][0] = xp[0]; xv[it][1] = xp[H]; xv[it][2] = xp[2 * H]; xv[it][3] = xp[3 * H];
      cv[it] = a.c[(size_t)b * H + unit];
    }
  }
  // k-steps are taken G at a time with the next group's weight and h fragments already in flight (double buffered in
  // registers: one wave per SIMD, so the register file is ours): the kernel is bound by L2/HBM latency, not by MFMA issue.
  constexpr int G = G_ > 0 ? G_ : 1;
  if (PF && ksteps % (2 * G) == 0) {
    uint4 wa[G][MT], ha[G][NT], wb[G][MT], hb[G][NT];
#define LSTM_LOAD(W, Hh, s0)                                                                      \
  _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                 \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) W[g][i] = wp[(DBG & 2) ? (size_t)0 : (size_t)(((s0) + g) * MT + i) * 64]; \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) Hh[g][j] = hp[(DBG & 1) ? (size_t)0 : (size_t)(((s0) + g) * NT + j) * 64]; \
  }
#define LSTM_MMA(W, Hh)                                                                           \
  _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                 \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                              \
      const f16x8 fb = *reinterpret_cast<f16x8*>(&Hh[g][j]);                                      \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                              \
        lstm_mfma<PIN>(acc[i][j], *reinterpret_cast<f16x8*>(&W[g][i]), fb);                        \
    }                                                                                             \
  }
    if constexpr (!PIN) {
      LSTM_LOAD(wa, ha, 0);
      for (int s0 = 0; s0 < ksteps; s0 += 2 * G) {
        LSTM_LOAD(wb, hb, s0 + G);
        LSTM_MMA(wa, ha);
        if (s0 + 2 * G < ksteps) { LSTM_LOAD(wa, ha, s0 + 2 * G); }
        LSTM_MMA(wb, hb);
      }
    } else {
    // (the last pair of groups is peeled: a prefetch condition inside the loop leaves the waitcnt pass with a merge point it can only
    // resolve with vmcnt(0) -- one group in flight instead of two)
    LSTM_LOAD(wa, ha, 0);
    int s0 = 0;
    // (the asm MFMAs carry no scheduling model, and under the register budget the scheduler sinks each group's loads down to
    // their first use -- one group in flight again; a scheduling barrier behind every load group keeps the double buffer)
#define LSTM_FENCE() __builtin_amdgcn_sched_barrier(0)
    for (; s0 + 2 * G < ksteps; s0 += 2 * G) {
      LSTM_LOAD(wb, hb, s0 + G);
      LSTM_FENCE();
      LSTM_MMA(wa, ha);
      LSTM_FENCE();
      LSTM_LOAD(wa, ha, s0 + 2 * G);
      LSTM_FENCE();
      LSTM_MMA(wb, hb);
      LSTM_FENCE();
    }
    LSTM_LOAD(wb, hb, s0 + G);
    LSTM_FENCE();
    LSTM_MMA(wa, ha);
    LSTM_FENCE();
    if (OWN && ob < B) {  // `wa` / `ha` are free now: the cell-update operands of round 0 take their place in flight
      const float* xp = a.xproj + ((size_t)a.t * B + ob) * (4 * H) + wg * UPW + ou;
#pragma unroll
      for (int g = 0; g < 4; ++g) oxv[g] = *reinterpret_cast<const float4*>(xp + (size_t)g * H);
      ocv = *reinterpret_cast<const float4*>(a.c + (size_t)ob * H + wg * UPW + ou);
    }
    LSTM_FENCE();
    LSTM_MMA(wb, hb);
    }
#undef LSTM_FENCE
#undef LSTM_LOAD
#undef LSTM_MMA
  } else {
    for (int s = 0; s < ksteps; ++s) {
      uint4 w[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) w[i] = wp[(size_t)(s * MT + i) * 64];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        uint4 hv = hp[(size_t)(s * NT + j) * 64];
        f16x8 fb = *reinterpret_cast<f16x8*>(&hv);
#pragma unroll
        for (int i = 0; i < MT; ++i) lstm_mfma<PIN>(acc[i][j], *reinterpret_cast<f16x8*>(&w[i]), fb);
      }
    }
    if (OWN && PIN && ob < B) {  // (narrow models whose k-loop takes this branch: the cell-update operands of round 0)
      const float* xp = a.xproj + ((size_t)a.t * B + ob) * (4 * H) + wg * UPW + ou;
#pragma unroll
      for (int g = 0; g < 4; ++g) oxv[g] = *reinterpret_cast<const float4*>(xp + (size_t)g * H);
      ocv = *reinterpret_cast<const float4*>(a.c + (size_t)ob * H + wg * UPW + ou);
    }
  }
  lstm_acc_settle<PIN>();
  LSTM_STAMP(1);
  if (OWN) {
    typedef float OwnT[3][MT][64][4];                    // [writer wave][slot: the writer's foreign tiles in order][gate tile][lane]
    OwnT* park = reinterpret_cast<OwnT*>(lstm_smem);
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int obr = ob + rd * 64;                      // this lane's batch row in this round
      if (rd && !(DBG & 4)) __syncthreads();             // every wave has read the previous round's partials
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j == q || (DBG & 4)) continue;               // (wave-uniform)
        const int sl = j - (j > q ? 1 : 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(&park[q][sl][i][lane][0]) = acc[i][rd * 4 + j];
      }
      f32x4 own[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) own[i] = q == 0 ? acc[i][rd * 4 + 0] : q == 1 ? acc[i][rd * 4 + 1] : q == 2 ? acc[i][rd * 4 + 2] : acc[i][rd * 4 + 3];
      // the next round's cell-update operands travel while this round is settled
      float4 nxv[4] = {}, ncv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rd + 1 < ROUNDS && obr + 64 < B) {
        const float* xp = a.xproj + ((size_t)a.t * B + obr + 64) * (4 * H) + wg * UPW + ou;
#pragma unroll
        for (int g = 0; g < 4; ++g) nxv[g] = *reinterpret_cast<const float4*>(xp + (size_t)g * H);
        ncv = *reinterpret_cast<const float4*>(a.c + (size_t)(obr + 64) * H + wg * UPW + ou);
      }
      if (!(DBG & 4)) __syncthreads();
      if (rd == 0) LSTM_STAMP(2);
      f32x4 z[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        f32x4 v[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int sl = q - (q > w ? 1 : 0);            // where writer w parked tile q (unused for w == q)
          v[w] = (w == q || (DBG & 4)) ? own[i] : *reinterpret_cast<const f32x4*>(&park[w][sl < 3 ? sl : 2][i][lane][0]);
        }
        z[i] = ((v[0] + v[1]) + v[2]) + v[3];
      }
      float hv[4];
      float4 cn4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float zi = z[0][r] + (&oxv[0].x)[r], zj = z[1][r] + (&oxv[1].x)[r], zf = z[2][r] + (&oxv[2].x)[r], zo = z[3][r] + (&oxv[3].x)[r];
        const float cn = lstm_cell_(zf, (&ocv.x)[r], zi, zj);
        (&cn4.x)[r] = cn;
        hv[r] = obr < B ? __fmul_rn(sigmoidf_(zo), tanhf_(cn)) : 0.0f;
      }
      const int unit0 = wg * UPW + ou;
      if (obr < B) {
        *reinterpret_cast<float4*>(a.c + (size_t)obr * H + unit0) = cn4;
        if (a.h_f32) *reinterpret_cast<float4*>(a.h_f32 + (size_t)obr * H + unit0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
      }
      // publish h: this lane's four units are one half of a 16-byte (8-unit) chunk of the fragment-ordered buffer
      const f16x4 hh = {(_Float16)hv[0], (_Float16)hv[1], (_Float16)hv[2], (_Float16)hv[3]};
      const int k0 = wg * UPW + (ou & ~7);               // first unit of the chunk
      const int ksg = k0 >> 5, grp = (k0 & 31) >> 3, half = (ou >> 2) & 1;
      reinterpret_cast<f16x4*>(a.hp_out)[(((size_t)ksg * NT + rd * 4 + q) * 64 + grp * 16 + (lane & 15)) * 2 + half] = hh;
      if (obr < B) *reinterpret_cast<f16x4*>(a.h_all + ((size_t)a.t * B + obr) * H + unit0) = hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) oxv[g] = nxv[g];
      ocv = ncv;
    }
    LSTM_STAMP(3);
    return;
  }
  // cell update: (unit u, batch row b): gate row r = g*UPW+u lives in tile r>>4, lane group (r&15)>>2, reg r&3
  // (the x-projection and cell-state operands were fetched before the k-loop: xv/cv)
#pragma unroll
  for (int ph = 0; ph < PH; ++ph) {
    if (ph) __syncthreads();  // the previous pass's sums have been read
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTP; ++j) *reinterpret_cast<f32x4*>(&red[q][i][j][lane][0]) = acc[i][ph * NTP + j];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < CU_ITERS; ++it) {
      const int p = tid + it * 256;
      if (p >= CU_ITEMS) break;
      const int b = p / UPW, u = p % UPW;
      const int nt = b >> 4;
      if (nt / NTP != ph) continue;  // (uniform per `it` for the shapes in use: 256 items = 16 rows of 16 units, or 32 rows of 8)
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = g * UPW + u;
        const int mt = r >> 4, rr = r & 15;
        const int ln = ((rr >> 2) << 4) + (b & 15);
        const int ntl = nt - ph * NTP;
        z[g] = red[0][mt][ntl][ln][rr & 3] + red[1][mt][ntl][ln][rr & 3] + red[2][mt][ntl][ln][rr & 3] + red[3][mt][ntl][ln][rr & 3];
      }
      float hval = 0.0f;
      if (b < B) {
        const int unit = wg * UPW + u;
        const float zi = z[0] + xv[it][0], zj = z[1] + xv[it][1], zf = z[2] + xv[it][2], zo = z[3] + xv[it][3];
        const float cn = lstm_cell_(zf, cv[it], zi, zj);
        a.c[(size_t)b * H + unit] = cn;
        hval = __fmul_rn(sigmoidf_(zo), tanhf_(cn));
        if (a.h_f32) a.h_f32[(size_t)b * H + unit] = hval;
      }
      hout[b][u] = (_Float16)hval;
    }
  }
  __syncthreads();
  // publish h: 16-byte chunks (8 units) per batch row, into next step's fragment-ordered buffer and the time-major h_all
  for (int e = tid; e < NT * 16 * (UPW / 8); e += 256) {
    const int b = e / (UPW / 8), half = e % (UPW / 8);
    const uint4 v = *reinterpret_cast<const uint4*>(&hout[b][half * 8]);
    const int k0 = wg * UPW + half * 8;
    const int ksg = k0 >> 5, grp = (k0 & 31) >> 3;
    reinterpret_cast<uint4*>(a.hp_out)[((size_t)ksg * NT + (b >> 4)) * 64 + grp * 16 + (b & 15)] = v;
    if (b < B) *reinterpret_cast<uint4*>(a.h_all + ((size_t)a.t * B + b) * H + k0) = v;
  }
}

#undef LSTM_STAMP
template <int NT, int G_, int MT, int PHS>
__global__ __launch_bounds__(256, 2) void lstm_step_kernel(LstmArgs a) {  // (<= 256 registers per lane: two 128-register GEMM waves per SIMD fit beside it)
  lstm_step_body<NT, G_, MT, PHS>(a);
}
// 128 rows: 128 accumulator registers + the operand double buffer do not fit in 256; one wave per SIMD may take more, as long as
// two eight-wave GEMM waves (72 each) still fit in the SIMD's 512 beside it.
#ifndef LSTM8_VGPRS
#define LSTM8_VGPRS 160  /* budget = 2 x this: 312 registers allocated (248 + 64 accumulator-file), no spill */
#endif
template <int G_>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(LSTM8_VGPRS))) void lstm_step8_kernel(LstmArgs a) {
  lstm_step_body<8, G_, 4, 3>(a);
}
// Timing probes (STTX_TestLstmSteps with the tunable lstm_probe; benchmarks/lstm_micro.py): the shipped 64- and 128-row steps with
// one operand stream served by the L1, and the 128-row step with two k-steps of prefetch and no register cap (runs alone only).
#ifdef STT_TEST_HOOKS   // (libstt_test.so only: the shipped library carries neither the probe kernels nor their launcher)
template <int DBG>
__global__ __launch_bounds__(256, 2) void lstm_probe4_kernel(LstmArgs a) { lstm_step_body<4, 2, 4, 3, DBG>(a); }
template <int DBG>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(LSTM8_VGPRS))) void lstm_probe8_kernel(LstmArgs a) { lstm_step_body<8, 1, 4, 3, DBG>(a); }
template <int G_>  // the 64-row step with pinned accumulators (64 accumulator registers + operands: G = 2 -> ~150, G = 4 -> ~215 VGPRs)
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(192))) void lstm_probe4pin_kernel(LstmArgs a) { lstm_step_body<4, G_, 4, 3, 0, true>(a); }
template <int DBG>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(184))) void lstm_probe8g2_kernel(LstmArgs a) { lstm_step_body<8, 2, 4, 3, DBG>(a); }
#endif  // STT_TEST_HOOKS

// h (f32 [B][H]) -> fragment-ordered f16 hp (used once per chunk to seed the recurrence from a carried state)
__global__ void pack_h_kernel(const float* h, _Float16* hp, int B, int H, int NT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over NT*16 * H
  if (idx >= NT * 16 * H) return;
  const int b = idx / H, k = idx - b * H;
  const float v = b < B ? h[(size_t)b * H + k] : 0.0f;
  const int ksg = k >> 5, grp = (k & 31) >> 3, e = k & 7;
  hp[(((size_t)ksg * NT + (b >> 4)) * 64 + grp * 16 + (b & 15)) * 8 + e] = (_Float16)v;
}

// streaming path: the frame list already contains the explicit zero context frames, so window t is the
// contiguous slice frames[t*n_input .. t*n_input + kw) (stt.cc:292-309)
template <typename OutT>
__global__ void window_rows_kernel(const float* frames, OutT* x1, int rows_valid, int rows_total, int n_input, int kw, int kp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows_total * kp) return;
  const int t = idx / kp, k = idx - t * kp;
  x1[idx] = (OutT)((k < kw && t < rows_valid) ? frames[(size_t)t * n_input + k] : 0.0f);
}
template <typename OutT>
__global__ void window_rows_batch_kernel(const float* const* frames_ptrs, const int* win_off, const int* take, OutT* x1, int B, int T, int n_input, int kw, int kp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * B * kp) return;
  const int row = idx / kp, k = idx - row * kp;
  const int t = row / B, b = row - t * B;
  float v = 0.0f;
  if (k < kw && t < take[b]) v = frames_ptrs[b][(size_t)(win_off[b] + t) * n_input + k];
  x1[idx] = (OutT)v;
}
void launch_window_rows_batch(const float* const* frames_ptrs, const int* win_off, const int* take, void* x1, int B, int T, int n_input, int kw, int kp,
                              hipStream_t st, bool f32) {
  const int n = T * B * kp;
  if (n <= 0) return;
  if (f32) hipLaunchKernelGGL(window_rows_batch_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, frames_ptrs, win_off, take, (float*)x1, B, T, n_input, kw, kp);
  else hipLaunchKernelGGL(window_rows_batch_kernel<_Float16>, dim3((n + 255) / 256), dim3(256), 0, st, frames_ptrs, win_off, take, (_Float16*)x1, B, T, n_input, kw, kp);
}
__global__ void gather_rows_kernel(const float* const* src, const unsigned char* valid, float* dst, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, k = idx - b * H;
  dst[idx] = (valid[b] && src[b]) ? src[b][k] : 0.0f;
}
__global__ void scatter_rows_kernel(float* const* dst, const float* src, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, k = idx - b * H;
  dst[b][k] = src[idx];
}
void launch_gather_rows(const float* const* src, const unsigned char* valid, float* dst, int B, int H, hipStream_t st) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, src, valid, dst, B, H);
}
void launch_scatter_rows(float* const* dst, const float* src, int B, int H, hipStream_t st) {
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, dst, src, B, H);
}
void launch_window_rows(const float* frames, void* x1, int rows_valid, int rows_total, int n_input, int kw, int kp, hipStream_t st, bool f32) {
  const int n = rows_total * kp;
  if (n <= 0) return;
  if (f32) hipLaunchKernelGGL(window_rows_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, frames, (float*)x1, rows_valid, rows_total, n_input, kw, kp);
  else hipLaunchKernelGGL(window_rows_kernel<_Float16>, dim3((n + 255) / 256), dim3(256), 0, st, frames, (_Float16*)x1, rows_valid, rows_total, n_input, kw, kp);
}

// softmax over the first C of ldl logits per row (deepspeech_model.py:357); row m = t*B+b -> probs[b][t][:].
// Output layers wider than the fused kernel handles (C > 256, e.g. a 6000-label alphabet): one wave per row, the
// row read and written with consecutive lanes on consecutive classes.
__global__ __launch_bounds__(256) void softmax_kernel(SoftmaxArgs a) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= a.M) return;
  const float* l = a.logits + (size_t)m * a.ldl;
  const int t = m / a.batch, b = m - t * a.batch;
  float mx = -3.0e38f;
  for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, l[c]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if (a.exact) {
    // The int8 path is stated bit for bit against oracle/am_hybrid.py, whose softmax -- like its LOGISTIC / TANH -- takes the correctly rounded
    // float of the real function: e = float(exp(l - max)) evaluated in float64, the sum of those floats in float64 (any order: the sum of
    // <= 8191 floats is exact to 2^-40 of its value), p = float(e / sum).  TFLite's own float kernel (exp, a float sum, one division) is
    // within an ulp or two of it; which ulp is build-specific (tests/test_gpu_hybrid.py: the second checker).
    double sd = 0.0;
    for (int c = lane; c < a.C; c += 64) sd += (double)(float)exp((double)(l[c] - mx));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sd += __shfl_xor(sd, d);
    float* o = a.probs + ((size_t)b * a.t_max + t) * a.C;
    for (int c = lane; c < a.C; c += 64) o[c] = (float)((double)(float)exp((double)(l[c] - mx)) / sd);
    return;
  }
  float s = 0.f;
  for (int c = lane; c < a.C; c += 64) s += expf(l[c] - mx);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
  float* o = a.probs + ((size_t)b * a.t_max + t) * a.C;
  const float inv = 1.0f / s;
  for (int c = lane; c < a.C; c += 64) o[c] = expf(l[c] - mx) * inv;
}

// =============================================================================================
// Output layer + softmax fused (deepspeech_model.py:250-252, 357): probs = softmax(x . W6 + b6).
// The output layer is a *skinny* GEMM (N = 29 classes for English): with 128x128 tiles it occupied
// 24 workgroups for a whole K = 2048 sweep.  Here one workgroup takes 16 rows of x and ALL classes
// (CT tiles of 16): its 4 waves split K, operands go straight from L2 into MFMA fragments (x rows
// and W6^T rows are both K-contiguous), partial sums meet in LDS, and 16 threads per row finish the
// softmax.  M/16 workgroups (192 for a 48-frame chunk of 64 utterances).
// =============================================================================================
template <int CT>
__global__ __launch_bounds__(256) void logits_softmax_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ wt, const float* __restrict__ bias,
                                                              float* __restrict__ probs, int M, int K, int C, int batch, int t_max) {
  __shared__ __attribute__((aligned(16))) float red[4][CT][64][4];
  __shared__ float lg[16][CT * 16 + 1];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int m0 = blockIdx.x * 16;
  const int kq = K / 4;  // this wave's K range
  int row = m0 + (lane & 15);
  if (row >= M) row = M - 1;
  const _Float16* xp = x + (size_t)row * K + q * kq + (lane >> 4) * 8;
  const _Float16* wp = wt + (size_t)(lane & 15) * K + q * kq + (lane >> 4) * 8;
  f32x4 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < kq; k += 32) {
    const f16x8 fb = *reinterpret_cast<const f16x8*>(xp + k);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const f16x8 fa = *reinterpret_cast<const f16x8*>(wp + (size_t)c * 16 * K + k);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) *reinterpret_cast<f32x4*>(&red[q][c][lane][0]) = acc[c];
  __syncthreads();
  // logits: lane group g = lane >> 4 holds classes g*4 .. g*4+3 of tile c for batch row (lane & 15)
  for (int e = tid; e < CT * 64; e += 256) {
    const int c = e >> 6, l = e & 63;
    const int r = l & 15, cls0 = c * 16 + (l >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      lg[r][cls0 + j] = red[0][c][l][j] + red[1][c][l][j] + red[2][c][l][j] + red[3][c][l][j] + bias[cls0 + j];
  }
  __syncthreads();
  // softmax: 16 threads per row
  const int r = tid >> 4, t16 = tid & 15;
  float mx = -3.0e38f;
  for (int c = t16; c < C; c += 16) mx = fmaxf(mx, lg[r][c]);
#pragma unroll
  for (int d = 8; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 16));
  float sum = 0.f;
  for (int c = t16; c < C; c += 16) { const float e = expf(lg[r][c] - mx); lg[r][c] = e; sum += e; }
#pragma unroll
  for (int d = 8; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 16);
  const int m = m0 + r;
  if (m < M) {
    const int t = m / batch, b = m - t * batch;
    float* o = probs + ((size_t)b * t_max + t) * C;
    const float inv = 1.0f / sum;
    for (int c = t16; c < C; c += 16) o[c] = lg[r][c] * inv;
  }
}
bool launch_logits_softmax(const _Float16* x, const _Float16* wt, const float* bias, float* probs, int M, int K, int C, int batch, int t_max, hipStream_t st) {
  if (K % 128 != 0 || C > 256) return false;  // wider output layers take the generic dense + softmax path
  const dim3 grid((M + 15) / 16), block(256);
  const int ct = (C + 15) / 16;
  if (ct <= 2) hipLaunchKernelGGL(logits_softmax_kernel<2>, grid, block, 0, st, x, wt, bias, probs, M, K, C, batch, t_max);
  else if (ct <= 4) hipLaunchKernelGGL(logits_softmax_kernel<4>, grid, block, 0, st, x, wt, bias, probs, M, K, C, batch, t_max);
  else if (ct <= 8) hipLaunchKernelGGL(logits_softmax_kernel<8>, grid, block, 0, st, x, wt, bias, probs, M, K, C, batch, t_max);
  else hipLaunchKernelGGL(logits_softmax_kernel<16>, grid, block, 0, st, x, wt, bias, probs, M, K, C, batch, t_max);
  return true;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
// byte mover for small tables and result blocks (engine.h: copy_h2d / copy_d2h): 16-byte lanes when both ends allow
__global__ void copy_bytes_kernel(unsigned char* dst, const unsigned char* src, size_t bytes, int wide) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  if (wide) {
    const size_t nv = bytes / 16;
    for (size_t k = i; k < nv; k += n) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
    for (size_t k = nv * 16 + i; k < bytes; k += n) dst[k] = src[k];
  } else {
    for (size_t k = i; k < bytes; k += n) dst[k] = src[k];
  }
}
void launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (!bytes) return;
  const int wide = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
  const size_t items = wide ? (bytes + 15) / 16 : bytes;
  const int blocks = (int)std::min<size_t>(64, (items + 255) / 256);
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(blocks), dim3(256), 0, st, (unsigned char*)dst, (const unsigned char*)src, bytes, wide);
}
// ---- which streams share a dispatch pipe?  (engine.cpp: place_engine_streams; measured: benchmarks/pipe_probe.hip, profiles/r06_pipe_probe.txt)
// A compute pipe of the command processor works on one dispatch at a time: a launch of more workgroups than the chip holds keeps its pipe
// until the last one is placed, and every other queue on that pipe waits.  HIP streams land on pipes round-robin in creation order (stream
// i and stream i + 4 share), so whether the recurrence's 250 short dependent launches per batch sit behind the GEMM engine's dispatches is
// an accident of what the process created before.  The probe makes it a measurement: a "hog" dispatch on one stream (8192 workgroups that
// sleep 200 us, LDS sized so that wave slots stay free on every CU), two one-wave launches on each candidate, events around both.
__global__ __launch_bounds__(256) void placement_hog_kernel(unsigned* sink, int ticks) {
  __shared__ unsigned lds[10240];   // 40 KiB: four workgroups per CU, 16 of its 32 wave slots
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[threadIdx.x] == 0xffffffffu) sink[0] = 1;
}
__global__ void placement_tick_kernel(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
// (8192 workgroups, 1024 resident at a time, 200 us each: the dispatch holds its pipe for ~1.6 ms -- several times what the host needs to issue
// the candidates' chains behind it, so "finished before the hog's dispatch ended" separates the pipes cleanly)
void launch_placement_hog(unsigned* scratch, hipStream_t st) { hipLaunchKernelGGL(placement_hog_kernel, dim3(8192), dim3(256), 0, st, scratch, 20000); }
void launch_placement_tick(unsigned* scratch, hipStream_t st) { hipLaunchKernelGGL(placement_tick_kernel, dim3(1), dim3(64), 0, st, scratch); }

void launch_dense_hybrid_i8(const signed char* q, const float* row_scale, const signed char* wq, const float* col_scale, int col_scale_n, const float* bias, void* y,
                            int M, int N, int K, hipStream_t st, int epi, float relu_clip, int ldy) {
  if (K % 128 != 0 || M < 1) throw std::runtime_error("launch_dense_hybrid_i8: K must be a multiple of 128");
  DenseArgs b{};
  b.wt = reinterpret_cast<const _Float16*>(wq); b.x = reinterpret_cast<const _Float16*>(q); b.bias = bias; b.y = y;
  b.M = M; b.N = N; b.K = K / 2; b.ldx = K / 2; b.ldy = ldy > 0 ? ldy : N;   // (K and ldx in 2-byte units: the tile loops of the f16 form, unchanged)
  b.row_scale = row_scale; b.col_scale = col_scale; b.col_scale_n = col_scale_n; b.relu_clip = relu_clip;
  if (M <= 16 && N % 64 == 0) {   // one stream's chunk: a wave per 16 output features, no LDS
    const dim3 grid(N / 64), block(256);
    switch (epi) {
      case DENSE_EPI_I8_RELU_F32: hipLaunchKernelGGL(dense_skinny_i8_kernel<DENSE_EPI_I8_RELU_F32>, grid, block, 0, st, b); break;
      case DENSE_EPI_I8_RAW: hipLaunchKernelGGL(dense_skinny_i8_kernel<DENSE_EPI_I8_RAW>, grid, block, 0, st, b); break;
      default: hipLaunchKernelGGL(dense_skinny_i8_kernel<DENSE_EPI_I8_F32>, grid, block, 0, st, b); break;
    }
    return;
  }
  if (N % 256 != 0) throw std::runtime_error("launch_dense_hybrid_i8: N must be a multiple of 256 (64 for M <= 16)");
  const int ntn = N / 256, ntm = (M + 127) / 128;
  int best = 1 << 30;
  b.xa = 1; b.xb = 8;
  for (int xa = 1; xa <= 8; xa *= 2) {
    const int xb = 8 / xa;
    const int Mx = (ntm + xa - 1) / xa, Nx = (ntn + xb - 1) / xb;
    const int cost = Mx + 2 * Nx;
    if (cost < best) { best = cost; b.xa = xa; b.xb = xb; }
  }
  const int per_xcd = ((ntm + b.xa - 1) / b.xa) * ((ntn + b.xb - 1) / b.xb);
  const size_t smem = 96 * 1024;
  static std::once_flag once[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 15], [&]() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_I8_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_I8_RELU_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_I8_RAW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  if (tune().dense_solo >= 4) {   // four stages of 24 KiB (K-tiles of 64 int8)
    static std::once_flag once4[16];
    std::call_once(once4[dev & 15], [&]() {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_I8_F32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_I8_RELU_F32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_I8_RAW, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    switch (epi) {
      case DENSE_EPI_I8_RELU_F32: hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_I8_RELU_F32, 4>), dim3(8 * per_xcd), dim3(512), smem, st, b); break;
      case DENSE_EPI_I8_RAW: hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_I8_RAW, 4>), dim3(8 * per_xcd), dim3(512), smem, st, b); break;
      default: hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_I8_F32, 4>), dim3(8 * per_xcd), dim3(512), smem, st, b); break;
    }
    return;
  }
  switch (epi) {
    case DENSE_EPI_I8_RELU_F32: hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_I8_RELU_F32>), dim3(8 * per_xcd), dim3(512), smem, st, b); break;
    case DENSE_EPI_I8_RAW: hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_I8_RAW>), dim3(8 * per_xcd), dim3(512), smem, st, b); break;
    default: hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_I8_F32>), dim3(8 * per_xcd), dim3(512), smem, st, b); break;
  }
}
template <int NFFT>
static void launch_mfcc_inst(const MfccArgs& a, int n_frames_total, hipStream_t st) {
  const size_t smem = (size_t)2 * NFFT * sizeof(double2) + (size_t)(NFFT / 2 + 2 + 40) * sizeof(double);
  static std::once_flag once;
  std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_kernel<NFFT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
  hipLaunchKernelGGL(mfcc_kernel<NFFT>, dim3(n_frames_total), dim3(NFFT / 2), smem, st, a);
}
void launch_mfcc(const MfccArgs& a, int n_frames_total, hipStream_t st) {
  switch (a.fft_len) {
    case 128: launch_mfcc_inst<128>(a, n_frames_total, st); break;
    case 256: launch_mfcc_inst<256>(a, n_frames_total, st); break;
    case 512: launch_mfcc_inst<512>(a, n_frames_total, st); break;
    case 1024: launch_mfcc_inst<1024>(a, n_frames_total, st); break;
    case 2048: launch_mfcc_inst<2048>(a, n_frames_total, st); break;      // 32 ms at 44.1 / 48 kHz (1411 / 1536 samples): 1024 threads
    default: throw std::runtime_error("launch_mfcc: unsupported FFT length");
  }
}
void launch_context(const ContextArgs& a, int rows, hipStream_t st) {
  hipLaunchKernelGGL(context_kernel, dim3(rows), dim3(256), 0, st, a);
}
template <int EPI, int T, int NS, bool W8 = false>
static void launch_dense_inst(const DenseArgs& b, int grid, hipStream_t st) {
  size_t smem = (size_t)2 * NS * (64 * T) * GT_BK * 2;  // stages x (W tile + X tile): 64 / 96 KiB (T = 2) or 128 KiB (T = 4)
  static std::once_flag once[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 15], [&]() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_kernel<EPI, T, NS, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if ((size_t)b.lds_floor > smem) smem = std::min<size_t>((size_t)b.lds_floor, 160 * 1024);
  hipLaunchKernelGGL((dense_kernel<EPI, T, NS, W8>), dim3(grid), dim3(W8 ? 512 : T * T * 64), smem, st, b);
}
void launch_dense(const DenseArgs& a, int epi, hipStream_t st) {
  if (a.M <= 16 && a.K % 32 == 0 && a.N % 64 == 0) {
    if (epi == DENSE_EPI_RELU_F16) hipLaunchKernelGGL(dense_skinny_kernel<DENSE_EPI_RELU_F16>, dim3(a.N / 64), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dense_skinny_kernel<DENSE_EPI_BIAS_F32>, dim3(a.N / 64), dim3(256), 0, st, a);
    return;
  }
  // 256-square tiles when the shape allows (N a multiple of 256 and enough rows to fill them), else 128-square
  // (measured: with 96 to 384 tiles for 256 CUs the 256-square tile is slower than 1536 small ones; from about two tiles per
  // CU on it wins -- the long chunks of the pipelined batch path.  Tunables dense_tile = 128 / 256 force a side,
  // dense_big_min moves the threshold)
  const int big_ok = tune().dense_tile, big_min = tune().dense_big_min;
  if (a.solo >= 3 && a.N % 256 == 0 && a.M >= 128) {  // the 128 x 256 eight-wave co-tenant form
    DenseArgs b = a;
    const int ntn = a.N / 256, ntm = (a.M + 127) / 128;
    int best = 1 << 30;
    b.xa = 1; b.xb = 8;
    for (int xa = 1; xa <= 8; xa *= 2) {
      const int xb = 8 / xa;
      const int Mx = (ntm + xa - 1) / xa, Nx = (ntn + xb - 1) / xb;
      const int cost = Mx + 2 * Nx;  // (a block column is a 256-row weight slice: twice the bytes of a block row)
      if (cost < best || (cost == best && Mx * Nx < ((ntm + b.xa - 1) / b.xa) * ((ntn + b.xb - 1) / b.xb))) { best = cost; b.xa = xa; b.xb = xb; }
    }
    const int per_xcd = ((ntm + b.xa - 1) / b.xa) * ((ntn + b.xb - 1) / b.xb);
    const size_t smem = 96 * 1024;
    static std::once_flag once[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 15], [&]() {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_RELU_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_BIAS_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    if (a.solo >= 4 && a.K % 32 == 0) {   // four stages of 24 KiB
      static std::once_flag once4[16];
      std::call_once(once4[dev & 15], [&]() {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_RELU_F16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wide_kernel<DENSE_EPI_BIAS_F32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      });
      if (epi == DENSE_EPI_RELU_F16) hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_RELU_F16, 4>), dim3(8 * per_xcd), dim3(512), smem, st, b);
      else hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_BIAS_F32, 4>), dim3(8 * per_xcd), dim3(512), smem, st, b);
      return;
    }
    if (epi == DENSE_EPI_RELU_F16) hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_RELU_F16>), dim3(8 * per_xcd), dim3(512), smem, st, b);
    else hipLaunchKernelGGL((dense_wide_kernel<DENSE_EPI_BIAS_F32>), dim3(8 * per_xcd), dim3(512), smem, st, b);
    return;
  }
  const bool big_fits = a.N % 256 == 0 && a.M >= 256;
  const bool big = big_fits && !a.solo && (big_ok >= 256 || (big_ok == 0 && ((a.M + 255) / 256) * (a.N / 256) >= big_min));
  const int side = big ? 256 : 128;
  // cut of the tile grid over the 8 XCDs: xa x xb blocks, minimising (block rows + block columns) = operand bytes per XCD
  DenseArgs b = a;
  const int ntn = a.N / side, ntm = (a.M + side - 1) / side;
  int best = 1 << 30;
  b.xa = 1; b.xb = 8;
  for (int xa = 1; xa <= 8; xa *= 2) {
    const int xb = 8 / xa;
    const int Mx = (ntm + xa - 1) / xa, Nx = (ntn + xb - 1) / xb;
    const int cost = Mx + Nx;
    if (cost < best || (cost == best && Mx * Nx < ((ntm + b.xa - 1) / b.xa) * ((ntn + b.xb - 1) / b.xb))) { best = cost; b.xa = xa; b.xb = xb; }
  }
  const int per_xcd = ((ntm + b.xa - 1) / b.xa) * ((ntn + b.xb - 1) / b.xb);
  if (big) {
    if (epi == DENSE_EPI_RELU_F16) launch_dense_inst<DENSE_EPI_RELU_F16, 4, 2>(b, 8 * per_xcd, st);
    else launch_dense_inst<DENSE_EPI_BIAS_F32, 4, 2>(b, 8 * per_xcd, st);
  } else if (a.solo >= 2) {
    if (epi == DENSE_EPI_RELU_F16) launch_dense_inst<DENSE_EPI_RELU_F16, 2, 3, true>(b, 8 * per_xcd, st);
    else launch_dense_inst<DENSE_EPI_BIAS_F32, 2, 3, true>(b, 8 * per_xcd, st);
  } else if (a.solo) {
    if (epi == DENSE_EPI_RELU_F16) launch_dense_inst<DENSE_EPI_RELU_F16, 2, 3>(b, 8 * per_xcd, st);
    else launch_dense_inst<DENSE_EPI_BIAS_F32, 2, 3>(b, 8 * per_xcd, st);
  } else {
    if (epi == DENSE_EPI_RELU_F16) launch_dense_inst<DENSE_EPI_RELU_F16, 2, 2>(b, 8 * per_xcd, st);
    else launch_dense_inst<DENSE_EPI_BIAS_F32, 2, 2>(b, 8 * per_xcd, st);
  }
}
// batch tiles per launch: up to 64 rows in every shape; 65 .. 128 rows (two batches advanced together, NT = 8) with 16 units per workgroup
int lstm_nt_for_batch(int B) { return B <= 16 ? 1 : B <= 32 ? 2 : B <= 64 ? 4 : B <= 128 ? 8 : -1; }
int lstm_max_rows(int H) { return lstm_units_per_wg(H) == 16 ? 128 : 64; }
// hidden units per workgroup of the recurrent kernel (and of the weight packing, pack_lstm_recurrent_host): 16 when the width
// allows, 8 otherwise (tunable lstm_upw = 8 keeps round 1's shape for A/B runs; it must not change while a model is loaded)
int lstm_units_per_wg(int H) { return (tune().lstm_upw >= 16 && H % 16 == 0) ? 16 : 8; }
template <int NT, int G, int MT, int PHS>
static void launch_lstm_inst2(const LstmArgs& a, hipStream_t st) {
  constexpr bool OWN = PHS == 3 && (NT == 4 || NT == 8) && MT == 4;
  const size_t smem = OWN ? (size_t)4 * 3 * MT * 64 * 16
                          : 4 * sizeof(float) * MT * ((NT >= 4 && PHS == 2) ? NT / 2 : NT) * 64 * 4 + (size_t)NT * 16 * MT * 4 * 2;
  const void* fn;
  if constexpr (NT == 8) fn = reinterpret_cast<const void*>(lstm_step8_kernel<G>);
  else fn = reinterpret_cast<const void*>(lstm_step_kernel<NT, G, MT, PHS>);
  static std::once_flag once[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 15], [&]() { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
  if constexpr (NT == 8) hipLaunchKernelGGL((lstm_step8_kernel<G>), dim3(a.n_hidden / (MT * 4)), dim3(256), smem, st, a);
  else hipLaunchKernelGGL((lstm_step_kernel<NT, G, MT, PHS>), dim3(a.n_hidden / (MT * 4)), dim3(256), smem, st, a);
}
template <int NT, int G, int MT>
static void launch_lstm_inst(const LstmArgs& a, hipStream_t st) {
  // a.passes: 0 / 1 = one pass (the fastest form when the step has the CU to itself: 2.88 ms per 250 steps against 2.95 for the
  // owner form), 2 = two passes, 3 = owner form (what fits beside a GEMM workgroup: the batch path's three engines).
  // The tunable lstm_form overrides every caller (A/B runs, form-vs-form tests).
  const int form = tune().lstm_form ? tune().lstm_form : (a.passes ? a.passes : 1);
  if constexpr (NT == 8) {
    launch_lstm_inst2<NT, G, MT, 3>(a, st);  // 128 rows: the owner form in two rounds is the only one that fits in LDS
  } else {
    if (NT == 4 && MT == 4 && form == 3) launch_lstm_inst2<NT, G, MT, (NT == 4 && MT == 4 ? 3 : 1)>(a, st);
    else if (NT >= 4 && form == 2) launch_lstm_inst2<NT, G, MT, (NT >= 4 ? 2 : 1)>(a, st);
    else launch_lstm_inst2<NT, G, MT, 1>(a, st);
  }
}
#ifdef STT_TEST_HOOKS
static bool launch_lstm_probe(const LstmArgs& a, int NT, hipStream_t st) {
  const int pr = tune().lstm_probe;  // 0 = off; 1..3 = DBG bits on the shipped kernel of this row count; 10 + DBG = the 128-row step with G = 2
  if (!pr || (NT != 4 && NT != 8) || lstm_units_per_wg(a.n_hidden) != 16) return false;
  const size_t smem = (size_t)4 * 3 * 4 * 64 * 16;
  const dim3 grid(a.n_hidden / 16), block(256);
#define PROBE(K) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); hipLaunchKernelGGL((K), grid, block, smem, st, a); return true; } while (0)
  if (NT == 4) {
    switch (pr) {
      case 1: PROBE(lstm_probe4_kernel<1>); case 2: PROBE(lstm_probe4_kernel<2>); case 3: PROBE(lstm_probe4_kernel<3>);
      case 20: PROBE(lstm_probe4pin_kernel<1>); case 21: PROBE(lstm_probe4pin_kernel<2>); case 22: PROBE(lstm_probe4pin_kernel<4>);
      default: return false;
    }
  }
  switch (pr) {
    case 1: PROBE(lstm_probe8_kernel<1>); case 2: PROBE(lstm_probe8_kernel<2>); case 3: PROBE(lstm_probe8_kernel<3>);
    case 4: PROBE(lstm_probe8_kernel<4>); case 7: PROBE(lstm_probe8_kernel<7>);
    case 10: PROBE(lstm_probe8g2_kernel<0>); case 11: PROBE(lstm_probe8g2_kernel<1>); case 12: PROBE(lstm_probe8g2_kernel<2>); case 13: PROBE(lstm_probe8g2_kernel<3>);
    default: return false;
  }
#undef PROBE
}
#endif  // STT_TEST_HOOKS
void launch_lstm_step(const LstmArgs& a, int NT, hipStream_t st) {
#ifdef STT_TEST_HOOKS
  if (a.probe && launch_lstm_probe(a, NT, st)) return;
#endif
  const int pg = tune().lstm_prefetch;
  if (lstm_units_per_wg(a.n_hidden) == 16) {
    switch (NT) {
      case 1: launch_lstm_inst<1, 4, 4>(a, st); break;
      case 2: launch_lstm_inst<2, 4, 4>(a, st); break;
      case 8: launch_lstm_inst<8, 1, 4>(a, st); break;
      default: if (pg >= 4) launch_lstm_inst<4, 4, 4>(a, st); else if (pg >= 2) launch_lstm_inst<4, 2, 4>(a, st); else launch_lstm_inst<4, 1, 4>(a, st); break;
    }
    return;
  }
  if (NT == 8) throw std::runtime_error("lstm step: 128-row batches need 16 hidden units per workgroup");
  switch (NT) {
    case 1: launch_lstm_inst<1, 4, 2>(a, st); break;
    case 2: launch_lstm_inst<2, 4, 2>(a, st); break;
    default:
      if (pg >= 4) launch_lstm_inst<4, 4, 2>(a, st);
      else if (pg >= 2) launch_lstm_inst<4, 2, 2>(a, st);
      else if (pg == 1) launch_lstm_inst<4, 1, 2>(a, st);
      else launch_lstm_inst<4, 0, 2>(a, st);
      break;
  }
}
void launch_pack_h(const float* h, void* hp, int B, int H, int NT, hipStream_t st) {
  const int n = NT * 16 * H;
  hipLaunchKernelGGL(pack_h_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h, reinterpret_cast<_Float16*>(hp), B, H, NT);
}
void launch_softmax(const SoftmaxArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(softmax_kernel, dim3((a.M + 3) / 4), dim3(256), 0, st, a);
}
