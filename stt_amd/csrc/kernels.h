// stt_amd/csrc/kernels.h -- argument blocks and launchers of the HIP kernels (internal header;
// the public boundary is include/coqui-stt.h + include/stt_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- features -----------------------------------------------------------------------------------
struct MfccArgs {
  const int16_t* audio;    // [rows][n_max]
  const int* rows;         // [B] row of utterance b in `audio` (null: row b)
  const int* n_samples;    // [B]
  const int* n_frames;     // [B]
  float* feats;            // [B][t_max][n_coef]
  float* const* feats_ptrs;  // optional: per-utterance output base (frames of utterance b go to feats_ptrs[b] + f*n_coef; frames >= n_frames[b] are not written)
  int n_max, t_max;
  int all_n_samples, all_n_frames;  // used for every utterance when n_samples / n_frames are null (one stream: no table upload)
  int win_len, win_step, n_coef, n_mel;
  int fft_len;             // NextPowerOfTwo(win_len): 128, 256, 512 or 1024
  const double* window;    // [win_len]
  const double2* twiddle;  // [fft_len / 2] (cos, -sin)(2 pi m / fft_len)
  const double* mel_w;     // [fft_len / 2 + 1]
  const int *mel_lo_begin, *mel_lo_end, *mel_hi_begin, *mel_hi_end;  // [n_mel] bin ranges
  const double* dct;       // [n_coef][n_mel]
};
struct ContextArgs {
  const float* feats;   // [B][t_max][n_coef]
  const int* n_frames;  // [B]
  _Float16* x1;         // [t_max*B][k_pad], row = t*B + b
  float* x1_f32;        // when set: the same rows as f32 (x1 unused) -- input of the int8 path's row quantisation
  int batch, t_max, n_coef, n_context, k_pad;
  int t0;               // first timestep of this launch (rows are local: row = (t - t0)*B + b)
};

// ---- dense --------------------------------------------------------------------------------------
enum { DENSE_EPI_RELU_F16 = 0, DENSE_EPI_BIAS_F32 = 1,
       // TFLite's hybrid FULLY_CONNECTED (launch_dense_hybrid_i8): f32 result / + clipped ReLU / the raw int32 sums
       DENSE_EPI_I8_F32 = 2, DENSE_EPI_I8_RELU_F32 = 3, DENSE_EPI_I8_RAW = 4 };
struct DenseArgs {
  const _Float16* wt;  // [N][K]
  const _Float16* x;   // [M][ldx]
  const float* bias;   // [N]
  void* y;             // f16 or f32 [M][ldy]
  int M, N, K, ldx, ldy;
  float relu_clip;
  int xa, xb;          // tile-grid cut over the 8 XCDs (xa blocks along M x xb along N; filled in by launch_dense)
  int lds_floor;       // ask for at least this much dynamic LDS (bytes; 0 = what the tile needs): > 80 KiB keeps the kernel at ONE
                       // workgroup per CU, so that a recurrent-step workgroup launched beside it always finds room (engine.cpp)
  const float* row_scale;  // DENSE_EPI_I8_F32: the activation rows' scaling factors [M] (quantize_rows) ...
  const float* col_scale;  // ... and the weight scale(s): [N] per output row, or [1] per tensor (col_scale_n)
  int col_scale_n;
  int solo;            // 1: the three-stage form of the 128-square tile (96 KiB: one workgroup per CU, two K-tiles in flight); 2: the same on eight waves;
                       // 3: the 128 x 256 eight-wave tile, two stages (96 KiB), where the shape allows (else as 2)
};

// TFLite's hybrid FULLY_CONNECTED (tensorflow/lite/kernels/fully_connected.cc EvalHybrid, restated in oracle/am_hybrid.py): every row of the
// f32 input quantised to int8 with its own scale max|x| / 127 (PortableSymmetricQuantizeFloats), int8 x int8 -> int32 dot products on
// v_mfma_i32_16x16x64_i8, y = bias + float(acc) * (row scale * weight scale).  The integer sums are exact, so the result differs from the
// reference CPU kernel only where float rounding of the rescale could (it cannot: same operations, same order).  x f32 [M][K] -> q int8
// [M][K] + scale f32 [M]; wq int8 [N][K]; y f32 [M][N].  K a multiple of 128; N a multiple of 256 (the 128 x 256 tile) or, for M <= 16
// rows, of 64 (the skinny form: one wave per 16 output features, operands straight from L2).  Since round 5 the released models' own
// arithmetic is a MODEL PATH (engine.cpp: acoustic_rows_i8): epi = DENSE_EPI_I8_F32 (bias only: the output layer), DENSE_EPI_I8_RELU_F32
// (layers 1-3, 5) or DENSE_EPI_I8_RAW (y = the int32 sums, no rescale: the x half of the cell's product).
// `range` (optional, [M]): max |x| of every row -- the cell's joint [x_t, h] quantisation compares it with max |h|.
// ldx: row stride of x in floats (>= K); the quantised rows are dense [M][K].
void launch_quantize_rows(const float* x, signed char* q, float* scale, int M, int K, hipStream_t st, float* range = nullptr, int ldx = 0);
void launch_dense_hybrid_i8(const signed char* q, const float* row_scale, const signed char* wq, const float* col_scale, int col_scale_n, const float* bias, void* y,
                            int M, int N, int K, hipStream_t st, int epi = DENSE_EPI_I8_F32, float relu_clip = 0.0f, int ldy = 0);

// ---- LSTM ---------------------------------------------------------------------------------------
struct LstmArgs {
  const _Float16* whp;    // packed recurrent weights, see pack_lstm_recurrent()
  const _Float16* hp_in;  // h_{t-1}, fragment order [H/32][NT][64][8]
  _Float16* hp_out;       // h_t, same order
  const float* xproj;     // [t_max*B][4H], row = t*B + b
  float* c;               // [B][H] cell state, updated in place
  float* h_f32;           // optional [B][H] copy of h_t in f32 (state hand-back); may be null
  _Float16* h_all;        // [t_max*B][H]
  int n_hidden, batch, t;
  int passes;             // form of the cross-wave reduction: 0 = default, 1 = one pass (64 KiB of LDS), 2 = two passes over the batch
                          // tiles (32 KiB), 3 = owner form (48 KiB, one barrier, cell update on registers); 2, 3: 64-row batches only
  int prio;               // wave priority (s_setprio 0..3) of the step's waves: beats age when other kernels' waves share the SIMDs
  unsigned long long* stamps;  // STTX_TestLstmSteps only: [steps][workgroups][4 waves][4] REFCLK stamps (null: none)
  int stamp_step;
  int probe;              // STTX_TestLstmSteps only: honour the tunable lstm_probe (timing probes with wrong results; never set by the engine)
};

// The recurrent step in the released models' own arithmetic (TFLite's hybrid FULLY_CONNECTED over concat([x_t, h_(t-1)]), oracle/am_hybrid.py:
// HybridModel.forward_batch): ONE scale per row for both halves, max(max |x_t|, max |h_(t-1)|) / 127.  |h| < 1 always, so whenever
// max |x_t| >= max |h_(t-1)| the joint scale is the x half's own and (a) the x half of the int32 sums is hoisted out of the recurrence
// (one int8 GEMM per chunk, DENSE_EPI_I8_RAW), (b) the step that PRODUCES h_(t-1) can quantise it for its consumer at once, with
// 127 / max |x_t| -- a speculation that is checked, not assumed: every workgroup compares its units' max |h| with the row's max |x_t| and
// flags the row when it is larger; a flagged row is computed again by the consuming step from the f32 x_t and h_(t-1) at the true joint
// scale (plain int8 dot products, a few microseconds, rare: layer 3's clipped ReLU over 2048 units is nearly always >= 1).
struct LstmI8Args {
  const signed char* whp;      // recurrent half of the kernel, int8, packed per (workgroup, k-step, gate tile, lane): pack_lstm_recurrent_i8_host()
  const signed char* hq_in;    // h_(t-1) quantised at the row's scale, MFMA B-fragment order [H/64][NT][64][16]
  signed char* hq_out;         // h_t, quantised for step t+1 (not written by the last step of a launch sequence)
  const int* accx;             // [T*B][4H] int32: x half of the dot products (row = t*B + b)
  const float* bias;           // [4H]
  const float* wscale;         // the kernel's scale(s): [1] per tensor or [4H] per output row
  int wscale_n;
  const float* xscale;         // [T*B] scaling factor of row (t, b)'s own quantisation: range / 127 (1 for an all-zero row)
  const float* xrange;         // [T*B] max |x_t| of row (t, b)
  float* c;                    // [B][H] cell state, updated in place
  float* h_all;                // [T*B][H] f32: h_t of every step (layer 5 quantises it row by row)
  float* h_last;               // [B][H] f32: h of the LAST step (t == T-1): the carried state
  float* pmax;                 // [2][NT*16][H/16]: max |h| of every (row, workgroup) of the previous / this step
  int* flag;                   // [2][NT*16]: flag[t & 1][row] == t + 1  <=>  row needs the slow path at step t
  const float* y3;             // [T*B][H] f32 layer-3 output (slow path only)
  const float* h_prev0;        // [B][H] f32: h before step 0 (slow path only; written by the prep kernel)
  const signed char* wxq;      // [4H][H] int8: x half of the kernel, rows = gate columns (the GEMM's operand; slow path only)
  const signed char* whq;      // [4H][H] int8: h half, same orientation (slow path only)
  float* zslow;                // [H/16][NT*16][64] f32 scratch of the slow path
  int n_hidden, batch, t, T;
  int prio;
  // Batch path only (null otherwise): frames of every row's utterance and the absolute time of step 0 of this launch sequence.  A row
  // beyond its utterance's end (t0 + t >= row_frames[row]) runs on zero windows whose results nobody reads: it is never flagged for the
  // slow path (a short utterance in a group of longer ones would otherwise pay ~7 us per step for nothing).  A STREAM's zero-padded steps
  // are NOT such rows: they perturb the carried state exactly as the reference's do (stt.cc:236-254) and are computed in full.
  const int* row_frames;
  int t0;
  unsigned int* slow_count;    // counts slow-path rows (diagnostics / tests; may be null)
  int probe;                   // STTX_TestHybridChain only: the tunable lstm_probe (timing probes with wrong results; never set by the batch / streaming paths)
};
// before step 0 of a launch sequence: h_src ([B][H] f32, null = zeros) -> hq (buffer of step 0), h_prev0, pmax / flag of step 0
void launch_lstm_i8_prep(const LstmI8Args& a, const float* h_src, int NT, hipStream_t st);
void launch_lstm_i8_step(const LstmI8Args& a, int NT, hipStream_t st, int rows_per_wg = 0);   // rows_per_wg: 16 / 32 / 64 below NT * 16 = that many rows per workgroup, NT * 16 / rows_per_wg workgroups per 16-unit slice
size_t lstm_i8_hq_bytes(int H, int NT);

struct SoftmaxArgs {
  const float* logits;  // [M][ldl]
  float* probs;         // [B][t_max][C]
  int M, C, ldl, batch, t_max;
  int exact;            // int8 path: every probability the correctly rounded float of exp(l - max) / sum (float64 evaluation, one rounding each)
};

// dst/src: device-addressable (HBM or mapped page-locked host memory); any size, any alignment
void launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t st);
void launch_placement_hog(unsigned* scratch, hipStream_t st);    // a dispatch that keeps its pipe for ~0.2 ms (engine.cpp: place_engine_streams)
void launch_placement_tick(unsigned* scratch, hipStream_t st);   // one wave, one atomic
void launch_mfcc(const MfccArgs& a, int n_frames_total, hipStream_t st);
void launch_context(const ContextArgs& a, int rows, hipStream_t st);   // a.x1_f32: rows as f32 (the int8 path quantises them itself)
void launch_dense(const DenseArgs& a, int epi, hipStream_t st);
int lstm_nt_for_batch(int B);
int lstm_max_rows(int H);      // rows one recurrent launch covers: 128 with 16 units per workgroup, else 64
int lstm_units_per_wg(int H);  // 16 or 8: shape of the recurrent kernel AND of the packed recurrent matrix
void launch_lstm_step(const LstmArgs& a, int NT, hipStream_t st);
void launch_pack_h(const float* h, void* hp, int B, int H, int NT, hipStream_t st);
void launch_softmax(const SoftmaxArgs& a, hipStream_t st);
// fused output layer + softmax for C <= 256 (returns false if the shape is not covered: caller uses dense + softmax)
bool launch_logits_softmax(const _Float16* x, const _Float16* wt, const float* bias, float* probs, int M, int K, int C, int batch, int t_max,
                           hipStream_t st);
// rows_valid windows are gathered, rows rows_valid .. rows_total-1 are written as zeros (the padded steps of a partial chunk)
void launch_window_rows(const float* frames, void* x1, int rows_valid, int rows_total, int n_input, int kw, int kp, hipStream_t st, bool f32 = false);
// batched streaming: window t of stream b = frames_ptrs[b][(win_off[b] + t) * n_input ...], rows t >= take[b] are zero; row = t*B + b
void launch_window_rows_batch(const float* const* frames_ptrs, const int* win_off, const int* take, void* x1, int B, int T, int n_input, int kw, int kp,
                              hipStream_t st, bool f32 = false);
// LSTM state of stream b <-> row b of a [B][H] matrix (src null or valid[b] == 0: zeros)
void launch_gather_rows(const float* const* src, const unsigned char* valid, float* dst, int B, int H, hipStream_t st);
void launch_scatter_rows(float* const* dst, const float* src, int B, int H, hipStream_t st);
