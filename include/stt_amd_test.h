/* stt_amd_test.h -- test hooks of the MI355X engine: NOT part of the shipped libstt.so.
 *
 * stt_amd/build.py links two libraries from the same objects: libstt.so (coqui-stt.h + stt_amd.h, nothing else) and libstt_test.so, which
 * adds the entry points below (sources compiled with -DSTT_TEST_HOOKS: api.cpp, fleet.cpp, and the timing-probe instantiations of the
 * recurrent step in kernels_am.hip / kernels_i8.hip).  They reach single kernels and internal state so that tests/ can compare each stage
 * with oracle/ -- a binding never needs them.  stt_amd/native.py loads libstt_test.so when STT_AMD_TEST_HOOKS=1 (tests/conftest.py sets it). */
#ifndef STT_AMD_TEST_H
#define STT_AMD_TEST_H

#include "stt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug / parity hook: the acoustic probabilities of a submitted, not yet collected batch exactly as the pipelined path computed
 * them (three engines, graph-replayed recurrence, ring slots, 64 or 128 rows per recurrent step) -- the block that batch's beam
 * search reads; stands in for the `logits` output of TFLiteModelState::infer (tflitemodelstate.cc:369-405) over the whole
 * utterance.  aProbs [aBatch][aMaxFrames][n_classes], aNumFrames[i] = frames of utterance i.  Waits for the batch. */
STTX_EXPORT int STTX_DebugBatchProbs(ModelState* aCtx, int aTicket, float* aProbs, unsigned int aMaxFrames, unsigned int* aNumFrames);
/* Test hooks.  STTX_TestFleetRecords (host only): pack / pad / concatenate / unpack the transcript records exactly as the fleet's two
 * all-gathers move them -- aTexts[i] decoded by shard aShardOf[i] of aShards -- and return the strings in the caller's order
 * (STTX_FreeStrings), NULL on a malformed record.  STTX_DebugFleetFailShard: the next fleet batch fails on that shard before it
 * decodes (-1 = off); the call must then return NULL on every device's thread instead of waiting inside a collective. */
STTX_EXPORT char** STTX_TestFleetRecords(const char* const* aTexts, const unsigned int* aShardOf, unsigned int aCount, unsigned int aShards);
STTX_EXPORT int STTX_DebugFleetFailShard(STTX_Fleet* aFleet, int aShard);
/* y = epi(x[M][K] . w[K][N] + bias): runs the MFMA dense kernel (f16 operands, f32 accumulate).  aEpilogue 0 = clipped
 * ReLU (y rounded to f16, returned as f32), 1 = bias only (f32). */
STTX_EXPORT int STTX_TestDense(int aM, int aN, int aK, const float* aX, const float* aW, const float* aBias, float aClip,
                              int aEpilogue, float* aY);
/* Test hook: TensorFlow Lite's hybrid FULLY_CONNECTED (what the reference's CPU path runs for the released, dynamic-range quantised
 * models: tflitemodelstate.cc:200, tensorflow/lite/kernels/fully_connected.cc EvalHybrid) on the int8 matrix cores -- every row of aX
 * (f32 [aM][aK]) quantised with its own scale max|x| / 127, int8 x int8 -> int32, aY = aBias + float(sum) * (row scale x weight scale);
 * aWq int8 [aN][aK], aWScale [aNScales = 1 or aN].  aQ / aRowScale (optional): the quantised rows and their scales.  aReps > 0: that many
 * timed repetitions, *aElapsedMs per repetition.  aK a multiple of 128, aN of 256 (of 64 for aM <= 16: the skinny form).  aEpi: 0 = bias only,
 * 1 = + the graph's clipped ReLU (aClip), as layers 1-3 and 5 run it.  The model path built from these kernels: STTX_GetAcousticMode. */
STTX_EXPORT int STTX_TestDenseHybrid(const float* aX, unsigned int aM, unsigned int aK, const signed char* aWq, const float* aWScale, unsigned int aNScales,
                                      const float* aBias, unsigned int aN, float* aY, signed char* aQ, float* aRowScale, unsigned int aReps, float* aElapsedMs,
                                      int aEpi, float aClip);
/* Test hook (int8 path): rows that took the recurrent step's slow path (max |h| above max |x_t|: both halves again at the joint scale) since the
 * model was created, over every engine form; waits for everything in flight.  0 rows on a float model. */
STTX_EXPORT int STTX_DebugSlowRows(ModelState* aCtx, unsigned int* aRows);
/* Test hook for the int8 path (aCtx must be in mode 1): aWindows f32 [aT * aB][19 x 26] context windows, row = t * aB + b, through layers
 * 1-3, the cell (state aC / aH [aB][n_hidden] f32, NULL = zeros), layers 5-6 and the softmax as ONE call of the engine's one-stream path.
 * Outputs (each may be NULL): aL3 [aT*aB][n_hidden] layer 3's f32 rows; aAccX [aT*aB][4 n_hidden] the x half of the cell's int32 sums;
 * aHAll [aT*aB][n_hidden] h_t; aLogits [aT*aB][n_classes]; aProbs [aB][aT][n_classes]; aNewC / aNewH [aB][n_hidden];
 * *aSlowRows = rows the recurrent steps computed again at the joint scale during this call (max |h| > max |x_t|); *aLstmMs (may be NULL) =
 * HIP-event time of the recurrence alone (the prep launch + aT step launches). */
STTX_EXPORT int STTX_TestHybridChain(ModelState* aCtx, const float* aWindows, unsigned int aB, unsigned int aT, const float* aC, const float* aH,
                                      float* aL3, int* aAccX, float* aHAll, float* aLogits, float* aProbs, float* aNewC, float* aNewH, unsigned int* aSlowRows,
                                      float* aLstmMs);
/* The recurrent step kernel alone (deepspeech_model.py:144-168, one LSTMCell step per launch) on the model's packed recurrent
 * matrix: aSteps steps from a zero state, step t adding x-projection block t % aPeriod (aXproj [aPeriod * aBatch][4 * n_hidden] f32,
 * row = block * aBatch + b).  aC, aH [aBatch][n_hidden]: the final state; aHAll (may be NULL) [aPeriod * aBatch][n_hidden] f16 bits:
 * h of the last aPeriod steps.  aGraph != 0: the launches are captured into one hipGraph and replayed (as the batch path does).
 * The kernel form is chosen with STTX_SetTuning("lstm_form" / "lstm_prefetch"); every form must give the same bits.
 * aElapsedMs (may be NULL): HIP-event time of the aSteps launches (or of the one graph launch). */
STTX_EXPORT int STTX_TestLstmSteps(ModelState* aCtx, unsigned int aBatch, unsigned int aSteps, unsigned int aPeriod, int aGraph,
                                  const float* aXproj, float* aC, float* aH, unsigned short* aHAll, float* aElapsedMs);
/* Device expf/logf/log_sum_exp of sttmath.h over arrays (aOp 0 = expf, 1 = logf, 2 = log_sum_exp(a, b)). */
STTX_EXPORT int STTX_TestMath(int aOp, const float* aA, const float* aB, float* aOut, unsigned int aCount);
/* Host only, test hook: label sequences walked through the dictionary tables of a scorer package as the engine parses them (the
 * minimised automaton of the package, or its unfolding into a tree -- tunable dict_tree_mb).  See stt_amd/csrc/api.cpp. */
STTX_EXPORT int STTX_TestDictionaryWalk(const char* aScorer, unsigned int aScorerBytes, int aSpaceLabel, const int* aLabels,
                                        unsigned int aNumSeq, unsigned int aLen, int* aOut);
/* KenLM FullScore (kenlm/lm/model.cc:170-176) over aNumWords words, the state carried from BeginSentence (aBos) or the null
 * context, on a bare KenLM trie binary: aProbs[i] = log10 probability, aLens[i] = matched n-gram length of word i
 * (lm::FullScoreReturn).  aMode 0 = the hashed n-gram index on the host (no GPU needed), 1 = the device trie walk,
 * 2 = the device index lookup (one lane per query, as in the search kernel). */
STTX_EXPORT int STTX_TestLm(const char* aLm, unsigned int aLmBytes, const char* const* aWords, unsigned int aNumWords, int aBos,
                           int aMode, float* aProbs, int* aLens);
/* Test hook: decoder arenas are sized for aFrames timesteps and never grow (0 = normal sizing), so that the overflow
 * reporting of the decode calls can be exercised. */
STTX_EXPORT int STTX_DebugLimitArena(int aFrames);

#ifdef __cplusplus
}
#endif
#endif /* STT_AMD_TEST_H */
