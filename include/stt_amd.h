/* include/stt_amd.h -- C-ABI additions of the MI355X engine next to coqui-stt.h.
 *
 * coqui-stt.h is batch-1 and host-buffer only (one utterance per call, like the reference).  A GPU
 * wants many utterances per launch, device-resident audio and stage-level entry points for parity
 * tests.  Everything here is additive: a binding that only knows coqui-stt.h keeps working.
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Which reference interface each entry point stands in for:
 *   STTX_SpeechToTextBatch*      evaluate_export.py:25-80 (one Model.stt() per utterance in worker processes);
 *                                semantically == calling STT_SpeechToText (stt.cc:655-662) on every element
 *   STTX_ComputeMfcc             ModelState::compute_mfcc (modelstate.h:36-42, tflitemodelstate.cc:407-436),
 *                                applied to every window of an utterance with stt.cc's framing
 *   STTX_AcousticProbs           the feedAudioContent -> infer chain (stt.cc:105-334) for a batch, probs only
 *   STTX_InferChunk              ModelState::infer (modelstate.h:44-54, tflitemodelstate.cc:369-405)
 *   STTX_Decoder*                DecoderState::{init,next,decode} (ctc_beam_search_decoder.h:14-87) and
 *                                ctc_beam_search_decoder_batch (ctc_beam_search_decoder.cpp:608-652)
 */
#ifndef STT_AMD_H
#define STT_AMD_H

#include "coqui-stt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STTX_EXPORT __attribute__((visibility("default")))

/* Selects the HIP device used by models created afterwards in this process (one process per GPU). */
STTX_EXPORT int STTX_SetDevice(int aDevice);
/* Number of visible HIP devices, or a negative STT error if the runtime is unusable. */
STTX_EXPORT int STTX_GetDeviceCount(void);

/* ---- batch recognition ---------------------------------------------------------------------- */
/* aBuffers[i] holds aBufferSizes[i] 16-bit mono samples (host memory).  Returns aBatch malloc'd strings
 * (array freed with STTX_FreeStrings) or NULL on error. */
STTX_EXPORT char** STTX_SpeechToTextBatch(ModelState* aCtx, const short* const* aBuffers, const unsigned int* aBufferSizes,
                                         unsigned int aBatch);
STTX_EXPORT Metadata** STTX_SpeechToTextBatchWithMetadata(ModelState* aCtx, const short* const* aBuffers,
                                                         const unsigned int* aBufferSizes, unsigned int aBatch,
                                                         unsigned int aNumResults);
/* Audio already resident in HBM: aDeviceAudio is a device pointer to [aBatch][aStride] int16 (row i holds
 * aBufferSizes[i] valid samples).  This is the entry point bench.py times. */
STTX_EXPORT char** STTX_SpeechToTextBatchDevice(ModelState* aCtx, const short* aDeviceAudio, unsigned int aStride,
                                               const unsigned int* aBufferSizes, unsigned int aBatch);
/* The same path as a pipeline driven by the caller (the reference's harness keeps its workers busy the same way:
 * evaluate_export.py:65-80 feeds a queue while results are collected).  Submit enqueues one batch of 1..64 utterances
 * (audio resident in HBM, as above; the array must stay valid until the batch is collected) and returns a ticket >= 0 without
 * waiting (negative: -STT_ERR_*); at most STTX_BatchPipelineDepthFor() batches may be in flight.  Collect waits for that batch
 * and returns its aCount transcripts (STTX_FreeStrings) in the order submitted, or NULL on failure.  The acoustic models of
 * the groups in flight run one after the other, each group's beam search beside the acoustic model and the searches of its
 * neighbours.  Where the recurrent kernel covers 128 rows, two consecutive submits form ONE group (the recurrent matrix is
 * streamed once per step for both): the first of the two is enqueued together with the second -- or alone, when it is collected
 * first.  Transcripts never depend on how batches were grouped.  Collect batches in the order they were submitted: slots are taken
 * round-robin, and a submit whose slot still holds an uncollected batch is refused ("the pipeline is full") even if another slot is
 * free.  A first half of a pair starts no GPU work until its partner is submitted or it is collected. */
STTX_EXPORT int STTX_BatchPipelineDepth(void);
/* ... for THIS model as configured now: a search-bound setup (code-point scorer, beam width beyond 512) takes four slots and
 * runs their searches side by side, everything else two. */
STTX_EXPORT int STTX_BatchPipelineDepthFor(ModelState* aCtx);
/* The same pipeline fed with HOST buffers, as every call of coqui-stt.h is (const short* aBuffer, unsigned int aBufferSize:
 * native_client/coqui-stt.h:294-297, stt.cc:641-688): the aBatch utterances (1..64) are gathered into page-locked memory, copied to HBM on a
 * queue of their own and enqueued behind that copy; returns the ticket (STTX_BatchCollect*), negative on error.  The buffers may be
 * reused as soon as the call returns.  This is the entry bench.py times; STTX_BatchSubmitDevice is for callers whose audio is already in HBM. */
STTX_EXPORT int STTX_BatchSubmit(ModelState* aCtx, const short* const* aBuffers, const unsigned int* aBufferSizes, unsigned int aBatch);
STTX_EXPORT int STTX_BatchSubmitDevice(ModelState* aCtx, const short* aDeviceAudio, unsigned int aStride,
                                       const unsigned int* aBufferSizes, unsigned int aBatch);
STTX_EXPORT char** STTX_BatchCollect(ModelState* aCtx, int aTicket, unsigned int* aCount);
/* ... the same with the best transcript's Metadata (tokens, timesteps, confidence; STT_SpeechToTextWithMetadata with
 * aNumResults = 1, stt.cc:349-365) per utterance instead of its string: free with STTX_FreeMetadataArray. */
STTX_EXPORT Metadata** STTX_BatchCollectWithMetadata(ModelState* aCtx, int aTicket, unsigned int* aCount);
/* ... the strings of STTX_BatchCollect plus, in aConfidence[0 .. *aCount) (room for 64), the best transcript's confidence
 * (CandidateTranscript::confidence, coqui-stt.h; 0 for an utterance without a result): what a caller needs to compare a
 * batch with the reference decoder's output without walking 64 Metadata structures. */
STTX_EXPORT char** STTX_BatchCollectScored(ModelState* aCtx, int aTicket, unsigned int* aCount, double* aConfidence);
/* The engine's tunables (stt_amd/csrc/tuning.h: one table, every entry with a measured default; none changes results).  Set them
 * between calls, with nothing in flight; STT_AMD_TUNING="name=value,..." seeds the table when the library is first used.
 * Returns STT_ERR_INVALID_SHAPE for an unknown name. */
STTX_EXPORT int STTX_SetTuning(const char* aName, int aValue);
STTX_EXPORT int STTX_GetTuning(const char* aName, int* aValue);
/* Call once BEFORE the process makes its first HIP call (before STT_CreateModel): asks the HIP runtime for 16 hardware queues
 * (GPU_MAX_HW_QUEUES, unless the caller set it), so that the streams of the batch path and of streaming replicas do not share queues.  Without it everything
 * still works; streams that share a queue run one after the other. */
STTX_EXPORT void STTX_ConfigureRuntime(void);
STTX_EXPORT void STTX_FreeStrings(char** aStrings, unsigned int aCount);
STTX_EXPORT void STTX_FreeMetadataArray(Metadata** aMetadata, unsigned int aCount);

/* ---- several GPUs of one node -------------------------------------------------------
 * The reference scales this path with one process per GPU and utterances dealt from a queue
 * (training/coqui_stt_training/transcribe.py:40-56,136-148).  A fleet does it inside the library: one replica of the model per
 * listed HIP device, one host thread per device, utterances dealt longest-processing-time-first, no communication while
 * decoding, transcripts gathered with RCCL over xGMI (all-gather of byte counts, then of the padded records).  librccl is
 * loaded with dlopen when the first fleet is created. */
typedef struct STTX_Fleet STTX_Fleet;
STTX_EXPORT int STTX_FleetCreate(const char* aModelPath, const int* aDevices, unsigned int aNumDevices, STTX_Fleet** retval);
STTX_EXPORT unsigned int STTX_FleetSize(const STTX_Fleet* aFleet);
STTX_EXPORT int STTX_FleetEnableExternalScorer(STTX_Fleet* aFleet, const char* aScorerPath);
STTX_EXPORT int STTX_FleetSetBeamWidth(STTX_Fleet* aFleet, unsigned int aBeamWidth);
/* aBatch malloc'd strings in the caller's order (STTX_FreeStrings), or NULL if any shard failed. */
STTX_EXPORT char** STTX_FleetSpeechToTextBatch(STTX_Fleet* aFleet, const short* const* aBuffers, const unsigned int* aBufferSizes,
                                              unsigned int aBatch);
STTX_EXPORT void STTX_FleetFree(STTX_Fleet* aFleet);
/* The dealing rule alone (host only, no GPU): aShardOf[i] = shard of utterance i; by descending length (stable), each to the
 * least loaded shard so far (lowest index on ties) -- the same rule as stt_amd/dist.py: shard_utterances. */
STTX_EXPORT int STTX_ShardUtterances(const unsigned int* aSizes, unsigned int aCount, unsigned int aShards, unsigned int* aShardOf);

/* ---- many streams at once (serving) -------------------------------------------------------------
 * The reference API advances one stream per call (stt.cc:553-639).  A server with many live streams calls these instead:
 * the semantics per stream are exactly those of STT_FeedAudioContent / STT_IntermediateDecode / STT_FinishStream, but the
 * windows that become ready in ANY of the streams go through the acoustic model and the beam search as one batch.
 * All streams must belong to the same model; streams that differ in beam width, scorer or hot words (captured at creation)
 * are processed one by one. */
STTX_EXPORT void STTX_FeedAudioContentBatch(StreamingState* const* aStreams, const short* const* aBuffers, const unsigned int* aBufferSizes,
                                           unsigned int aCount);
/* ... the same, and for every stream with aLast[i] != 0 this buffer is the stream's FINAL audio: what STT_FinishStream's flush does in
 * an acoustic pass of its own (the partial window, the trailing context frames, the last partial batch of windows: stt.cc:236-254)
 * goes through the model in the SAME launches as the other streams' hop.  STT_FinishStream / STT_FinishStreamWithMetadata /
 * STTX_FinishStreamBatch on such a stream then only rank and back-track.  (A server that keeps a fixed number of streams live finishes a
 * few of them in every hop; their flushes cost a whole pass each otherwise.)  aLast[i] == 2: the same, but whatever the flush leaves
 * after the call's first pass (at most n_steps - 1 windows: the flush adds n_context + 1 frames to a full hop) is not given a pass of
 * its own: it rides in the stream's NEXT STTX_FeedAudioContentBatch(Ex) call (pass the stream with an empty buffer, beside the live
 * streams' hop) or is processed by its finish.  aLast may be NULL (= STTX_FeedAudioContentBatch).
 * After a stream's final audio only decodes and the finish mean anything: audio fed to it later (here or through STT_FeedAudioContent) is
 * IGNORED -- the call only drains a deferred tail -- and the ...FlushBuffers decodes add no further partial-window frame. */
STTX_EXPORT void STTX_FeedAudioContentBatchEx(StreamingState* const* aStreams, const short* const* aBuffers, const unsigned int* aBufferSizes,
                                              const unsigned char* aLast, unsigned int aCount);
/* aCount malloc'd strings (free with STTX_FreeStrings), or NULL on error. */
STTX_EXPORT char** STTX_IntermediateDecodeBatch(StreamingState* const* aStreams, unsigned int aCount);
/* Like STT_FinishStream on every stream: the streams are destroyed. */
STTX_EXPORT char** STTX_FinishStreamBatch(StreamingState* const* aStreams, unsigned int aCount);
/* Both in one ranking + back-tracking launch: STT_IntermediateDecode for the streams with aFinish[i] == 0, STT_FinishStream (the
 * stream is destroyed, also on error) for those with aFinish[i] != 0 -- a server's hop has some of each.  aCount strings or NULL. */
STTX_EXPORT char** STTX_DecodeStreamsBatch(StreamingState* const* aStreams, const unsigned char* aFinish, unsigned int aCount);

/* Per-stage GPU time of the last batch call, measured with HIP events on the engine's own stream.
 * aMs receives up to aCap floats: [0] features, [1] dense layers 1-3 + x-projection, [2] LSTM recurrence,
 * [3] layers 5-6 + softmax, [4] decoder next, [5] decoder decode + D2H, [6] LSTM launches, [7] timesteps.
 * aEnable: 0 = off, 1 = stage events + decoder counters, 2 = also the search kernel's per-phase shader-cycle
 * counters (STTX_GetDecoderPhaseCycles; they cost a few percent of the search kernel, so not inside timed runs), 3 = ONLY the events
 * around the recurrence's launches on its own stream ([2], [6], [7] are filled): the one live measurement a timed run needs for its
 * roofline -- every event is a barrier packet on its queue, and all of them together cost the batch pipeline ~3 %. */
STTX_EXPORT int STTX_SetProfiling(ModelState* aCtx, int aEnable);
STTX_EXPORT int STTX_GetStageTimes(ModelState* aCtx, float* aMs, int aCap);
/* Decoder counters accumulated over the last batch call: steps, candidates, lm queries, lm memory probes. */
STTX_EXPORT int STTX_GetDecoderStats(ModelState* aCtx, unsigned long long* aOut4);
/* Shader cycles spent per decoder phase (summed over streams) in the last batch call: emissions + hash, expand
 * prefix-sum, expand items, LM, merge, select, rank + write, end of step. */
STTX_EXPORT int STTX_GetDecoderPhaseCycles(ModelState* aCtx, unsigned long long* aOut8);
/* Profiling level 2: shader cycles between the fine-grained stamps of the search step (ctc.hip: DecParams::stamps), summed over
 * the streams of the last batch call ([0..15] arrival of each wave at the end of the expand phase, [16..31] its wait there, ...). */
STTX_EXPORT int STTX_GetDecoderStamps(ModelState* aCtx, unsigned long long* aOut64);

/* ---- stage-level entry points (host buffers in and out) --------------------------------------- */
/* aOut: [aCapFrames][n_input] floats; *aNumFrames = frames produced for aNumSamples samples. */
STTX_EXPORT int STTX_ComputeMfcc(ModelState* aCtx, const short* aBuffer, unsigned int aNumSamples, float* aOut,
                                unsigned int aCapFrames, unsigned int* aNumFrames);
/* aProbs: [aBatch][aMaxFrames][n_classes] floats; aNumFrames[i] = frames of utterance i. */
STTX_EXPORT int STTX_AcousticProbs(ModelState* aCtx, const short* const* aBuffers, const unsigned int* aBufferSizes,
                                  unsigned int aBatch, float* aProbs, unsigned int aMaxFrames, unsigned int* aNumFrames);
/* One infer() call: aMfcc [aNumFrames][n_input*(2*n_context+1)], state vectors [n_hidden]; aProbs [aNumFrames][n_classes]. */
STTX_EXPORT int STTX_InferChunk(ModelState* aCtx, const float* aMfcc, unsigned int aNumFrames, const float* aStateC,
                               const float* aStateH, float* aProbs, float* aNewStateC, float* aNewStateH);
STTX_EXPORT int STTX_GetGeometry(const ModelState* aCtx, int* aOut10);

/* ---- decoder on caller-supplied emissions ----------------------------------------------------- */
typedef struct STTX_Decoder STTX_Decoder;
/* aNumStreams independent DecoderStates sharing the model's alphabet, scorer and hot words (captured now).
 * Threads: a decoder has result blocks of its own and runs on one of the model's decoder streams (tunable decoder_streams: 1 = the model's own
 * stream, the default; 2..4 = a pool dealt round-robin at creation), so DIFFERENT decoders of one model may be driven from different host
 * threads at the same time -- side by side on the GPU with decoder_streams > 1 (one decoder = one workgroup per stream: 64 streams are a
 * quarter of an MI355X; bench.py's decoder-stage workloads keep four decoders in flight).  One decoder is used from one thread at a time, and no decoder call may overlap a call that
 * changes the model (scorer, hot words, tunables) -- the reference's rule for a model, applied per decoder. */
STTX_EXPORT int STTX_DecoderCreate(ModelState* aCtx, unsigned int aNumStreams, unsigned int aBeamWidth, double aCutoffProb,
                                  unsigned int aCutoffTopN, STTX_Decoder** retval);
/* aProbs: [aNumStreams][aStride][n_classes] floats; stream i consumes its first aNumFrames[i] rows. */
STTX_EXPORT int STTX_DecoderNext(STTX_Decoder* aDec, const float* aProbs, unsigned int aStride, const unsigned int* aNumFrames);
/* Writes for stream i, result r: tokens/timesteps [i][r][aMaxLen], lens [i][r], confidences [i][r]; aNumResultsOut[i]. */
STTX_EXPORT int STTX_DecoderDecode(const STTX_Decoder* aDec, unsigned int aNumResults, unsigned int aMaxLen,
                                  unsigned int* aTokens, unsigned int* aTimesteps, int* aLens, double* aConfidences,
                                  int* aNumResultsOut);
/* Raw beam of one stream after the last next(): up to aCap entries of (score, log_prob_b_prev, log_prob_nb_prev, character). */
STTX_EXPORT int STTX_DecoderBeam(const STTX_Decoder* aDec, unsigned int aStream, float* aScore, float* aPb, float* aPnb,
                                int* aChar, unsigned int aCap);
STTX_EXPORT int STTX_DecoderStats(const STTX_Decoder* aDec, unsigned long long* aOut4);
/* Profiling of a standalone decoder (benchmarks/search_micro.py): level 1 = HIP-event time of the search launches, 2 = also the
 * kernel's own phase cycle counters (8, summed over streams) and fine-grained stamps (64).  No equivalent in the reference. */
/* The OR of the streams' search-state error bits (0 = every stream's state is intact): 0x1 path arena, 0x2 time arena, 0x4 candidates, 0x8 scorer
 * cache, 0x10 an intra-workgroup wait timed out, 0x20 two prefixes with one path key (stt_amd/csrc/ctc.hip: child_key).  A stream with a
 * bit set yields no results (STTX_DecoderDecode fails): the reference has no such states (heap trie), a refused result stands in for them. */
STTX_EXPORT int STTX_DecoderErrorBits(const STTX_Decoder* aDec, int* aBits);
STTX_EXPORT int STTX_DecoderSetProfiling(STTX_Decoder* aDec, int aLevel);
STTX_EXPORT int STTX_DecoderGetProfile(const STTX_Decoder* aDec, unsigned long long* aPhase8, unsigned long long* aStamps64, float* aSearchMs);
STTX_EXPORT void STTX_DecoderFree(STTX_Decoder* aDec);

/* ---- model files (host only, no GPU needed) ---------------------------------------------------- */
/* STT_CreateModel accepts two containers: the reference's `.tflite` export (float or hybrid int8; read without TensorFlow
 * Lite, replaces TFLiteModelState::init, native_client/tflitemodelstate.cc:161-338) and this engine's raw container
 * (stt_amd/modelfile.py).  These two calls parse a model image the same way STT_CreateModel does and hand out what was
 * found; `python -m stt_amd.convert` builds the .tflite -> .sttw converter on them. */
typedef struct STTX_ModelInfo {
  int n_input, n_context, n_hidden, n_classes, n_steps, sample_rate, win_len, win_step, beam_width;
  float relu_clip;
  unsigned int alphabet_bytes;
  int is_tflite;
  int hybrid_int8;                 /* 1: the file's six matrices are symmetric int8 and the model will run TFLite's hybrid arithmetic (STTX_GetAcousticMode 1) */
  int asymmetric_quantize_inputs;  /* 1: a FULLY_CONNECTED asks for asymmetric input quantisation: not restated, the weights are de-quantised (f16 path) */
} STTX_ModelInfo;
STTX_EXPORT int STTX_InspectModel(const char* aModelBuffer, unsigned int aBufferSize, STTX_ModelInfo* aInfo);
/* aIndex 0..11: layer_1/weights, layer_1/bias, layer_2/weights, layer_2/bias, layer_3/weights, layer_3/bias, lstm/kernel,
 * lstm/bias, layer_5/weights, layer_5/bias, layer_6/weights, layer_6/bias as f32 (matrices [inputs][outputs]); 12: the
 * serialised alphabet.  Writes min(size, aCapBytes) bytes to aOut (may be NULL) and the full size to *aBytes. */
STTX_EXPORT int STTX_ReadModelTensor(const char* aModelBuffer, unsigned int aBufferSize, int aIndex, void* aOut,
                                    unsigned long long aCapBytes, unsigned long long* aBytes);

/* (the kernel-level test hooks -- STTX_Test*, STTX_Debug* -- are declared in include/stt_amd_test.h and exist in libstt_test.so only) */
/* Which arithmetic the acoustic model of aCtx runs in: 0 = f16 MFMA operands / f32 accumulate (north_star's; int8 weights of a quantised
 * file are de-quantised), 1 = the released models' own (TensorFlow Lite's hybrid int8 FULLY_CONNECTED end to end: int8 activations per row,
 * int32 sums; taken for a dynamic-range quantised `.tflite`, or with the tunable am_i8 = 1).  Replaces nothing in coqui-stt.h: the
 * reference's CPU path has only the second one (native_client/tflitemodelstate.cc:200,369-405). */
STTX_EXPORT int STTX_GetAcousticMode(const ModelState* aCtx);
/* Host-side packing of the recurrent matrix (no GPU needed): aKernel [2H][4H] f32 -> aOut [4H*H] f16 bits. */
STTX_EXPORT int STTX_PackLstmRecurrent(const float* aKernel, int aHidden, unsigned short* aOut);

#ifdef __cplusplus
}
#endif
#endif /* STT_AMD_H */
