/* include/coqui-stt.h -- the drop-in boundary of the MI355X engine.
 *
 * This header re-declares, symbol for symbol, the C ABI that every binding of the reference
 * links against (reference: native_client/coqui-stt.h; implementation it replaces:
 * native_client/stt.cc:336-731, native_client/modelstate.cc:32-76,
 * native_client/stt_errors.cc:4-19).  Types, field order, argument order and error values
 * are ABI and therefore identical; the text around them is ours.  libstt.so built from
 * stt_amd/csrc exports exactly these 29 functions (tests/test_abi.py checks the list).
 *
 * Ownership rules (same as the reference):
 *   - every returned char*       -> STT_FreeString      (malloc/strdup'd)
 *   - every returned Metadata*   -> STT_FreeMetadata
 *   - STT_FinishStream* / STT_FreeStream destroy the stream; STT_FreeModel destroys the model
 *   - a scorer / hot-word set is captured by a stream when the stream is created
 */
#ifndef COQUI_STT_H
#define COQUI_STT_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SWIG
#if defined _MSC_VER
#define STT_EXPORT __declspec(dllexport)
#else
#define STT_EXPORT __attribute__((visibility("default")))
#endif
#else
#define STT_EXPORT
#endif

#ifndef SWIG_ERRORS_ONLY

typedef struct ModelState ModelState;         /* opaque: model + (optional) scorer, resident in HBM */
typedef struct StreamingState StreamingState; /* opaque: one resumable utterance */

/* one decoded label; reference coqui-stt.h:29-38 */
typedef struct TokenMetadata {
  const char* const text;       /* UTF-8 text of the label */
  const unsigned int timestep;  /* index of the 20 ms frame the label was emitted at */
  const float start_time;       /* timestep * win_step / sample_rate, seconds */
} TokenMetadata;

/* one candidate transcript; reference coqui-stt.h:43-56 */
typedef struct CandidateTranscript {
  const TokenMetadata* const tokens;
  const unsigned int num_tokens;
  const double confidence; /* sum of acoustic log-probs plus scorer terms; relative, not a probability */
} CandidateTranscript;

/* raw per-frame class probabilities; reference coqui-stt.h:61-73 */
typedef struct AcousticModelEmissions {
  int num_symbols;
  const char** symbols; /* num_symbols + 1 entries, the last one is the blank ("\t") */
  int num_timesteps;
  const double* emissions; /* [num_timesteps][num_symbols + 1] */
} AcousticModelEmissions;

/* N-best list; reference coqui-stt.h:78-86 */
typedef struct Metadata {
  const CandidateTranscript* const transcripts;
  const unsigned int num_transcripts;
  const AcousticModelEmissions* const emissions; /* NULL unless ...WithEmissions was called */
} Metadata;

#endif /* SWIG_ERRORS_ONLY */

/* Error codes and messages; values are ABI (reference coqui-stt.h:92-124). */
// sphinx-doc: error_code_listing_start
#define STT_FOR_EACH_ERROR(APPLY) \
  APPLY(STT_ERR_OK,                      0x0000, "No error.") \
  APPLY(STT_ERR_NO_MODEL,                0x1000, "Missing model information.") \
  APPLY(STT_ERR_INVALID_ALPHABET,        0x2000, "Invalid alphabet embedded in model. (Data corruption?)") \
  APPLY(STT_ERR_INVALID_SHAPE,           0x2001, "Invalid model shape.") \
  APPLY(STT_ERR_INVALID_SCORER,          0x2002, "Invalid scorer file.") \
  APPLY(STT_ERR_MODEL_INCOMPATIBLE,      0x2003, "Incompatible model.") \
  APPLY(STT_ERR_SCORER_NOT_ENABLED,      0x2004, "External scorer is not enabled.") \
  APPLY(STT_ERR_SCORER_UNREADABLE,       0x2005, "Could not read scorer file.") \
  APPLY(STT_ERR_SCORER_INVALID_LM,       0x2006, "Could not recognize language model header in scorer.") \
  APPLY(STT_ERR_SCORER_NO_TRIE,          0x2007, "Reached end of scorer file before loading vocabulary trie.") \
  APPLY(STT_ERR_SCORER_INVALID_TRIE,     0x2008, "Invalid magic in trie header.") \
  APPLY(STT_ERR_SCORER_VERSION_MISMATCH, 0x2009, "Scorer file version does not match expected version.") \
  APPLY(STT_ERR_FAIL_INIT_MMAP,          0x3000, "Failed to initialize memory mapped model.") \
  APPLY(STT_ERR_FAIL_INIT_SESS,          0x3001, "Failed to initialize the session.") \
  APPLY(STT_ERR_FAIL_INTERPRETER,        0x3002, "Interpreter failed.") \
  APPLY(STT_ERR_FAIL_RUN_SESS,           0x3003, "Failed to run the session.") \
  APPLY(STT_ERR_FAIL_CREATE_STREAM,      0x3004, "Error creating the stream.") \
  APPLY(STT_ERR_FAIL_READ_PROTOBUF,      0x3005, "Error reading the proto buffer model file.") \
  APPLY(STT_ERR_FAIL_CREATE_SESS,        0x3006, "Failed to create session.") \
  APPLY(STT_ERR_FAIL_CREATE_MODEL,       0x3007, "Could not allocate model state.") \
  APPLY(STT_ERR_FAIL_INSERT_HOTWORD,     0x3008, "Could not insert hot-word.") \
  APPLY(STT_ERR_FAIL_CLEAR_HOTWORD,      0x3009, "Could not clear hot-words.") \
  APPLY(STT_ERR_FAIL_ERASE_HOTWORD,      0x3010, "Could not erase hot-word.")
// sphinx-doc: error_code_listing_end

enum STT_Error_Codes {
#define DEFINE(NAME, VALUE, DESC) NAME = VALUE,
  STT_FOR_EACH_ERROR(DEFINE)
#undef DEFINE
};

#ifndef SWIG_ERRORS_ONLY

/* ---- model life cycle (stt.cc:336-412) ------------------------------------------------- */
/* Loads an acoustic model (see DESIGN.md "model container") and uploads it to HBM. */
STT_EXPORT int STT_CreateModel(const char* aModelPath, ModelState** retval);
/* Same from memory; the caller keeps aModelBuffer alive for the life of the model. */
STT_EXPORT int STT_CreateModelFromBuffer(const char* aModelBuffer, unsigned int aBufferSize, ModelState** retval);
STT_EXPORT unsigned int STT_GetModelBeamWidth(const ModelState* aCtx);
STT_EXPORT int STT_SetModelBeamWidth(ModelState* aCtx, unsigned int aBeamWidth);
STT_EXPORT int STT_GetModelSampleRate(const ModelState* aCtx);
STT_EXPORT void STT_FreeModel(ModelState* ctx);

/* ---- external scorer and hot words (stt.cc:414-517) ------------------------------------ */
STT_EXPORT int STT_EnableExternalScorer(ModelState* aCtx, const char* aScorerPath);
STT_EXPORT int STT_EnableExternalScorerFromBuffer(ModelState* aCtx, const char* aScorerBuffer, unsigned int aBufferSize);
STT_EXPORT int STT_AddHotWord(ModelState* aCtx, const char* word, float boost);
STT_EXPORT int STT_EraseHotWord(ModelState* aCtx, const char* word);
STT_EXPORT int STT_ClearHotWords(ModelState* aCtx);
STT_EXPORT int STT_DisableExternalScorer(ModelState* aCtx);
STT_EXPORT int STT_SetScorerAlphaBeta(ModelState* aCtx, float aAlpha, float aBeta);

/* ---- one-shot recognition (stt.cc:641-688): aBufferSize counts 16-bit mono samples ------ */
STT_EXPORT char* STT_SpeechToText(ModelState* aCtx, const short* aBuffer, unsigned int aBufferSize);
STT_EXPORT Metadata* STT_SpeechToTextWithMetadata(ModelState* aCtx, const short* aBuffer, unsigned int aBufferSize,
                                                  unsigned int aNumResults);
STT_EXPORT Metadata* STT_SpeechToTextWithEmissions(ModelState* aCtx, const short* aBuffer, unsigned int aBufferSize,
                                                   unsigned int aNumResults);

/* ---- streaming (stt.cc:519-639) --------------------------------------------------------- */
STT_EXPORT int STT_CreateStream(ModelState* aCtx, StreamingState** retval);
STT_EXPORT void STT_FeedAudioContent(StreamingState* aSctx, const short* aBuffer, unsigned int aBufferSize);
/* does not disturb the stream */
STT_EXPORT char* STT_IntermediateDecode(const StreamingState* aSctx);
STT_EXPORT Metadata* STT_IntermediateDecodeWithMetadata(const StreamingState* aSctx, unsigned int aNumResults);
/* flushes the partial audio window and the partial batch through the model first (changes LSTM state) */
STT_EXPORT char* STT_IntermediateDecodeFlushBuffers(StreamingState* aSctx);
STT_EXPORT Metadata* STT_IntermediateDecodeWithMetadataFlushBuffers(StreamingState* aSctx, unsigned int aNumResults);
/* final result; frees the stream */
STT_EXPORT char* STT_FinishStream(StreamingState* aSctx);
STT_EXPORT Metadata* STT_FinishStreamWithMetadata(StreamingState* aSctx, unsigned int aNumResults);
STT_EXPORT void STT_FreeStream(StreamingState* aSctx);

/* ---- housekeeping (stt.cc:696-737, stt_errors.cc:4-19) ---------------------------------- */
STT_EXPORT void STT_FreeMetadata(Metadata* m);
STT_EXPORT void STT_FreeString(char* str);
STT_EXPORT char* STT_Version();
STT_EXPORT char* STT_ErrorCodeToErrorMessage(int aErrorCode);

#endif /* SWIG_ERRORS_ONLY */

#undef STT_EXPORT

#ifdef __cplusplus
}
#endif

#endif /* COQUI_STT_H */
