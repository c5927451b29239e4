// stt_amd/csrc/tuning.h -- the engine's tunables in ONE table (internal header).
//
// Every value has a measured default; nothing here changes results, only what runs beside what
// and which of several bit-identical kernel forms is launched.  Two ways in, both for A/B runs and tests:
//   STTX_SetTuning("name", value)        between calls, nothing in flight (include/stt_amd.h)
//   STT_AMD_TUNING="name=value,name=value"   read once, when the table is first used
// The reference has no equivalent: its only run-time switches are the model's beam width and the scorer's alpha/beta.
#pragma once

#define STT_TUNING_FIELDS(X)                                                                                                       \
  X(pipeline, 0, "group slots of the batch path (0 = by configuration: 2, or 4 for a search-bound setup; 1..4)")                   \
  X(active, 0, "slots whose beam searches may run side by side (0 = by configuration)")                                           \
  X(pair, 1, "1: two consecutive 64-utterance batches share one recurrence (128 rows per recurrent step); 0: 64 rows")            \
  X(chunk0, 16, "blocking call: frames of the first time-chunk")                                                                   \
  X(chunk, 48, "blocking call: frames of the later time-chunks")                                                                   \
  X(pchunk0, 16, "batches in flight: frames of the first time-chunk")                                                              \
  X(pchunk, 48, "batches in flight: frames of the later time-chunks")                                                              \
  X(am_pipe, 1, "acoustic model of the batch path as three engines (0: one stream)")                                               \
  X(dense_lds_kb, 82, "LDS floor of the two-stage GEMM form when it runs beside the recurrence (one workgroup per CU)")            \
  X(dense_solo, 3, "GEMM form beside the recurrence: 4 = 128x256 eight-wave in four stages of K = 32 (three K-tiles in flight), 3 = 128x256 eight-wave, 2 = 128-square three-stage eight-wave, 1 = four-wave, 0 = padded") \
  X(dense_solo_test, -1, "STTX_TestDense: force DenseArgs::solo (-1 = off)")                                                       \
  X(dense_tile, 0, "force the GEMM tile side (128 / 256; 0 = by shape)")                                                           \
  X(dense_big_min, 480, "256-square tiles from this many tiles on")                                                                \
  X(lstm_graph, 1, "recurrence chunks replayed as hipGraphs")                                                                      \
  X(lstm_passes, 3, "form of the recurrent step beside the GEMMs (1 one pass, 2 two passes, 3 owner form)")                        \
  X(lstm_form, 0, "force the form of EVERY recurrent step (0 = as the caller asks)")                                               \
  X(lstm_prefetch, 2, "k-steps per prefetch group of the 64-row recurrent step (1, 2, 4)")                                         \
  X(lstm_prio, 1, "recurrent step's waves at s_setprio 3")                                                                         \
  X(lstm_probe, 0, "STTX_TestLstmSteps only: timing probes of the recurrent step (kernels_am.hip: launch_lstm_probe); wrong results")                                                                         \
  X(lstm_cotenant, 0, "STTX_TestLstmSteps only: this many x-projection GEMMs (6144 x 8192 x 2048, form dense_solo) run beside the steps") \
  X(lstm_stamps, 0, "STTX_TestLstmSteps only: in-kernel REFCLK stamps, summary on stderr")                                         \
  X(am_i8, -1, "acoustic model in the released models' own arithmetic (TFLite's hybrid int8 FULLY_CONNECTED: int8 activations per row, int32 sums): -1 = when the file is a dynamic-range quantised .tflite, 0 = never (int8 weights are de-quantised to f16), 1 = always (float weights are quantised at load as the converter does); read when a model is loaded") \
  X(am_place, 1, "three-engine form: the recurrence's and the output engine's streams are PLACED -- candidates probed for which dispatch pipe they share with the GEMM engine's and the searches' streams (engine.cpp: place_engine_streams; 0 = take whatever the runtime hands out, round 5's behaviour)") \
  X(am_placed, 0, "counter, not a knob: placements made so far; bits 8.. of the last one: candidates found sharing a pipe with the GEMM engine's stream") \
  X(am_moves, 6, "three-engine form: how often a model may move its recurrence and output engine to fresh streams when a chunk's steps are picked up late twice in a row (0 = never watch)") \
  X(am_slow_us, 0, "three-engine form: microseconds per recurrent step above which a chunk counts as slow (0 = 31 us x max(1, (n_hidden / 2048)^2))") \
  X(am_moved, 0, "counter, not a knob: moves of the engines to fresh streams so far (all models)") \
  X(am_step_us_x10, 0, "counter, not a knob: the last watched chunk's microseconds per recurrent step, times ten") \
  X(search_cus, 0, "beam-search streams confined to the compute units of the first N bits of a CU mask (hipExtStreamCreateWithCUMask; bit i = XCD i % 8: benchmarks/cumask_probe.hip), 0 = no mask; read when a model is loaded") \
  X(am_cus, 0, "with search_cus: which acoustic engines are confined to the OTHER compute units (bit 0 the GEMM engine, bit 1 the recurrence, bit 2 the output engine), 0 = none; read when a model is loaded") \
  X(lstm_i8_rows, 64, "int8 recurrent step: rows per workgroup (16 / 32 / 64 / 128); a step of more rows runs that many row groups per 16-unit slice") \
  X(am_i8_pipe, 1, "int8 path: batches in flight through three acoustic engines like the f16 path (1: 3.0 - 3.2 ms per batch once the engines' queues sit well -- the placement watch, am_moves, sees to that) or on one acoustic stream (0: 3.8 ms wherever the queues sit)") \
  X(lstm_upw, 16, "hidden units per recurrent workgroup (16 or 8); read when a model is loaded")                                   \
  X(copy_kernel, 1, "small tables / result blocks through a copy kernel and mapped page-locked memory (0: copy engine)")           \
  X(search_lds_kb, 160, "LDS budget of the search kernel's layout (96..160)")                                                      \
  X(search_step, 2, "word-mode search step: 2 = label bitmaps + indexed FullScore where it applies, 0 = generic step")              \
  X(lm_prio, 3, "s_setprio of the search step's language-model waves during their queries (0..3)")                               \
  X(lm_waves, 0, "language-model waves of the search step (0 = by beam width)")                                                    \
  X(exp_waves, 0, "expand waves of the search step (0 = all the others)")                                                          \
  X(wait_spins, 0, "test hook: polls (of 256 cycles) an intra-workgroup counter wait of the search step may take before it gives up with error bit 0x10 (0 = 4 M, about half a second)") \
  X(debug_key_bits, 0, "test hook: path keys of the search truncated to this many bits (4..62; 0 = all 63): forces key collisions, which the guard must flag (error bit 0x20)") \
  X(item_table_cap, 0, "test hook: items per pass of the bitmap step's expand table (0 = what fits; small values force the several-pass path)") \
  X(stream_graph, 1, "batched streaming: the acoustic + search pass of a hop replayed as one hipGraph per live-set shape (0: launch by launch)") \
  X(stream_frames, 256, "frames (20 ms each) a new stream's search arenas are laid out for; longer utterances grow them (an allocation and a copy in the middle of a hop): a server sets its longest expected utterance") \
  X(decode_cache, 1, "streams: a decode with one result walks the best path back only to where it meets the previously decoded one (0: the whole path every time); read when a stream is created") \
  X(lm_memo, 1, "code-point scorer: FullScore memo table (0 off, 1 = 2^24 entries of 32 B -- measured on the code-point bench scorer: 2^18 66.4 ms per batch, 2^22 54.1, 2^24 49.3 --, 10..26 = log2 of the entry count); read when a scorer is loaded")                                                                         \
  X(unit_bounds, 1, "code-point scorer: upper bounds of the LM score (candidates that cannot reach the beam skip FullScore): 1 = the largest over all units, 2 = also a table by code point; read when a scorer is loaded") \
  X(dict_tree_mb, 2048, "dictionary unfolded into a tree (no arc reads in the search): byte cap in MiB, 0 = keep the automaton")           \
  X(cp_blocks, 1, "code-point scorer: bigram blocks (context x 64 consecutive code points -> one table entry + one slice of records; unigram records by code point): a FullScore of a unit that is one code point of the vocabulary goes through them -- no hash of its bytes, no vocabulary probe, no memo (round 6: the bytes workload 50.1 -> 45.6 ms per batch, the LM phase on flat emissions 1.36 M -> 0.87 M cycles per stream-timestep); 0 = memo + index only; read when a scorer is loaded")   \
  X(cp_index, 1, "code-point scorer: a FullScore that misses the memo goes through the hashed n-gram index (one bucket read per order) instead of the trie walk (an interpolation search per order); read when a scorer is loaded")   \
  X(lm_index_mb, 4096, "hashed n-gram index: byte cap in MiB (larger models take the trie walk)")                                  \
  X(decoder_streams, 1, "standalone decoders (STTX_Decoder*): 1 = every decoder on the model's own stream; 2..4 = a pool of that many streams per model, dealt round-robin when a decoder is created, so that decoders can be driven side by side from several host threads (bench.py's decoder-stage workloads: 4); keep the process at or below GPU_MAX_HW_QUEUES streams -- engine.h, INTEGRATION.md") \
  X(debug_poison, 0, "test hook: every new device buffer is filled with this byte pattern first (1 = 0xFF, 2 = 0xA5; 0 = left as the allocator returns it): a read of memory nobody wrote then yields absurd indices instead of whatever the previous owner left there") \
  X(debug_scribble, 0, "test hook: bit 0 = before every search launch a kernel on the same stream overwrites the LDS of every compute unit, the launch queue's scratch memory and the vector registers with a pattern (a read of LDS / scratch / a register the search kernel did not write then sees garbage instead of what its own previous launch left there); bit 1 = the decoder pool may hold 16 streams per model (decoder_streams up to 16: the configuration in which round 6's race in the code-point step showed every time -- DESIGN.md 10.10); bits 2 - 6: variants of the scribbler for benchmarks/r06_scribble_fuzz.sh (no scratch, no LDS / registers, one workgroup, a sleep after each decoder launch, once per stream); the scribbler exists in libstt_test.so only") \
  X(debug_scribble_lo, 0, "test hook: first 4-byte word of every lane's scratch memory the scribbler writes") \
  X(debug_scribble_hi, 512, "test hook: one past the last word the scribbler writes (bisection of a read of uninitialised scratch)") \
  X(arena_shrink, 1, "test hook: divides the optimistic arena sizes of a batch group (forces the overflow -> decode-again path)")        \
  X(hop_replays, 0, "counter, not a knob: streaming hops whose acoustic + search pass was replayed from a captured graph")                 \
  X(arena_retries, 0, "counter, not a knob: batch groups decoded again with full-size arenas after an overflow flag")                 \
  X(dump_marks, 0, "profiling: raw HIP-event timeline of the batch path on stderr")

struct Tuning {
#define X(name, def, doc) int name = def;
  STT_TUNING_FIELDS(X)
#undef X
};
Tuning& tune();                                // the table (seeded from STT_AMD_TUNING on first use)
void tuning_model_count(int delta);            // models alive: the knobs that are read when a model / scorer is loaded cannot change under one
int tuning_set(const char* name, int value);   // 0 = ok, -1 = no such name
int tuning_get(const char* name, int* value);
