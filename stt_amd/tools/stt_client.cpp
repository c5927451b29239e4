// stt_amd/tools/stt_client.cpp -- `stt` command-line client on top of include/coqui-stt.h.
//
// Same options and output formats as the reference client (native_client/args.h:47-204, native_client/client.cc:
// plain transcript, --extended, --json with per-word timings and alternatives, --stream N / --extended_stream N with
// intermediate results, --hot_words w:boost,..., --init_from_bytes, -t, --version), so the scenarios of
// ci_scripts/asserts.sh can be replayed against libstt.so.  Differences, on purpose:
//   * WAV files are read by walking the RIFF chunks (fmt / data); the reference's no-SoX fallback assumes a 44-byte
//     header (client.cc:390-426).  16-bit mono PCM at the model's sample rate only (no resampler: SoX is not used).
//   * a directory is decoded as ONE batch through STTX_SpeechToTextBatch (all utterances in flight on the GPU) instead
//     of file by file, unless a streaming mode was asked for.
//   * -t reports wall-clock time (std::chrono), not clock() CPU time.
#include <dirent.h>
#include <getopt.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/stt_amd.h"

namespace {

struct Options {
  std::string model, scorer, audio, hot_words;
  bool set_beam = false, set_ab = false, times = false, extended = false, json = false, emissions = false, from_bytes = false;
  int beam = 0, candidates = 3, stream = 0, ext_stream = 0;
  float alpha = 0.f, beta = 0.f;
};

void usage(const char* bin) {
  std::cout << "Usage: " << bin << " --model MODEL [--scorer SCORER] --audio AUDIO [-t] [-e]\n\n"
               "Running Coqui STT inference (MI355X engine).\n\n"
               "\t--model MODEL\t\t\tPath to the model file\n"
               "\t--scorer SCORER\t\t\tPath to the external scorer file\n"
               "\t--audio AUDIO\t\t\tPath to the audio file (16-bit mono WAV) or a directory of them\n"
               "\t--beam_width BEAM_WIDTH\t\tValue for decoder beam width (int)\n"
               "\t--lm_alpha LM_ALPHA\t\tValue for language model alpha param (float)\n"
               "\t--lm_beta LM_BETA\t\tValue for language model beta param (float)\n"
               "\t-t\t\t\t\tRun in benchmark mode, output inference time\n"
               "\t--extended\t\t\tOutput string from extended metadata\n"
               "\t--keep_emissions\t\tSave the output of the acoustic model\n"
               "\t--json\t\t\t\tExtended output, shows word timings as JSON\n"
               "\t--candidate_transcripts NUMBER\tNumber of candidate transcripts to include in JSON output\n"
               "\t--stream size\t\t\tRun in stream mode, output intermediate results\n"
               "\t--extended_stream size\t\tRun in stream mode using metadata output, output intermediate results\n"
               "\t--hot_words\t\t\tHot-words and their boosts. Word:Boost pairs are comma-separated\n"
               "\t--init_from_bytes\t\tInit model and scorer from arrays of bytes\n"
               "\t--help\t\t\t\tShow help\n"
               "\t--version\t\t\tPrint version and exits\n";
  char* v = STT_Version();
  std::cerr << "Coqui STT " << v << "\n";
  STT_FreeString(v);
}

bool parse(int argc, char** argv, Options& o) {
  static const option longopts[] = {
      {"model", required_argument, nullptr, 'm'}, {"scorer", required_argument, nullptr, 'l'}, {"audio", required_argument, nullptr, 'a'},
      {"beam_width", required_argument, nullptr, 'b'}, {"lm_alpha", required_argument, nullptr, 'c'}, {"lm_beta", required_argument, nullptr, 'd'},
      {"t", no_argument, nullptr, 't'}, {"extended", no_argument, nullptr, 'e'}, {"keep_emissions", no_argument, nullptr, 'L'},
      {"json", no_argument, nullptr, 'j'}, {"init_from_bytes", no_argument, nullptr, 'B'}, {"candidate_transcripts", required_argument, nullptr, 150},
      {"stream", required_argument, nullptr, 's'}, {"extended_stream", required_argument, nullptr, 'S'}, {"hot_words", required_argument, nullptr, 'w'},
      {"version", no_argument, nullptr, 'v'}, {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
  bool version = false;
  for (int c; (c = getopt_long(argc, argv, "m:l:a:b:c:d:tejs:w:vh", longopts, nullptr)) != -1;) {
    switch (c) {
      case 'm': o.model = optarg; break;
      case 'l': o.scorer = optarg; break;
      case 'a': o.audio = optarg; break;
      case 'b': o.set_beam = true; o.beam = atoi(optarg); break;
      case 'c': o.set_ab = true; o.alpha = (float)atof(optarg); break;
      case 'd': o.set_ab = true; o.beta = (float)atof(optarg); break;
      case 't': o.times = true; break;
      case 'e': o.extended = true; break;
      case 'L': o.emissions = true; break;
      case 'j': o.json = true; break;
      case 'B': o.from_bytes = true; break;
      case 150: o.candidates = atoi(optarg); break;
      case 's': o.stream = atoi(optarg); break;
      case 'S': o.ext_stream = atoi(optarg); break;
      case 'w': o.hot_words = optarg; break;
      case 'v': version = true; break;
      default: usage(argv[0]); return false;
    }
  }
  if (version) { char* v = STT_Version(); std::cout << "Coqui STT " << v << "\n"; STT_FreeString(v); return false; }
  if (o.model.empty() || o.audio.empty()) { usage(argv[0]); return false; }
  if ((o.stream < 0 || o.stream % 160) || (o.ext_stream < 0 || o.ext_stream % 160)) {  // args.h:186-196
    std::cout << "Stream buffer size must be multiples of 160\n";
    return false;
  }
  return true;
}

std::string slurp(const std::string& path) { std::ifstream f(path, std::ios::binary); std::stringstream s; s << f.rdbuf(); return s.str(); }

// 16-bit mono PCM samples of a RIFF/WAVE file; empty + message on anything else
bool read_wav(const std::string& path, int want_rate, std::vector<short>& out) {
  const std::string d = slurp(path);
  auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, d.data() + o, 2); return v; };
  auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, d.data() + o, 4); return v; };
  if (d.size() < 12 || d.compare(0, 4, "RIFF") || d.compare(8, 4, "WAVE")) { std::cerr << path << ": not a RIFF/WAVE file\n"; return false; }
  bool have_fmt = false;
  for (size_t o = 12; o + 8 <= d.size();) {
    const std::string id = d.substr(o, 4);
    const size_t len = u32(o + 4), body = o + 8;
    if (id == "fmt " && body + 16 <= d.size()) {
      const int fmt = u16(body), ch = u16(body + 2), bits = u16(body + 14);
      const int rate = (int)u32(body + 4);
      if (fmt != 1 || ch != 1 || bits != 16 || rate != want_rate) {
        std::cerr << path << ": need 16-bit mono PCM at " << want_rate << " Hz (got format " << fmt << ", " << ch << " ch, " << bits << " bit, " << rate << " Hz)\n";
        return false;
      }
      have_fmt = true;
    } else if (id == "data" && have_fmt) {
      const size_t n = std::min(len, d.size() - body) / 2;
      out.resize(n);
      memcpy(out.data(), d.data() + body, n * 2);
      return true;
    }
    o = body + len + (len & 1);
  }
  std::cerr << path << ": no fmt/data chunk\n";
  return false;
}

std::string transcript_text(const CandidateTranscript& t) {
  std::string s;
  for (unsigned i = 0; i < t.num_tokens; ++i) s += t.tokens[i].text;
  return s;
}

// words with start time and duration from the token timings (client.cc:64-108)
std::string transcript_json(const CandidateTranscript& t) {
  std::ostringstream o;
  o << "\"metadata\":{\"confidence\":" << t.confidence << "},\"words\":[";
  std::string word; float start = 0; bool first = true;
  for (unsigned i = 0; i < t.num_tokens; ++i) {
    const TokenMetadata& k = t.tokens[i];
    const bool space = strcmp(k.text, " ") == 0;
    if (!space) { if (word.empty()) start = k.start_time; word += k.text; }
    if (space || i + 1 == t.num_tokens) {
      const float dur = std::max(0.0f, k.start_time - start);
      o << (first ? "" : ",") << "{\"word\":\"" << word << "\",\"time\":" << start << ",\"duration\":" << dur << "}";
      first = false; word.clear(); start = 0;
    }
  }
  o << "]";
  return o.str();
}

std::string metadata_json(const Metadata* m, bool with_emissions) {
  std::ostringstream o;
  o << "{\n";
  for (unsigned j = 0; j < m->num_transcripts; ++j) {
    if (j == 0) { o << transcript_json(m->transcripts[0]); if (m->num_transcripts > 1) o << ",\n\"alternatives\":[\n"; }
    else { o << "{" << transcript_json(m->transcripts[j]) << "}" << (j + 1 < m->num_transcripts ? ",\n" : "\n]"); }
  }
  if (with_emissions && m->emissions) {
    const AcousticModelEmissions* e = m->emissions;
    const int C = e->num_symbols + 1;
    o << ",\n\"alphabet\":[";
    for (int i = 0; i < C; ++i) o << "\"" << e->symbols[i] << "\"" << (i + 1 < C ? ", " : "");
    o << "],\n\"emissions\":[\n";
    // The reference prints num_symbols values per row and steps the flat [timesteps][num_symbols + 1] array by num_symbols
    // (client.cc:169-180): rows drift against the blank column.  Kept as is: scripts that parse the reference's output
    // see the same bytes (tests/test_gpu_refclient.py compares the two clients byte for byte).
    const int S = e->num_symbols;
    for (int t = 0; t < e->num_timesteps; ++t) {
      o << "[";
      for (int c = 0; c < S; ++c) o << e->emissions[(size_t)t * S + c] << (c + 1 < S ? ", " : "");
      o << "]" << (t + 1 < e->num_timesteps ? "," : "") << "\n";
    }
    o << "\n]";
  }
  o << "\n}\n";
  return o.str();
}

std::string run_one(ModelState* ctx, const Options& o, const std::vector<short>& pcm) {
  const short* buf = pcm.data();
  const unsigned n = (unsigned)pcm.size();
  if (o.emissions) { Metadata* m = STT_SpeechToTextWithEmissions(ctx, buf, n, (unsigned)o.candidates); std::string s = m ? metadata_json(m, true) : ""; STT_FreeMetadata(m); return s; }
  if (o.extended) { Metadata* m = STT_SpeechToTextWithMetadata(ctx, buf, n, 1); std::string s = m && m->num_transcripts ? transcript_text(m->transcripts[0]) : ""; STT_FreeMetadata(m); return s; }
  if (o.json) { Metadata* m = STT_SpeechToTextWithMetadata(ctx, buf, n, (unsigned)o.candidates); std::string s = m ? metadata_json(m, false) : ""; STT_FreeMetadata(m); return s; }
  if (o.stream > 0 || o.ext_stream > 0) {
    StreamingState* st = nullptr;
    if (STT_CreateStream(ctx, &st) != STT_ERR_OK) return "";
    const unsigned hop = (unsigned)(o.stream > 0 ? o.stream : o.ext_stream);
    std::string last; bool have_last = false;
    for (unsigned off = 0; off < n; off += hop) {
      STT_FeedAudioContent(st, buf + off, std::min(hop, n - off));
      std::string partial;
      if (o.stream > 0) { char* p = STT_IntermediateDecode(st); partial = p ? p : ""; STT_FreeString(p); }
      else { Metadata* m = STT_IntermediateDecodeWithMetadata(st, 1); partial = m && m->num_transcripts ? transcript_text(m->transcripts[0]) : ""; STT_FreeMetadata(m); }
      if (!have_last || partial != last) { printf("%s\n", partial.c_str()); last = partial; have_last = true; }  // only changes are printed (client.cc:216-222)
    }
    if (o.stream > 0) { char* p = STT_FinishStream(st); std::string s = p ? p : ""; STT_FreeString(p); return s; }
    Metadata* m = STT_FinishStreamWithMetadata(st, 1);
    std::string s = m && m->num_transcripts ? transcript_text(m->transcripts[0]) : "";
    STT_FreeMetadata(m);
    return s;
  }
  char* p = STT_SpeechToText(ctx, buf, n);
  std::string s = p ? p : "";
  STT_FreeString(p);
  return s;
}

}  // namespace

int main(int argc, char** argv) {
  Options o;
  if (!parse(argc, argv, o)) return 1;
  STTX_ConfigureRuntime();   // before the first HIP call: a hardware queue for each of the engine's streams (include/stt_amd.h)
  ModelState* ctx = nullptr;
  std::string model_bytes, scorer_bytes;  // must outlive the model when created from buffers (client.cc:491,519)
  int status;
  if (o.from_bytes) { model_bytes = slurp(o.model); status = STT_CreateModelFromBuffer(model_bytes.data(), (unsigned)model_bytes.size(), &ctx); }
  else status = STT_CreateModel(o.model.c_str(), &ctx);
  if (status != STT_ERR_OK) { char* e = STT_ErrorCodeToErrorMessage(status); fprintf(stderr, "Could not create model: %s\n", e); STT_FreeString(e); return 1; }
  if (o.set_beam && STT_SetModelBeamWidth(ctx, (unsigned)o.beam) != STT_ERR_OK) { fprintf(stderr, "Could not set model beam width.\n"); return 1; }
  if (!o.scorer.empty()) {
    if (o.from_bytes) { scorer_bytes = slurp(o.scorer); status = STT_EnableExternalScorerFromBuffer(ctx, scorer_bytes.data(), (unsigned)scorer_bytes.size()); }
    else status = STT_EnableExternalScorer(ctx, o.scorer.c_str());
    if (status != STT_ERR_OK) { char* e = STT_ErrorCodeToErrorMessage(status); fprintf(stderr, "Could not enable external scorer: %s\n", e); STT_FreeString(e); return 1; }
    if (o.set_ab && STT_SetScorerAlphaBeta(ctx, o.alpha, o.beta) != STT_ERR_OK) { fprintf(stderr, "Error setting scorer alpha and beta.\n"); return 1; }
  }
  if (!o.hot_words.empty()) {  // word:boost,word:boost (client.cc:539-555)
    std::stringstream ss(o.hot_words);
    for (std::string item; std::getline(ss, item, ',');) {
      const size_t colon = item.find(':');
      const std::string boost = colon == std::string::npos ? "" : item.substr(colon + 1);
      const bool valid = !boost.empty() && boost.find_first_not_of("-.0123456789") == std::string::npos;
      if (!valid || STT_AddHotWord(ctx, item.substr(0, colon).c_str(), strtof(boost.c_str(), nullptr)) != STT_ERR_OK) { fprintf(stderr, "Could not enable hot-word.\n"); return 1; }
    }
  }
  const int rate = STT_GetModelSampleRate(ctx);
  struct stat st;
  if (stat(o.audio.c_str(), &st) != 0) { printf("Error on stat: %s\n", o.audio.c_str()); STT_FreeModel(ctx); return 1; }
  std::vector<std::string> files;
  const bool is_dir = S_ISDIR(st.st_mode);
  if (is_dir) {
    printf("Running on directory %s\n", o.audio.c_str());
    if (DIR* dir = opendir(o.audio.c_str())) {
      while (dirent* e = readdir(dir)) { const std::string f = e->d_name; if (f.find(".wav") != std::string::npos) files.push_back(o.audio + "/" + f); }
      closedir(dir);
    }
    std::sort(files.begin(), files.end());
  } else files.push_back(o.audio);

  const bool plain = !(o.emissions || o.extended || o.json || o.stream > 0 || o.ext_stream > 0);
  const auto t0 = std::chrono::steady_clock::now();
  if (is_dir && plain && files.size() > 1) {  // one GPU batch for the whole directory
    std::vector<std::vector<short>> pcm(files.size());
    std::vector<const short*> ptrs; std::vector<unsigned> sizes; std::vector<size_t> idx;
    for (size_t i = 0; i < files.size(); ++i) if (read_wav(files[i], rate, pcm[i])) { ptrs.push_back(pcm[i].data()); sizes.push_back((unsigned)pcm[i].size()); idx.push_back(i); }
    char** out = ptrs.empty() ? nullptr : STTX_SpeechToTextBatch(ctx, ptrs.data(), sizes.data(), (unsigned)ptrs.size());
    for (size_t k = 0; out && k < idx.size(); ++k) printf("> %s\n%s\n", files[idx[k]].c_str(), out[k] ? out[k] : "");
    if (out) STTX_FreeStrings(out, (unsigned)ptrs.size());
  } else {
    for (const std::string& f : files) {
      std::vector<short> pcm;
      if (is_dir) printf("> %s\n", f.c_str());
      if (!read_wav(f, rate, pcm)) continue;
      const auto f0 = std::chrono::steady_clock::now();
      const std::string text = run_one(ctx, o, pcm);
      printf("%s\n", text.c_str());
      if (o.times) printf("wall_time_overall=%.05f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - f0).count());
    }
  }
  if (o.times && is_dir) printf("wall_time_directory=%.05f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  STT_FreeModel(ctx);
  return 0;
}
