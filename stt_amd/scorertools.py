"""stt_amd/scorertools.py -- Python face of stt_amd/lib/stt_scorer_tools (stt_amd/tools/scorer_tools.cpp).

    generate_scorer_package(...)   restates native_client/generate_scorer_package.cpp:18-106 (same option names)
    synth_lm(...)                  synthetic KenLM `-a 255 -q 8 -v trie` binary for benchmarks (no corpus / lmplz offline)
"""
import os
import subprocess

from . import build as _build


def _tool():
    if not os.path.exists(_build.TOOLS_BIN):
        _build.build_tools(verbose=False)
    return _build.TOOLS_BIN


def generate_scorer_package(lm, vocab, package, alphabet=None, force_bytes_output_mode=False, default_alpha=0.0, default_beta=0.0):
    cmd = [_tool(), "package", "--lm", lm, "--vocab", vocab, "--package", package,
           "--default_alpha", repr(float(default_alpha)), "--default_beta", repr(float(default_beta))]
    cmd += ["--bytes"] if force_bytes_output_mode else ["--alphabet", alphabet]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return package


def synth_lm(out, vocab_out, words=100000, order=5, seed=1, avg=None, codepoints=False):
    """codepoints=True: the units are `words` distinct three-byte code points (a code-point level LM for a bytes-output scorer)."""
    cmd = [_tool(), "synth-lm", "--words", str(int(words)), "--order", str(int(order)), "--seed", str(int(seed)),
           "--out", out, "--vocab-out", vocab_out]
    if codepoints:
        cmd += ["--codepoints", "1"]
    for n, v in (avg or {}).items():
        cmd += ["--avg%d" % int(n), repr(float(v))]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out
