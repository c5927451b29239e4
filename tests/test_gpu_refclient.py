"""Reference-side code on the drop-in boundary.

oracle/_ref/ref_client is the REFERENCE's own command-line client -- native_client/client.cc compiled where it lies, against
the reference's own coqui-stt.h, with its built-in RIFF reader (-DNO_SOX, client.cc:390-426) -- linked with
stt_amd/lib/libstt.so (oracle/Makefile: refclient; built in the container that has /root/reference, the binary travels
with the snapshot).  The scenarios of ci_scripts/asserts.sh:393-604 are replayed through it and through this repository's
`stt` client; stdout must agree byte for byte, and with the Python mirror of the binding.  No released model exists
offline, so the transcripts are those of the seeded synthetic model, not the LDC93S1 sentence.

native_client/test/concurrent_streams.py (asserts.sh:434-452) is restated on the ctypes binding: two interleaved streams
on one model give the transcripts of the two files decoded alone."""
import json
import os
import subprocess
import wave

import numpy as np
import pytest

from conftest import ROOT
from stt_amd import build, modelfile, synth

pytestmark = pytest.mark.gpu
REF_CLIENT = os.path.join(ROOT, "oracle", "_ref", "ref_client")


@pytest.fixture(scope="module")
def setup(tmp_path_factory, fix):
    if not os.path.exists(REF_CLIENT):
        pytest.skip("oracle/_ref/ref_client not built (needs /root/reference once: make -C oracle refclient)")
    from stt_amd import Model
    d = tmp_path_factory.mktemp("refcli")
    w = synth.synth_weights(21, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    path = str(d / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=64)
    build.build_tools(verbose=False)
    m = Model(path)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    wavdir = d / "wavs"; wavdir.mkdir()
    for i, n in enumerate([24000, 40000]):
        with wave.open(str(wavdir / ("u%d.wav" % i)), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(synth.synth_audio(n, seed=60 + i).tobytes())
    return m, path, os.path.join(fix, "pruned_lm.scorer"), str(wavdir), os.path.join(fix, "LDC93S1_pcms16le_1_16000.wav")


def _run(binary, args):
    r = subprocess.run([binary] + args, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_reference_client_runs_unmodified_on_libstt(setup):
    m, model, scorer, wavdir, ldc = setup
    with wave.open(ldc, "rb") as f:
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    base = ["--model", model, "--audio", ldc]
    lm = ["--scorer", scorer]
    scenarios = {
        "nolm": base,                                                              # asserts.sh:400-404
        "nolm_extended": base + ["--extended"],                                    # :406-410
        "lm": base + lm,                                                           # :412-416
        "lm_json": base + lm + ["--json", "--candidate_transcripts", "3"],
        "lm_stream": base + lm + ["--stream", "5120"],
        "lm_extended_stream": base + lm + ["--extended_stream", "5120"],
        "lm_hot_words": base + lm + ["--hot_words", "she:5.0,dark:-2.5"],
        "lm_init_from_bytes": base + lm + ["--init_from_bytes"],
        "lm_beam_alpha_beta": base + lm + ["--beam_width", "32", "--lm_alpha", "0.9", "--lm_beta", "1.2"],
        "lm_keep_emissions": base + lm + ["--json", "--keep_emissions"],
    }
    for name, args in scenarios.items():
        rc_r, out_r, err_r = _run(REF_CLIENT, args)
        rc_o, out_o, err_o = _run(build.CLIENT_BIN, args)
        assert rc_r == 0 and rc_o == 0, (name, rc_r, rc_o, err_r[-300:], err_o[-300:])
        assert out_r == out_o, (name, out_r[:400], out_o[:400])                    # byte for byte
        assert "TensorFlow:" in err_r and "Coqui STT:" in err_r                    # the version lines CI greps (asserts.sh:284-321)
    # and the binding agrees with what the reference client printed
    rc, out, _ = _run(REF_CLIENT, base + lm)
    assert out.rstrip("\n") == m.stt(pcm)
    rc, out, _ = _run(REF_CLIENT, base)
    m.disableExternalScorer()
    try:
        assert out.rstrip("\n") == m.stt(pcm)
    finally:
        m.enableExternalScorer(scorer)
    rc, out, _ = _run(REF_CLIENT, base + lm + ["--json", "--candidate_transcripts", "3"])
    j = json.loads(out)
    md = m.sttWithMetadata(pcm, 3)
    assert len(j["alternatives"]) == len(md["transcripts"]) - 1
    # directory mode (client.cc:597-625; readdir order is the file system's, so compare per file)
    rc, out, _ = _run(REF_CLIENT, ["--model", model] + lm + ["--audio", wavdir])
    lines = out.rstrip("\n").split("\n")
    assert rc == 0 and lines[0].startswith("Running on directory")
    got = {lines[i][2:]: lines[i + 1] for i in range(1, len(lines), 2)}
    for p, text in got.items():
        with wave.open(p, "rb") as f:
            assert text == m.stt(np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16))
    # error contract seen by the reference client (client.cc:455-461, 505-511)
    rc, _, err = _run(REF_CLIENT, ["--model", "/no/such/model", "--audio", ldc])
    assert rc == 1 and "Could not create model" in err
    rc, _, err = _run(REF_CLIENT, ["--model", model, "--scorer", "/no/such/scorer", "--audio", ldc])
    assert rc == 1 and "Could not enable external scorer" in err


def test_concurrent_streams_like_the_reference_test(setup):
    """native_client/test/concurrent_streams.py:43-54: two streams of one model fed alternately in ten parts each."""
    m, model, scorer, wavdir, ldc = setup
    with wave.open(ldc, "rb") as f:
        audio1 = np.frombuffer(f.readframes(f.getnframes()), np.int16)
    audio2 = synth.synth_audio(52000, seed=77)
    s1, s2 = m.createStream(), m.createStream()
    for p1, p2 in zip(np.array_split(audio1, 10), np.array_split(audio2, 10)):
        s1.feedAudioContent(p1)
        s2.feedAudioContent(p2)
    assert s1.finishStream() == m.stt(audio1)
    assert s2.finishStream() == m.stt(audio2)
