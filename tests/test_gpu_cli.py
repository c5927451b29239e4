"""The `stt` command-line client (stt_amd/tools/stt_client.cpp) against the Python mirror on the same model:
the scenarios of ci_scripts/asserts.sh (plain, --extended, --json, --stream, directory, hot words, bytes init)."""
import json
import os
import subprocess
import wave

import numpy as np
import pytest

from stt_amd import build, modelfile, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(tmp_path_factory, fix):
    from stt_amd import Model
    d = tmp_path_factory.mktemp("cli")
    w = synth.synth_weights(21, n_hidden=256)
    w["layer_6/weights"] = (w["layer_6/weights"] * 6.0).astype(np.float32)
    path = str(d / "m.sttw")
    modelfile.write_model(path, w, synth.ENGLISH_LABELS, beam_width=64)
    build.build_tools(verbose=False)
    assert os.path.exists(build.CLIENT_BIN)
    m = Model(path)
    m.enableExternalScorer(os.path.join(fix, "pruned_lm.scorer"))
    wavdir = d / "wavs"; wavdir.mkdir()
    audios = {}
    for i, n in enumerate([24000, 40000, 9000]):
        a = synth.synth_audio(n, seed=60 + i)
        p = str(wavdir / ("u%d.wav" % i))
        with wave.open(p, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(a.tobytes())
        audios[p] = a
    return m, path, os.path.join(fix, "pruned_lm.scorer"), str(wavdir), audios, os.path.join(fix, "LDC93S1_pcms16le_1_16000.wav")


def _run(args):
    r = subprocess.run([build.CLIENT_BIN] + args, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_cli_plain_extended_json_and_stream(setup):
    m, model, scorer, wavdir, audios, ldc = setup
    with wave.open(ldc, "rb") as f:
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    want = m.stt(pcm)
    rc, out, err = _run(["--model", model, "--scorer", scorer, "--audio", ldc])
    assert rc == 0 and out.rstrip("\n") == want
    assert "Coqui STT" in err and "TensorFlow" in err                         # version lines on stderr (asserts.sh:284-321)
    rc, out, _ = _run(["--model", model, "--scorer", scorer, "--audio", ldc, "--extended", "-t"])
    lines = out.rstrip("\n").split("\n")
    assert rc == 0 and lines[0] == want and lines[1].startswith("wall_time_overall=")
    rc, out, _ = _run(["--model", model, "--scorer", scorer, "--audio", ldc, "--json", "--candidate_transcripts", "2"])
    j = json.loads(out)
    assert rc == 0 and " ".join(w["word"] for w in j["words"]).strip() == want.strip() and len(j["alternatives"]) == 1
    md = m.sttWithMetadata(pcm, 2)
    assert abs(j["metadata"]["confidence"] - md["transcripts"][0]["confidence"]) < 1e-3
    rc, out, _ = _run(["--model", model, "--scorer", scorer, "--audio", ldc, "--stream", "5120"])
    assert rc == 0 and out.rstrip("\n").split("\n")[-1] == want
    rc, out, _ = _run(["--model", model, "--scorer", scorer, "--audio", ldc, "--extended_stream", "5120", "--init_from_bytes"])
    assert rc == 0 and out.rstrip("\n").split("\n")[-1] == want
    m.addHotWord("she", 5.0)
    boosted = m.stt(pcm)
    m.clearHotWords()
    rc, out, _ = _run(["--model", model, "--scorer", scorer, "--audio", ldc, "--hot_words", "she:5.0"])
    assert rc == 0 and out.rstrip("\n") == boosted


def test_cli_directory_is_one_batch_and_errors(setup):
    m, model, scorer, wavdir, audios, ldc = setup
    rc, out, _ = _run(["--model", model, "--scorer", scorer, "--audio", wavdir])
    assert rc == 0
    lines = out.rstrip("\n").split("\n")
    assert lines[0].startswith("Running on directory")
    got = {lines[i][2:]: lines[i + 1] for i in range(1, len(lines), 2)}
    assert got == {p: m.stt(a) for p, a in audios.items()}
    rc, _, err = _run(["--model", "/no/such/model", "--audio", ldc])
    assert rc == 1 and "Could not create model" in err
    rc, _, err = _run(["--model", model, "--scorer", "/no/such/scorer", "--audio", ldc])
    assert rc == 1 and "Could not enable external scorer" in err
    rc, out, _ = _run(["--model", model, "--audio", ldc, "--stream", "1000"])
    assert rc == 1 and "multiples of 160" in out
