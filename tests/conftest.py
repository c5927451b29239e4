"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, ABI surface (no GPU needed).
`-m gpu`: parity of the HIP path against the oracle through the C-ABI (needs a MI355X)."""
import json
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("STT_AMD_TEST_HOOKS", "1")      # before stt_amd.native is imported: the tests load libstt_test.so (libstt.so + include/stt_amd_test.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIX = os.path.join(ROOT, "tests", "golden", "fixtures")
GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "gpurun_out")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_port():
    from oracle import port
    if not port.available():
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    return port


@pytest.fixture(scope="session")
def port():
    return _ensure_port()


@pytest.fixture(scope="session")
def ref():
    from oracle import ref as r
    if not r.available():
        pytest.skip("oracle/_ref not built (needs /root/reference once: make -C oracle ref)")
    return r


@pytest.fixture(scope="session")
def fix():
    return FIX


@pytest.fixture(scope="session")
def english(port):
    labels, space = port.parse_alphabet_file(os.path.join(FIX, "alphabet.txt"))
    return labels, space


@pytest.fixture(scope="session")
def decoder_cases():
    with open(os.path.join(GOLD, "decoder_cases.json")) as f:
        cases = json.load(f)
    gold = np.load(os.path.join(GOLD, "decoder_golden.npz"))
    return cases, gold


def case_emissions(case):
    from stt_amd import synth
    s = case["sentence"]
    if case["mode"] == "bytes":
        lab, C, blank = [b - 1 for b in s.encode()], 256, 255
    else:
        lab, C, blank = [0 if ch == " " else (27 if ch == "'" else ord(ch) - ord("a") + 1) for ch in s], 29, 28
    T = case["T"] or (30 + 5 * len(lab))
    return synth.peaky_emissions(lab, T, C, blank, seed=case["seed"], noise=case["noise"])


def golden_results(gold, name):
    conf, lens = gold[name + "/conf"], gold[name + "/lens"]
    tok, ts = gold[name + "/tokens"], gold[name + "/timesteps"]
    out, o = [], 0
    for c, l in zip(conf, lens):
        out.append((float(c), tuple(int(x) for x in tok[o:o + l]), tuple(int(x) for x in ts[o:o + l])))
        o += l
    return out


def canon(results):
    """Order-insensitive form of an N-best list: the reference leaves the order of exact (score, character) ties to libstdc++."""
    return sorted((float(c), tuple(int(x) for x in t), tuple(int(x) for x in ts)) for c, t, ts in results)


def dump(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
