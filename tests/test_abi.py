"""The C-ABI shared library: loads without a GPU, exports every symbol the headers declare, keeps the reference's
error contract for calls that fail before any GPU work."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def native():
    from stt_amd import native as n
    if not os.path.exists(n.LIB_PATH):
        from stt_amd import build
        build.build(verbose=False)
    return n


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"STTX?_EXPORT\s+[\w\s\*]*?\b(STTX?_[A-Za-z]+)\s*\(", src)))


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_exports_every_declared_symbol(native):
    """libstt.so (what ships) exports coqui-stt.h + stt_amd.h and NOTHING else -- no test hook, no probe kernel; libstt_test.so (what the
    tests load) the same plus include/stt_amd_test.h."""
    a, b, t = _declared("coqui-stt.h"), _declared("stt_amd.h"), _declared("stt_amd_test.h")
    assert len(a) == 29, a            # reference coqui-stt.h:136-504 exports 29 functions
    assert sorted(a) == sorted(native.COQUI_STT_H)
    assert sorted(b) == sorted(native.STT_AMD_H)
    assert sorted(t) == sorted(native.STT_AMD_TEST_H) and all(re.match(r"STTX_(Test|Debug)", n) for n in t)
    assert not [n for n in b if re.match(r"STTX_(Test|Debug)", n)]
    prod = _exported(native.PRODUCT_LIB_PATH)
    assert {s for s in prod if s.startswith(("STT_", "STTX_"))} == set(a + b)
    assert {s for s in _exported(native.TEST_LIB_PATH) if s.startswith(("STT_", "STTX_"))} == set(a + b + t)
    # the shipped library carries no timing-probe kernels either (they are -DSTT_TEST_HOOKS instantiations: wrong results by design)
    blob = open(native.PRODUCT_LIB_PATH, "rb").read()
    assert b"lstm_probe4" not in blob and b"lstm_probe8" not in blob and b"lstm_i8_probe" not in blob      # (mangled kernel names; "lstm_probe" alone is a tunable's name)
    assert b"lstm_probe8" in open(native.TEST_LIB_PATH, "rb").read()
    # the tests run on the hooks build, everything else on the product
    assert native.TEST_HOOKS and native.LIB_PATH == native.TEST_LIB_PATH


def test_error_messages_and_version(native):
    L = native.lib()
    assert native.take_string(L.STT_Version()) == b"1.4.0"
    assert native.error_message(0x0000) == "No error."
    assert native.error_message(0x2004) == "External scorer is not enabled."
    assert native.error_message(0x3010) == "Could not erase hot-word."
    assert native.error_message(0x7777).startswith("Unknown error")


def test_create_model_errors_without_gpu_work(native, capfd):
    L = native.lib()
    h = C.c_void_p(123)
    assert L.STT_CreateModel(b"", C.byref(h)) == 0x1000 and not h.value      # STT_ERR_NO_MODEL, stt.cc:354-357
    assert L.STT_CreateModel(b"/definitely/not/here", C.byref(h)) != 0 and not h.value
    err = capfd.readouterr().err
    assert "TensorFlow:" in err and "Coqui STT:" in err                      # version lines, stt.cc:344-345
    assert L.STT_CreateModelFromBuffer(None, 0, C.byref(h)) == 0x1000


def test_struct_layout_matches_header(native):
    # field order/types are ABI (coqui-stt.h:29-86)
    assert [f[0] for f in native.TokenMetadata._fields_] == ["text", "timestep", "start_time"]
    assert C.sizeof(native.TokenMetadata) == 16 and C.sizeof(native.CandidateTranscript) == 24
    assert C.sizeof(native.Metadata) == 24 and C.sizeof(native.AcousticModelEmissions) == 32
