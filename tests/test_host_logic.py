"""Host-side logic that needs no GPU: alphabet formats, model container, frame bookkeeping, weight packing."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import am_ref
from conftest import ROOT
from stt_amd import modelfile, synth


def test_alphabet_text_formats(port, fix):
    labels, space = port.parse_alphabet_file(os.path.join(fix, "alphabet.txt"))
    assert len(labels) == 28 and space == 0 and labels[1] == b"a" and labels[27] == b"'"
    # tests/test_text.py of the reference: the three line-ending conventions parse identically
    got = [port.parse_alphabet_file(os.path.join(fix, "alphabet_%s.txt" % k))[0] for k in ("unix", "macos", "windows")]
    assert got[0] == got[1] == got[2] == [b"a", b"b", b"c"]
    assert synth.ENGLISH_LABELS == labels


def test_alphabet_binary_format_matches_reference(ref, fix):
    A = ref.Alphabet(os.path.join(fix, "alphabet.txt"))
    assert modelfile.serialize_alphabet(synth.ENGLISH_LABELS) == A.serialize()   # alphabet.cc:102-131


def test_frame_count_follows_stt_cc():
    # 5 s -> 250, LDC93S1 -> 146, < 512 samples -> 1
    assert [am_ref.n_frames_for(n) for n in (80000, 46797, 0, 100, 511, 512, 831, 832, 240000)] == [250, 146, 1, 1, 1, 2, 2, 3, 750]
    # explicit simulation of the feedAudioContent / flushBuffers buffer logic (stt.cc:105-128, 236-254)
    for n in (0, 1, 511, 512, 513, 832, 5000, 46797):
        buf, frames = 0, 0
        for _ in range(n):
            buf += 1
            if buf == 512:
                frames += 1
                buf -= 320
        frames += 1  # flush
        assert frames == am_ref.n_frames_for(n), n


def test_model_container_layout():
    w = synth.synth_weights(0, n_hidden=128)
    blob = modelfile.model_bytes(w, synth.ENGLISH_LABELS, beam_width=77)
    assert blob[:8] == b"STTAMDW1"
    hdr = struct.unpack("<10I", blob[8:48])
    assert hdr == (1, 26, 9, 128, 29, 16, 16000, 512, 320, 77)
    alen = struct.unpack("<I", blob[52:56])[0]
    off = 64 + ((alen + 7) // 8) * 8
    l1 = np.frombuffer(blob, dtype="<f4", count=494 * 128, offset=off).reshape(494, 128)
    assert np.array_equal(l1, w["layer_1/weights"])
    assert len(blob) == off + 4 * sum(int(np.prod(w[k].shape)) for k in modelfile.TENSOR_ORDER)


def _emulate_lstm_kernel(packed, hp, xproj_t, c, H, B, NT, UPW):
    """numpy emulation of lstm_step_kernel's index arithmetic under the documented MFMA 16x16x32 fragment layout:
    A[i = lane&15][k = 8*(lane>>4)+e], B[k][j = lane&15], D[i = 4*(lane>>4)+r][j = lane&15].  UPW hidden units (MT = UPW/4
    gate tiles of 16 rows) per workgroup."""
    ksteps = H // 128
    MT = UPW // 4
    z = np.zeros((H // UPW, MT * 16, NT * 16), dtype=np.float64)
    lane = np.arange(64)
    for wg in range(H // UPW):
        for q in range(4):
            for s in range(ksteps):
                for mt in range(MT):
                    a = packed[(((wg * 4 + q) * ksteps + s) * MT + mt)].astype(np.float64)   # [64 lanes][8]
                    A = np.zeros((16, 32)); A[(lane & 15)[:, None], (lane >> 4)[:, None] * 8 + np.arange(8)[None, :]] = a
                    for nt in range(NT):
                        bfr = hp[(q * ksteps + s) * NT + nt].astype(np.float64)
                        Bm = np.zeros((32, 16)); Bm[(lane >> 4)[:, None] * 8 + np.arange(8)[None, :], (lane & 15)[:, None]] = bfr
                        z[wg, mt * 16:(mt + 1) * 16, nt * 16:(nt + 1) * 16] += A @ Bm
    sig = lambda x: 1 / (1 + np.exp(-x))
    h_new = np.zeros((B, H)); c_new = np.zeros((B, H))
    for wg in range(H // UPW):
        for u in range(UPW):
            unit = wg * UPW + u
            for b in range(B):
                zi, zj, zf, zo = (z[wg, g * UPW + u, b] + xproj_t[b, g * H + unit] for g in range(4))
                cn = sig(zf) * c[b, unit] + sig(zi) * np.tanh(zj)
                c_new[b, unit] = cn; h_new[b, unit] = sig(zo) * np.tanh(cn)
    return h_new, c_new


def test_lstm_weight_packing_matches_kernel_indexing():
    """pack_lstm_recurrent_host (model.cpp) + the hp fragment order reproduce h.K[H:] under the kernel's indexing."""
    from stt_amd import native
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libstt.so not built")
    H, B, NT = 128, 5, 1
    rng = np.random.default_rng(3)
    kernel = rng.standard_normal((2 * H, 4 * H)).astype(np.float32)
    out = np.zeros(4 * H * H, dtype=np.uint16)
    assert native.lib().STTX_PackLstmRecurrent(kernel.ctypes.data, H, out.ctypes.data) == 0
    packed = out.view(np.float16).reshape(-1, 64, 8)
    h = rng.standard_normal((B, H)).astype(np.float16)
    hp = np.zeros((H // 32, NT, 64, 8), dtype=np.float16)   # pack_h_kernel's order
    for b in range(B):
        for k in range(H):
            hp[k >> 5, b >> 4, ((k & 31) >> 3) * 16 + (b & 15), k & 7] = h[b, k]
    xproj = rng.standard_normal((B, 4 * H))
    c = rng.standard_normal((B, H))
    upw = native.get_tuning("lstm_upw")                     # the library packs for the kernel shape it will launch (kernels.h: lstm_units_per_wg)
    upw = 16 if upw >= 16 else 8
    h_new, c_new = _emulate_lstm_kernel(packed, hp.reshape(-1, 64, 8), xproj, c, H, B, NT, upw)
    Kh = kernel[H:].astype(np.float16).astype(np.float64)
    z = xproj + h.astype(np.float64) @ Kh
    i, j, f, o = np.split(z, 4, axis=1)
    sig = lambda x: 1 / (1 + np.exp(-x))
    c_ref = sig(f) * c + sig(i) * np.tanh(j)
    h_ref = sig(o) * np.tanh(c_ref)
    np.testing.assert_allclose(c_new, c_ref, atol=1e-9)
    np.testing.assert_allclose(h_new, h_ref, atol=1e-9)


def test_am_oracle_self_consistency():
    """MFCC slow (op-order faithful) vs vectorised path; chunked (n_steps=16, carried state) vs whole-utterance forward."""
    spec = am_ref.MfccSpec()
    a = synth.synth_audio(9000, seed=5)
    np.testing.assert_allclose(am_ref.mfcc_utterance(a, spec), spec.frames_fast(a), atol=2e-5)
    w = am_ref.synth_weights(1, n_hidden=128)
    win = am_ref.context_windows(spec.frames_fast(a))
    full, cF, hF = am_ref.am_forward(win, w)
    parts, c, h = [], None, None
    for i in range(0, len(win), 16):
        p, c, h = am_ref.am_forward(win[i:i + 16], w, c0=c, h0=h)
        parts.append(p)
    np.testing.assert_allclose(np.concatenate(parts), full, atol=1e-6)
    assert np.allclose(full.sum(1), 1.0, atol=1e-5)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under stt_amd/ (python or C++) or include/ may import, include or link it; only
    tests/, bench.py's cpu_baseline leg and __graft_entry__ do."""
    import re
    bad = []
    for base in ("stt_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "lib" in dirpath.split(os.sep) or "__pycache__" in dirpath:
                continue
            for f in files:
                if not f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                    continue
                src = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<][^\">]*oracle/|libstt_oracle|libctcdecode_ref", src, re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle import") >= 1
    # ... and in bench.py only inside the cpu_baseline leg: cpu_baseline() (the timed CPU baseline) and cpu_baseline_reference_decoder()
    # (the same reference build as the after-the-clock checker of the timed batches)
    inside = 0
    for chunk in bench.split("\ndef ")[1:]:
        if chunk.startswith("cpu_baseline"):
            inside += chunk.count("from oracle import")
    assert bench.count("from oracle import") == inside
    # the checker runs after the clock: its only call site sits below the line that stops the timed region
    measure = bench[bench.index("def measure"):]
    assert measure.count("cpu_baseline_reference_decoder(") == 1
    assert measure.index("cpu_baseline_reference_decoder(") > measure.index("elapsed = time.perf_counter() - t0")


def test_bench_checks_fail_the_run_on_a_forged_mismatch():
    """bench.py's verdict logic, without a GPU: an utterance that differs from the reference is accepted only when the restatement shows a
    boundary tie AND gives the timed output; anything else is `unexplained`, `verified` false, exit code 3 -- also for a side workload."""
    import bench
    good = [{"id": i, "got_text": "a b", "got_conf": -1.5, "want_text": "a b", "want_conf": -1.5, "against": "reference"} for i in range(4)]
    ok, counts, mm = bench.judge_against_reference(good, {})
    assert ok and counts["equal"] == 4 and counts["unexplained"] == 0 and not mm
    # a forged transcript mismatch, no tie information: unexplained
    forged = [dict(good[0], got_text="a c")] + good[1:]
    ok, counts, mm = bench.judge_against_reference(forged, {})
    assert not ok and counts["unexplained"] == 1 and mm[0]["explained_by_a_boundary_tie"] is False
    # the same difference WITH a boundary tie and the restatement agreeing with the timed output: explained
    ok, counts, mm = bench.judge_against_reference(forged, {0: (2, "a c", -1.5)})
    assert ok and counts["differ_with_a_boundary_tie_and_equal_to_the_restatement"] == 1 and counts["unexplained"] == 0
    # ... with the reference-order restatement's verdict: it must reproduce the REFERENCE's output for the difference to count as the order effect
    ok, counts, _ = bench.judge_against_reference(forged, {0: (2, "a c", -1.5, "a b", -1.5)})
    assert ok and counts["of_those_the_reference_reproduced_in_reference_order"] == 1
    ok, counts, _ = bench.judge_against_reference(forged, {0: (2, "a c", -1.5, "a x", -1.5)})
    assert not ok and counts["unexplained"] == 1
    # ... a tie, but the restatement does not give the timed output either: a real mismatch
    ok, counts, _ = bench.judge_against_reference(forged, {0: (2, "a d", -1.5)})
    assert not ok and counts["unexplained"] == 1
    # ... the restatement agrees but saw no tie: a real mismatch (the restatement itself is off the reference without its one excuse)
    ok, counts, _ = bench.judge_against_reference(forged, {0: (0, "a c", -1.5)})
    assert not ok
    # a confidence that differs in the last bit is a mismatch as well
    ok, counts, _ = bench.judge_against_reference([dict(good[0], got_conf=-1.5000000000000002)], {})
    assert not ok
    # differences against a blocking call can never be excused by a tie
    ok, _, _ = bench.judge_against_reference([dict(good[0], got_text="x", against="blocking")], {0: (1, "x", -1.5)})
    assert not ok
    line = {"verified": True, "verify_counts": {"unexplained": 0}, "workloads": {"peaky": {"verified": True, "verify_counts": {"unexplained": 0}}}}
    assert bench.exit_code(line) == 0
    assert bench.exit_code(dict(line, verified=False)) == 3
    assert bench.exit_code(dict(line, verified=None)) == 3
    assert bench.exit_code(dict(line, verify_counts={"unexplained": 1})) == 3
    assert bench.exit_code(dict(line, workloads={"stream": {"verified": False}})) == 3
    assert bench.exit_code(dict(line, workloads={"bytes": {"error": "RuntimeError('x')"}})) == 3


def test_bench_final_line_stays_short_enough_for_the_driver():
    """Round 5's line grew to 40 KB and the driver recorded `parsed: null`.  The LAST stdout line is compact_line(res): always JSON, always
    under bench.FINAL_LINE_LIMIT, always with the contract's keys, `roofline` and `cpu_baseline` -- whatever the checks found (here: round 5's
    own full result with 64 forged mismatch records of two 100-character transcripts in every line and sub-line)."""
    import json
    import bench
    res = json.load(open(os.path.join(ROOT, "profiles", "r05_c_bench_line.json")))
    mm = [{"id": [i, 0, i], "against": "reference", "got": "g" * 100, "want": "w" * 100, "got_confidence": -1.0 * i, "want_confidence": -2.0 * i,
           "boundary_tie_steps": 3, "equals_the_restatement": True, "reference_reproduced_by_the_reference_order_restatement": True,
           "explained_by_a_boundary_tie": True} for i in range(64)]
    res["verify_mismatches"] = mm
    res["verified_what"] = "x" * 5000
    for r in res["workloads"].values():
        r["verify_mismatches"] = list(mm)
        r["verified_what"] = "y" * 5000
    res["workloads"]["broken"] = {"error": "RuntimeError(%r)" % ("z" * 4000)}
    assert len(json.dumps(res)) > 100000
    line = json.dumps(bench.compact_line(res))
    assert len(line) < bench.FINAL_LINE_LIMIT < 8192
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "verified", "verify_counts", "p50_utterance_latency_ms", "roofline", "cpu_baseline", "workloads"):
        assert k in back, k
    assert back["value"] == round(res["value"], 6) and back["ms_per_step"] == round(res["ms_per_step"], 6)
    assert isinstance(back["config"]["workload"], str) and "model" not in back["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"):
        assert k in back["roofline"], k
    assert back["roofline"]["bound"] in ("hbm", "mfma") and abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-6
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["cpu_baseline"]["kind"] in ("reference", "port") and "value" in back["cpu_baseline"]["end_to_end"]
    for w in ("batch_i8", "ragged", "stream", "bytes", "peaky", "peaky_bytes"):
        for k in ("value", "ms_per_step", "verified", "verified_against", "unexplained", "roofline_frac"):
            assert k in back["workloads"][w], (w, k)
    assert "error" in back["workloads"]["broken"]
    # what exit_code() reads survives the cut: a failed check is still visible in the short line
    assert back["verify_counts"]["unexplained"] == res["verify_counts"]["unexplained"]
    # a much smaller limit still yields valid JSON with the headline
    tiny = bench.compact_line(res, limit=3500)
    assert len(json.dumps(tiny)) <= 3500 and tiny["value"] == back["value"] and "roofline" in tiny and "cpu_baseline" in tiny
    # and main() prints it LAST, after the detail
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('print("bench_detail: " + detail)') < src.index("print(json.dumps(compact_line(res)")
