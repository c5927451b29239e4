"""TensorFlow Lite's hybrid FULLY_CONNECTED on the int8 matrix cores (kernels.h: launch_quantize_rows + launch_dense_hybrid_i8, through the
STTX_TestDenseHybrid hook) against its restatement oracle/am_hybrid.py (fully_connected.cc EvalHybrid; portable_tensor_utils.cc
PortableSymmetricQuantizeFloats / MatrixBatchVectorMultiplyAccumulate).  Integer dot products are exact and the float operations around
them are the reference's, in its order: the bar is BIT EQUALITY of the quantised rows, their scales and the f32 outputs."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import am_hybrid

pytestmark = pytest.mark.gpu


def _run(x, wq, wscale, bias, reps=0, epi=0, clip=0.0):
    from stt_amd import native
    L = native.lib()
    M, K = x.shape
    N = wq.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float32); wq = np.ascontiguousarray(wq, dtype=np.int8)
    wscale = np.ascontiguousarray(wscale, dtype=np.float32); bias = np.ascontiguousarray(bias, dtype=np.float32)
    y = np.zeros((M, N), dtype=np.float32); q = np.zeros((M, K), dtype=np.int8); rs = np.zeros(M, dtype=np.float32)
    ms = C.c_float(0)
    rc = L.STTX_TestDenseHybrid(x.ctypes.data, M, K, wq.ctypes.data, wscale.ctypes.data, len(wscale), bias.ctypes.data, N, y.ctypes.data, q.ctypes.data, rs.ctypes.data,
                                  reps, C.byref(ms), epi, C.c_float(clip))
    assert rc == 0, hex(rc)
    return y, q, rs, float(ms.value)


@pytest.fixture(params=[3, 4], ids=["two stages of K = 128", "four stages of K = 64"])
def wide_form(request):
    """Both staging forms of the 128 x 256 tile (tunable dense_solo; kernels_am.hip: dense_wide_kernel<EPI, NS>): integer sums, same bits."""
    from stt_amd import native
    native.set_tuning("dense_solo", request.param)
    yield request.param
    native.set_tuning("dense_solo", 3)


@pytest.mark.parametrize("per_channel", [False, True])
def test_hybrid_fully_connected_is_bit_equal_to_the_restatement(per_channel, wide_form):
    rng = np.random.default_rng(21)
    M, K, N = 300, 2048, 512                      # (M not a multiple of the 128-row tile)
    x = (rng.standard_normal((M, K)) * rng.uniform(0.01, 6.0, size=(M, 1))).astype(np.float32)
    x[7] = 0.0                                    # IsZeroVector / range == 0: zeros, scale 1 -> the output row is the bias
    x[11, 5] = 20.0                               # layer outputs are clipped at 20 (relu_clip)
    x[13] = np.round(x[13] * 4) / 4               # many exact .5 products: round half away from zero, not to even
    x[13, 0] = 127.0 / 4
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    wq, wscale = am_hybrid.quantize_weights(w, per_channel)
    bias = rng.standard_normal(N).astype(np.float32)
    y, q, rs, _ = _run(x, wq, wscale, bias)
    q_ref, sf_ref = am_hybrid.symmetric_quantize_rows(x)
    assert np.array_equal(q.astype(np.float64), q_ref)
    assert np.array_equal(rs, sf_ref)
    want = am_hybrid.fully_connected_hybrid(x, wq, wscale, bias)
    assert np.array_equal(y[7], bias)
    assert np.array_equal(y, want), float(np.abs(y - want).max())


def test_hybrid_gemm_at_the_bench_shape(wide_form):
    """The x-projection of one 48-frame chunk of 128 rows (M = 6144, K = 2048, N = 8192), int32 sums beyond 2^24 included (the
    int -> float conversion rounds like the reference's); timed, and written beside the f16 form's figure for DESIGN.md 7.1."""
    rng = np.random.default_rng(22)
    M, K, N = 6144, 2048, 8192
    x = np.minimum(np.maximum(rng.standard_normal((M, K)) * 3.0, 0.0), 20.0).astype(np.float32)     # what layer 3 hands on: clipped ReLU
    x[:64] = 20.0 * (rng.random((64, K)) > 0.02)                                                      # nearly saturated rows: |sum| > 2^24
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    w[:, :64] = 0.2 + 0.01 * rng.standard_normal((K, 64))          # quantised to ~100 of 127: sums of 2048 x 127 x 100 = 2.6e7 > 2^24
    wq, wscale = am_hybrid.quantize_weights(w, False)
    bias = rng.standard_normal(N).astype(np.float32)
    y, q, rs, ms = _run(x, wq, wscale, bias, reps=20)
    rows = np.r_[0:96, rng.choice(M, 160, replace=False)]
    want = am_hybrid.fully_connected_hybrid(x[rows], wq, wscale, bias)
    acc = am_hybrid.symmetric_quantize_rows(x[:64])[0] @ wq[:64].astype(np.float64).T
    assert np.abs(acc).max() > 2 ** 24
    assert np.array_equal(y[rows], want), float(np.abs(y[rows] - want).max())
    out = {"M": M, "K": K, "N": N, "ms_quantise_plus_product": ms, "int8_TOP_s": 2.0 * M * K * N / (ms * 1e-3) / 1e12,
           "note": "row quantisation + 128 x 256 tile on v_mfma_i32_16x16x64_i8, alone on the chip; the f16 form of the same product: benchmarks / DESIGN.md 8.3 (0.94 PF/s alone)"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out["dense_solo"] = wide_form
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "hybrid_i8_gemm_form%d.json" % wide_form), "w"))


@pytest.mark.parametrize("M", [16, 5, 40])
def test_hybrid_layer_forms_with_the_clipped_relu(M):
    """What layers 1-3 and 5 of the int8 model path launch: the skinny form (M <= 16: one stream's chunk) and the 128 x 256 tile with
    rows clamped (a few streams), clipped-ReLU epilogue; bit-equal to the restatement's _dense."""
    rng = np.random.default_rng(23 + M)
    K, N = 512, 256
    x = (rng.standard_normal((M, K)) * rng.uniform(0.1, 9.0, size=(M, 1))).astype(np.float32)
    w = (rng.standard_normal((K, N)) * 0.2).astype(np.float32)
    wq, wscale = am_hybrid.quantize_weights(w, False)
    bias = rng.standard_normal(N).astype(np.float32)
    y, q, rs, _ = _run(x, wq, wscale, bias, epi=1, clip=20.0)
    want = np.minimum(np.maximum(am_hybrid.fully_connected_hybrid(x, wq, wscale, bias), np.float32(0)), np.float32(20))
    assert (want == 20).any() and (want == 0).any()
    assert np.array_equal(y, want), float(np.abs(y - want).max())
