"""oracle/glibc_flt.h (restated glibc expf/logf) against the host libm -- the pin for the decoder's float arithmetic."""
import ctypes as C

import numpy as np


def _libm():
    m = C.CDLL("libm.so.6")
    m.expf.restype = C.c_float; m.expf.argtypes = [C.c_float]
    m.logf.restype = C.c_float; m.logf.argtypes = [C.c_float]
    return m


def test_expf_logf_match_host_libm(port):
    L = port.lib()
    rng = np.random.default_rng(0)
    # the decoder's operating ranges plus raw random bit patterns
    xs = np.concatenate([
        -np.abs(rng.standard_normal(200000) * 30).astype(np.float32),      # exp(x - xmax), x - xmax <= 0
        rng.uniform(-110, 90, 200000).astype(np.float32),
        rng.integers(0, 2**32, 400000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, -87.5, -88.0, -103.9, -104.0, 88.7, float.fromhex("-0x1.f8cbb2p+5"), float.fromhex("0x1.04845ep+5")], dtype=np.float32),
    ])
    xs = xs[~np.isnan(xs)]
    out = np.zeros_like(xs)
    L.port_expf_array(xs.ctypes.data, out.ctypes.data, len(xs))
    m = _libm()
    ref = np.array([m.expf(float(x)) for x in xs[:50000]], dtype=np.float32)
    assert np.array_equal(out[:50000].view(np.uint32), ref.view(np.uint32))
    # numpy's float32 exp is not glibc's; use libm for everything that is cheap enough, spot check the rest
    idx = rng.integers(0, len(xs), 50000)
    ref2 = np.array([m.expf(float(x)) for x in xs[idx]], dtype=np.float32)
    assert np.array_equal(out[idx].view(np.uint32), ref2.view(np.uint32))

    ys = np.concatenate([
        rng.uniform(1.0, 2.0, 200000).astype(np.float32),                   # log(exp(a) + exp(b)) argument
        (rng.uniform(0, 1, 200000).astype(np.float32) + np.float32(1.17549435e-38)),  # log(prob + FLT_MIN)
        np.abs(rng.integers(0, 2**31, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)),
        np.array([1.17549435e-38, 1.0, 2.0, 1e-45, 3.4e38], dtype=np.float32),
    ])
    ys = ys[~np.isnan(ys) & (ys > 0)]
    out = np.zeros_like(ys)
    L.port_logf_array(ys.ctypes.data, out.ctypes.data, len(ys))
    idx = rng.integers(0, len(ys), 60000)
    ref3 = np.array([m.logf(float(y)) for y in ys[idx]], dtype=np.float32)
    assert np.array_equal(out[idx].view(np.uint32), ref3.view(np.uint32))
