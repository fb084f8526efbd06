"""include/e3d_libm.h -- the bit-defined atanf / atan2f / sinf / cosf / tanf / log2f that kernels, host code and oracle share.

CPU part: the host evaluation (through the oracle's test hook) against (i) the f64 library function rounded once to f32
(what "correctly rounded" looks like up to double rounding) and (ii) the C library's float functions.
GPU part: the HIP kernel returns the same bits as the host for every input, special values included.
"""
import numpy as np
import pytest


def _inputs(fn, n=400_000, seed=1):
    rs = np.random.RandomState(seed)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38,
                        -3.4028235e38, 0.4375, 0.6875, 1.1875, 2.4375, np.pi / 4, np.pi / 2, np.pi, 1e-8, 7.0, 1e5], np.float32)
    if fn in ("sinf", "cosf", "tanf"):
        x = np.concatenate([rs.uniform(-np.pi, np.pi, n // 2), rs.uniform(-200, 200, n // 4), rs.normal(0, 1e-3, n // 8),
                            rs.uniform(-1e5, 1e5, n // 16), np.exp(rs.uniform(0, 88, n // 16)) * rs.choice([-1, 1], n // 16)])
    elif fn == "log2f":
        x = np.concatenate([np.exp(rs.uniform(-80, 80, n // 2)), rs.uniform(0.5, 2.0, n // 4), rs.uniform(0, 100, n // 4)])
    else:
        x = np.concatenate([rs.normal(0, 1, n // 2), rs.uniform(-20, 20, n // 4), np.exp(rs.uniform(-40, 40, n // 4)) * rs.choice([-1, 1], n // 4)])
    x = np.concatenate([special, x.astype(np.float32)])
    y = np.concatenate([special[::-1], rs.normal(0, 1, x.size - special.size).astype(np.float32)])
    if fn == "atan2f":       # the pipeline's call: atan2(r, 1)
        y[special.size + 1000:special.size + 50000] = 1.0
    return x, y


REF64 = {"atanf": lambda x, y: np.arctan(x.astype(np.float64)), "atan2f": lambda x, y: np.arctan2(x.astype(np.float64), y.astype(np.float64)),
         "sinf": lambda x, y: np.sin(x.astype(np.float64)), "cosf": lambda x, y: np.cos(x.astype(np.float64)),
         "tanf": lambda x, y: np.tan(x.astype(np.float64)), "log2f": lambda x, y: np.log2(x.astype(np.float64))}
REF32 = {"atanf": lambda x, y: np.arctan(x), "atan2f": lambda x, y: np.arctan2(x, y), "sinf": lambda x, y: np.sin(x),
         "cosf": lambda x, y: np.cos(x), "tanf": lambda x, y: np.tan(x), "log2f": lambda x, y: np.log2(x)}


def _ulp_diff(a, b):
    """distance in units of the last place between two f32 arrays (NaN == NaN, +0 == -0 only if same sign bit)."""
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai); bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    d = np.abs(ai - bi)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


@pytest.mark.parametrize("fn", ["atanf", "atan2f", "sinf", "cosf", "tanf", "log2f"])
def test_host_libm_is_the_rounded_f64_value(ob, fn):
    x, y = _inputs(fn)
    with np.errstate(all="ignore"):
        got = ob.libm_eval(fn, x, y)
        want = REF64[fn](x, y).astype(np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    d = _ulp_diff(got, want)
    # f64 library value rounded once: identical except where two < 1 ulp(f64) results straddle an f32 rounding boundary
    assert d.max() <= 1, (fn, d.max(), x[d.argmax()], y[d.argmax()])
    assert (d > 0).mean() < 1e-5, (fn, (d > 0).mean())
    # glibc's float functions (what the reference calls) are within 2 ulp of these values and mostly identical
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    f = getattr(libm, fn)
    f.restype = C.c_float
    f.argtypes = [C.c_float, C.c_float] if fn == "atan2f" else [C.c_float]
    idx = np.flatnonzero(np.isfinite(x) & np.isfinite(y))
    idx = idx[np.linspace(0, idx.size - 1, 40000).astype(np.int64)]
    with np.errstate(all="ignore"):
        libc = np.array([f(float(x[i]), float(y[i])) if fn == "atan2f" else f(float(x[i])) for i in idx], np.float32)
    dl = _ulp_diff(got[idx], libc)
    assert dl.max() <= 2, (fn, dl.max(), x[idx][dl.argmax()])
    assert (dl > 0).mean() < 0.25, (fn, (dl > 0).mean())
    # signed zeros and exact cases
    if fn == "log2f":
        p = np.float32(2.0) ** np.arange(-120, 120, dtype=np.float32)
        assert np.array_equal(ob.libm_eval(fn, p), np.arange(-120, 120, dtype=np.float32))
    if fn in ("atanf", "sinf", "tanf"):
        z = ob.libm_eval(fn, np.array([0.0, -0.0], np.float32))
        assert z.view(np.uint32).tolist() == [0, 0x80000000]


@pytest.mark.gpu
@pytest.mark.parametrize("fn", ["atanf", "atan2f", "sinf", "cosf", "tanf", "log2f"])
def test_device_libm_bit_identical_to_host(e3d, ob, fn):
    x, y = _inputs(fn, n=2_000_000, seed=7)
    # denser coverage of the arguments the kernels actually see
    if fn in ("sinf", "cosf"):
        x = np.concatenate([x, np.linspace(0, np.pi / 3, 1_000_001, dtype=np.float32)]); y = np.concatenate([y, np.zeros(1_000_001, np.float32)])
    if fn in ("atan2f", "atanf", "tanf"):
        x = np.concatenate([x, np.linspace(0, 3.0, 1_000_001, dtype=np.float32)]); y = np.concatenate([y, np.ones(1_000_001, np.float32)])
    dev = e3d.libm_eval(fn, x, y)
    host = ob.libm_eval(fn, x, y)
    nan = np.isnan(host)
    assert np.array_equal(np.isnan(dev), nan)
    assert np.array_equal(dev.view(np.uint32)[~nan], host.view(np.uint32)[~nan])
