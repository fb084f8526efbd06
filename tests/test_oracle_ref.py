"""The oracle against the reference itself, for the two headers of path (B) that compile from their own sources in this image
(oracle/_ref/libe3d_ref.so, built by oracle/Makefile from /root/reference/src/opt/robust_weighting.h and descriptor.h where the
reference is present; the prebuilt library travels to the GPU box).  Nothing here reads /root/reference."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding as ob
from oracle import reg_binding as rb


@pytest.fixture(scope="module")
def ref():
    ob.build()
    path = ob.ref_lib_path()
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libe3d_ref.so not built (the reference is not present in this environment)")
    L = C.CDLL(path)
    L.e3d_ref_robust_many.restype = None
    L.e3d_ref_robust_many.argtypes = [C.c_int, C.c_float, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    L.e3d_ref_compute_descriptor.restype = C.c_float
    L.e3d_ref_compute_descriptor.argtypes = [C.c_float, C.c_float]
    return L


def _oracle_many(kind, parameter, residuals):
    L = rb.lib()
    L.oracle_reg_robust_weight.restype = C.c_float; L.oracle_reg_robust_weight.argtypes = [C.c_int, C.c_float, C.c_float]
    L.oracle_reg_robust_residual.restype = C.c_float; L.oracle_reg_robust_residual.argtypes = [C.c_int, C.c_float, C.c_float]
    w = np.array([L.oracle_reg_robust_weight(kind, parameter, float(r)) for r in residuals], np.float32)
    p = np.array([L.oracle_reg_robust_residual(kind, parameter, float(r)) for r in residuals], np.float32)
    return w, p


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("parameter", [float(np.float32(30 * np.sqrt(5) / np.sqrt(2))), 5.0, 0.02, 30.0])
def test_robust_weighting_equals_the_reference(ref, kind, parameter):
    """RobustWeighting::CalculateWeight / CalculateRobustResidual (robust_weighting.h:59-110) -- the weights of every colour and depth
    residual in AccumulateHAndBAndResidualsForObservations and CostCalculator: the oracle's restatement returns the reference's bits,
    around the threshold, at zero, for huge and tiny residuals."""
    rng = np.random.RandomState(kind * 7 + int(parameter * 10) % 5)
    p32 = np.float32(parameter)
    r = np.concatenate([rng.normal(0, parameter, 4000), rng.normal(0, 100 * parameter, 500), rng.normal(0, 1e-3 * parameter, 500),
                        [0.0, -0.0, parameter, -parameter, np.nextafter(p32, np.float32(0)), np.nextafter(p32, np.float32(np.inf)), 1e30, -1e30,
                         1e-30, np.inf]]).astype(np.float32)
    w_ref = np.zeros(len(r), np.float32); p_ref = np.zeros(len(r), np.float32)
    ref.e3d_ref_robust_many(kind, parameter, r.ctypes.data, len(r), w_ref.ctypes.data, p_ref.ctypes.data)
    w, p = _oracle_many(kind, parameter, r)
    assert np.array_equal(w.view(np.uint32), w_ref.view(np.uint32))
    assert np.array_equal(p.view(np.uint32), p_ref.view(np.uint32))
    assert kind == 0 or (w_ref < 1).sum() > 100                      # the robust branch was exercised


def test_descriptor_equals_the_reference(ref):
    """opt::ComputeDescriptor (descriptor.h:36-38): neighbour intensity minus centre intensity, as the accumulate / cost / colour kernels form it"""
    rng = np.random.RandomState(3)
    a = rng.uniform(0, 255, 1000).astype(np.float32); b = rng.uniform(0, 255, 1000).astype(np.float32)
    got = np.array([ref.e3d_ref_compute_descriptor(float(x), float(y)) for x, y in zip(a, b)], np.float32)
    assert np.array_equal(got.view(np.uint32), (b - a).view(np.uint32))
