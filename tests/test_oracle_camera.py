"""Pins the oracle's camera models (oracle/oracle_camera.h) with the reference's own camera tests, restated:
src/camera/test/test_camera.cc:40-138 (undistort/distort round trips), :225-262 (ImageDerivativeByWorld vs central
differences, tolerance 0.25), :264-343 (ImageDerivativeByIntrinsics vs central differences, tolerance 2.5e-3), with the
parameter sets of TEST(Camera, Pinhole / FisheyeFOV / PolynomialTangential / FisheyePolynomial4 / Benchmark)
:408-426,464-468,470-475,483-489,499-506."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reg_binding as rb  # noqa: E402

W, H = 640, 480
CAMERAS = {
    "PINHOLE": (rb.PINHOLE, [250.0, 200.0, 319.5, 239.5]),
    "OPENCV": (rb.OPENCV, [340.926, 341.124, 302.4, 201.6, -0.101082, 0.0703954, 0.000438661, -0.000680887]),
    "THIN_PRISM_FISHEYE": (rb.THIN_PRISM_FISHEYE, [340.926, 341.124, 302.4, 201.6, 0.221184, 0.128597, 0.000531602, -0.000388873,
                                                   0.0623079, 0.20419, -0.000805024, 4.07704e-05]),
    "OPENCV_FISHEYE": (rb.OPENCV_FISHEYE, [340.926, 341.124, 302.4, 201.6, 0.221184, 0.128597, 0.0623079, 0.20419]),
    "FOV": (rb.FOV, [250.0, 200.0, 319.5, 239.5, 1.0]),
    # test_camera.cc:428-462,491-497 (kK1 = 0.13, kK2 = -0.66, kK3 = 0.64): the remaining classes of camera_base.cc:66-77
    "SIMPLE_PINHOLE": (rb.SIMPLE_PINHOLE, [250.0, 319.5, 239.5]),
    "RADIAL": (rb.RADIAL, [250.0, 319.5, 239.5, 0.13, -1e-2]),
    "SIMPLE_RADIAL": (rb.SIMPLE_RADIAL, [450.0, 319.5, 239.5, 0.13]),
    "POLYNOMIAL_3": (rb.POLYNOMIAL_3, [250.0, 200.0, 319.5, 239.5, 0.13, -0.66, 0.64]),
    "FISHEYE_POLYNOMIAL_2_TANGENTIAL_2": (rb.FISHEYE_POLYNOMIAL_2_TANGENTIAL_2, [340.926, 341.124, 302.4, 201.6, -0.101082, 0.0703954, 0.000438661,
                                                                               -0.000680887]),
    # test_camera.cc:440-444,452-456,477-481: the three classes the factory never creates, with the reference's own test parameters
    "RADIAL_FISHEYE_CLASS": (rb.RADIAL_FISHEYE_CLASS, [250.0, 319.5, 239.5, -0.13, 0.66]),
    "SIMPLE_RADIAL_FISHEYE_CLASS": (rb.SIMPLE_RADIAL_FISHEYE_CLASS, [450.0, 319.5, 239.5, 0.13]),
    "FULL_OPENCV": (rb.FULL_OPENCV, [340.926, 341.124, 302.4, 201.6, -0.101082, 0.0703954, 0.0438661, -0.0680887, -0.00101082, 0.1, 0.001, -0.001]),
}


@pytest.fixture(scope="module", params=sorted(CAMERAS))
def cam(request):
    t, p = CAMERAS[request.param]
    return rb.make_camera(W, H, np.array(p, np.float32), t)


def test_parameter_storage(cam):
    t = cam.type
    n = rb.PARAM_COUNT[t]
    p = np.array([1 + 10 * i for i in range(n)], np.float32)
    c = rb.make_camera(10, 10, p, t)
    assert np.array_equal(c.params(), p)


def test_undistort_then_distort_image_corners(cam):
    if cam.type == rb.FOV:
        pytest.skip("test_camera.cc:383-388: not run for the FOV camera, the corners lie outside its image circle")
    for x, y in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)):
        nxy = np.array([cam.fx_inv * np.float32(x) + cam.cx_inv, cam.fy_inv * np.float32(y) + cam.cy_inv], np.float32)
        u, _ = rb.cam_undistort(cam, float(nxy[0]), float(nxy[1]))
        r = rb.cam_distort(cam, float(u[0]), float(u[1]))
        assert abs(r[0] - nxy[0]) <= 1e-5 and abs(r[1] - nxy[1]) <= 1e-5


def test_distort_then_undistort(cam):
    for fx, fy in ((0, 0), (1, 1), (0, 1), (1, 0), (.5, .5), (.1, .2), (.8, .9), (.5, .6), (.1, .9)):
        x, y = np.float32(fx) * W, np.float32(fy) * H
        nxy = np.array([cam.fx_inv * x + cam.cx_inv, cam.fy_inv * y + cam.cy_inv], np.float32)
        d = rb.cam_distort(cam, float(nxy[0]), float(nxy[1]))
        u, _ = rb.cam_undistort(cam, float(d[0]), float(d[1]))
        assert abs(u[0] - nxy[0]) <= 1e-5 and abs(u[1] - nxy[1]) <= 1e-5


def test_image_derivative_by_world(cam):
    step = np.float32(0.001)
    for at in ((0.0, 0.0, 3.0), (1.0, 3.0, 8.0), (-0.1, 0.7, -0.8)):
        at = np.array(at, np.float32)
        if cam.type == rb.FOV and at[0] == 0 and at[1] == 0:
            continue          # test_camera.cc:261-264: the FOV model's derivative at the principal axis is the identity, not the limit
        J = rb.cam_deriv_by_world(cam, at)
        for a in range(3):
            plus, minus = at.copy(), at.copy()
            plus[a] += step; minus[a] -= step
            num = (rb.cam_project(cam, plus) - rb.cam_project(cam, minus)) / (2 * step)
            assert abs(J[0, a] - num[0]) <= 0.25 and abs(J[1, a] - num[1]) <= 0.25


def test_image_derivative_by_intrinsics(cam):
    checked = 0
    for at in ((0.0, 0.0, 3.0), (1.0, 2.5, 4.0), (1.0, 3.0, 8.0), (-0.1, 0.4, 0.8)):
        at = np.array(at, np.float32)
        pxy = rb.cam_project(cam, at)
        if pxy[0] < 0 or pxy[1] < 0 or pxy[0] >= W or pxy[1] >= H:
            continue
        J = rb.cam_deriv_by_intrinsics(cam, at)
        for c in range(cam.n_params):
            d = np.float32(0.01)
            pp, pm = cam.params(), cam.params()
            pp[c] += d; pm[c] += np.float32(-1.0) * d
            cp, cm = rb.make_camera(W, H, pp, cam.type), rb.make_camera(W, H, pm, cam.type)
            num = (rb.cam_project(cp, at) - rb.cam_project(cm, at)) / (2 * d)
            assert abs(J[0, c] - num[0]) <= 2.5e-3 and abs(J[1, c] - num[1]) <= 2.5e-3, (c, J[:, c], num)
            checked += 1
    assert checked > 0


def test_cutoffs():
    for name in ("PINHOLE", "SIMPLE_PINHOLE"):
        t, p = CAMERAS[name]
        assert np.isinf(rb.make_camera(W, H, np.array(p, np.float32), t).cutoff2)      # their constructors never call InitCutoff
    # SIMPLE_RADIAL: closed form -1 / (3 k) for k < 0 only (camera_simple_radial.cc:51-57)
    assert np.isinf(rb.make_camera(W, H, np.array(CAMERAS["SIMPLE_RADIAL"][1], np.float32), rb.SIMPLE_RADIAL).cutoff2)
    c = rb.make_camera(W, H, np.array([450.0, 319.5, 239.5, -0.2], np.float32), rb.SIMPLE_RADIAL)
    assert c.cutoff2 == np.float32(-1.0) / (np.float32(3) * np.float32(-0.2))
    # RADIAL / POLYNOMIAL_3: RadialBase::InitCutoff on the camera itself -- r * factor(r^2) = farthest corner radius, * 1.01
    for name in ("RADIAL", "POLYNOMIAL_3"):
        t, p = CAMERAS[name]
        c = rb.make_camera(W, H, np.array(p, np.float32), t)
        nd = 2 if name == "RADIAL" else 3
        k = np.array(list(p[-nd:]) + [0.0], np.float64)
        corner = max(np.hypot(c.fx_inv * x + c.cx_inv, c.fy_inv * y + c.cy_inv) for x, y in ((0, 0), (W, 0), (0, H), (W, H)))
        if np.isfinite(c.cutoff2):
            r = np.sqrt(np.float64(c.cutoff2) / np.float64(np.float32(1.01)))
            fac = 1 + r**2 * (k[0] + r**2 * (k[1] + r**2 * k[2]))
            # either the innermost solution (times 1.01) or the squared second-best radius bounds it
            assert abs(r * fac - corner) <= 2e-5 * corner or c.cutoff2 < r * r * 1.01 + 1e-6
        assert np.isinf(c.inner_cutoff2)
    t, p = CAMERAS["FISHEYE_POLYNOMIAL_2_TANGENTIAL_2"]
    c = rb.make_camera(W, H, np.array(p, np.float32), t)
    assert np.isinf(c.cutoff2) and np.isfinite(c.inner_cutoff2) and c.inner_cutoff2 > 0
    inner = rb.make_camera(W, H, np.array(p, np.float32), rb.OPENCV)                   # its inner PolynomialTangentialCamera
    assert c.inner_cutoff2 == inner.cutoff2
    t, p = CAMERAS["OPENCV"]
    c = rb.make_camera(W, H, np.array(p, np.float32), t)
    assert np.isfinite(c.cutoff2) and c.cutoff2 > 0
    # every image corner un-distorts to inside the cut-off
    for x, y in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)):
        u, conv = rb.cam_undistort(c, c.fx_inv * x + c.cx_inv, c.fy_inv * y + c.cy_inv)
        assert conv and float(u[0]) ** 2 + float(u[1]) ** 2 <= c.cutoff2
    t, p = CAMERAS["THIN_PRISM_FISHEYE"]
    c = rb.make_camera(W, H, np.array(p, np.float32), t)
    assert np.isinf(c.cutoff2) and np.isfinite(c.inner_cutoff2) and c.inner_cutoff2 > 0
    # a ray far outside the field of view projects to infinity
    assert not np.all(np.isfinite(rb.cam_project(c, np.array([50.0, 0.0, 0.1], np.float32))))
    # OPENCV_FISHEYE: the inner Polynomial4Camera's RadialBase::InitCutoff (farthest corner, 1-D Gauss-Newton)
    t, p = CAMERAS["OPENCV_FISHEYE"]
    c = rb.make_camera(W, H, np.array(p, np.float32), t)
    assert np.isinf(c.cutoff2) and np.isfinite(c.inner_cutoff2) and c.inner_cutoff2 > 0
    k = np.array(p[4:], np.float64)
    corner = max(np.hypot(c.fx_inv * x + c.cx_inv, c.fy_inv * y + c.cy_inv) for x, y in ((0, 0), (W, 0), (0, H), (W, H)))
    # the cut-off radius r satisfies r * factor(r^2) = corner radius (f64 check of the f32 solver), then * 1.01
    r = np.sqrt(c.inner_cutoff2 / np.float64(np.float32(1.01)))
    fac = 1 + r**2 * (k[0] + r**2 * (k[1] + r**2 * (k[2] + r**2 * k[3])))
    assert abs(r * fac - corner) <= 2e-5 * corner


def test_fov_closed_forms():
    """FisheyeFOVCamera: no cut-off, derived constants of its constructor (camera_fisheye_fov.cc:37-51), closed-form Undistort that
    returns infinity past image_radius_ (camera_fisheye_fov.h:76-86)"""
    t, p = CAMERAS["FOV"]
    c = rb.make_camera(W, H, np.array(p, np.float32), t)
    assert np.isinf(c.cutoff2)
    assert c.p[5] == np.float32(2.0) * np.tan(np.float32(0.5), dtype=np.float32) and c.p[6] == np.float32(np.pi / 2.0)
    u, conv = rb.cam_undistort(c, 1.2, 1.2)                  # r = 1.697 > pi/2
    assert not conv and np.all(np.isinf(u))
    u, conv = rb.cam_undistort(c, 0.3, -0.4)
    assert conv and np.allclose(u, np.array([0.3, -0.4]) * np.tan(0.5) / (0.5 * 2 * np.tan(0.5)), rtol=1e-6)
    d = rb.cam_distort(c, 0.0, 0.0)
    assert d[0] == 0 and d[1] == 0
    assert np.array_equal(rb.cam_deriv_by_world(c, np.array([0, 0, 2], np.float32)),
                          np.array([[125.0, 0, 0], [0, 100.0, 0]], np.float32))


def test_scaled_camera_matches_scaledby():
    t, p = CAMERAS["OPENCV"]
    c = rb.make_camera(W, H, np.array(p, np.float32), t)
    lv = rb.camera_pyramid(c, 3)
    assert (lv[1].width, lv[1].height) == (320, 240) and (lv[2].width, lv[2].height) == (160, 120)
    assert lv[1].p[0] == np.float32(p[0]) * np.float32(0.5)
    assert lv[1].p[2] == np.float32(0.5) * (np.float32(p[2]) + np.float32(0.5)) - np.float32(0.5)
    assert list(lv[1].p[4:8]) == list(c.p[4:8])
