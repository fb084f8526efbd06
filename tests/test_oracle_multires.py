"""Known-answer tests of the reference's multi-scale point cloud code (src/opt/test/test_multi_scale_point_cloud.cc:37-289),
run against the CPU oracle (oracle/multires.py); tests/test_gpu_multires.py runs the same cases through the HIP path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import multires as mr              # noqa: E402
from oracle import reg_binding as rb           # noqa: E402

# MergeClosePoints KAT (test_multi_scale_point_cloud.cc:37-105): three points in a row get merged (the middle one's colour is ignored
# because it comes from another scan), a fourth point of the other scan stays.
MERGE_KAT = dict(pts=np.array([[0.1, 0, 0], [0.5, 0, 0], [0.9, 0, 0], [0.5, 0, 2]], np.float32), colors=np.array([0, 44, 2, 99], np.float32),
                 scans=np.array([0, 1, 0, 1], np.uint8), max_radius=np.array([13, 12, 11, 47], np.float32), merge_distance=1.0, num_scans=2)


def check_merge_kat(out):
    p, c, s, m = out
    assert len(p) == len(c) == len(s) == len(m) == 2
    assert sorted(s.tolist()) == [0, 1]
    for i in range(2):
        if s[i] == 0:
            np.testing.assert_allclose(p[i], [0.5, 0.0, 0.0], rtol=4e-7, atol=1e-7)          # EXPECT_FLOAT_EQ: 4 ulp
            assert c[i] == 1 and m[i] == 13
        else:
            assert np.array_equal(p[i], MERGE_KAT["pts"][3]) and c[i] == 99 and m[i] == 47


def test_merge_close_points_reference_kat():
    K = MERGE_KAT
    check_merge_kat(mr.merge_close_points(K["merge_distance"], K["num_scans"], K["pts"], K["colors"], K["scans"], K["max_radius"]))


def test_preprocess_scans_reference_kat():
    """:107-150: one grey point per scan; colour = the grey value, scan index = position in the list."""
    scans = [(np.array([[1, 2, 3]], np.float32), np.array([[5, 5, 5]], np.uint8)), (np.array([[7, 8, 9]], np.float32), np.array([[11, 11, 11]], np.uint8))]
    pts, col, idx = mr.preprocess_scans(scans)
    assert len(pts) == len(col) == len(idx) == 2 and sorted(idx.tolist()) == [0, 1]
    for i in range(2):
        np.testing.assert_allclose(pts[i], scans[idx[i]][0][0], rtol=4e-7)
        np.testing.assert_allclose(col[i], [5, 11][idx[i]], rtol=4e-7)


def multi_scale_kat_inputs():
    """:164-242: a 640x480 pinhole camera (f = 640 / 480, 3 image scales) at the identity pose looking at a grey image; one point 2 m
    in front of it and one 2 m behind."""
    w, h = 640, 480
    images = {0: dict(intr=0, pyr=[np.full((h >> l, w >> l), 100, np.uint8) for l in range(3)], masks=None,
                      q=np.array([1, 0, 0, 0], np.float32), t=np.zeros(3, np.float32))}
    intr = {0: dict(w=w, h=h, params=np.array([w, h, w / 2 - 0.5, h / 2 - 0.5], np.float32), min=0, n=3, model=0)}
    pts = np.array([[0, 0, 2], [0, 0, -2]], np.float32)
    return images, intr, pts, np.array([12, 33], np.float32), np.zeros(2, np.uint8)


def check_multi_scale_kat(scales, observation_scale_of):
    """:244-288: the point in front of the camera at two scales, observed once per scale, at image scales in [0, 1) and [1, 2)."""
    assert len(scales) == 2
    seen = set()
    for radius, p, c, s in scales:
        assert len(p) == len(c) == len(s) == 1 and c[0] == 12 and s[0] == 0 and np.array_equal(p[0], [0, 0, 2])
        sc = observation_scale_of(p, float(radius))
        assert len(sc) == 1
        seen.add(int(np.floor(sc[0])))
    assert seen == {0, 1}


def test_create_multi_scale_point_cloud_reference_kat():
    images, intr, pts, colors, sidx = multi_scale_kat_inputs()
    mn, mx = mr.point_radius_minmax(pts, images, intr, pts, 3)
    assert np.isinf(mn[1]) and np.isfinite(mn[0])                       # the point behind the camera is never observed
    scales = mr.create_multi_scale_point_cloud(pts, colors, sidx, 1, mn, mx)
    levels = rb.camera_pyramid(rb.make_camera(640, 480, intr[0]["params"], 0), 3)

    def obs_scale(p, radius):
        depth = rb.splat_depth(pts, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), levels[0], 0.03)
        o = rb.observe(p, radius, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), levels, 0, images[0]["pyr"], None, depth, 0, 0, 0, 3)
        return o[3]
    check_multi_scale_kat(scales, obs_scale)
