"""CPU tests of the normal-estimation oracle (path A').  The reference has no test for src/geometry, so these pin the
restatement against independent numpy linear algebra and the committed goldens ("parity unpinned" vs PCL itself)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_normals_golden(ob):
    g = np.load(os.path.join(GOLD, "nn_and_normals.npz"))
    n, c, knn = ob.normals(g["cloud"], k=16, viewpoint=(0, 0, 0), return_knn=True)
    assert np.array_equal(knn, g["knn"])
    assert np.abs(n - g["normals"]).max() <= 1e-6 and np.abs(c - g["curvature"]).max() <= 1e-6


def test_knn_matches_bruteforce(ob):
    rng = np.random.RandomState(0)
    P = rng.uniform(-1, 1, (800, 3)).astype(np.float32)
    idx, dist = ob.knn(P, P, 9)
    D = ((P[:, None, :].astype(np.float32) - P[None, :, :]) ** 2)
    D = (D[..., 0] + D[..., 1]) + D[..., 2]                     # L2_Simple order in f32
    order = np.lexsort((np.broadcast_to(np.arange(800), D.shape), D), axis=1)[:, :9]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.all(idx[:, 0] == np.arange(800))                 # the query itself is its own first neighbour
    assert np.array_equal(dist, np.take_along_axis(D, order, 1))


def test_normals_against_numpy_eigh(ob, synth):
    s = synth.make_scene(1, 5000, seed=9)[0]
    P = s["xyz"].numpy()
    n, c, knn = ob.normals(P, k=24, viewpoint=(0, 0, 0), return_knn=True)
    for i in range(0, 5000, 97):
        nb = P[knn[i]].astype(np.float64)
        C = np.cov(nb.T, bias=True)
        w, v = np.linalg.eigh(C)
        ref = v[:, 0]
        if np.dot(-P[i], ref) < 0:
            ref = -ref
        if w[1] - w[0] > 1e-3 * w[2]:                           # skip (near-)degenerate neighbourhoods
            assert abs(np.dot(ref, n[i])) > 1 - 1e-4
            assert np.dot(-P[i].astype(np.float64), n[i]) >= -1e-6     # flipped towards the viewpoint (origin)
            assert abs(c[i] - w[0] / w.sum()) < 1e-3


def test_normals_degenerate(ob):
    P = np.array([[0, 0, 0], [1, 0, 0]], np.float32)
    n, c = ob.normals(P, k=8)
    assert np.all(np.isnan(n)) and np.all(np.isnan(c))           # < 3 neighbours -> NaN (two_pass_normal_3d.h:97-103)
    P = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0]], np.float32)
    n, c = ob.normals(P, k=5, viewpoint=(0, 0, 5))
    assert np.allclose(np.abs(n[:, 2]), 1, atol=1e-6) and np.all(n[:, 2] > 0)
    assert np.allclose(c, 0, atol=1e-7)
    with pytest.raises(ValueError):
        ob.normals(P, k=0, radius=-1)


def test_normals_radius_search(ob):
    rng = np.random.RandomState(4)
    P = np.c_[rng.uniform(0, 1, (2000, 2)), 0.001 * rng.normal(size=2000)].astype(np.float32)
    n, c = ob.normals(P, radius=0.1, viewpoint=(0, 0, 1))
    ok = ~np.isnan(n[:, 0])
    assert ok.sum() > 1900 and np.all(n[ok, 2] > 0.99)


def test_local_outlier_removal_restatement():
    """LocalStatisticalOutlierRemoval (local_statistical_outlier_removal.hpp:71-172) against a numpy restatement on a
    brute-force kNN: a noisy plane plus uniform clutter; clutter goes, the plane stays."""
    from scipy.spatial import cKDTree
    from oracle import binding as ob
    rng = np.random.RandomState(3)
    plane = np.stack([rng.uniform(-1, 1, 4000), rng.uniform(-1, 1, 4000), 0.002 * rng.normal(size=4000)], 1)
    pts = np.concatenate([plane, rng.uniform(-1, 1, (80, 3))]).astype(np.float32)
    for mean_k, factor in ((8, 2.0), (20, 1.5)):
        inl, md = ob.local_outlier_removal(pts, mean_k, factor)
        d, i = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=mean_k + 1)
        m = (np.sqrt((d[:, 1:] ** 2).astype(np.float32)).astype(np.float64).sum(1) / mean_k).astype(np.float32)
        assert np.abs(m - md).max() <= 2e-7
        nb = md[i[:, 1:]].astype(np.float64)
        thr = nb.sum(1) / mean_k * factor
        exp = ~(md > thr)
        assert (exp != inl).sum() <= 2                       # float ties at the threshold only
        assert inl[:4000].mean() > 0.97 and inl[4000:].mean() < 0.8     # the filter is local: clutter near clutter stays
    inl_neg, _ = ob.local_outlier_removal(pts, 8, 2.0, negative=True)
    inl_pos, _ = ob.local_outlier_removal(pts, 8, 2.0)
    assert np.array_equal(inl_neg, ~inl_pos)


def test_local_outlier_removal_small_cloud():
    """Fewer points than mean_k + 1 (a late --filter pass of PointCloudCleaner): pcl::KdTreeFLANN clamps the search to the cloud,
    the first pass still divides by mean_k, the second walks the shorter lists; an empty cloud is a no-op."""
    from oracle import binding as ob
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 4], [10, 10, 10]], np.float32)
    inl, md = ob.local_outlier_removal(pts, 8, 2.0)
    d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1).astype(np.float32))
    exp = np.array([np.sort(d[i])[1:].astype(np.float64).sum() / 8 for i in range(5)], np.float32)
    assert np.allclose(md, exp, rtol=1e-6)
    thr = np.array([md[[j for j in range(5) if j != i]].astype(np.float64).sum() / 4 * 2.0 for i in range(5)])
    assert np.array_equal(inl, ~(md > thr)) and not inl[4] and inl[:4].all()
    inl, md = ob.local_outlier_removal(pts[:1], 8, 2.0)
    assert inl.tolist() == [True] and md.tolist() == [0.0]        # no neighbour: mean = 0 / 0, every comparison false -> kept
    inl, md = ob.local_outlier_removal(np.zeros((0, 3), np.float32), 8, 2.0)
    assert inl.shape == (0,)


def test_normals_sample_equals_the_full_run():
    """oracle_normals_sample (the checker of the 20 M point at-size test): the listed points searched in the WHOLE cloud give what
    oracle_normals gives for them -- kNN lists, normals, curvature, bit for bit; k search and radius search."""
    from oracle import binding as ob
    rng = np.random.RandomState(11)
    pts = (rng.normal(size=(6000, 3)) * np.array([2.0, 1.0, 0.03])).astype(np.float32)
    sample = np.sort(rng.choice(len(pts), 700, replace=False))
    n, c, knn = ob.normals(pts, k=12, viewpoint=(0, 0, 3), return_knn=True)
    ns, cs, ks = ob.normals_sample(pts, sample, k=12, viewpoint=(0, 0, 3))
    assert np.array_equal(knn[sample], ks)
    assert np.array_equal(n[sample].view(np.uint32), ns.view(np.uint32)) and np.array_equal(c[sample].view(np.uint32), cs.view(np.uint32))
    n, c = ob.normals(pts, radius=0.15, viewpoint=(0, 0, 3))
    ns, cs, _ = ob.normals_sample(pts, sample, radius=0.15, viewpoint=(0, 0, 3))
    same_nan = np.isnan(n[sample, 0]) == np.isnan(ns[:, 0])
    assert same_nan.all()
    v = ~np.isnan(ns[:, 0])
    assert np.array_equal(n[sample][v].view(np.uint32), ns[v].view(np.uint32)) and np.array_equal(c[sample][v].view(np.uint32), cs[v].view(np.uint32))


def test_scanner_sampled_scan_lies_on_the_room():
    """synth.make_scan_angular (bench / DESIGN 4.3b): rays uniform in angle, first hit -- every point lies on a surface of the room
    (to the range noise), the density falls with the range, normals are unit vectors turned to the scanner."""
    import importlib
    synth = importlib.import_module("dataset-pipeline_amd.synth")
    origin, yaw = synth.SCAN_POSES[0]
    xyz, nrm, T = synth.make_scan_angular(60000, origin, yaw, 3, sigma=0.0)
    assert xyz.shape == (60000, 3) and abs(float((nrm.norm(dim=1) - 1).abs().max())) < 1e-6
    P = xyz.numpy().astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3].astype(np.float64)      # world frame
    W, D, H = 10.0, 10.0, 3.0
    d_planes = np.minimum.reduce([np.abs(P[:, 2]), np.abs(P[:, 0]), np.abs(P[:, 0] - W), np.abs(P[:, 1]), np.abs(P[:, 1] - D)])
    d_cyl = np.minimum.reduce([np.abs(np.hypot(P[:, 0] - cx, P[:, 1] - cy) - r) for cx, cy, r in synth._CYL])
    assert np.minimum(d_planes, d_cyl).max() < 1e-5
    assert ((-xyz.numpy() * nrm.numpy()).sum(1) >= -1e-6).all()                     # turned to the scanner (the local origin)
    r = np.linalg.norm(xyz.numpy(), axis=1)
    ru = np.linalg.norm(synth.make_scan(60000, origin, yaw, 3, sigma=0.0)[0].numpy(), axis=1)
    assert (r < 2.0).mean() > 3 * (ru < 2.0).mean()                                 # far denser near the scanner than uniform per area
