"""CPU tests of the oracle (path A): the reference's own known-answer cases restated, brute-force / scipy cross
checks of the exact NN semantics, and the committed regression goldens.  No GPU needed."""
import importlib
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.linalg import expm

from conftest import identical_cloud_case, plane_case, pose_error

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(ob, clouds, d, iters, thr):
    o = ob.OracleICP()
    ids = [o.add_point_cloud(x, n, T, f) for (x, n, T, f) in clouds]
    conv = o.run(d, 0, iters, thr, False)
    return o, ids, conv


def _check_golden(ob, name, clouds, d, iters, thr):
    g = np.load(os.path.join(GOLD, name))
    o, ids, conv = _run(ob, clouds, d, iters, thr)
    recs = np.array([(r[0], r[1], r[2], r[3]) for r in o.pair_records()], dtype=np.int64)
    assert bool(g["converged"]) == conv
    assert np.array_equal(recs, g["pair_records"])
    poses = np.stack([o.get_result_global_T_cloud(i) for i in ids if i >= 0])
    assert np.abs(poses - g["poses"]).max() <= 1e-6
    return o, ids


# ---- reference KATs: src/opt/test/test_icp.cc ---------------------------------------------------------------------
def test_plane_case_success(ob):
    """PointToPlaneICP.PlaneCaseSuccess (test_icp.cc:111-172): final poses equal within 1e-5 per matrix entry."""
    xyz, nrm, T0, T1 = plane_case()
    o, ids = _check_golden(ob, "icp_plane_case.npz", [(xyz, nrm, T0, False), (xyz, nrm, T1, False)], 1.5, 100, 1e-7)
    A, B = o.get_result_global_T_cloud(ids[0]), o.get_result_global_T_cloud(ids[1])
    assert np.abs(A - B).max() <= 1e-5
    # impl cloud 0 (first movable cloud when nothing is fixed) never moves [QUIRK]
    assert np.array_equal(A, T0)


def test_identical_cloud_alignment(ob):
    """PointToPlaneICP.IdenticalCloudAlignment (test_icp.cc:39-109)."""
    P, N, Ts = identical_cloud_case()
    o, ids = _check_golden(ob, "icp_identical_clouds.npz", [(P, N, T, False) for T in Ts],
                           np.float32(0.15) * np.sqrt(3), 100, 1e-7)
    T0 = o.get_result_global_T_cloud(ids[0])
    for i in ids[1:]:
        assert np.abs(o.get_result_global_T_cloud(i) - T0).max() <= 1e-5


def test_room_with_fixed_cloud_golden(ob, synth):
    scans = synth.make_scene(3, 4000, seed=42)
    clouds = [(s["xyz"].numpy(), s["normals"].numpy(), s["T_init"], i == 0) for i, s in enumerate(scans)]
    o, ids = _check_golden(ob, "icp_room_fixed_plus_two.npz", clouds, 0.3, 5, 1e-9)
    assert ids == [-1, 0, 1]
    # pairs are reported in the sequential order of the reference's ik loop; fixed pairs use -1
    first = [(r[1], r[2]) for r in o.pair_records() if r[0] == 0]
    assert first == [(1, -1), (-1, 1), (1, 2), (2, 1), (2, -1), (-1, 2)]


def test_alignment_improves(ob, synth):
    scans = synth.make_scene(2, 20000, seed=5)
    o, ids, _ = _run(ob, [(s["xyz"].numpy(), s["normals"].numpy(), s["T_init"], False) for s in scans], 0.15, 10, 1e-9)
    a0, t0 = pose_error(scans[1]["T_init"], scans[1]["T_true"])
    a1, t1 = pose_error(o.get_result_global_T_cloud(1), scans[1]["T_true"])
    assert t1 < 0.25 * t0 and a1 < 0.25 * a0


# ---- FindCorrespondencesFast semantics ----------------------------------------------------------------------------
@pytest.mark.parametrize("seed,ns,nt,d", [(0, 400, 300, 0.2), (1, 50, 2000, 0.08), (2, 1000, 7, 1.0)])
def test_kdtree_equals_bruteforce(ob, seed, ns, nt, d):
    rng = np.random.RandomState(seed)
    src = rng.uniform(-1, 1, (ns, 3)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (nt, 3)).astype(np.float32)
    a = ob.find_correspondences(src, tgt, d)
    b = ob.find_correspondences(src, tgt, d, brute=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_nn_against_scipy(ob):
    rng = np.random.RandomState(3)
    src = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    d = 0.1
    iq, im, sd = ob.find_correspondences(src, tgt, d)
    dist, idx = cKDTree(tgt.astype(np.float64)).query(src.astype(np.float64), k=1)
    # away from the radius boundary the sets agree; f32 vs f64 rounding only matters within 1e-6 of d
    clear_in = dist < d * (1 - 1e-5)
    clear_out = dist > d * (1 + 1e-5)
    found = np.zeros(len(src), bool)
    found[iq] = True
    assert np.all(found[clear_in]) and not np.any(found[clear_out])
    m = np.full(len(src), -1)
    m[iq] = im
    assert np.all(m[clear_in] == idx[clear_in])
    assert np.allclose(sd, dist[iq] ** 2, rtol=1e-5, atol=1e-9)


def test_nn_golden(ob):
    g = np.load(os.path.join(GOLD, "nn_and_normals.npz"))
    iq, im, sd = ob.find_correspondences(g["src"], g["tgt"], 0.2)
    assert np.array_equal(iq, g["iq"]) and np.array_equal(im, g["im"])
    assert np.array_equal(sd.view(np.uint32), g["sd"].view(np.uint32))


def test_nn_ties_and_strict_radius(ob):
    tgt = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [1, 0, 0]], np.float32)
    src = np.zeros((1, 3), np.float32)
    iq, im, sd = ob.find_correspondences(src, tgt, 1.5)
    assert list(im) == [0] and sd[0] == 1.0            # lowest index among equidistant neighbours
    iq, im, sd = ob.find_correspondences(src, tgt, 1.0)
    assert len(iq) == 0                                # dist^2 < r^2 is strict (FLANN KNNRadiusResultSet)
    iq, im, sd = ob.find_correspondences(src, np.zeros((0, 3), np.float32), 1.0)
    assert len(iq) == 0


# ---- small-matrix / Lie-group arithmetic ----------------------------------------------------------------------------
def test_se3_update_matches_matrix_exponential(ob):
    rng = np.random.RandomState(0)
    for scale in (1e-12, 1e-6, 1e-2, 0.7):
        x = rng.normal(size=6) * scale
        q, t = ob.se3_update(x, [1, 0, 0, 0], [0, 0, 0])
        R = ob.quat_to_R(q)
        xi = -x
        Om = np.array([[0, -xi[5], xi[4]], [xi[5], 0, -xi[3]], [-xi[4], xi[3], 0]])
        M = np.zeros((4, 4)); M[:3, :3] = Om; M[:3, 3] = xi[:3]
        E = expm(M)
        assert np.abs(R - E[:3, :3]).max() < 2e-7 and np.abs(t - E[:3, 3]).max() < 1e-7 * max(1, scale)


def test_ldlt_upper_only(ob):
    rng = np.random.RandomState(1)
    for n in (6, 42, 114):
        A = rng.normal(size=(n, n)); A = A @ A.T + 0.1 * np.eye(n)
        b = rng.normal(size=n)
        U = np.triu(A) + np.tril(rng.normal(size=(n, n)), -1) * 100     # garbage below the diagonal is never read
        x = ob.ldlt_solve_upper(U, b)
        assert np.abs(x - np.linalg.solve(A, b)).max() < 1e-9 * np.abs(x).max() + 1e-12


def test_pair_system_quirk_and_symmetry(ob):
    """J1_src = -J1_tgt etc. only in exact arithmetic: the 12x12 pair system is PSD and b = J^T r."""
    rng = np.random.RandomState(2)
    S = rng.uniform(-1, 1, (50, 3)).astype(np.float32); Sn = rng.normal(size=(50, 3)).astype(np.float32)
    T = rng.uniform(-1, 1, (60, 3)).astype(np.float32); Tn = rng.normal(size=(60, 3)).astype(np.float32)
    iq = rng.randint(0, 50, 200); im = rng.randint(0, 60, 200)
    H, b, c = ob.pair_system(S, Sn, T, Tn, iq, im, [1, 0, 0, 0], [0, 0, 0], [1, 0, 0, 0], [0, 0, 0])
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > -1e-9 * np.abs(H).max()
    assert c > 0
    # source and target blocks nearly mirror each other
    assert np.abs(H[:6, :6] - H[6:, 6:]).max() < 1e-4 * np.abs(H).max()


def test_errors(ob):
    o = ob.OracleICP()
    with pytest.raises(RuntimeError):
        o.run(0.1, 0, 1, 1e-6)
    with pytest.raises(IndexError):
        o.get_result_global_T_cloud(0)
