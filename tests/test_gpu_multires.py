"""GPU parity tests of the multi-resolution point cloud construction against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mr():
    from oracle import multires
    multires.lib()
    return multires


def _scan_like(n, seed, spacing=0.004):
    """points on two walls in scan-line order (structured input: long greedy dependency chains) + noise"""
    rng = np.random.RandomState(seed)
    side = int(np.sqrt(n / 2))
    u, v = np.meshgrid(np.arange(side) * spacing, np.arange(side) * spacing, indexing="ij")
    a = np.stack([u.ravel(), v.ravel(), np.zeros(side * side)], 1)
    b = np.stack([np.zeros(side * side), u.ravel(), v.ravel() + 0.3], 1)
    P = np.concatenate([a, b]) + rng.normal(0, spacing * 0.15, (2 * side * side, 3))
    return P.astype(np.float32)


@pytest.mark.parametrize("n,dist,scans", [(20000, 0.009, 1), (60000, 0.013, 3), (3000, 0.5, 2)])
def test_merge_close_points_matches_oracle(e3d, mr, n, dist, scans):
    rng = np.random.RandomState(n)
    P = _scan_like(n, n)
    col = rng.uniform(0, 255, len(P)).astype(np.float32)
    sidx = rng.randint(0, scans, len(P)).astype(np.uint8)
    mxr = rng.uniform(0.001, 0.1, len(P)).astype(np.float32)
    g = e3d.merge_close_points(dist, scans, P, col, sidx, mxr)
    o = mr.merge_close_points(dist, scans, P, col, sidx, mxr)
    assert len(g[0]) == len(o[0]) and len(g[0]) < len(P)                       # the same centres, in the same order
    assert np.array_equal(g[2], o[2])                                          # majority scan incl. the tie rule
    assert np.array_equal(g[3].view(np.uint32), o[3].view(np.uint32))          # max of max_radius: exact
    assert np.abs(g[0] - o[0]).max() <= 1e-5 * max(1.0, np.abs(P).max())       # f32 means, different summation order
    assert np.abs(g[1] - o[1]).max() <= 1e-3


def test_merge_close_points_edge_cases(e3d, mr):
    P = np.array([[0, 0, 0]], np.float32)
    g = e3d.merge_close_points(0.1, 1, P, np.array([7], np.float32), np.array([0], np.uint8), np.array([0.5], np.float32))
    assert len(g[0]) == 1 and np.array_equal(g[0][0], P[0]) and g[1][0] == 7 and g[3][0] == 0.5
    # duplicates and a strict radius: points exactly merge_distance apart are NOT merged
    P = np.array([[0, 0, 0], [0, 0, 0], [0.5, 0, 0], [1.0, 0, 0]], np.float32)
    args = (np.arange(4, dtype=np.float32), np.zeros(4, np.uint8), np.ones(4, np.float32))
    g = e3d.merge_close_points(0.5, 1, P, *args); o = mr.merge_close_points(0.5, 1, P, *args)
    assert len(g[0]) == len(o[0]) == 3 and np.allclose(g[0], o[0])
    with pytest.raises(e3d.E3DError):
        e3d.merge_close_points(0.0, 1, P, *args)
    with pytest.raises(e3d.E3DError):
        e3d.merge_close_points(0.1, 99, P, *args)


@pytest.mark.parametrize("model", [0, 1, 2, 3, 4, 10, 11, 12])
def test_point_radius_minmax_matches_oracle(e3d, mr, model):
    from reg_util import make_multi_image_scene
    M = make_multi_image_scene(n_points=8000, n_images=3, seed=21, model=model)
    G = e3d.RegProblem(e3d.default_reg_params(image_scale_count=M["n_levels"], point_neighbor_count=M["K"]))
    G.set_intrinsics(0, M["width"], M["height"], M["params"], 0, M["n_levels"], camera_type=model)
    G.set_splat_points(M["pts"])
    images = {}
    for i, im in enumerate(M["images"]):
        G.set_image(i, 0, im["pyr"]); G.set_image_pose(i, im["q_true"], im["t_true"])
        images[i] = dict(intr=0, pyr=im["pyr"], masks=None, q=im["q_true"], t=im["t_true"])
    intr = {0: dict(w=M["width"], h=M["height"], params=M["params"], min=0, n=M["n_levels"], model=model)}
    # some points far outside every frustum stay unobserved
    pts = np.concatenate([M["pts"], np.array([[50, 3, 0], [0, -5, 0]], np.float32)])
    gmn, gmx = G.point_radius_minmax(pts)
    omn, omx = mr.point_radius_minmax(pts, images, intr, M["pts"], M["n_levels"])
    seen_o = np.isfinite(omn)
    assert np.array_equal(np.isfinite(gmn), seen_o) and seen_o[:-2].sum() > 4000 and not seen_o[-2:].any()
    assert np.array_equal(np.isfinite(gmx), np.isfinite(omx))
    tol = 1e-6                      # every model: shared elementary functions (include/e3d_libm.h)
    assert np.abs(gmn[seen_o] - omn[seen_o]).max() <= tol * omn[seen_o].max()
    assert np.abs(gmx[seen_o] - omx[seen_o]).max() <= tol * omx[seen_o].max()
    # sanity of the magnitude: half a pixel at depth z and focal length f is about z / (2 f)
    assert 0.5 * 2.5 / 210 / 2 < np.median(gmn[seen_o]) < 2 * 3.5 / 210 / 2


# ---- the reference's own known-answer tests (src/opt/test/test_multi_scale_point_cloud.cc) through the HIP path -------------------
def test_merge_close_points_reference_kat(e3d):
    from test_oracle_multires import MERGE_KAT as K, check_merge_kat
    check_merge_kat(e3d.merge_close_points(K["merge_distance"], K["num_scans"], K["pts"], K["colors"], K["scans"], K["max_radius"]))


def test_create_multi_scale_point_cloud_reference_kat(e3d, mr):
    """:164-289 with the radius range, the observations and their scales from the GPU (the scale loop itself is host code, shared
    with the oracle here; the tool's C++ version of it is covered by tests/test_gpu_cli_reg.py)."""
    from test_oracle_multires import check_multi_scale_kat, multi_scale_kat_inputs
    images, intr, pts, colors, sidx = multi_scale_kat_inputs()
    G = e3d.RegProblem(e3d.default_reg_params(image_scale_count=3, point_neighbor_count=5))
    G.set_intrinsics(0, 640, 480, intr[0]["params"], 0, 3)
    G.set_image(0, 0, images[0]["pyr"]); G.set_image_pose(0, images[0]["q"], images[0]["t"])
    G.set_splat_points(pts)
    mn, mx = G.point_radius_minmax(pts)
    omn, omx = mr.point_radius_minmax(pts, images, intr, pts, 3)
    assert np.array_equal(mn.view(np.uint32), omn.view(np.uint32)) and np.array_equal(mx.view(np.uint32), omx.view(np.uint32))
    scales = mr.create_multi_scale_point_cloud(pts, colors, sidx, 1, mn, mx)

    def obs_scale(p, radius):
        nbr = np.zeros((len(p), 5), np.uint32)
        G.set_point_scale(0, p, radius, nbr, np.zeros((len(p), 5), np.float32))
        G.render_depth(0, 0)
        n = G.observe(0, 0, 0, 0)
        return G.get_observations(0, 0, n)[3]
    check_multi_scale_kat(scales, obs_scale)
