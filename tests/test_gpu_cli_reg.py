"""ImageRegistrator CLI end to end on a synthetic dataset written in the pipeline's own file formats (MeshLab project + PLY
scans, COLMAP cameras.txt / images.txt, rigs.json, PNG images, multi-resolution point cloud cache): the tool's exported
state must equal what the same optimisation gives when driven through the Python binding of the same C-ABI, level by
level, and the photometric cost must go down."""
import importlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from cli_util import BIN, ROOT, read_mlp, write_mlp, write_ply_xyz
from reg_util import make_multi_image_scene, make_rig_scene

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def _write_png(path, img):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.ascontiguousarray(img, np.uint8), mode="L").save(path)


def _write_dataset(tmp, M, names, rigs=None, model_name="PINHOLE"):
    from tools.make_multires_cache import write_cache
    d = str(tmp)
    write_ply_xyz(os.path.join(d, "scan.ply"), M["pts"], rgb=np.full((len(M["pts"]), 3), 128, np.uint8))
    write_mlp(os.path.join(d, "scans.mlp"), [("scan", "scan.ply", np.eye(4))])
    os.makedirs(os.path.join(d, "state"), exist_ok=True)
    from reg_util import UNIQUE_FOCAL
    ci = 1 if M.get("model", 0) in UNIQUE_FOCAL else 2                          # cx, cy follow ONE focal length in those models
    p = M["params"].astype(np.float64).copy(); p[ci] += 0.5; p[ci + 1] += 0.5   # COLMAP's pixel-corner convention
    with open(os.path.join(d, "state", "cameras.txt"), "w") as f:
        f.write("# cameras\n7 %s %d %d %s\n" % (model_name, M["width"], M["height"], " ".join("%.9g" % v for v in p)))
    with open(os.path.join(d, "state", "images.txt"), "w") as f:
        f.write("# images\n")
        for i, (im, name) in enumerate(zip(M["images"], names)):
            q, t = im.get("q_init", im.get("q_true")), im.get("t_init", im.get("t_true"))
            f.write("%d %s %s 7 %s\n\n" % (10 + i, " ".join("%.9g" % v for v in q), " ".join("%.9g" % v for v in t), name))
            _write_png(os.path.join(d, "images", name), im["pyr"][0])
    if rigs is not None:
        import json
        json.dump(rigs, open(os.path.join(d, "state", "rigs.json"), "w"), indent=4)
    write_cache(os.path.join(d, "cache"), [(sc["radius"], sc["pts"], sc["tex"], sc["nbr"]) for sc in _point_scales(M)],
                neighbor_count=M["K"], candidate_count=25)
    return d


def _point_scales(M):
    """Two point scales like a real multi-resolution cloud: all points at the fine radius, every third point at twice the
    radius (observed one image scale coarser).  The stored intensity is the scene texture, from which the tool derives the
    fixed descriptors (problem.cc:549-572)."""
    from scipy.spatial import cKDTree
    from reg_util import texture
    out = []
    for radius, pts in ((M["point_radius"], M["pts"]), (2 * M["point_radius"], M["pts"][::3])):
        pts = np.ascontiguousarray(pts, np.float32)
        _, nn = cKDTree(pts).query(pts, k=M["K"] + 1)
        nbr = nn[:, 1:].astype(np.uint32)
        tex = texture(pts[:, 0].astype(np.float64), pts[:, 2].astype(np.float64)).astype(np.float32)
        out.append(dict(radius=radius, pts=pts, nbr=nbr, tex=tex, fixed=(tex[nbr] - tex[:, None]).astype(np.float32)))
    return out


def _run_tool(d, extra=()):
    cmd = [os.path.join(BIN, "ImageRegistrator"), "--scan_alignment_path", os.path.join(d, "scans.mlp"), "--multi_res_point_cloud_directory_path",
           os.path.join(d, "cache"), "--image_base_path", os.path.join(d, "images"), "--state_path", os.path.join(d, "state"),
           "--output_folder_path", os.path.join(d, "out"), "--observations_cache_path", os.path.join(d, "obs_cache"),
           "--max_iterations", "4", "--max_initial_image_area_in_pixels", "3000"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _read_images_txt(path):
    out = {}
    for line in open(path):
        v = line.split()
        if len(v) == 10 and not line.startswith("#"):
            out[int(v[0])] = (np.array(v[1:5], np.float64), np.array(v[5:8], np.float64), int(v[8]), v[9])
    return out


def test_image_registrator_cli_matches_binding(tmp_path, e3d):
    M = make_multi_image_scene(n_points=6000, n_images=3, seed=12, perturb=0.006)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = _write_dataset(tmp_path, M, names)
    out = _run_tool(d)
    assert "--- Optimizing at scaling factor 0.5 ---" in out and "--- Optimizing at scaling factor 1 ---" in out and "Finished!" in out
    # the same run through the Python binding: 3 image scales (area 43200 / 3000 -> 1 + ceil(log4(14.4)) = 3), scales 1 then 0
    G = e3d.RegProblem(e3d.default_reg_params(image_scale_count=3, point_neighbor_count=M["K"]))
    G.set_intrinsics(0, M["width"], M["height"], M["params"], 0, 3)
    for s_i, sc in enumerate(_point_scales(M)):
        G.set_point_scale(s_i, sc["pts"], sc["radius"], sc["nbr"], sc["fixed"])
    G.set_splat_points(M["pts"])
    from reg_util import pyramid_u8
    for i, im in enumerate(M["images"]):
        G.set_image(i, 0, pyramid_u8(im["pyr"][0], 3)); G.set_image_pose(i, im["q_init"], im["t_init"])
    costs = []
    for scale in (1, 0):
        prm = G.params; prm.current_image_scale = scale; G.set_params(prm)
        G.set_cache_observations(scale != 1)           # enabled after the first scale (image_registrator.cc:230-235)
        costs.append(G.run_on_current_scale(4, 0.0, 15, False)[1])
        st = _read_images_txt(os.path.join(d, "out", "scale_%s_state" % ("0.5" if scale == 1 else "1"), "images.txt"))
        assert sorted(st) == [0, 1, 2]
        for i in range(3):
            q, t = G.get_image_pose(i)
            assert np.abs(st[i][0] - q).max() <= 2e-5 and np.abs(st[i][1] - t).max() <= 2e-5 and st[i][2] == 0 and st[i][3] == names[i]
        cam = open(os.path.join(d, "out", "scale_%s_state" % ("0.5" if scale == 1 else "1"), "cameras.txt")).read().split("\n")[3].split()
        p = G.intrinsics_level(0, 0)[2].astype(np.float64); p[2] += 0.5; p[3] += 0.5
        assert cam[:4] == ["0", "PINHOLE", str(M["width"]), str(M["height"])]
        assert np.abs(np.array(cam[4:], np.float64) - p).max() <= 2e-3
        meta = open(os.path.join(d, "out", "scale_%s_state" % ("0.5" if scale == 1 else "1"), "metadata.txt")).read()
        assert "optimum_cost " in meta and "point_neighbor_count 5" in meta
        assert np.isfinite(costs[-1]) and abs(float(meta.strip().split("optimum_cost ")[1]) - costs[-1]) <= 1e-4 * costs[-1]
    # the `.observed_indices` files written before the second scale (observations_cache.cc:127-158) hold the lists of the binding
    from oracle.reg_driver import observed_indices_path, read_observed_indices
    lists = {}
    for i in range(3):
        fn = observed_indices_path(os.path.join(d, "obs_cache"), os.path.join(d, "images", names[i]))
        assert fn.endswith("obs_cache/dslr/img_%d.png.observed_indices" % i) and os.path.isfile(fn)
        lists[i] = read_observed_indices(fn)
        assert len(lists[i]) == 2 and len(lists[i][0]) > 1000
        for ps in range(2):
            assert np.array_equal(lists[i][ps], G.get_observed_indices(i, ps))
    # second run on the existing cache folder with --cache_observations 1: the lists are loaded (observations_cache.cc:70-102)
    # and drive every scale
    out2 = _run_tool(d, ["--cache_observations", "1"])
    assert "Finished!" in out2
    G2 = e3d.RegProblem(e3d.default_reg_params(image_scale_count=3, point_neighbor_count=M["K"]))
    G2.set_intrinsics(0, M["width"], M["height"], M["params"], 0, 3)
    for s_i, sc in enumerate(_point_scales(M)):
        G2.set_point_scale(s_i, sc["pts"], sc["radius"], sc["nbr"], sc["fixed"])
    G2.set_splat_points(M["pts"])
    for i, im in enumerate(M["images"]):
        G2.set_image(i, 0, pyramid_u8(im["pyr"][0], 3)); G2.set_image_pose(i, im["q_init"], im["t_init"])
        for ps in range(2):
            G2.set_observed_indices(i, ps, lists[i][ps])
    G2.set_cache_observations(True)
    for scale in (1, 0):
        prm = G2.params; prm.current_image_scale = scale; G2.set_params(prm)
        G2.run_on_current_scale(4, 0.0, 15, False)
    st = _read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt"))
    for i in range(3):
        q, t = G2.get_image_pose(i)
        assert np.abs(st[i][0] - q).max() <= 2e-5 and np.abs(st[i][1] - t).max() <= 2e-5
    # a cache folder that lacks a file is fatal, like in the reference
    os.remove(observed_indices_path(os.path.join(d, "obs_cache"), os.path.join(d, "images", names[1])))
    cmd_fail = subprocess.run([os.path.join(BIN, "ImageRegistrator"), "--scan_alignment_path", os.path.join(d, "scans.mlp"),
                               "--multi_res_point_cloud_directory_path", os.path.join(d, "cache"), "--image_base_path", os.path.join(d, "images"),
                               "--state_path", os.path.join(d, "state"), "--output_folder_path", os.path.join(d, "out"), "--observations_cache_path",
                               os.path.join(d, "obs_cache"), "--max_iterations", "2", "--max_initial_image_area_in_pixels", "3000",
                               "--cache_observations", "1"], capture_output=True, text=True, timeout=600)
    assert cmd_fail.returncode != 0 and "Missing file for observed point indices" in cmd_fail.stderr


def test_image_registrator_cli_camera_and_image_masks(tmp_path, e3d):
    """masks_for_cameras/<folder>.png (Image::GetCameraMaskPath) goes to the library as the camera's own mask, masks_for_images/... as the
    image's; the tool's optimum equals the binding's with set_camera_mask / per-image masks on the same data."""
    M = make_multi_image_scene(n_points=6000, n_images=3, seed=31, perturb=0.005)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = _write_dataset(tmp_path, M, names)
    H, W = M["height"], M["width"]
    cmask = np.zeros((H, W), np.uint8); cmask[:, : W // 4] = 2              # kEvalObs over the left quarter, every image of the camera
    imask = np.zeros((H, W), np.uint8); imask[H // 2:, W // 2:] = 1         # kObs, image 1 only
    _write_png(os.path.join(d, "images", "masks_for_cameras", "dslr.png"), cmask)
    _write_png(os.path.join(d, "images", "masks_for_images", "dslr", "img_1.png"), imask)
    out = _run_tool(d)
    assert "Finished!" in out
    from reg_util import pyramid_u8

    def mask_pyr(m):
        lv = [m]
        for _ in range(2):
            a = lv[-1]; h, w = (a.shape[0] // 2) * 2, (a.shape[1] // 2) * 2
            lv.append(a[0:h:2, 0:w:2] | a[0:h:2, 1:w:2] | a[1:h:2, 0:w:2] | a[1:h:2, 1:w:2])
        return lv
    G = e3d.RegProblem(e3d.default_reg_params(image_scale_count=3, point_neighbor_count=M["K"]))
    G.set_intrinsics(0, W, H, M["params"], 0, 3)
    G.set_camera_mask(0, mask_pyr(cmask))
    for s_i, sc in enumerate(_point_scales(M)):
        G.set_point_scale(s_i, sc["pts"], sc["radius"], sc["nbr"], sc["fixed"])
    G.set_splat_points(M["pts"])
    for i, im in enumerate(M["images"]):
        G.set_image(i, 0, pyramid_u8(im["pyr"][0], 3), mask_pyr(imask) if i == 1 else None); G.set_image_pose(i, im["q_init"], im["t_init"])
    for scale in (1, 0):
        prm = G.params; prm.current_image_scale = scale; G.set_params(prm)
        G.set_cache_observations(scale != 1)
        cost = G.run_on_current_scale(4, 0.0, 15, False)[1]
        meta = open(os.path.join(d, "out", "scale_%s_state" % ("0.5" if scale == 1 else "1"), "metadata.txt")).read()
        assert abs(float(meta.strip().split("optimum_cost ")[1]) - cost) <= 1e-4 * cost
    st = _read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt"))
    for i in range(3):
        q, t = G.get_image_pose(i)
        assert np.abs(st[i][0] - q).max() <= 2e-5 and np.abs(st[i][1] - t).max() <= 2e-5


@pytest.mark.parametrize("incomplete", [False, True])
def test_image_registrator_cli_with_rig(tmp_path, e3d, incomplete):
    """incomplete: the second camera's image of frame 1 is not registered in images.txt (its file exists): AssignRigs adds it at
    the pose the rig gives it (rig.cc:236-252), so the run ends with all four images again."""
    M = make_rig_scene(n_points=6000, seed=13)
    # image ids 2f / 2f+1 = cameras "cam0" / "cam1" of frame f; frames share the file name across the two folders
    names = ["cam%d/frame_%d.png" % (i % 2, i // 2) for i in range(4)]
    # COLMAP input carries full poses for every image (no rig knowledge): use the true dependent poses perturbed via the frame
    from oracle import reg_binding as rb
    for f in range(2):
        ref, dep = M["images"][2 * f], M["images"][2 * f + 1]
        dep["q_init"], dep["t_init"] = rb.se3_mul(*M["rig_init"][1], ref["q_init"], ref["t_init"])
    rigs = [{"ref_camera_id": 7, "cameras": [{"camera_id": 7, "image_prefix": "cam0"}, {"camera_id": 7, "image_prefix": "cam1"}]}]
    d = _write_dataset(tmp_path, M, names, rigs=rigs)
    if incomplete:
        lines = open(os.path.join(d, "state", "images.txt")).read().split("\n")
        keep = [l for i, l in enumerate(lines) if not (l.startswith("13 ") or (i > 0 and lines[i - 1].startswith("13 ")))]
        assert len(keep) == len(lines) - 2
        open(os.path.join(d, "state", "images.txt"), "w").write("\n".join(keep))
    out = _run_tool(d, ["--max_initial_image_area_in_pixels", "32000"])
    assert "AssignRigs(): assigned 4 out of 4 images to rig(s)" in out and "Finished!" in out
    st = _read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt"))
    assert len(st) == 4
    rj = open(os.path.join(d, "out", "scale_1_state", "rigs.json")).read()
    import json
    assert json.loads(rj) == [{"ref_camera_id": 0, "cameras": [{"camera_id": 0, "image_prefix": "cam0"}, {"camera_id": 0, "image_prefix": "cam1"}]}]
    # rig consistency of the exported poses: image_T_global(cam1) * global_T_image(cam0) is the same in both frames
    from scipy.spatial.transform import Rotation

    def T(q, t):
        m = np.eye(4); m[:3, :3] = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(); m[:3, 3] = t; return m
    rel = [T(*st[2 * f + 1][:2]) @ np.linalg.inv(T(*st[2 * f][:2])) for f in range(2)]
    assert np.abs(rel[0] - rel[1]).max() <= 1e-4
    # and the costs printed by the tool decreased
    costs = [float(l.split(":")[-1]) for l in out.splitlines() if "Cost (considering occlusions) is" in l]
    assert len(costs) >= 3 and min(costs) < costs[0]


@pytest.mark.parametrize("model,name,n_params", [(3, "OPENCV_FISHEYE", 8), (2, "THIN_PRISM_FISHEYE", 12), (4, "FOV", 5)])
def test_image_registrator_cli_fisheye_models(tmp_path, e3d, model, name, n_params):
    """The distorted camera models through the tool: model name and parameter count survive the COLMAP round trip and the
    photometric cost goes down."""
    M = make_multi_image_scene(n_points=6000, n_images=3, seed=23, perturb=0.004, model=model)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = _write_dataset(tmp_path, M, names, model_name=name)
    out = _run_tool(d)
    assert "Finished!" in out
    cam = open(os.path.join(d, "out", "scale_1_state", "cameras.txt")).read().split("\n")[3].split()
    assert cam[1] == name and len(cam) == 4 + n_params
    p = np.array(cam[4:], np.float64)
    # (the high-order coefficients are barely observable in this small field of view and may wander)
    assert np.isfinite(p).all() and abs(p[4] - M["params"][4]) < 5e-2 and abs(p[0] - M["params"][0]) < 5.0
    costs = [float(l.split(":")[-1]) for l in out.splitlines() if "Cost (considering occlusions) is" in l]
    assert len(costs) >= 4 and np.isfinite(costs).all() and min(costs) < costs[0]


@pytest.mark.parametrize("model,name,out_name,n_params", [(5, "SIMPLE_PINHOLE", "SIMPLE_PINHOLE", 3), (6, "SIMPLE_RADIAL_FISHEYE", "SIMPLE_RADIAL", 4),
                                                         (7, "RADIAL", "RADIAL", 5), (7, "RADIAL_FISHEYE", "RADIAL", 5),
                                                         (8, "POLYNOMIAL_3", "POLYNOMIAL_3", 7),
                                                         (9, "FISHEYE_POLYNOMIAL_2_TANGENTIAL_2", "FISHEYE_POLYNOMIAL_2_TANGENTIAL_2", 8)])
def test_image_registrator_cli_remaining_colmap_models(tmp_path, e3d, model, name, out_name, n_params):
    """The rest of camera_base.cc:66-77 through the tool.  The one-focal-length models keep their [f cx cy ...] parameter order; the
    names RADIAL_FISHEYE / SIMPLE_RADIAL_FISHEYE construct the plain RADIAL / SIMPLE_RADIAL classes in the reference's factory
    ([QUIRK] camera_base.cc:73-74) and are written back under those names."""
    M = make_multi_image_scene(n_points=6000, n_images=3, seed=29, perturb=0.004, model=model)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = _write_dataset(tmp_path, M, names, model_name=name)
    out = _run_tool(d)
    assert "Finished!" in out
    cam = open(os.path.join(d, "out", "scale_1_state", "cameras.txt")).read().split("\n")[3].split()
    assert cam[1] == out_name and len(cam) == 4 + n_params
    p = np.array(cam[4:], np.float64)
    assert np.isfinite(p).all() and abs(p[0] - M["params"][0]) < 5.0
    ci = 1 if model in (5, 6, 7) else 2
    assert abs(p[ci] - (M["params"][ci] + 0.5)) < 3.0 and abs(p[ci + 1] - (M["params"][ci + 1] + 0.5)) < 3.0
    costs = [float(l.split(":")[-1]) for l in out.splitlines() if "Cost (considering occlusions) is" in l]
    assert len(costs) >= 4 and np.isfinite(costs).all() and min(costs) < costs[0]
    # the same optimisation through the binding ends at the same intrinsics and poses
    G = e3d.RegProblem(e3d.default_reg_params(image_scale_count=3, point_neighbor_count=M["K"]))
    G.set_intrinsics(0, M["width"], M["height"], M["params"], 0, 3, camera_type=model)
    for s_i, sc in enumerate(_point_scales(M)):
        G.set_point_scale(s_i, sc["pts"], sc["radius"], sc["nbr"], sc["fixed"])
    G.set_splat_points(M["pts"])
    from reg_util import pyramid_u8
    for i, im in enumerate(M["images"]):
        G.set_image(i, 0, pyramid_u8(im["pyr"][0], 3)); G.set_image_pose(i, im["q_init"], im["t_init"])
    for scale in (1, 0):
        prm = G.params; prm.current_image_scale = scale; G.set_params(prm)
        G.set_cache_observations(scale != 1)
        G.run_on_current_scale(4, 0.0, 15, False)
    pg = G.intrinsics_level(0, 0)[2].astype(np.float64); pg[ci] += 0.5; pg[ci + 1] += 0.5
    assert np.abs(p - pg).max() <= 2e-3 * max(1.0, np.abs(pg).max())
    st = _read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt"))
    for i in range(3):
        q, t = G.get_image_pose(i)
        assert np.abs(st[i][0] - q).max() <= 2e-5 and np.abs(st[i][1] - t).max() <= 2e-5


def test_image_registrator_cli_errors(tmp_path):
    r = subprocess.run([os.path.join(BIN, "ImageRegistrator")], capture_output=True, text=True)
    assert r.returncode != 0 and "Please specify all the required paths." in r.stderr


def _read_cache(d):
    meta = open(os.path.join(d, "metadata.txt")).read().split()
    assert meta[:2] == ["version", "1"]
    nscales = int(meta[meta.index("point_scale_count") + 1])
    radii = [float(meta[i + 1]) for i, v in enumerate(meta) if v == "point_radius"]
    K = int(meta[meta.index("neighbor_count") + 1])
    scales = []
    nb_raw = np.fromfile(os.path.join(d, "neighbor_point_indices"), np.uint64)
    off = 0
    for s in range(nscales):
        data = open(os.path.join(d, "points_of_scale_%d.ply" % s), "rb").read()
        end = data.index(b"end_header\n") + len(b"end_header\n")
        n = int(data[:end].decode().split("element vertex ")[1].split()[0])
        rec = np.frombuffer(data, dtype=[("p", "<f4", 3), ("i", "<f4")], count=n, offset=end)
        scales.append(dict(radius=radii[s], pts=rec["p"].copy(), colors=rec["i"].copy(), nbr=nb_raw[off:off + n * K].reshape(n, K).astype(np.int64)))
        off += n * K
    assert off == len(nb_raw)
    return scales


def test_image_registrator_computes_multires_cloud(tmp_path, e3d):
    """No cache directory: the tool builds the multi-resolution point cloud from the coloured scans (radius ranges and
    merging on the GPU), saves it in the reference's cache format and optimises on it.  Compared with the oracle pipeline."""
    from oracle import multires as mr
    from reg_util import texture
    from tools.make_multires_cache import write_cache  # noqa: F401  (format documented there)
    M = make_multi_image_scene(n_points=9000, n_images=3, seed=14, perturb=0.004)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = _write_dataset(tmp_path, M, names)
    import shutil
    shutil.rmtree(os.path.join(d, "cache"))
    # two coloured scans (grey = the scene texture), identity scan poses
    tex = texture(M["pts"][:, 0].astype(np.float64), M["pts"][:, 2].astype(np.float64)).clip(0, 255)
    rgb = np.repeat(np.rint(tex).astype(np.uint8)[:, None], 3, 1)
    half = len(M["pts"]) // 2
    write_ply_xyz(os.path.join(d, "scan_a.ply"), M["pts"][:half], rgb=rgb[:half])
    write_ply_xyz(os.path.join(d, "scan_b.ply"), M["pts"][half:], rgb=rgb[half:])
    write_mlp(os.path.join(d, "scans.mlp"), [("a", "scan_a.ply", np.eye(4)), ("b", "scan_b.ply", np.eye(4))])
    out = _run_tool(d)          # --max_initial_image_area_in_pixels 3000 -> 3 image scales
    assert "ComputeMultiResPointCloud(): Creating multi-res point cloud ..." in out and "Finished!" in out and "#Image scales: 3" in out
    got = _read_cache(os.path.join(d, "cache"))
    # oracle pipeline on the same inputs (initial poses, as the tool sees them)
    images = {i: dict(intr=0, pyr=im["pyr"], masks=None, q=im["q_init"], t=im["t_init"]) for i, im in enumerate(M["images"])}
    intr = {0: dict(w=M["width"], h=M["height"], params=M["params"], min=0, n=3, model=0)}
    scans = [(M["pts"][:half], rgb[:half]), (M["pts"][half:], rgb[half:])]
    exp = mr.compute_multi_res_point_cloud(scans, images, intr, image_scale_count=3)
    assert len(got) == len(exp) >= 1
    for g, o in zip(got, exp):
        assert abs(g["radius"] - float(o["radius"])) <= 5.1e-6 * float(o["radius"])      # metadata.txt carries 6 significant digits (ostream default)
        assert abs(len(g["pts"]) - len(o["pts"])) <= max(2, len(o["pts"]) // 200)      # merged means differ in the last bits
        assert g["nbr"].max() < len(g["pts"]) and np.all(g["nbr"] != np.arange(len(g["pts"]))[:, None])
        if len(g["pts"]) == len(o["pts"]):
            assert np.abs(g["pts"] - o["pts"]).max() <= 1e-4 and np.abs(g["colors"] - o["colors"]).max() <= 1e-2
            assert (g["nbr"] == o["nbr"].astype(np.int64)).mean() > 0.98
    costs = [float(l.split(":")[-1]) for l in out.splitlines() if "Cost (considering occlusions) is" in l]
    assert len(costs) >= 2 and np.isfinite(costs).all() and min(costs) < costs[0]
    # second run: the saved cache is picked up
    shutil.rmtree(os.path.join(d, "out"))
    out2 = _run_tool(d)
    assert "Loaded existing multi-res point cloud." in out2 and "Creating multi-res point cloud" not in out2


@pytest.mark.parametrize("binary", [True, False])
def test_image_registrator_cli_with_occlusion_mesh(tmp_path, e3d, binary):
    """--occlusion_mesh_path: the wall as a triangle mesh (PLY, binary or ASCII) replaces the scan-point splats as
    occlusion geometry; the tool must run through both image scales and lower the photometric cost."""
    from cli_util import write_ply_mesh
    M = make_multi_image_scene(n_points=6000, n_images=3, seed=15, perturb=0.005)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = _write_dataset(tmp_path, M, names)
    gx, gz = np.meshgrid(np.linspace(-1.6, 1.6, 33), np.linspace(-1.3, 1.3, 27), indexing="ij")
    verts = np.stack([gx.ravel(), np.full(gx.size, 3.0), gz.ravel()], 1)
    tris = []
    for i in range(32):
        for j in range(26):
            a, b, c, dd = i * 27 + j, (i + 1) * 27 + j, (i + 1) * 27 + j + 1, i * 27 + j + 1
            tris += [(a, b, c), (a, c, dd)]
    write_ply_mesh(os.path.join(d, "occlusion.ply"), verts, tris, binary=binary)
    out = _run_tool(d, ["--occlusion_mesh_path", os.path.join(d, "occlusion.ply")])
    assert "adding mesh" in out and "computing edges" in out and "No occlusion meshes given" not in out and "Finished!" in out
    costs = [float(l.split(":")[-1]) for l in out.splitlines() if "Cost (considering occlusions) is" in l]
    assert len(costs) >= 4 and np.isfinite(costs).all() and min(costs) < costs[0]
    assert os.path.exists(os.path.join(d, "out", "scale_1_state", "images.txt"))


# ---- GroundTruthCreator (f4) -----------------------------------------------------------------------------------------------------
def _run_gt(d, extra=()):
    cmd = [os.path.join(BIN, "GroundTruthCreator"), "--scan_alignment_path", os.path.join(d, "scans.mlp"), "--image_base_path",
           os.path.join(d, "images"), "--state_path", os.path.join(d, "state"), "--output_folder_path", os.path.join(d, "gt")] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _read_ply_xyz_pcl(path):
    raw = open(path, "rb").read()
    h = raw.index(b"end_header\n") + 11
    header = raw[:h].decode()
    n = int(header.split("element vertex ")[1].split()[0])
    assert "comment PCL generated" in header and "element camera 1" in header and len(raw) - h - 12 * n == 4 * 12 + 4 * 5 + 4 * 2 + 4 * 2
    return np.frombuffer(raw, "<f4", 3 * n, h).reshape(n, 3)


def test_ground_truth_creator_cli_matches_oracle(tmp_path, e3d):
    """Counts -> trimmed scans, ground-truth depth and occlusion depth maps (raw and gzip) against the oracle; identity scan
    pose, so the upright rotation is the identity and every number is bit-exact."""
    import gzip
    from oracle import reg_binding as rb
    from reg_util import quat_to_R
    M = make_multi_image_scene(n_points=20000, n_images=3, seed=19, perturb=0.0)
    rng = np.random.RandomState(6)
    bx, bz = np.meshgrid(np.arange(-0.3, 0.2, 0.01), np.arange(-0.2, 0.2, 0.01))
    blocker = np.stack([bx.ravel(), np.full(bx.size, 1.6), bz.ravel()], 1)
    M["pts"] = np.concatenate([M["pts"], blocker, rng.uniform(-4, 4, (3000, 3))]).astype(np.float32)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = str(tmp_path)
    write_ply_xyz(os.path.join(d, "scan.ply"), M["pts"], rgb=np.full((len(M["pts"]), 3), 128, np.uint8))
    write_mlp(os.path.join(d, "scans.mlp"), [("scan", "scan.ply", np.eye(4))])
    os.makedirs(os.path.join(d, "state"), exist_ok=True)
    p = M["params"].astype(np.float64).copy(); p[2] += 0.5; p[3] += 0.5
    with open(os.path.join(d, "state", "cameras.txt"), "w") as f:
        f.write("# cameras\n7 PINHOLE %d %d %s\n" % (M["width"], M["height"], " ".join("%.9g" % v for v in p)))
    with open(os.path.join(d, "state", "images.txt"), "w") as f:
        for i, (im, name) in enumerate(zip(M["images"], names)):
            f.write("%d %s %s 7 %s\n\n" % (10 + i, " ".join("%.9g" % v for v in im["q_true"]), " ".join("%.9g" % v for v in im["t_true"]), name))
            _write_png(os.path.join(d, "images", name), im["pyr"][0])
    mask = np.zeros((M["height"], M["width"]), np.uint8)
    mask[40:90, 60:140] = 2; mask[100:120, 10:50] = 1
    _write_png(os.path.join(d, "images", "masks_for_images", "dslr", "img_0.png"), mask)     # <image dir>/../masks_for_images/<folder>/
    out = _run_gt(d)
    assert "Writing COLMAP state file ..." in out and "Writing Point Cloud ..." in out and "Writing depth maps ..." in out
    # oracle: the tool's state = the text it read (float parse of %.9g is exact for f32)
    from oracle.reg_driver import OracleRegProblem
    O = OracleRegProblem(K=M["K"], image_scale_count=3)
    O.set_intrinsics(0, M["width"], M["height"], M["params"], 0, 3)
    cam = O.intr[0]["levels"][0]
    counts = np.zeros(len(M["pts"]), np.int32)
    occ = []
    for i, im in enumerate(M["images"]):
        O.set_image(i, 0, [np.zeros((1, 1), np.uint8)] * 3); O.set_image_pose(i, im["q_true"], im["t_true"])
        R = O._R(O.images[i]); t = O.images[i]["t"]
        occ.append(rb.splat_depth(M["pts"], R, t, cam, 0.03))
        rb.scan_visibility(M["pts"], R, t, cam, occ[i], counts, mask=mask if i == 0 else None)
    trimmed = _read_ply_xyz_pcl(os.path.join(d, "gt", "points", "scan.ply"))
    assert np.array_equal(trimmed, M["pts"][counts >= 2]) and 5000 < len(trimmed) < len(M["pts"])
    mlp = read_mlp(os.path.join(d, "gt", "points", "scan_alignment.mlp"))
    assert len(mlp) == 1 and mlp[0][1] == "scan.ply" and np.allclose(mlp[0][2], np.eye(4))
    for i, im in enumerate(M["images"]):
        R = O._R(O.images[i]); t = O.images[i]["t"]
        ogt = rb.scan_visibility(M["pts"], R, t, cam, occ[i], counts, mask=mask if i == 0 else None, mode=1, min_count=2)
        gt = np.fromfile(os.path.join(d, "gt", "ground_truth_depth", "dslr", "img_%d.png" % i), np.float32).reshape(M["height"], M["width"])
        go = np.fromfile(os.path.join(d, "gt", "occlusion_depth", "dslr", "img_%d.png" % i), np.float32).reshape(M["height"], M["width"])
        # poses pass through the (identity) upright rotation: a quaternion product + renormalisation may move them by an ulp
        for a, b in ((gt, ogt), (go, occ[i])):
            fin = np.isfinite(a) & np.isfinite(b)
            assert (np.isfinite(a) != np.isfinite(b)).mean() < 1e-3 and (np.abs(a[fin] - b[fin]) > 1e-5).mean() < 1e-3
        assert np.isfinite(gt).sum() > 2000
    cal = _read_images_txt(os.path.join(d, "gt", "calibration", "images.txt"))
    assert sorted(cal) == [0, 1, 2] and np.abs(cal[1][0] - M["images"][1]["q_true"]).max() < 1e-6
    # compressed variant, depth maps only
    shutil.rmtree(os.path.join(d, "gt"))
    _run_gt(d, ["--compress_depth_maps", "1", "--write_point_cloud", "0", "--write_occlusion_depth", "0"])
    assert not os.path.exists(os.path.join(d, "gt", "points")) and not os.path.exists(os.path.join(d, "gt", "occlusion_depth"))
    gz = np.frombuffer(gzip.open(os.path.join(d, "gt", "ground_truth_depth", "dslr", "img_2.png.gz")).read(), np.float32)
    fin = np.isfinite(gz) & np.isfinite(ogt.ravel())
    assert gz.shape == (M["height"] * M["width"],) and fin.sum() > 2000 and (np.abs(gz[fin] - ogt.ravel()[fin]) > 1e-5).mean() < 1e-3


def test_ground_truth_creator_scan_renderings(tmp_path, e3d):
    """--write_scan_renderings (ground_truth_creator.cc:149,175-199): every image painted with the visible scan points seen in >= 2
    images, squares of 2 * scan_point_radius + 1 pixels in point order, written under the image's own name -- a JPEG image gives a JPEG
    (libjpeg quality 95 = what cv::imwrite writes), a PNG a PNG.  Expected: the oracle's sequential painting over Pillow's decoding of
    the same files, compared pixel for pixel (PNG) and byte for byte (JPEG, against Pillow's encoder)."""
    from PIL import Image
    from oracle import reg_binding as rb
    from oracle.reg_driver import OracleRegProblem
    M = make_multi_image_scene(n_points=20000, n_images=3, seed=21, perturb=0.0)
    rng = np.random.RandomState(7)
    M["pts"] = np.concatenate([M["pts"], rng.uniform(-4, 4, (2000, 3))]).astype(np.float32)
    colors = rng.randint(0, 256, (len(M["pts"]), 3)).astype(np.uint8)
    names = ["dslr/img_0.png", "dslr/img_1.jpg", "dslr/img_2.png"]
    d = str(tmp_path)
    write_ply_xyz(os.path.join(d, "scan.ply"), M["pts"], rgb=colors)
    write_mlp(os.path.join(d, "scans.mlp"), [("scan", "scan.ply", np.eye(4))])
    os.makedirs(os.path.join(d, "state"), exist_ok=True)
    os.makedirs(os.path.join(d, "images", "dslr"), exist_ok=True)
    p = M["params"].astype(np.float64).copy(); p[2] += 0.5; p[3] += 0.5
    with open(os.path.join(d, "state", "cameras.txt"), "w") as f:
        f.write("# cameras\n7 PINHOLE %d %d %s\n" % (M["width"], M["height"], " ".join("%.9g" % v for v in p)))
    yy, xx = np.mgrid[0:M["height"], 0:M["width"]]
    with open(os.path.join(d, "state", "images.txt"), "w") as f:
        for i, (im, name) in enumerate(zip(M["images"], names)):
            f.write("%d %s %s 7 %s\n\n" % (10 + i, " ".join("%.9g" % v for v in im["q_true"]), " ".join("%.9g" % v for v in im["t_true"]), name))
            col = np.stack([im["pyr"][0], (im["pyr"][0].astype(int) + xx) % 256, (yy * 2) % 256], -1).astype(np.uint8)      # a colour image
            if name.endswith(".jpg"):
                Image.fromarray(col, "RGB").save(os.path.join(d, "images", name), "JPEG", quality=90, subsampling=2)
            else:
                Image.fromarray(col, "RGB").save(os.path.join(d, "images", name))
    out = _run_gt(d, ["--write_scan_renderings", "1", "--scan_point_radius", "1", "--write_point_cloud", "0"])
    assert "Writing scan renderings ..." in out
    O = OracleRegProblem(K=M["K"], image_scale_count=3)
    O.set_intrinsics(0, M["width"], M["height"], M["params"], 0, 3)
    cam = O.intr[0]["levels"][0]
    counts = np.zeros(len(M["pts"]), np.int32)
    occ = []
    for i, im in enumerate(M["images"]):
        O.set_image(i, 0, [np.zeros((1, 1), np.uint8)] * 3); O.set_image_pose(i, im["q_true"], im["t_true"])
        occ.append(rb.splat_depth(M["pts"], O._R(O.images[i]), O.images[i]["t"], cam, 0.03))
        rb.scan_visibility(M["pts"], O._R(O.images[i]), O.images[i]["t"], cam, occ[i], counts)
    for i, name in enumerate(names):
        win = rb.scan_rendering(M["pts"], O._R(O.images[i]), O.images[i]["t"], cam, occ[i], counts, 1)
        exp = np.array(Image.open(os.path.join(d, "images", name)).convert("RGB"))
        painted = win > 0
        exp[painted] = colors[win[painted] - 1]
        assert painted.sum() > 5000 and (~painted).sum() > 5000
        path = os.path.join(d, "gt", "scan_rendering", name)
        if name.endswith(".png"):
            got = np.array(Image.open(path).convert("RGB"))
            assert (got != exp).any(-1).mean() < 2e-3          # the poses pass through the (identity) upright rotation: an ulp may move a point
        else:
            Image.fromarray(exp, "RGB").save(os.path.join(d, "exp.jpg"), "JPEG", quality=95, subsampling=2)
            a, b = open(path, "rb").read(), open(os.path.join(d, "exp.jpg"), "rb").read()
            if a != b:                                         # same caveat: then at least nearly all decoded pixels agree
                ga, gb = np.array(Image.open(path).convert("RGB")).astype(int), np.array(Image.open(os.path.join(d, "exp.jpg")).convert("RGB")).astype(int)
                assert (np.abs(ga - gb).max(-1) > 8).mean() < 5e-3
    assert os.path.exists(os.path.join(d, "gt", "ground_truth_depth", "dslr", "img_1.jpg"))


def test_ground_truth_creator_rotates_first_scan_upright(tmp_path, e3d):
    """A tilted first scan: scans, cameras and the written scan_alignment.mlp are all moved by U = (R0^-1, t0 - R0^-1 t0)
    (ground_truth_creator.cc:275-291, :333-339); visibility is invariant under that rigid motion."""
    from scipy.spatial.transform import Rotation
    from reg_util import quat_to_R
    M = make_multi_image_scene(n_points=20000, n_images=3, seed=20, perturb=0.0)
    names = ["dslr/img_%d.png" % i for i in range(3)]
    d = str(tmp_path)
    T0 = np.eye(4); T0[:3, :3] = Rotation.from_euler("xyz", [0.2, -0.1, 0.4]).as_matrix(); T0[:3, 3] = [0.3, -0.2, 0.1]
    local = (M["pts"].astype(np.float64) - T0[:3, 3]) @ T0[:3, :3]            # R0^T (p - t0)
    write_ply_xyz(os.path.join(d, "scan.ply"), local.astype(np.float32), rgb=np.full((len(local), 3), 128, np.uint8))
    write_mlp(os.path.join(d, "scans.mlp"), [("scan", "scan.ply", T0)])
    os.makedirs(os.path.join(d, "state"), exist_ok=True)
    p = M["params"].astype(np.float64).copy(); p[2] += 0.5; p[3] += 0.5
    with open(os.path.join(d, "state", "cameras.txt"), "w") as f:
        f.write("7 PINHOLE %d %d %s\n" % (M["width"], M["height"], " ".join("%.9g" % v for v in p)))
    with open(os.path.join(d, "state", "images.txt"), "w") as f:
        for i, (im, name) in enumerate(zip(M["images"], names)):
            f.write("%d %s %s 7 %s\n\n" % (10 + i, " ".join("%.9g" % v for v in im["q_true"]), " ".join("%.9g" % v for v in im["t_true"]), name))
            _write_png(os.path.join(d, "images", name), im["pyr"][0])
    _run_gt(d, ["--write_occlusion_depth", "0"])
    n_up = len(_read_ply_xyz_pcl(os.path.join(d, "gt", "points", "scan.ply")))
    gt_up = np.fromfile(os.path.join(d, "gt", "ground_truth_depth", "dslr", "img_1.png"), np.float32)
    cal_up = _read_images_txt(os.path.join(d, "gt", "calibration", "images.txt"))
    mlp_up = read_mlp(os.path.join(d, "gt", "points", "scan_alignment.mlp"))
    shutil.rmtree(os.path.join(d, "gt"))
    _run_gt(d, ["--rotate_first_scan_upright", "0", "--write_occlusion_depth", "0"])
    n_plain = len(_read_ply_xyz_pcl(os.path.join(d, "gt", "points", "scan.ply")))
    gt_plain = np.fromfile(os.path.join(d, "gt", "ground_truth_depth", "dslr", "img_1.png"), np.float32)
    cal_plain = _read_images_txt(os.path.join(d, "gt", "calibration", "images.txt"))
    mlp_plain = read_mlp(os.path.join(d, "gt", "points", "scan_alignment.mlp"))
    assert n_plain > 10000 and abs(n_up - n_plain) <= n_plain // 200
    both = np.isfinite(gt_up) & np.isfinite(gt_plain)
    assert (np.isfinite(gt_up) != np.isfinite(gt_plain)).mean() < 5e-3 and np.abs(gt_up[both] - gt_plain[both]).max() < 1e-3
    # the first scan ends up without rotation, at its old position; the plain run keeps T0
    assert np.abs(mlp_plain[0][2] - T0).max() < 1e-5
    assert np.abs(mlp_up[0][2][:3, :3] - np.eye(3)).max() < 1e-5 and np.abs(mlp_up[0][2][:3, 3] - T0[:3, 3]).max() < 1e-5
    # cameras: image_T_global' = image_T_global * U^-1
    U = np.eye(4); U[:3, :3] = T0[:3, :3].T; U[:3, 3] = T0[:3, 3] - T0[:3, :3].T @ T0[:3, 3]
    for i in range(3):
        def mat(q, t):
            m = np.eye(4); m[:3, :3] = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(); m[:3, 3] = t; return m
        assert np.abs(mat(*cal_up[i][:2]) - mat(*cal_plain[i][:2]) @ np.linalg.inv(U)).max() < 1e-5


# ---- the reference's own two-frame alignment test on its own data (src/opt/test/test_alignment.cc, TestPairAlignment) --------------
@pytest.mark.parametrize("case", ["identical_images", "small_offset"])
def test_reference_pair_alignment_thresholds(tmp_path, case):
    """ProcessOnePair (test_alignment_util.cc:122-250) through the drop-in tool: the point cloud of image a's depth map is the
    scan, both images start at the identity pose, the multi-resolution cloud is computed from the scan, all image scales are
    optimised; the recovered relative pose must meet the reference test's thresholds (translation error <= 1e-2 of the scene
    depth, rotation error <= 1 degree).  Data: tests/golden/alignment_test_data.npz (the reference's test_data)."""
    from PIL import Image
    from scipy.spatial.transform import Rotation
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alignment_test_data.npz"))
    w, h, fx, fy, cx, cy, depth_factor = G[case + "_calibration"]
    w, h = int(w), int(h)
    files = [str(v) for v in G[case + "_files"]]
    a_img, a_depth, b_img = G[files[0]], G[files[1]], G[files[2]]
    # unprojection as in the test: depth = depth_factor (float) * u16; p = (depth * nxy, depth), nxy = ImageToNormalized(x, y)
    fx32, fy32, cx32, cy32 = np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy)
    fx_inv, fy_inv = np.float32(1.0 / np.float64(fx32)), np.float32(1.0 / np.float64(fy32))
    cx_inv, cy_inv = np.float32(-1.0 * np.float64(cx32) / np.float64(fx32)), np.float32(-1.0 * np.float64(cy32) / np.float64(fy32))
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (np.float32(depth_factor) * a_depth.astype(np.float32)).astype(np.float64)
    nx = (fx_inv * xx.astype(np.float32) + cx_inv).astype(np.float64); ny = (fy_inv * yy.astype(np.float32) + cy_inv).astype(np.float64)
    keep = depth != 0
    pts = np.stack([depth * nx, depth * ny, depth], -1)[keep].astype(np.float32)
    rgb = a_img[keep]
    d = str(tmp_path)
    write_ply_xyz(os.path.join(d, "scan.ply"), pts, rgb=rgb)
    write_mlp(os.path.join(d, "scans.mlp"), [("scan", "scan.ply", np.eye(4))])
    os.makedirs(os.path.join(d, "state")); os.makedirs(os.path.join(d, "images", "cam"))
    with open(os.path.join(d, "state", "cameras.txt"), "w") as f:
        f.write("1 PINHOLE %d %d %.9g %.9g %.9g %.9g\n" % (w, h, fx, fy, cx + 0.5, cy + 0.5))
    with open(os.path.join(d, "state", "images.txt"), "w") as f:
        f.write("1 1 0 0 0 0 0 0 1 cam/a.png\n\n2 1 0 0 0 0 0 0 1 cam/b.png\n\n")
    Image.fromarray(a_img, "RGB").save(os.path.join(d, "images", "cam", "a.png"))
    Image.fromarray(b_img, "RGB").save(os.path.join(d, "images", "cam", "b.png"))
    cmd = [os.path.join(BIN, "ImageRegistrator"), "--scan_alignment_path", os.path.join(d, "scans.mlp"), "--multi_res_point_cloud_directory_path",
           os.path.join(d, "cache"), "--image_base_path", os.path.join(d, "images"), "--state_path", os.path.join(d, "state"),
           "--output_folder_path", os.path.join(d, "out"), "--observations_cache_path", os.path.join(d, "obs_cache"),
           "--max_iterations", "300", "--max_initial_image_area_in_pixels", str(80 * 60)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "#Image scales: 3" in r.stdout and "--- Optimizing at scaling factor 1 ---" in r.stdout
    st = _read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt"))

    def T(q, t):
        m = np.eye(4); m[:3, :3] = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(); m[:3, 3] = t; return m
    a_T_b = T(*st[0][:2]) @ np.linalg.inv(T(*st[1][:2]))            # model.image_T_global * query.global_T_image
    gt = G[case + "_a_t_b"]
    t_err = np.linalg.norm(a_T_b[:3, 3] - gt[:, 3]) / float(G[case + "_average_scene_depth"])
    Rd = a_T_b[:3, :3].T @ gt[:, :3]
    r_err = np.degrees(np.arccos(np.clip((np.trace(Rd) - 1) / 2, -1, 1)))
    print(case, "translation error / scene depth", t_err, "rotation error deg", r_err)
    assert t_err <= 1e-2 and r_err <= 1.0, (t_err, r_err)


# ---- the reference's four-frame alignment tests (src/opt/test/test_alignment.cc:87-634, TEST(Alignment, FourFrame_*) :649-697) -----
from reg_util import se3_log as _se3_log


@pytest.mark.parametrize("use_variable_colors,use_rig", [(False, False), (True, False), (False, True), (True, True)])
def test_reference_four_frame_alignment_thresholds(tmp_path, use_variable_colors, use_rig):
    """FourFrame_FixedColorsOnly / _FixedAndVariableColors, each without and with a rig, through the drop-in tool: two recordings
    of a two-camera rig see a colour-interpolated heightmap; the scan is 30 % of the pixels of the four depth maps; every pose
    starts 2 mm (camera 0) or 6 mm (camera 1) off in x and y.  After optimising all image scales with the test's parameters every
    component of log(result * ground_truth^-1) must be <= 0.0016 and the mean optical flow between the ground-truth and the
    resulting projections <= 0.07 px (test_alignment.cc:541-603).  The scene comes from numpy's generator instead of std::mt19937,
    the images from a software renderer instead of OpenGL (tests/reg_util.py:make_four_frame_scene); the depth-residual variant
    (FourFrame_DepthResidualVerification) feeds depth maps through the library and lives in tests/test_gpu_reg.py."""
    import json
    from PIL import Image
    from scipy.spatial.transform import Rotation
    from reg_util import make_four_frame_scene
    S = make_four_frame_scene(seed=0)
    d = str(tmp_path)
    write_ply_xyz(os.path.join(d, "scan.ply"), S["pts"], rgb=S["rgb"])
    write_mlp(os.path.join(d, "scans.mlp"), [("scan", "scan.ply", np.eye(4))])
    os.makedirs(os.path.join(d, "state"))
    fx, fy, cx, cy = S["params"]
    with open(os.path.join(d, "state", "cameras.txt"), "w") as f:
        f.write("1 PINHOLE %d %d %.9g %.9g %.9g %.9g\n" % (S["width"], S["height"], fx, fy, cx + 0.5, cy + 0.5))
    keys = [(0, 0), (0, 1), (1, 0), (1, 1)]                      # (rig_image_set, camera_index), the order the test adds them in
    with open(os.path.join(d, "state", "images.txt"), "w") as f:
        for i, (s, c) in enumerate(keys):
            name = "camera%d/image%d.png" % (c, s)
            os.makedirs(os.path.join(d, "images", "camera%d" % c), exist_ok=True)
            Image.fromarray(S["images"][(s, c)]["color"], "RGB").save(os.path.join(d, "images", name))
            f.write("%d 1 0 0 0 %s 1 %s\n\n" % (i + 1, " ".join("%.9g" % v for v in S["images"][(s, c)]["t_init"]), name))
    if use_rig:
        json.dump([{"ref_camera_id": 1, "cameras": [{"camera_id": 1, "image_prefix": "camera0"}, {"camera_id": 1, "image_prefix": "camera1"}]}],
                  open(os.path.join(d, "state", "rigs.json"), "w"), indent=4)
    cmd = [os.path.join(BIN, "ImageRegistrator"), "--scan_alignment_path", os.path.join(d, "scans.mlp"), "--multi_res_point_cloud_directory_path",
           os.path.join(d, "cache"), "--image_base_path", os.path.join(d, "images"), "--state_path", os.path.join(d, "state"),
           "--output_folder_path", os.path.join(d, "out"), "--observations_cache_path", os.path.join(d, "obs_cache"),
           "--max_iterations", "500", "--point_neighbor_count", "5", "--point_neighbor_candidate_count", "25",
           "--min_mean_intensity_difference_for_points", "0", "--robust_weighting_type", "tukey", "--robust_weighting_parameter", "5",
           "--max_initial_image_area_in_pixels", str(64 * 64), "--occlusion_depth_threshold", "0.05",
           "--fixed_residuals_weight", "1", "--variable_residuals_weight", "1" if use_variable_colors else "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "#Image scales: 3" in r.stdout and "--- Optimizing at scaling factor 1 ---" in r.stdout
    if use_rig:
        assert "AssignRigs(): assigned 4 out of 4 images to rig(s)" in r.stdout
    st = _read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt"))
    cam = open(os.path.join(d, "out", "scale_1_state", "cameras.txt")).read().split("\n")[3].split()
    rfx, rfy, rcx, rcy = [float(v) for v in cam[4:8]]
    rcx -= 0.5; rcy -= 0.5
    worst = 0.0
    flow_sum = flow_count = 0
    by_name = {v[3]: v for v in st.values()}
    for (s, c) in keys:
        q, t = by_name["camera%d/image%d.png" % (c, s)][:2]
        Tr = np.eye(4); Tr[:3, :3] = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(); Tr[:3, 3] = t
        im = S["images"][(s, c)]
        Tg = np.eye(4); Tg[:3, :3] = im["R"]; Tg[:3, 3] = im["t"]
        delta = _se3_log(Tr @ np.linalg.inv(Tg))
        worst = max(worst, np.abs(delta).max())
        ys, xs = np.nonzero(im["depth"] > 0)
        dd = im["depth"][ys, xs].astype(np.float64)
        P = np.stack([dd * (xs - cx) / fx, dd * (ys - cy) / fy, dd, np.ones_like(dd)], 0)
        Q = Tr @ np.linalg.inv(Tg) @ P
        ok = Q[2] > 0
        flow = np.hypot(rfx * Q[0, ok] / Q[2, ok] + rcx - xs[ok], rfy * Q[1, ok] / Q[2, ok] + rcy - ys[ok])
        flow_sum += flow.sum(); flow_count += ok.sum()
    mean_flow = flow_sum / flow_count
    print("variable", use_variable_colors, "rig", use_rig, "worst log component", worst, "mean flow px", mean_flow)
    assert worst <= 0.0016 and mean_flow <= 0.07, (worst, mean_flow)


def test_image_registrator_arrow_solver_matches_dense(tmp_path, e3d):
    """Large problems solve the arrow-structured normal equations through the Schur complement instead of the dense LDLT
    (E3D_REG_SOLVER forces either): same run, same result to f64 rounding -- rig scene, so that the shared block holds intrinsics
    and rig extrinsics and the pose blocks belong to rig frames."""
    M = make_rig_scene(n_points=6000, seed=13)
    names = ["cam%d/frame_%d.png" % (i % 2, i // 2) for i in range(4)]
    from oracle import reg_binding as rb
    for f in range(2):
        ref, dep = M["images"][2 * f], M["images"][2 * f + 1]
        dep["q_init"], dep["t_init"] = rb.se3_mul(*M["rig_init"][1], ref["q_init"], ref["t_init"])
    rigs = [{"ref_camera_id": 7, "cameras": [{"camera_id": 7, "image_prefix": "cam0"}, {"camera_id": 7, "image_prefix": "cam1"}]}]
    res = {}
    for solver in ("dense", "arrow"):
        os.makedirs(str(tmp_path / solver))
        d = _write_dataset(tmp_path / solver, M, names, rigs=rigs)
        os.environ["E3D_REG_SOLVER"] = solver
        try:
            out = _run_tool(d, ["--max_initial_image_area_in_pixels", "32000", "--max_iterations", "6"])
        finally:
            del os.environ["E3D_REG_SOLVER"]
        assert "Finished!" in out
        res[solver] = (_read_images_txt(os.path.join(d, "out", "scale_1_state", "images.txt")),
                       [float(l.split(":")[-1]) for l in out.splitlines() if "Cost (considering occlusions) is" in l])
    a, b = res["dense"], res["arrow"]
    assert len(a[1]) == len(b[1]) and np.allclose(a[1], b[1], rtol=1e-6)
    for k in a[0]:
        assert np.abs(a[0][k][0] - b[0][k][0]).max() < 1e-6 and np.abs(a[0][k][1] - b[0][k][1]).max() < 1e-6
