"""End-to-end runs of the drop-in binaries on the GPU box: same inputs through ICPScanAligner / NormalEstimator and
through the CPU oracle; stdout contract (correspondence counts), output poses and output PLY are compared."""
import os
import re
import subprocess

import numpy as np
import pytest

from cli_util import BIN, read_mlp, read_ply_normals, write_mlp, write_ply_xyz
from conftest import pose_error

pytestmark = pytest.mark.gpu


def _project(tmp_path, synth, n_scans, n_points, seed, binary=True, rgb=False):
    scans = synth.make_scene(n_scans, n_points, seed=seed)
    entries = []
    for i, s in enumerate(scans):
        fn = "scan%d.ply" % i
        colors = (np.random.RandomState(i).randint(0, 256, (n_points, 3)).astype(np.uint8)) if rgb else None
        write_ply_xyz(str(tmp_path / fn), s["xyz"].numpy(), colors, binary=binary)
        s["rgb"] = colors
        entries.append(("scan%d" % i, fn, s["T_init"].astype(np.float64)))
    write_mlp(str(tmp_path / "in.mlp"), entries)
    return scans


def test_icp_scan_aligner_cli_matches_oracle(tmp_path, synth, ob, e3d):
    scans = _project(tmp_path, synth, 3, 20000, seed=77, binary=True)
    out = tmp_path / "out.mlp"
    iters = 4
    r = subprocess.run([os.path.join(BIN, "ICPScanAligner"), "-i", str(tmp_path / "in.mlp"), "-o", str(out), "-d", "0.15",
                        "--max_iterations", str(iters), "--convergence_threshold", "1e-10",
                        "--objects_to_optimize", "scan1.ply;scan2.ply", "--normal_estimation_neighbor_count", "16"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the oracle with the same steps as the tool: normals (k = 16, viewpoint origin) then ICP, scan0 fixed
    o = ob.OracleICP()
    ids = []
    for i, s in enumerate(scans):
        P = s["xyz"].numpy()
        n, _ = ob.normals(P, k=16, viewpoint=(0, 0, 0))
        # the .mlp text carries 9 significant digits -> poses as the tool parses them
        ids.append(o.add_point_cloud(P, n, np.array(["%.9g" % v for v in s["T_init"].ravel()], np.float64).reshape(4, 4).astype(np.float32), i == 0))
    for it in range(iters):
        o.run(0.15, it, 1, 1e-10, False)
    # stdout contract: iteration headers + one "found correspondences" line per directed pair with the exact counts
    assert r.stdout.count("-- Alignment iteration") == iters and "Starting ICP ..." in r.stdout and r.stdout.rstrip().endswith("Finished!")
    got = re.findall(r"found correspondences from (fixed clouds|\d+) to (fixed clouds|\d+): (\d+)", r.stdout)
    exp = [(("fixed clouds" if a < 0 else str(a)), ("fixed clouds" if b < 0 else str(b)), str(c)) for (_, a, b, c, _) in o.pair_records()]
    # normals are bit-identical to the oracle's (shared elementary functions), so every iteration's counts are exact
    assert got == exp
    m = read_mlp(str(out))
    assert [x[1] for x in m] == ["scan0.ply", "scan1.ply", "scan2.ply"]
    assert np.allclose(m[0][2], scans[0]["T_init"], atol=1e-5)          # fixed scan untouched
    for i in (1, 2):
        ang, tr = pose_error(m[i][2], o.get_result_global_T_cloud(ids[i]))
        assert ang <= 2e-5 and tr <= 1e-4 + 1e-5 * np.abs(m[i][2][:3, 3]).max()   # .mlp text keeps 6 significant digits


def test_icp_scan_aligner_multiscale_runs(tmp_path, synth):
    _project(tmp_path, synth, 2, 30000, seed=78, binary=False)            # ascii PLY input
    out = tmp_path / "out.mlp"
    r = subprocess.run([os.path.join(BIN, "ICPScanAligner"), "-i", str(tmp_path / "in.mlp"), "-o", str(out), "-d", "0.05",
                        "--max_iterations", "3", "--number_of_scales", "2", "--downscale_step", "4"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Optimizing at scale 0" in r.stdout and "Optimizing at scale 1" in r.stdout
    assert r.stdout.count("Loaded mesh: scan0.ply") == 2                 # re-read from disk at every scale
    # scale 0 uses every 4th point and twice the search distance
    first = re.findall(r"found correspondences from 0 to 1: (\d+)", r.stdout)
    assert len(first) >= 2 and int(first[0]) <= 7500
    assert os.path.exists(out)


def test_normal_estimator_cli(tmp_path, synth, ob):
    scans = _project(tmp_path, synth, 2, 15000, seed=79, binary=True, rgb=True)
    out = tmp_path / "normals.ply"
    r = subprocess.run([os.path.join(BIN, "NormalEstimator"), "-i", str(tmp_path / "in.mlp"), "-o", str(out), "--neighbor_count", "8"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Finished!" in r.stdout, r.stdout + r.stderr
    header, rec, tail = read_ply_normals(str(out))
    assert "property float nx" in header and "property uchar blue" in header and "element camera 1" in header
    assert rec.shape[0] == 30000 and tail == 4 * 12 + 4 * 5 + 4 * 2 + 4 * 2      # PCL's camera block follows the vertices
    off = 0
    for s in scans:
        T = np.array(["%.9g" % v for v in s["T_init"].ravel()], np.float64).reshape(4, 4).astype(np.float32)
        gx, _, _, _ = ob.transform_cloud(s["xyz"].numpy(), s["xyz"].numpy(), T)
        n = s["xyz"].shape[0]
        # unit-scale project: scale_factor = 1/scale(first matrix) ~ 1 -> positions equal the transformed cloud to f32 rounding
        assert np.abs(rec["p"][off:off + n] - gx).max() <= 2e-5
        assert np.array_equal(rec["c"][off:off + n], s["rgb"])
        on, _ = ob.normals(rec["p"][off:off + n].copy(), k=8, viewpoint=tuple(T[:3, 3]))
        good = np.abs(np.einsum("ij,ij->i", rec["n"][off:off + n], on)) > 1 - 1e-3
        assert good.mean() > 0.995
        off += n


def test_point_cloud_cleaner_cli(tmp_path, ob):
    """PointCloudCleaner: two filter passes; inliers / outliers files (x y z + rgb, PCL header) match the oracle filter chain."""
    rng = np.random.RandomState(21)
    plane = np.stack([rng.uniform(-2, 2, 40000), rng.uniform(-2, 2, 40000), 0.003 * rng.normal(size=40000)], 1)
    pts = np.concatenate([plane, rng.uniform(-2, 2, (800, 3))]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    rgb = rng.randint(0, 256, (len(pts), 3)).astype(np.uint8)
    path = str(tmp_path / "scan.ply")
    write_ply_xyz(path, pts, rgb=rgb)
    r = subprocess.run([os.path.join(BIN, "PointCloudCleaner"), "--in", path, "--filter", "8,2.0", "--filter", "19.6,1.5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Applying filter with knn = 8, factor = 2 ..." in r.stderr and "Applying filter with knn = 20, factor = 1.5 ..." in r.stderr

    def read(p):
        raw = open(p, "rb").read()
        h = raw.index(b"end_header\n") + 11
        header = raw[:h].decode()
        n = int(header.split("element vertex ")[1].split()[0])
        rec = np.frombuffer(raw, np.dtype([("p", "<f4", 3), ("c", "u1", 3)]), n, h)
        assert len(raw) - h - 15 * n == 4 * 12 + 4 * 5 + 4 * 2 + 4 * 2 and "property uchar red" in header and "element camera 1" in header
        return rec
    inl, outl = read(path + ".inliers.ply"), read(path + ".outliers.ply")
    k1, _ = ob.local_outlier_removal(pts, 8, 2.0)
    p1, c1 = pts[k1], rgb[k1]
    k2, _ = ob.local_outlier_removal(p1, 20, 1.5)
    assert np.array_equal(inl["p"], p1[k2]) and np.array_equal(inl["c"], c1[k2])
    exp_out_p = np.concatenate([pts[~k1], p1[~k2]]); exp_out_c = np.concatenate([rgb[~k1], c1[~k2]])
    assert np.array_equal(outl["p"], exp_out_p) and np.array_equal(outl["c"], exp_out_c)
    assert len(inl) + len(outl) == len(pts) and 300 < len(outl) < 5000


def test_icp_scan_aligner_gpus_flag(tmp_path, synth, e3d):
    """--gpus N (one host thread per GPU, the library's RCCL communicator): same correspondence counts on stdout and the same
    result poses (1e-5 rad / 1e-4 m) as the single-GPU run.  Needs more than one visible GPU; asking for more GPUs than there
    are is refused with a message."""
    _project(tmp_path, synth, 3, 20000, seed=79, binary=True)
    n_dev = int(e3d.lib().e3d_init(0))
    base = [os.path.join(BIN, "ICPScanAligner"), "-i", str(tmp_path / "in.mlp"), "-d", "0.15", "--max_iterations", "3",
            "--convergence_threshold", "1e-10", "--normal_estimation_neighbor_count", "16"]
    r = subprocess.run(base + ["-o", str(tmp_path / "too_many.mlp"), "--gpus", str(n_dev + 1)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "HIP device" in (r.stdout + r.stderr)
    if n_dev < 2:
        pytest.skip("one visible GPU: the multi-GPU tool path needs at least two")
    r1 = subprocess.run(base + ["-o", str(tmp_path / "one.mlp")], capture_output=True, text=True, timeout=300)
    rn = subprocess.run(base + ["-o", str(tmp_path / "many.mlp"), "--gpus", str(n_dev)], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and rn.returncode == 0, rn.stdout + rn.stderr
    pat = r"found correspondences from (fixed clouds|\d+) to (fixed clouds|\d+): (\d+)"
    assert re.findall(pat, r1.stdout) == re.findall(pat, rn.stdout)
    for a, b in zip(read_mlp(str(tmp_path / "one.mlp")), read_mlp(str(tmp_path / "many.mlp"))):
        ang, tr = pose_error(a[2], b[2])
        assert ang <= 1e-5 and tr <= 1e-4
