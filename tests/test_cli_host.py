"""Host-side (no GPU) behaviour of the drop-in tools: flag handling, .mlp / PLY I/O, early-out paths."""
import os
import subprocess

import numpy as np

from cli_util import BIN, read_mlp, write_mlp, write_ply_xyz


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(BIN), "csrc", "host")])


def test_usage_and_exit_codes(tmp_path):
    _build()
    r = subprocess.run([os.path.join(BIN, "ICPScanAligner")], capture_output=True, text=True)
    assert r.returncode != 0 and "Please provide input and output MeshLab project paths with -i and -o." in r.stdout
    r = subprocess.run([os.path.join(BIN, "NormalEstimator"), "-i", "x.mlp"], capture_output=True, text=True)
    assert r.returncode != 0 and "Please provide input paths." in r.stdout
    r = subprocess.run([os.path.join(BIN, "ICPScanAligner"), "-i", str(tmp_path / "missing.mlp"), "-o", str(tmp_path / "o.mlp")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot load MeshLab project" in r.stdout


def test_single_object_early_out_roundtrips_the_project(tmp_path):
    """0 or 1 movable object and nothing fixed: the tool only rewrites the project (icp_scan_aligner.cc:262-272).
    Exercises the .mlp reader/writer (entities, label/filename, 6-significant-digit matrix text with trailing
    spaces) without touching the GPU."""
    _build()
    T = np.eye(4); T[:3, 3] = [1.5, -2.25, 0.333333333]; T[0, 0] = 0.999999999
    write_ply_xyz(str(tmp_path / "a.ply"), np.zeros((3, 3), np.float32))
    write_mlp(str(tmp_path / "in.mlp"), [("scan &amp; one", "a.ply", T), ("two", "b.ply", np.eye(4))])
    out = tmp_path / "out.mlp"
    r = subprocess.run([os.path.join(BIN, "ICPScanAligner"), "-i", str(tmp_path / "in.mlp"), "-o", str(out),
                        "--objects_to_optimize", "a.ply", "--objects_to_ignore", "b.ply", "--unknown_flag", "7"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Starting alignment with the following parameters:" in r.stdout
    assert "  max_num_iterations: 50" in r.stdout and "  max_correspondence_distance: 0.1" in r.stdout
    assert "  optimizing a.ply" in r.stdout and "  ignoring b.ply" in r.stdout
    assert "Warning: Not enough active objects" in r.stdout
    m = read_mlp(str(out))
    assert [(x[0], x[1]) for x in m] == [("scan &amp; one", "a.ply"), ("two", "b.ply")]
    assert np.allclose(m[0][2], T, atol=1e-5) and np.allclose(m[1][2], np.eye(4))
    # text contract: four rows, every row ends with a space before the newline (MeshLab needs it)
    rows = m[0][3].strip("\n").split("\n")
    assert len(rows) == 4 and all(row.endswith(" ") for row in rows) and rows[3] == "0 0 0 1 "
    assert rows[0].split()[3] == "1.5" and rows[1].split()[3] == "-2.25" and rows[2].split()[3] == "0.333333"


def test_image_registrator_usage(tmp_path):
    """Argument validation happens before the HIP library is loaded (no GPU needed): same message and exit code as the
    reference (src/exe/image_registrator.cc:118-128)."""
    import subprocess
    from cli_util import BIN
    r = subprocess.run([os.path.join(BIN, "ImageRegistrator"), "--state_path", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "Please specify all the required paths." in r.stderr
    r = subprocess.run([os.path.join(BIN, "ImageRegistrator"), "--robust_weighting_type", "bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "--robust_weighting_type parameter not recognized" in r.stderr


def test_point_cloud_cleaner_usage(tmp_path):
    _build()
    r = subprocess.run([os.path.join(BIN, "PointCloudCleaner")], capture_output=True, text=True)
    assert r.returncode != 0 and "--in <file.ply> --filter <knn,factor>" in r.stderr
    write_ply_xyz(str(tmp_path / "a.ply"), np.zeros((3, 3), np.float32))
    r = subprocess.run([os.path.join(BIN, "PointCloudCleaner"), "--in", str(tmp_path / "a.ply")], capture_output=True, text=True)
    assert r.returncode != 0 and "One or more --filter knn,factor parameter values must be given." in r.stderr
    r = subprocess.run([os.path.join(BIN, "PointCloudCleaner"), "--in", str(tmp_path / "a.ply"), "--filter", "8"], capture_output=True, text=True)
    assert r.returncode != 0 and "different than 2" in r.stderr


def test_ground_truth_creator_usage(tmp_path):
    _build()
    r = subprocess.run([os.path.join(BIN, "GroundTruthCreator")], capture_output=True, text=True)
    assert r.returncode != 0 and "Please specify all the required paths." in r.stderr
    r = subprocess.run([os.path.join(BIN, "GroundTruthCreator"), "--scan_alignment_path", "a", "--image_base_path", "b", "--state_path", "c",
                        "--output_folder_path", str(tmp_path / "o"), "--write_scan_renderings", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "--write_scan_renderings is not part of this build." in r.stderr


def _imread(path, tmp_path):
    out = str(tmp_path / "o.pgm")
    r = subprocess.run([os.path.join(BIN, "e3d_imread_gray"), path, out], capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr
    raw = open(out, "rb").read()
    parts = raw.split(b"\n", 3)
    w, h = [int(v) for v in parts[1].split()]
    return np.frombuffer(parts[3], np.uint8, w * h).reshape(h, w), ""


def test_jpeg_decoder_matches_libjpeg_golden(tmp_path):
    """cv::imread(path, IMREAD_GRAYSCALE) on JPEG = libjpeg's luminance plane (islow IDCT).  Golden: Pillow / libjpeg-turbo
    (tests/golden/make_jpeg_golden.py); 4:2:0 / 4:2:2 / 4:4:4, optimised Huffman tables, grey files, restart markers, sizes
    that are not MCU multiples -- bit for bit; the eight EXIF orientations, which cv::imread applies (expected: Pillow's
    exif_transpose of the luminance plane)."""
    _build()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.load(os.path.join(gold, "jpeg_golden.npz"))
    assert len(exp.files) == 14 and exp["exif_6"].shape == (37, 29)
    for name in exp.files:
        img, err = _imread(os.path.join(gold, "jpeg_%s.jpg" % name), tmp_path)
        assert img is not None, err
        assert img.shape == exp[name].shape and np.array_equal(img, exp[name]), name
    img, err = _imread(os.path.join(gold, "jpeg_progressive.jpg"), tmp_path)
    assert img is None and "progressive JPEG is not supported" in err


def test_png_decoder_roundtrip(tmp_path):
    """The PNG path of imread_gray against Pillow-written files: 8-bit grey (all filter types via a noisy image), RGB -> grey with
    the libpng weights OpenCV uses, 16-bit grey (high byte)."""
    from PIL import Image
    _build()
    rng = np.random.RandomState(0)
    g = rng.randint(0, 256, (37, 53)).astype(np.uint8)
    Image.fromarray(g, "L").save(str(tmp_path / "g.png"))
    img, err = _imread(str(tmp_path / "g.png"), tmp_path)
    assert img is not None and np.array_equal(img, g), err
    rgb = rng.randint(0, 256, (20, 31, 3)).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(str(tmp_path / "c.png"))
    img, _ = _imread(str(tmp_path / "c.png"), tmp_path)
    r64 = rgb.astype(np.int64)
    exp = ((9797 * r64[..., 0] + 19234 * r64[..., 1] + 3737 * r64[..., 2] + 16384) >> 15).astype(np.uint8)
    assert np.array_equal(img, exp)
    g16 = rng.randint(0, 65536, (9, 14)).astype(np.uint16)
    Image.fromarray(g16).save(str(tmp_path / "h.png"))
    img, _ = _imread(str(tmp_path / "h.png"), tmp_path)
    assert np.array_equal(img, (g16 >> 8).astype(np.uint8))


def test_arrow_system_matches_dense_ldlt(tmp_path):
    """e3d::ArrowSystem (block-sparse normal equations of IntrinsicsAndPoseOptimizer, Schur-complement solve) against the dense
    pivoted LDLT the reference uses, on random systems with the same sparsity pattern; also the packed exchange layout and that
    an entry coupling two different poses is refused."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "arrow_system_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(root, "tests", "cpp", "arrow_system_test.cc")])
    for args in (("12", "40", "1", "-1"), ("20", "100", "2", "7"), ("0", "5", "3", "-1"), ("30", "3", "4", "0")):
        r = subprocess.run([exe, *args], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        err, mx, pack_ok = r.stdout.split()
        assert float(err) <= 1e-12 * max(1.0, float(mx)) and pack_ok == "1", (args, r.stdout)
