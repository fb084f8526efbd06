"""Host-side (no GPU) behaviour of the drop-in tools: flag handling, .mlp / PLY I/O, early-out paths."""
import os
import subprocess
import sys

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
    r = subprocess.run([os.path.join(BIN, "GroundTruthCreator"), "--scan_alignment_path", str(tmp_path / "none.mlp"), "--image_base_path", "b",
                        "--state_path", "c", "--output_folder_path", str(tmp_path / "o"), "--write_scan_renderings", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot read scan poses from" in r.stderr


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
    that are not MCU multiples, progressive files (spectral selection + successive approximation, colour and grey, with restart
    markers) -- bit for bit; the eight EXIF orientations, which cv::imread applies (expected: Pillow's exif_transpose of the
    luminance plane)."""
    _build()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.load(os.path.join(gold, "jpeg_golden.npz"))
    assert len(exp.files) == 19 and exp["exif_6"].shape == (37, 29) and sum(n.startswith("prog") for n in exp.files) == 5
    for name in exp.files:
        img, err = _imread(os.path.join(gold, "jpeg_%s.jpg" % name), tmp_path)
        assert img is not None, err
        assert img.shape == exp[name].shape and np.array_equal(img, exp[name]), name
    # damaged scan headers are refused with a message: a Huffman table selector beyond the four tables (found by tools/fuzz/image_readers.cc),
    # a second frame header
    raw = bytearray(open(os.path.join(gold, "jpeg_prog_420_q40_opt.jpg"), "rb").read())
    sos = raw.index(b"\xff\xda")
    raw[sos + 6] = 0xEE
    open(str(tmp_path / "sel.jpg"), "wb").write(bytes(raw))
    img, err = _imread(str(tmp_path / "sel.jpg"), tmp_path)
    assert img is None and "bad SOS" in err
    raw = open(os.path.join(gold, "jpeg_420_q90.jpg"), "rb").read()
    sof = raw.index(b"\xff\xc0")
    seg = raw[sof: sof + 2 + int.from_bytes(raw[sof + 2: sof + 4], "big")]
    open(str(tmp_path / "two.jpg"), "wb").write(raw[:sof] + seg + raw[sof:])
    img, err = _imread(str(tmp_path / "two.jpg"), tmp_path)
    assert img is None and "more than one frame header" in err
    # a file that ends exactly at an SOS segment of length 2 (no payload byte to read the component count from)
    for name in ("jpeg_prog_420_q40_opt.jpg", "jpeg_420_q90.jpg"):
        raw = open(os.path.join(gold, name), "rb").read()
        sos = raw.index(b"\xff\xda")
        open(str(tmp_path / "sos2.jpg"), "wb").write(raw[:sos] + b"\xff\xda\x00\x02")
        img, err = _imread(str(tmp_path / "sos2.jpg"), tmp_path)
        assert img is None and "bad SOS" in err, err
    # a progressive file cut in the middle of a scan is refused, not decoded to garbage
    raw = open(os.path.join(gold, "jpeg_prog_420_q40_opt.jpg"), "rb").read()
    open(str(tmp_path / "cut.jpg"), "wb").write(raw[: len(raw) // 3])
    img, err = _imread(str(tmp_path / "cut.jpg"), tmp_path)
    assert img is None or img.shape == exp["prog_420_q40_opt"].shape


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


def test_ldlt_lower_triangle_form_is_bit_identical(tmp_path):
    """e3d::ldlt_solve_upper keeps the lower triangle only (round 6); the plain full-matrix form of the same pivoted LDL^T (restated in
    the test program) gives the same solution and pivot order bit for bit: positive definite, indefinite, rank-deficient systems and
    systems with zero rows (unknowns without residuals), at the sizes of the ICP (6 ... 90) and registration (150, 384) systems."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ldlt_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(root, "tests", "cpp", "ldlt_test.cc")])
    for kind in range(4):
        for n, seed in ((1, 1), (6, 2), (42, 3), (90, 4), (150, 5), (384, 6)):
            r = subprocess.run([exe, str(n), str(seed), str(kind)], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            assert r.stdout.split()[0] == "0", (n, kind, r.stdout)


def test_ply_readers_mixed_types_and_endianness(tmp_path):
    """loadPLYFile / loadPLYMesh (csrc/host/io_ply.h) on binary files as scanners and MeshLab write them: double coordinates,
    properties in any order, colours, normals, intensity, extra properties, big-endian files, face lists with different count /
    index types and a per-face scalar, a non-triangle face (rejected)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ply_reader_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "ply_reader_test.cc")])
    rng = np.random.RandomState(3)
    n = 257
    for endian, tag in (("<", "binary_little_endian"), (">", "binary_big_endian")):
        v = np.zeros(n, dtype=[("nz", endian + "f4"), ("x", endian + "f8"), ("red", "u1"), ("y", endian + "f8"), ("conf", endian + "i2"), ("z", endian + "f8"),
                               ("green", "u1"), ("blue", "u1"), ("nx", endian + "f4"), ("ny", endian + "f4"), ("intensity", endian + "f4")])
        for k in ("x", "y", "z", "nx", "ny", "nz", "intensity"):
            v[k] = rng.uniform(-5, 5, n)
        for k in ("red", "green", "blue"):
            v[k] = rng.randint(0, 256, n)
        v["conf"] = rng.randint(-300, 300, n)
        names = {"f4": "float", "f8": "double", "u1": "uchar", "i2": "short"}
        hdr = "ply\nformat %s 1.0\ncomment test\nelement vertex %d\n" % (tag, n)
        for name in v.dtype.names:
            hdr += "property %s %s\n" % (names[v.dtype[name].str[-2:]], name)
        hdr += "end_header\n"
        path = str(tmp_path / ("cloud_%s.ply" % tag))
        open(path, "wb").write(hdr.encode() + v.tobytes())
        r = subprocess.run([exe, "cloud", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().split("\n")
        assert lines[0] == "%d 1 1 1" % n
        got = np.array([[float(t) for t in l.split()] for l in lines[1:]])
        exp = np.stack([v["x"].astype(np.float32), v["y"].astype(np.float32), v["z"].astype(np.float32), v["red"], v["green"], v["blue"],
                        v["nx"], v["ny"], v["nz"], v["intensity"]], 1).astype(np.float64)
        assert np.array_equal(got.astype(np.float32), exp.astype(np.float32)), tag
        # mesh: float vertices + colour, faces = (uchar count, int indices, float quality) or (int count, uint indices)
        mv = np.zeros(50, dtype=[("x", endian + "f4"), ("y", endian + "f4"), ("z", endian + "f4"), ("red", "u1")])
        for k in ("x", "y", "z"):
            mv[k] = rng.uniform(-1, 1, 50)
        for cnt_t, idx_t, cn, in_ in (("u1", "i4", "uchar", "int"), ("i4", "u4", "int", "uint")):
            f = np.zeros(31, dtype=[("k", endian.replace("<", "<") + cnt_t if cnt_t != "u1" else "u1"), ("i", endian + idx_t, 3), ("q", endian + "f4")])
            f["k"] = 3; f["i"] = rng.randint(0, 50, (31, 3)); f["q"] = 0.5
            hdr = ("ply\nformat %s 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                   "element face 31\nproperty list %s %s vertex_indices\nproperty float quality\nend_header\n" % (tag, cn, in_))
            path = str(tmp_path / ("mesh_%s_%s.ply" % (tag, cn)))
            open(path, "wb").write(hdr.encode() + mv.tobytes() + f.tobytes())
            r = subprocess.run([exe, "mesh", path], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            lines = r.stdout.strip().split("\n")
            assert lines[0] == "50 31"
            gv = np.array([[float(t) for t in l.split()] for l in lines[1:51]], np.float32)
            gt = np.array([[int(t) for t in l.split()] for l in lines[51:]])
            assert np.array_equal(gv, np.stack([mv["x"], mv["y"], mv["z"]], 1).astype(np.float32)) and np.array_equal(gt, f["i"].astype(np.int64))
    # a quad is refused
    f = np.zeros(1, dtype=[("k", "u1"), ("i", "<i4", 4)]); f["k"] = 4
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
    path = str(tmp_path / "quad.ply")
    open(path, "wb").write(hdr.encode() + mv.astype(mv.dtype.newbyteorder("<")).tobytes() + f.tobytes())
    r = subprocess.run([exe, "mesh", path], capture_output=True, text=True)
    assert r.returncode != 0 and "only triangle meshes are supported" in r.stderr


def test_ply_readers_refuse_damaged_files(tmp_path):
    """Counts the file cannot hold (a negative vertex count, a list count beyond the end of the file, a negative list count) and
    truncated bodies return -1 with a message instead of a multi-GB resize or a walk over garbage."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ply_reader_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "ply_reader_test.cc")])
    v = np.zeros(10, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    vh = "property float x\nproperty float y\nproperty float z\n"

    def run(kind, name, blob):
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        return subprocess.run([exe, kind, path], capture_output=True, text=True, timeout=60)
    # element vertex -1 (wraps to 2^64 - 1 as size_t)
    for kind in ("cloud", "mesh"):
        r = run(kind, "neg.ply", ("ply\nformat binary_little_endian 1.0\nelement vertex -1\n" + vh + "end_header\n").encode() + v.tobytes())
        assert r.returncode == 1 and "beyond the end" in r.stderr, r.stderr
    # vertex body shorter than the header says
    r = run("cloud", "short.ply", ("ply\nformat binary_little_endian 1.0\nelement vertex 10\n" + vh + "end_header\n").encode() + v.tobytes()[:50])
    assert r.returncode == 1 and "truncated" in r.stderr
    fh = "ply\nformat binary_little_endian 1.0\nelement vertex 10\n" + vh + "element face 2\nproperty list %s int vertex_indices\nend_header\n"
    tri = np.array([0, 1, 2], "<i4").tobytes()
    # uint32 list count of 4 G entries in a 200-byte file; negative int32 count; file ending inside a list
    for cnt_type, cnt, what in (("uint", np.array([0xFFFFFFF0], "<u4").tobytes(), "bad list count"),
                                ("int", np.array([-3], "<i4").tobytes(), "bad list count"),
                                ("uchar", b"\x03", None)):
        body = v.tobytes() + (b"\x03" if cnt_type == "uchar" else np.array([3], "<u4").tobytes()) + tri + cnt + (tri[:5] if what is None else tri)
        r = run("mesh", "list_%s.ply" % cnt_type, (fh % cnt_type).encode() + body)
        assert r.returncode == 1 and ("truncated" in r.stderr), r.stderr
        if what:
            assert what in r.stderr
    # faces BEFORE the vertices: loadPLYFile walks the list element record by record
    hdr = "ply\nformat binary_little_endian 1.0\nelement face 2\nproperty list uchar int vertex_indices\nelement vertex 10\n" + vh + "end_header\n"
    good = run("cloud", "faces_first.ply", hdr.encode() + (b"\x03" + tri) * 2 + v.tobytes())
    assert good.returncode == 0 and good.stdout.split("\n")[0].startswith("10 ")
    bad = run("cloud", "faces_first_bad.ply", hdr.encode() + b"\x03" + tri + b"\xff" + tri)
    assert bad.returncode == 1 and "truncated" in bad.stderr
    # ascii: negative list count
    r = run("mesh", "ascii_neg.ply", ("ply\nformat ascii 1.0\nelement vertex 3\n" + vh + "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
                                      "0 0 0\n1 0 0\n0 1 0\n-3 0 1 2\n").encode())
    assert r.returncode == 1 and "bad list count" in r.stderr


def test_affine_rotation_of_a_reflection(tmp_path):
    """Affine3f::rotation() (Eigen's Transform::rotation()): for a linear part with negative determinant the sign flip goes to the
    direction of the SMALLEST singular value, wherever the unsorted one-sided Jacobi leaves it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ply_reader_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "ply_reader_test.cc")])
    rng = np.random.RandomState(4)
    for trial in range(40):
        U, _ = np.linalg.qr(rng.normal(size=(3, 3))); V, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        sig = rng.permutation([1.0, 0.6 + 0.3 * rng.rand(), 0.1 + 0.3 * rng.rand()])
        A = ((U * sig) @ V.T).astype(np.float32)
        if trial % 2 == 0:
            A = -A                                                  # det < 0 for every other trial
        r = subprocess.run([exe, "rotation"] + ["%.9g" % x for x in A.ravel()], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        R = np.array([float(t) for t in r.stdout.split()]).reshape(3, 3)
        Us, s, Vt = np.linalg.svd(A.astype(np.float64))
        D = np.diag([1, 1, np.sign(np.linalg.det(Us @ Vt))])
        assert np.abs(R - Us @ D @ Vt).max() < 1e-9, (trial, R, Us @ D @ Vt)
        assert abs(np.linalg.det(R) - 1) < 1e-9


def test_png_decoder_refuses_damaged_headers(tmp_path):
    """A short IHDR chunk at the end of the file and dimensions the compressed data cannot fill are refused with a message (no read
    past the buffer, no multi-GB allocation) -- ADVICE round 1."""
    import struct
    import zlib
    _build()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    sig = bytes([137, 80, 78, 71, 13, 10, 26, 10])
    open(str(tmp_path / "short_ihdr.png"), "wb").write(sig + chunk(b"IHDR", b"\x00\x00\x00\x10"))
    img, err = _imread(str(tmp_path / "short_ihdr.png"), tmp_path)
    assert img is None and "IHDR" in err
    ihdr = struct.pack(">IIBBBBB", 60000, 60000, 8, 0, 0, 0, 0)            # 3.6 GB of pixels promised by a 100-byte file
    open(str(tmp_path / "huge.png"), "wb").write(sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\x00" * 16)) + chunk(b"IEND", b""))
    img, err = _imread(str(tmp_path / "huge.png"), tmp_path)
    assert img is None and "too short" in err
    ihdr = struct.pack(">IIBBBBB", 1 << 24, 4, 8, 0, 0, 0, 0)
    open(str(tmp_path / "wide.png"), "wb").write(sig + chunk(b"IHDR", ihdr) + chunk(b"IEND", b""))
    img, err = _imread(str(tmp_path / "wide.png"), tmp_path)
    assert img is None and "out of range" in err


def _imread_color_to(path, out):
    r = subprocess.run([os.path.join(BIN, "e3d_imread_gray"), "--color", path, out], capture_output=True, text=True)
    return r.returncode, r.stderr


def test_color_decoder_matches_libjpeg(tmp_path):
    """cv::imread(path) on JPEG = libjpeg's RGB output: chroma planes, triangle-filter ("fancy") upsampling for 4:2:2 / 4:2:0, the
    fixed-point YCbCr -> RGB tables.  Baseline and progressive files, odd sizes, a grey file; expected = Pillow / libjpeg-turbo, bit
    for bit.  PNG (RGB, RGBA, palette, grey) and PPM through the same entry point."""
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_jpeg_golden import scene
    _build()
    cases = [("420", (67, 45), dict(quality=90, subsampling=2)), ("422", (67, 45), dict(quality=80, subsampling=1)),
             ("444", (40, 52), dict(quality=95, subsampling=0)), ("p420", (131, 99), dict(quality=75, subsampling=2, progressive=True)),
             ("p444", (61, 47), dict(quality=92, subsampling=0, progressive=True)), ("p422r", (90, 70), dict(quality=85, subsampling=1, progressive=True,
                                                                                                               restart_marker_blocks=2)),
             ("odd", (33, 17), dict(quality=88, subsampling=2)), ("tiny", (3, 2), dict(quality=90, subsampling=2)),
             ("r420", (75, 60), dict(quality=80, subsampling=2, restart_marker_blocks=3))]
    for i, (name, (w, h), kw) in enumerate(cases):
        src = str(tmp_path / (name + ".jpg"))
        scene(w, h, 40 + i).save(src, "JPEG", **kw)
        rc, err = _imread_color_to(src, str(tmp_path / "o.ppm"))
        assert rc == 0, err
        got = np.array(Image.open(str(tmp_path / "o.ppm")))
        exp = np.array(Image.open(src).convert("RGB"))
        assert got.shape == exp.shape and np.array_equal(got, exp), name
    scene(50, 40, 3).convert("L").save(str(tmp_path / "g.jpg"), "JPEG", quality=85)
    assert _imread_color_to(str(tmp_path / "g.jpg"), str(tmp_path / "o.ppm"))[0] == 0
    assert np.array_equal(np.array(Image.open(str(tmp_path / "o.ppm"))), np.array(Image.open(str(tmp_path / "g.jpg")).convert("RGB")))
    rgb = scene(37, 29, 5)
    for mode, fname in (("RGB", "c.png"), ("RGBA", "a.png"), ("P", "p.png"), ("L", "l.png"), ("RGB", "c.ppm")):
        im = rgb.convert(mode) if mode != "P" else rgb.quantize(32)
        im.save(str(tmp_path / fname))
        assert _imread_color_to(str(tmp_path / fname), str(tmp_path / "o.ppm"))[0] == 0
        assert np.array_equal(np.array(Image.open(str(tmp_path / "o.ppm"))), np.array(im.convert("RGB"))), fname


def test_jpeg_encoder_writes_libjpeg_bytes(tmp_path):
    """cv::imwrite(path.jpg, image) = libjpeg at quality 95, 4:2:0, standard tables: the files of io_jpeg_write.h equal the ones
    Pillow / libjpeg-turbo writes with those settings byte for byte -- colour conversion, edge padding (right edge before, bottom edge
    after downsampling), forward DCT, quantisation, dummy blocks, Huffman coding, headers.  PNG output round-trips."""
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_jpeg_golden import scene
    _build()
    rng = np.random.RandomState(1)
    sizes = [(129, 70), (70, 129), (64, 48), (17, 2), (1, 1), (2, 2), (15, 31), (250, 198), (16, 16)] + [tuple(int(v) for v in rng.randint(1, 90, 2)) for _ in range(10)]
    for (w, h) in sizes:
        img = scene(w, h, 3 * w + h)
        img.save(str(tmp_path / "src.png"))
        rc, err = _imread_color_to(str(tmp_path / "src.png"), str(tmp_path / "ours.jpg"))
        assert rc == 0, err
        img.save(str(tmp_path / "pil.jpg"), "JPEG", quality=95, subsampling=2)
        assert open(str(tmp_path / "ours.jpg"), "rb").read() == open(str(tmp_path / "pil.jpg"), "rb").read(), (w, h)
    assert _imread_color_to(str(tmp_path / "src.png"), str(tmp_path / "o.png"))[0] == 0
    assert np.array_equal(np.array(Image.open(str(tmp_path / "o.png"))), np.array(Image.open(str(tmp_path / "src.png")).convert("RGB")))
    rc, err = _imread_color_to(str(tmp_path / "src.png"), str(tmp_path / "o.bmp"))
    assert rc != 0 and "unsupported file extension" in err
