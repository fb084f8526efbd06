"""Helpers for the CLI tests: write PLY / .mlp inputs the way the ETH3D pipeline stores them."""
import os
import re
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dataset-pipeline_amd", "bin")


def write_ply_xyz(path, xyz, rgb=None, binary=True):
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    with open(path, "wb") as f:
        hdr = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % (
            "binary_little_endian" if binary else "ascii", n)
        if rgb is not None:
            hdr += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
        hdr += "element face 0\nproperty list uchar int vertex_indices\nend_header\n"
        f.write(hdr.encode())
        if binary:
            if rgb is None:
                f.write(xyz.tobytes())
            else:
                rec = np.zeros(n, dtype=[("p", "<f4", 3), ("c", "u1", 3)])
                rec["p"] = xyz; rec["c"] = rgb
                f.write(rec.tobytes())
        else:
            for i in range(n):
                line = "%.9g %.9g %.9g" % tuple(xyz[i])
                if rgb is not None:
                    line += " %d %d %d" % tuple(rgb[i])
                f.write((line + "\n").encode())


def write_mlp(path, entries):
    """entries: list of (label, filename, 4x4 matrix)."""
    with open(path, "w") as f:
        f.write("<!DOCTYPE MeshLabDocument>\n<MeshLabProject>\n <MeshGroup>\n")
        for label, fn, T in entries:
            f.write('  <MLMesh label="%s" filename="%s">\n   <MLMatrix44>\n' % (label, fn))
            for r in range(4):
                f.write(" ".join("%.9g" % v for v in np.asarray(T)[r]) + " \n")
            f.write("</MLMatrix44>\n  </MLMesh>\n")
        f.write(" </MeshGroup>\n <RasterGroup/>\n</MeshLabProject>\n")


def read_mlp(path):
    txt = open(path).read()
    out = []
    for m in re.finditer(r'<MLMesh label="([^"]*)" filename="([^"]*)">\s*<MLMatrix44>(.*?)</MLMatrix44>', txt, re.S):
        vals = np.array(m.group(3).split(), dtype=np.float64).reshape(4, 4)
        out.append((m.group(1), m.group(2), vals, m.group(3)))
    return out


def read_ply_normals(path):
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode()
    n = int(re.search(r"element vertex (\d+)", header).group(1))
    rec = np.frombuffer(data, dtype=[("p", "<f4", 3), ("n", "<f4", 3), ("c", "u1", 3)], count=n, offset=end)
    tail = len(data) - end - 27 * n
    return header, rec, tail


def write_ply_mesh(path, vertices, triangles, binary=True):
    vertices = np.asarray(vertices, np.float32); triangles = np.asarray(triangles, np.int32)
    with open(path, "wb") as f:
        f.write(("ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face %d\n"
                 "property list uchar int vertex_indices\nend_header\n" % ("binary_little_endian" if binary else "ascii", len(vertices), len(triangles))).encode())
        if binary:
            f.write(vertices.tobytes())
            rec = np.zeros(len(triangles), dtype=[("c", "u1"), ("i", "<i4", 3)])
            rec["c"] = 3; rec["i"] = triangles
            f.write(rec.tobytes())
        else:
            for v in vertices:
                f.write(("%.9g %.9g %.9g\n" % tuple(v)).encode())
            for t in triangles:
                f.write(("3 %d %d %d\n" % tuple(t)).encode())
