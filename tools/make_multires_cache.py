"""Writes a multi-resolution point cloud cache directory in the reference's format (Problem::SaveMultiResPointCloud,
src/opt/problem.cc:364-411): metadata.txt, points_of_scale_<s>.ply (x y z intensity, f32), neighbor_point_indices (u64).
ImageRegistrator loads it (Problem::LoadMultiResPointCloud); computing it from raw scans (ComputeMultiResPointCloud) is
SURVEY f1 and not built yet, so datasets that do not come with a cache need this helper or the reference's tool.

    from tools.make_multires_cache import write_cache
    write_cache(dir, [(radius, xyz, intensity, neighbors), ...], neighbor_count, candidate_count)
"""
import os

import numpy as np


def write_cache(directory, scales, neighbor_count=5, candidate_count=25):
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "metadata.txt"), "w") as f:
        f.write("version 1\nneighbor_candidate_count %d\nneighbor_count %d\npoint_scale_count %d\n" % (candidate_count, neighbor_count, len(scales)))
        for radius, _, _, _ in scales:
            f.write("point_radius %.9g\n" % radius)
    with open(os.path.join(directory, "neighbor_point_indices"), "wb") as nf:
        for s, (_, xyz, intensity, nbr) in enumerate(scales):
            xyz = np.asarray(xyz, np.float32); intensity = np.asarray(intensity, np.float32)
            rec = np.zeros(len(xyz), dtype=[("p", "<f4", 3), ("i", "<f4")])
            rec["p"] = xyz; rec["i"] = intensity
            with open(os.path.join(directory, "points_of_scale_%d.ply" % s), "wb") as f:
                f.write(("ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex %d\nproperty float x\nproperty float y\n"
                         "property float z\nproperty float intensity\nend_header\n" % len(xyz)).encode())
                f.write(rec.tobytes())
            nf.write(np.ascontiguousarray(nbr, np.uint64).tobytes())
