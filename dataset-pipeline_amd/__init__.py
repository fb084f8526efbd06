"""dataset-pipeline_amd -- MI355X (gfx950) implementation of the ETH3D dataset-pipeline scan-alignment
hot path (ICPScanAligner / NormalEstimator), behind the reference's own class surface.

The compute lives in lib/libe3dhip.so (hand-written HIP kernels + C-ABI, include/e3d_hip.h).  This
Python package is only the host-side mirror used by tests and bench.py; it contains no CPU fallback:
every entry point raises if the HIP library or a GPU is missing.

The directory name contains a hyphen (it is the name the task prescribes), so import it with
    import importlib; e3d = importlib.import_module("dataset-pipeline_amd")
"""
from .capi import (Comm, E3DError, PointToPlaneICP, RegParams, RegProblem, default_reg_params, determine_point_neighbors,
                   find_correspondences, icp_pair_system, lib, lib_path, libm_eval, local_outlier_removal, release_workspaces, merge_close_points, normals_knn, normals_radius, transform_cloud)

__all__ = ["Comm", "E3DError", "PointToPlaneICP", "RegParams", "RegProblem", "default_reg_params", "determine_point_neighbors", "find_correspondences",
           "icp_pair_system", "lib", "lib_path", "libm_eval", "local_outlier_removal", "release_workspaces", "merge_close_points", "normals_knn", "normals_radius", "transform_cloud"]
