"""CPU oracle of the multi-resolution point cloud construction (TEST INFRASTRUCTURE ONLY).

Orchestrates oracle_multires.c / oracle_shuffle.cc as the reference's host code does:
  PreprocessScans, CreateMultiScalePointCloud          src/opt/multi_scale_point_cloud.cc:182-369
  Problem::ComputeMultiResPointCloud                   src/opt/problem.cc:160-362
Inputs are plain arrays (scans already in the global frame); images are described like in oracle/reg_driver.py.
"""
import ctypes as C

import numpy as np

from . import binding as ob
from . import reg_binding as rb


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_READY = False


def lib():
    global _READY
    L = rb.lib()
    if not _READY:
        fp, u8p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        cp = C.POINTER(rb.Camera)
        L.oracle_undistortion_lookup.argtypes = [cp, fp]
        L.oracle_image_to_normalized.argtypes = [cp, fp, C.c_float, C.c_float, fp]
        L.oracle_point_radius_minmax.argtypes = [fp, C.c_size_t, fp, fp, cp, C.c_int, C.c_int, cp, fp, u8p, u8p, fp, C.c_float, C.c_float,
                                                 C.c_double, fp, fp]
        L.oracle_merge_close_points.argtypes = [C.c_float, C.c_int, fp, fp, u8p, fp, C.c_size_t, fp, fp, u8p, fp]
        L.oracle_merge_close_points.restype = C.c_size_t
        L.oracle_determine_point_neighbors.argtypes = [fp, C.c_size_t, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u32p]
        _READY = True
    return L


def undistortion_lookup(cam):
    out = np.zeros((cam.height, cam.width, 2), np.float32)
    lib().oracle_undistortion_lookup(C.byref(cam), _p(out, C.c_float))
    return out


def image_to_normalized(cam, lookup, x, y):
    o = np.zeros(2, np.float32)
    lib().oracle_image_to_normalized(C.byref(cam), _p(np.ascontiguousarray(lookup, np.float32), C.c_float), x, y, _p(o, C.c_float))
    return o


def merge_close_points(merge_distance, num_scans, pts, colors, scan_idx, max_radius):
    pts = np.ascontiguousarray(pts, np.float32); colors = np.ascontiguousarray(colors, np.float32)
    scan_idx = np.ascontiguousarray(scan_idx, np.uint8); max_radius = np.ascontiguousarray(max_radius, np.float32)
    n = len(pts)
    op = np.zeros((n + 1, 3), np.float32); oc = np.zeros(n + 1, np.float32); osc = np.zeros(n + 1, np.uint8); om = np.zeros(n + 1, np.float32)
    m = lib().oracle_merge_close_points(np.float32(merge_distance), num_scans, _p(pts, C.c_float), _p(colors, C.c_float), _p(scan_idx, C.c_uint8),
                                        _p(max_radius, C.c_float), n, _p(op, C.c_float), _p(oc, C.c_float), _p(osc, C.c_uint8), _p(om, C.c_float))
    return op[:m].copy(), oc[:m].copy(), osc[:m].copy(), om[:m].copy()


def determine_point_neighbors(pts, neighbor_count, candidate_count, scan_idx=None, scan_count=1):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros((len(pts), neighbor_count), np.uint32)
    si = np.ascontiguousarray(scan_idx, np.uint8) if scan_idx is not None else None
    r = lib().oracle_determine_point_neighbors(_p(pts, C.c_float), len(pts), _p(si, C.c_uint8) if si is not None else None, scan_count,
                                               1 if si is not None else 0, neighbor_count, candidate_count, _p(out, C.c_uint32))
    if r != 0:
        raise ValueError("a cloud / scan has fewer points than candidates + 1")
    return out


def preprocess_scans(scans):
    """scans: list of (xyz f32 n x 3, rgb u8 n x 3) -> points, colours (0.299 r + 0.587 g + 0.114 b, double -> float), scan indices"""
    pts = np.concatenate([np.asarray(s[0], np.float32) for s in scans])
    col = np.concatenate([(0.299 * s[1][:, 0].astype(np.float64) + 0.587 * s[1][:, 1].astype(np.float64)
                           + 0.114 * s[1][:, 2].astype(np.float64)).astype(np.float32) for s in scans])
    idx = np.concatenate([np.full(len(s[0]), i, np.uint8) for i, s in enumerate(scans)])
    return pts, col, idx


def point_radius_minmax(pts, images, intrinsics, splat_points, image_scale_count, occlusion_threshold=0.01, max_valid_intensity=252.0,
                        splat_radius=0.03):
    """min / max radius of every point over all images (CreateMultiScalePointCloud :236-262).  images: {id: dict(intr, pyr,
    masks, q, t)}, intrinsics: {id: dict(w, h, params, min, n, model)} as in OracleRegProblem."""
    n = len(pts)
    mn = np.full(n, np.inf, np.float32); mx = np.full(n, -np.inf, np.float32)
    min_scaling = 2.0 ** (-1 * (image_scale_count - 1))
    for iid in sorted(images):
        im = images[iid]; I = intrinsics[im["intr"]]
        levels = rb.camera_pyramid(rb.make_camera(I["w"], I["h"], I["params"], I.get("model", 0)), I["n"])
        scale = I["min"]                                   # best available scale of max(min_occlusion_check_image_scale, 0)
        cam = levels[0]
        lookup = undistortion_lookup(cam)
        R = ob.quat_to_R(im["q"])
        depth = rb.splat_depth(splat_points, R, im["t"], cam, splat_radius)
        img = np.ascontiguousarray(im["pyr"][0], np.uint8)
        mask = np.ascontiguousarray(im["masks"][0], np.uint8) if im.get("masks") is not None else None
        lib().oracle_point_radius_minmax(_p(np.ascontiguousarray(pts, np.float32), C.c_float), n, _p(np.ascontiguousarray(im["q"], np.float32), C.c_float),
                                         _p(np.ascontiguousarray(im["t"], np.float32), C.c_float), C.byref(cam), scale, I["min"], C.byref(cam),
                                         _p(lookup, C.c_float), _p(img, C.c_uint8), _p(mask, C.c_uint8) if mask is not None else None,
                                         _p(np.ascontiguousarray(depth, np.float32), C.c_float), occlusion_threshold, max_valid_intensity,
                                         min_scaling, _p(mn, C.c_float), _p(mx, C.c_float))
    return mn, mx


def create_multi_scale_point_cloud(pts, colors, scan_idx, num_scans, min_radius, max_radius, min_radius_bias=1.05, merge_distance_factor=4.0):
    """CreateMultiScalePointCloud :264-369 -> list of (radius, points, colours, scan indices)."""
    min_radius_value = np.float32(min_radius.min()); max_radius_value = np.float32(max_radius.max())
    radius = float(np.float32(min_radius_value * np.float32(min_radius_bias)))          # float product, then double
    sel = radius >= min_radius.astype(np.float64)
    last = (pts[sel], colors[sel], scan_idx[sel], max_radius[sel])
    last_radius = -1.0
    out = []
    while True:
        if last_radius > 0:
            keep = radius <= last[3].astype(np.float64)
            add = (np.float32(last_radius) < min_radius) & (radius >= min_radius.astype(np.float64))
            last = tuple(np.concatenate([a[keep], b[add]]) for a, b in zip(last, (pts, colors, scan_idx, max_radius)))
        merged = merge_close_points(np.float32(np.float64(np.float32(merge_distance_factor)) * radius), num_scans, *last)
        out.append((np.float32(radius), merged[0], merged[1], merged[2]))
        last_radius = float(np.float32(radius))
        radius *= 2
        if radius >= float(max_radius_value * np.float32(0.99)):
            break
        last = merged
    return out


def compute_multi_res_point_cloud(scans, images, intrinsics, image_scale_count, K=5, candidates=25, min_mean_intensity_difference=5.0,
                                  use_fixed_scan_colors=True, **kw):
    """Problem::ComputeMultiResPointCloud -> list of dict(radius, pts, colors, nbr)."""
    pts, col, sidx = preprocess_scans(scans)
    num_scans = len(scans)
    mn, mx = point_radius_minmax(pts, images, intrinsics, pts, image_scale_count, **kw)
    scales = create_multi_scale_point_cloud(pts, col, sidx, num_scans, mn, mx)

    def enough(s):
        if use_fixed_scan_colors:
            return all(int((s[3] == k).sum()) >= candidates + 1 for k in range(num_scans))
        return len(s[1]) >= candidates + 1
    scales = [s for s in scales if enough(s)]
    out = []
    for radius, p, c, si in scales:
        nbr = determine_point_neighbors(p, K, candidates, si if use_fixed_scan_colors else None, num_scans)
        diff = np.zeros(len(p), np.float32)
        for k in range(K):                                               # float sum in neighbour order
            diff = diff + np.abs(c[nbr[:, k]] - c)
        delete1 = (diff / np.float32(K)) < np.float32(min_mean_intensity_difference)
        keep = ~delete1
        keep2 = keep.copy()
        keep2[nbr[keep].ravel()] = True                                  # neighbours of kept points stay as well
        out.append((radius, p[keep2], c[keep2], si[keep2]))
    out = [s for s in out if enough(s)]
    res = []
    for radius, p, c, si in out:
        nbr = determine_point_neighbors(p, K, candidates, si if use_fixed_scan_colors else None, num_scans)
        res.append(dict(radius=radius, pts=p, colors=c, scan=si, nbr=nbr))
    return res
