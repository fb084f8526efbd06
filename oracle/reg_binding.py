"""ctypes binding of the path-(B) oracle (oracle/oracle_reg.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import binding as _b


class Camera(C.Structure):
    _fields_ = [("type", C.c_int), ("width", C.c_int), ("height", C.c_int), ("n_params", C.c_int), ("p", C.c_float * 12),
                ("cutoff2", C.c_float), ("inner_cutoff2", C.c_float), ("fx_inv", C.c_float), ("fy_inv", C.c_float),
                ("cx_inv", C.c_float), ("cy_inv", C.c_float)]

    def params(self):
        """The model's own parameter vector (GetParameters order); inside the struct every model is [fx fy cx cy q...]."""
        if self.type in (5, 6, 7, 11, 12):   # one focal length: [f cx cy q...]
            return np.array([self.p[0], self.p[2], self.p[3]] + list(self.p[4:self.n_params + 1]), np.float32)
        return np.array(self.p[:self.n_params], np.float32)


class RigLink(C.Structure):
    """A non-reference image of a rig frame: image_T_rig quaternion of its camera + pose of the frame's reference image."""
    _fields_ = [("q_image_T_rig", C.c_float * 4), ("q_rig_T_global", C.c_float * 4), ("t_rig_T_global", C.c_float * 3)]


def rig_link(q_image_T_rig, q_rig_T_global, t_rig_T_global):
    r = RigLink()
    r.q_image_T_rig[:] = [float(v) for v in q_image_T_rig]
    r.q_rig_T_global[:] = [float(v) for v in q_rig_T_global]
    r.t_rig_T_global[:] = [float(v) for v in t_rig_T_global]
    return r


PINHOLE, OPENCV, THIN_PRISM_FISHEYE, OPENCV_FISHEYE, FOV = 0, 1, 2, 3, 4
SIMPLE_PINHOLE, SIMPLE_RADIAL, RADIAL, POLYNOMIAL_3, FISHEYE_POLYNOMIAL_2_TANGENTIAL_2 = 5, 6, 7, 8, 9
FULL_OPENCV, RADIAL_FISHEYE_CLASS, SIMPLE_RADIAL_FISHEYE_CLASS = 10, 11, 12     # classes of src/camera the reference's factory never creates
PARAM_COUNT = {0: 4, 1: 8, 2: 12, 3: 8, 4: 5, 5: 3, 6: 4, 7: 5, 8: 7, 9: 8, 10: 12, 11: 5, 12: 4}


_READY = False


def lib():
    global _READY
    L = _b.lib()
    if not _READY:
        fp, u8p, u32p, dp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
        i32p, i64p, ip = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int)
        u8pp = C.POINTER(u8p)
        cp = C.POINTER(Camera)
        L.oracle_reg_camera_init.argtypes = [cp, C.c_int, C.c_int, C.c_int, fp]
        L.oracle_reg_camera_scaled.argtypes = [cp, C.c_float, cp]
        L.oracle_reg_camera_distort.argtypes = [cp, C.c_float, C.c_float, fp]
        L.oracle_reg_camera_undistort.argtypes = [cp, C.c_float, C.c_float, fp, ip]
        L.oracle_reg_camera_project.argtypes = [cp, fp, fp]
        L.oracle_reg_camera_deriv_by_world.argtypes = [cp, fp, fp]
        L.oracle_reg_camera_deriv_by_intrinsics.argtypes = [cp, fp, fp]
        for suffix, tp in (("u8", u8p), ("f32", fp)):
            getattr(L, "oracle_interp_trilinear_" + suffix).argtypes = [tp, C.c_int, tp, C.c_int, C.c_float, C.c_float, C.c_float, fp]
            getattr(L, "oracle_interp_trilinear_d_" + suffix).argtypes = [tp, C.c_int, tp, C.c_int, C.c_float, C.c_float, C.c_float, fp, fp, fp, fp]
        L.oracle_reg_robust_residual.argtypes = [C.c_int, C.c_float, C.c_float]; L.oracle_reg_robust_residual.restype = C.c_float
        L.oracle_reg_robust_weight.argtypes = [C.c_int, C.c_float, C.c_float]; L.oracle_reg_robust_weight.restype = C.c_float
        L.oracle_reg_splat_depth.argtypes = [fp, C.c_size_t, fp, fp, cp, C.c_float, fp]
        L.oracle_reg_observe.argtypes = [fp, C.c_size_t, C.c_float, u32p, C.c_size_t, fp, fp, cp, C.c_int, C.c_int, u8pp, u8pp, fp,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, u32p, fp, fp, fp]
        L.oracle_reg_observe.restype = C.c_size_t
        L.oracle_reg_neighbors_observed.argtypes = [C.c_size_t, u32p, C.c_size_t, u32p, C.c_int, u8p]
        L.oracle_reg_pass1.argtypes = [fp, C.c_float, cp, C.c_int, u8pp, ip, fp, fp, u32p, fp, fp, fp, C.c_size_t, fp, fp, fp]
        rp = C.POINTER(RigLink)
        L.oracle_reg_pass1_rig.argtypes = [fp, C.c_float, cp, C.c_int, u8pp, ip, fp, fp, u32p, fp, fp, fp, C.c_size_t, rp, fp, fp, fp, fp]
        L.oracle_reg_accumulate_rig.argtypes = [fp, C.c_size_t, C.c_float, u32p, C.c_int, fp, fp, i32p, cp, C.c_int, u8pp, ip, fp, fp,
                                                u32p, fp, fp, fp, u8p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_float, rp, dp, dp, dp, i64p]
        L.oracle_se3_mul.argtypes = [fp, fp, fp, fp, fp, fp]
        L.oracle_reg_accumulate.argtypes = [fp, C.c_size_t, C.c_float, u32p, C.c_int, fp, fp, i32p, cp, C.c_int, u8pp, ip, fp, fp,
                                            u32p, fp, fp, fp, u8p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_float, dp, dp, dp, i64p]
        L.oracle_reg_cost.argtypes = [C.c_size_t, u32p, C.c_int, fp, fp, i32p, C.c_int, u8pp, ip, u32p, fp, fp, fp, u8p, C.c_size_t,
                                      C.c_int, C.c_float, C.c_float, C.c_float, dp, i64p]
        L.oracle_reg_color_accumulate.argtypes = [C.c_size_t, u32p, C.c_int, C.c_int, u8pp, ip, u32p, fp, fp, fp, u8p, C.c_size_t, fp, i32p]
        L.oracle_reg_color_finish.argtypes = [C.c_size_t, C.c_int, fp, i32p]
        _READY = True
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def make_camera(w, h, params, ctype=0):
    c = Camera()
    pr = np.ascontiguousarray(params, np.float32)
    lib().oracle_reg_camera_init(C.byref(c), ctype, w, h, _p(pr, C.c_float))
    return c


def cam_distort(cam, nx, ny):
    o = np.zeros(2, np.float32); lib().oracle_reg_camera_distort(C.byref(cam), nx, ny, _p(o, C.c_float)); return o


def cam_undistort(cam, dx, dy):
    o = np.zeros(2, np.float32); conv = C.c_int()
    lib().oracle_reg_camera_undistort(C.byref(cam), dx, dy, _p(o, C.c_float), C.byref(conv)); return o, bool(conv.value)


def cam_project(cam, P):
    P = np.ascontiguousarray(P, np.float32); o = np.zeros(2, np.float32)
    lib().oracle_reg_camera_project(C.byref(cam), _p(P, C.c_float), _p(o, C.c_float)); return o


def cam_deriv_by_world(cam, P):
    P = np.ascontiguousarray(P, np.float32); o = np.zeros((2, 3), np.float32)
    lib().oracle_reg_camera_deriv_by_world(C.byref(cam), _p(P, C.c_float), _p(o, C.c_float)); return o


def cam_deriv_by_intrinsics(cam, P):
    P = np.ascontiguousarray(P, np.float32); o = np.zeros((2, cam.n_params), np.float32)
    lib().oracle_reg_camera_deriv_by_intrinsics(C.byref(cam), _p(P, C.c_float), _p(o, C.c_float)); return o


def camera_pyramid(cam, n_levels):
    """Intrinsics::BuildModelPyramid: successive ScaledBy(0.5)."""
    arr = (Camera * n_levels)()
    arr[0] = cam
    for i in range(1, n_levels):
        lib().oracle_reg_camera_scaled(C.byref(arr[i - 1]), 0.5, C.byref(arr[i]))
    return arr


def _img_ptrs(images):
    keep = [np.ascontiguousarray(im, np.uint8) if im is not None else None for im in images]
    arr = (C.POINTER(C.c_uint8) * len(keep))()
    for i, im in enumerate(keep):
        arr[i] = _p(im, C.c_uint8) if im is not None else None
    widths = np.array([im.shape[1] if im is not None else 0 for im in keep], np.int32)
    return arr, widths, keep


def trilinear(img0, img1, x0, y0, z, derivs=False):
    f32 = img0.dtype != np.uint8
    a0 = np.ascontiguousarray(img0, np.float32 if f32 else np.uint8)
    a1 = np.ascontiguousarray(img1, np.float32 if f32 else np.uint8)
    t = C.c_float if f32 else C.c_uint8
    suffix = "f32" if f32 else "u8"
    v, dx, dy, dz = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    if derivs:
        getattr(lib(), "oracle_interp_trilinear_d_" + suffix)(_p(a0, t), a0.shape[1], _p(a1, t), a1.shape[1], x0, y0, z,
                                                              C.byref(v), C.byref(dx), C.byref(dy), C.byref(dz))
        return v.value, dx.value, dy.value, dz.value
    getattr(lib(), "oracle_interp_trilinear_" + suffix)(_p(a0, t), a0.shape[1], _p(a1, t), a1.shape[1], x0, y0, z, C.byref(v))
    return v.value


def splat_depth(pts, R, t, cam, point_radius):
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    d = np.zeros((cam.height, cam.width), np.float32)
    lib().oracle_reg_splat_depth(_p(pts, C.c_float), pts.shape[0], _p(R, C.c_float), _p(t, C.c_float), C.byref(cam), point_radius, _p(d, C.c_float))
    return d


def observe(pts, point_radius, R, t, levels, min_image_scale, images, masks, occlusion, image_scale, border,
            current_image_scale, image_scale_count, occlusion_threshold=0.01, max_valid_intensity=252.0, indices=None):
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    n = pts.shape[0]
    ip, widths, keep = _img_ptrs(images)
    mp, _, keepm = _img_ptrs(masks if masks is not None else [None] * len(images))
    idx = np.ascontiguousarray(indices, np.uint32) if indices is not None else None
    cap = len(idx) if idx is not None else n
    oi = np.zeros(cap + 1, np.uint32); ox = np.zeros(cap + 1, np.float32); oy = np.zeros(cap + 1, np.float32); os_ = np.zeros(cap + 1, np.float32)
    occ = np.ascontiguousarray(occlusion, np.float32) if occlusion is not None else np.zeros((1, 1), np.float32)
    c = lib().oracle_reg_observe(_p(pts, C.c_float), n, point_radius, _p(idx, C.c_uint32) if idx is not None else None, cap if idx is not None else 0,
                                 _p(R, C.c_float), _p(t, C.c_float), levels, min_image_scale, len(levels), ip, mp if masks is not None else None,
                                 _p(occ, C.c_float), image_scale, border, current_image_scale, image_scale_count,
                                 occlusion_threshold, max_valid_intensity, _p(oi, C.c_uint32), _p(ox, C.c_float), _p(oy, C.c_float), _p(os_, C.c_float))
    return oi[:c].copy(), ox[:c].copy(), oy[:c].copy(), os_[:c].copy()


def neighbors_observed(n_pts, obs_idx, nbr, K):
    obs_idx = np.ascontiguousarray(obs_idx, np.uint32); nbr = np.ascontiguousarray(nbr, np.uint32)
    f = np.zeros(len(obs_idx) + 1, np.uint8)
    lib().oracle_reg_neighbors_observed(n_pts, _p(obs_idx, C.c_uint32), len(obs_idx), _p(nbr, C.c_uint32), K, _p(f, C.c_uint8))
    return f[:len(obs_idx)].copy()


def se3_mul(qa, ta, qb, tb):
    a = [np.ascontiguousarray(v, np.float32) for v in (qa, ta, qb, tb)]
    q = np.zeros(4, np.float32); t = np.zeros(3, np.float32)
    lib().oracle_se3_mul(*[_p(v, C.c_float) for v in a], _p(q, C.c_float), _p(t, C.c_float))
    return q, t


def pass1(pts, point_radius, cam_min, min_image_scale, images, R, t, obs, rig=None):
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    ip, widths, keep = _img_ptrs(images)
    oi, ox, oy, os_ = [np.ascontiguousarray(a) for a in obs]
    n = len(oi)
    I = np.zeros(n + 1, np.float32); JI = np.zeros((n + 1, cam_min.n_params), np.float32); JP = np.zeros((n + 1, 6), np.float32)
    JR = np.zeros((n + 1, 6), np.float32)
    lib().oracle_reg_pass1_rig(_p(pts, C.c_float), point_radius, C.byref(cam_min), min_image_scale, ip, _p(widths, C.c_int), _p(R, C.c_float),
                               _p(t, C.c_float), _p(oi, C.c_uint32), _p(ox, C.c_float), _p(oy, C.c_float), _p(os_, C.c_float), n,
                               C.byref(rig) if rig is not None else None, _p(I, C.c_float), _p(JI, C.c_float), _p(JP, C.c_float),
                               _p(JR, C.c_float))
    if rig is not None:
        return I[:n].copy(), JI[:n].copy(), JP[:n].copy(), JR[:n].copy()
    return I[:n].copy(), JI[:n].copy(), JP[:n].copy()


def accumulate(pts, point_radius, nbr, K, fixed_desc, var_desc, obs_counts, cam_min, min_image_scale, images, R, t, obs, flags,
               robust_type, robust_param, fixed_weight, var_weight, rig=None):
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    nbr = np.ascontiguousarray(nbr, np.uint32); fd = np.ascontiguousarray(fixed_desc, np.float32); vd = np.ascontiguousarray(var_desc, np.float32)
    oc = np.ascontiguousarray(obs_counts, np.int32); fl = np.ascontiguousarray(flags, np.uint8)
    ip, widths, keep = _img_ptrs(images)
    oi, ox, oy, os_ = [np.ascontiguousarray(a) for a in obs]
    V = cam_min.n_params + (12 if rig is not None else 6)
    H = np.zeros((V, V)); b = np.zeros(V); sums = np.zeros(2); counts = np.zeros(2, np.int64)
    lib().oracle_reg_accumulate_rig(_p(pts, C.c_float), pts.shape[0], point_radius, _p(nbr, C.c_uint32), K, _p(fd, C.c_float), _p(vd, C.c_float),
                                    _p(oc, C.c_int32), C.byref(cam_min), min_image_scale, ip, _p(widths, C.c_int), _p(R, C.c_float), _p(t, C.c_float),
                                    _p(oi, C.c_uint32), _p(ox, C.c_float), _p(oy, C.c_float), _p(os_, C.c_float), _p(fl, C.c_uint8), len(oi),
                                    robust_type, robust_param, fixed_weight, var_weight, C.byref(rig) if rig is not None else None,
                                    _p(H, C.c_double), _p(b, C.c_double), _p(sums, C.c_double), _p(counts, C.c_int64))
    return H, b, sums, counts


def cost(n_pts, nbr, K, fixed_desc, var_desc, obs_counts, min_image_scale, images, obs, flags, robust_type, robust_param,
         fixed_weight, var_weight):
    nbr = np.ascontiguousarray(nbr, np.uint32); fd = np.ascontiguousarray(fixed_desc, np.float32); vd = np.ascontiguousarray(var_desc, np.float32)
    oc = np.ascontiguousarray(obs_counts, np.int32); fl = np.ascontiguousarray(flags, np.uint8)
    ip, widths, keep = _img_ptrs(images)
    oi, ox, oy, os_ = [np.ascontiguousarray(a) for a in obs]
    sums = np.zeros(2); counts = np.zeros(2, np.int64)
    lib().oracle_reg_cost(n_pts, _p(nbr, C.c_uint32), K, _p(fd, C.c_float), _p(vd, C.c_float), _p(oc, C.c_int32), min_image_scale, ip,
                          _p(widths, C.c_int), _p(oi, C.c_uint32), _p(ox, C.c_float), _p(oy, C.c_float), _p(os_, C.c_float), _p(fl, C.c_uint8),
                          len(oi), robust_type, robust_param, fixed_weight, var_weight, _p(sums, C.c_double), _p(counts, C.c_int64))
    return sums, counts


def _depth_ptrs(depth_maps):
    keep = [np.ascontiguousarray(d, np.float32) for d in depth_maps]
    arr = (C.POINTER(C.c_float) * len(keep))()
    for i, d in enumerate(keep):
        arr[i] = _p(d, C.c_float)
    widths = np.array([d.shape[1] for d in keep], np.int32)
    return arr, widths, keep


def depth_rows(pts, point_radius, cam_min, min_image_scale, depth_maps, R, t, q, obs):
    """depth residual and its Jacobian rows per observation: (residuals[n], j_intrinsics[n, I], j_pose[n, 6])"""
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    dp, widths, keep = _depth_ptrs(depth_maps)
    oi, ox, oy, os_ = [np.ascontiguousarray(a) for a in obs]
    n = len(oi)
    res = np.zeros(n + 1, np.float32); JI = np.zeros((n + 1, cam_min.n_params), np.float32); JP = np.zeros((n + 1, 6), np.float32)
    f = lib().oracle_reg_depth_rows
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    f(pts.ctypes.data, point_radius, C.addressof(cam_min), min_image_scale, C.cast(dp, C.c_void_p), widths.ctypes.data, R.ctypes.data,
      t.ctypes.data, q.ctypes.data, oi.ctypes.data, ox.ctypes.data, oy.ctypes.data, os_.ctypes.data, n, res.ctypes.data, JI.ctypes.data,
      JP.ctypes.data)
    return res[:n].copy(), JI[:n].copy(), JP[:n].copy()


def depth_accumulate(residuals, j_intr, j_pose, robust_type, robust_param, depth_weight):
    res = np.ascontiguousarray(residuals, np.float32); JI = np.ascontiguousarray(j_intr, np.float32); JP = np.ascontiguousarray(j_pose, np.float32)
    I = JI.shape[1]; V = I + 6
    H = np.zeros((V, V)); b = np.zeros(V); sm = C.c_double(0); cn = C.c_int64(0)
    f = lib().oracle_reg_depth_accumulate
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_void_p]
    f(res.ctypes.data, JI.ctypes.data, JP.ctypes.data, len(res), I, robust_type, robust_param, depth_weight, H.ctypes.data, b.ctypes.data,
      C.addressof(sm), C.addressof(cn))
    return H, b, sm.value, cn.value


def depth_cost(pts, min_image_scale, depth_maps, q, t, obs, robust_type, robust_param):
    pts = np.ascontiguousarray(pts, np.float32); q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    dp, widths, keep = _depth_ptrs(depth_maps)
    oi, ox, oy, os_ = [np.ascontiguousarray(a) for a in obs]
    sm = C.c_double(0); cn = C.c_int64(0)
    f = lib().oracle_reg_depth_cost
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                  C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    f(pts.ctypes.data, min_image_scale, C.cast(dp, C.c_void_p), widths.ctypes.data, q.ctypes.data, t.ctypes.data, oi.ctypes.data, ox.ctypes.data,
      oy.ctypes.data, os_.ctypes.data, len(oi), robust_type, robust_param, C.addressof(sm), C.addressof(cn))
    return sm.value, cn.value


def color_accumulate(n_pts, nbr, K, min_image_scale, images, obs, flags, descriptors, obs_counts):
    nbr = np.ascontiguousarray(nbr, np.uint32); fl = np.ascontiguousarray(flags, np.uint8)
    ip, widths, keep = _img_ptrs(images)
    oi, ox, oy, os_ = [np.ascontiguousarray(a) for a in obs]
    assert descriptors.dtype == np.float32 and obs_counts.dtype == np.int32 and descriptors.flags.c_contiguous
    lib().oracle_reg_color_accumulate(n_pts, _p(nbr, C.c_uint32), K, min_image_scale, ip, _p(widths, C.c_int), _p(oi, C.c_uint32),
                                      _p(ox, C.c_float), _p(oy, C.c_float), _p(os_, C.c_float), _p(fl, C.c_uint8), len(oi),
                                      _p(descriptors, C.c_float), _p(obs_counts, C.c_int32))


def color_finish(K, descriptors, obs_counts):
    lib().oracle_reg_color_finish(len(obs_counts), K, _p(descriptors, C.c_float), _p(obs_counts, C.c_int32))


def scan_visibility(pts, R, t, cam, occlusion, counts, mask=None, excluded=2, mode=0, min_count=2, occlusion_threshold=0.01):
    """GroundTruthCreator: mode 0 increments `counts` (int32, in place) for visible scan points; mode 1 returns the ground-truth
    depth map (min z over visible points with counts >= min_count, +inf elsewhere)."""
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    occ = np.ascontiguousarray(occlusion, np.float32)
    assert counts.dtype == np.int32 and counts.flags.c_contiguous
    gt = np.full((cam.height, cam.width), np.inf, np.float32)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    f = lib().oracle_scan_visibility
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int,
                  C.c_void_p, C.c_void_p]
    f(pts.ctypes.data, pts.shape[0], R.ctypes.data, t.ctypes.data, C.addressof(cam), occ.ctypes.data, occlusion_threshold,
      m.ctypes.data if m is not None else None, excluded, mode, min_count, counts.ctypes.data, gt.ctypes.data)
    return gt


def scan_rendering(pts, R, t, cam, occlusion, counts, radius, mask=None, excluded=2, min_count=2, occlusion_threshold=0.01):
    """GroundTruthCreator --write_scan_renderings: (height, width) uint32, index + 1 of the scan point painted last over each pixel."""
    pts = np.ascontiguousarray(pts, np.float32); R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
    occ = np.ascontiguousarray(occlusion, np.float32); counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros((cam.height, cam.width), np.uint32)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    f = lib().oracle_scan_rendering
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                  C.c_int, C.c_void_p]
    f(pts.ctypes.data, pts.shape[0], R.ctypes.data, t.ctypes.data, C.addressof(cam), occ.ctypes.data, occlusion_threshold,
      m.ctypes.data if m is not None else None, excluded, min_count, counts.ctypes.data, radius, out.ctypes.data)
    return out
