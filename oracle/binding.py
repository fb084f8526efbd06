"""ctypes binding of the CPU oracle (oracle/libe3d_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def variant():
    """"" (default: the bit-defined elementary functions shared with the kernels) or "glibc" (E3D_ORACLE_VARIANT=glibc: the C
    library's, as the reference calls them -- oracle_libm_select.h; used by tests/test_oracle_glibc.py in a subprocess)."""
    return "glibc" if os.environ.get("E3D_ORACLE_VARIANT", "") == "glibc" else ""


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    glibc = variant() == "glibc"
    so = os.path.join(_HERE, "libe3d_oracle_glibc.so" if glibc else "libe3d_oracle.so")
    if force or not os.path.exists(so) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
        for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".cc"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["glibc"] if glibc else []))
    build_ref()
    return so


def ref_lib_path():
    return os.path.join(_HERE, "_ref", "libe3d_ref.so")


def build_ref():
    """oracle/_ref: the reference's own dependency-free sources, compiled in place when /root/reference exists (here, not on the GPU box)."""
    if os.path.exists("/root/reference/src/opt/robust_weighting.h"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref"])
    return ref_lib_path() if os.path.exists(ref_lib_path()) else None


class PairRecord(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("src", C.c_int32), ("tgt", C.c_int32),
                ("count", C.c_int64), ("distance_sum", C.c_float)]


class IterRecord(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("inner_iterations", C.c_int32),
                ("accumulate_passes", C.c_int32), ("cost_passes", C.c_int32),
                ("correspondences", C.c_int64), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("t_transform_s", C.c_double),
                ("t_nn_s", C.c_double), ("t_lm_s", C.c_double)]


def effective_cores():
    """CPUs this process can really use: the scheduler affinity, capped by the cgroup's CPU quota (a container on a 256-core host may
    own a fraction of it: 256 OpenMP threads on a quota of 20 CPUs thrash instead of scaling)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())            # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_num_threads(n):
    """OpenMP threads of the oracle's parallel regions from now on (libgomp's omp_set_num_threads)."""
    lib()
    g = C.CDLL("libgomp.so.1")
    g.omp_set_num_threads(int(max(1, n)))
    return int(max(1, n))


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        dp = C.POINTER(C.c_double)
        L.oracle_icp_create.restype = C.c_void_p
        L.oracle_icp_destroy.argtypes = [C.c_void_p]
        L.oracle_icp_add_cloud.argtypes = [C.c_void_p, fp, fp, C.c_size_t, fp, C.c_int]
        L.oracle_icp_run.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int]
        L.oracle_icp_get_pose.argtypes = [C.c_void_p, C.c_int, fp]
        L.oracle_icp_set_max_inner_iterations.argtypes = [C.c_void_p, C.c_int]
        L.oracle_icp_set_all_core.argtypes = [C.c_void_p, C.c_int]
        L.oracle_icp_num_pair_records.argtypes = [C.c_void_p]
        L.oracle_icp_num_pair_records.restype = C.c_size_t
        L.oracle_icp_pair_records.argtypes = [C.c_void_p]
        L.oracle_icp_pair_records.restype = C.POINTER(PairRecord)
        L.oracle_icp_num_iter_records.argtypes = [C.c_void_p]
        L.oracle_icp_num_iter_records.restype = C.c_size_t
        L.oracle_icp_iter_records.argtypes = [C.c_void_p]
        L.oracle_icp_iter_records.restype = C.POINTER(IterRecord)
        for name in ("oracle_find_correspondences", "oracle_find_correspondences_brute"):
            f = getattr(L, name)
            f.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_float, ip, ip, fp]
            f.restype = C.c_int64
        L.oracle_transform_cloud.argtypes = [fp, fp, C.c_size_t, fp, fp, fp, fp, fp]
        L.oracle_icp_pair_system.argtypes = [fp, fp, fp, fp, ip, ip, C.c_int64, fp, fp, fp, fp, dp, dp, dp]
        L.oracle_se3_update.argtypes = [dp, fp, fp, fp, fp]
        L.oracle_quat_to_R.argtypes = [fp, fp]
        L.oracle_ldlt_solve_upper.argtypes = [dp, C.c_int, dp, dp]
        L.oracle_normals.argtypes = [fp, C.c_size_t, C.c_int, C.c_float, fp, fp, fp, ip]
        L.oracle_normals.restype = C.c_int
        L.oracle_point_normal.argtypes = [fp, ip, C.c_int, fp, fp]
        L.oracle_knn.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, ip, fp]
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c32(a, shape_last=3):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == shape_last
    return a


class OracleICP:
    """Mirror of icp::PointToPlaneICP (src/icp/icp_point_to_plane.h:39-80) on the oracle."""

    def __init__(self):
        self._lib = lib()                      # (kept on the object: module globals may be gone when __del__ runs at interpreter exit)
        self._h = self._lib.oracle_icp_create()

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_lib", None) is not None:
            self._lib.oracle_icp_destroy(self._h)
            self._h = None

    def add_point_cloud(self, xyz, nrm, T, fixed):
        xyz = _c32(xyz); nrm = _c32(nrm)
        T = np.ascontiguousarray(np.asarray(T, dtype=np.float32)[:3, :4])
        return lib().oracle_icp_add_cloud(self._h, _f(xyz), _f(nrm), xyz.shape[0], _f(T), int(bool(fixed)))

    def run(self, max_correspondence_distance, initial_iteration, max_num_iterations,
            convergence_threshold_max_movement, print_progress=False):
        r = lib().oracle_icp_run(self._h, np.float32(max_correspondence_distance), initial_iteration,
                                 max_num_iterations, np.float32(convergence_threshold_max_movement),
                                 int(bool(print_progress)))
        if r < 0:
            raise RuntimeError("oracle_icp_run failed (no clouds)")
        return bool(r)

    def get_result_global_T_cloud(self, idx):
        T = np.zeros((3, 4), dtype=np.float32)
        if lib().oracle_icp_get_pose(self._h, idx, _f(T)) != 0:
            raise IndexError(idx)
        out = np.eye(4, dtype=np.float32)
        out[:3] = T
        return out

    def set_max_inner_iterations(self, n):
        lib().oracle_icp_set_max_inner_iterations(self._h, n)

    def set_all_core(self, on):
        lib().oracle_icp_set_all_core(self._h, int(bool(on)))

    def pair_records(self):
        n = lib().oracle_icp_num_pair_records(self._h)
        p = lib().oracle_icp_pair_records(self._h)
        return [(p[i].iteration, p[i].src, p[i].tgt, p[i].count, p[i].distance_sum) for i in range(n)]

    def iter_records(self):
        n = lib().oracle_icp_num_iter_records(self._h)
        p = lib().oracle_icp_iter_records(self._h)
        names = [f[0] for f in IterRecord._fields_]
        return [{k: getattr(p[i], k) for k in names} for i in range(n)]


def find_correspondences(src, tgt, d, brute=False):
    src = _c32(src); tgt = _c32(tgt)
    n = src.shape[0]
    iq = np.zeros(n + 1, np.int32); im = np.zeros(n + 1, np.int32); sd = np.zeros(n + 1, np.float32)
    fn = lib().oracle_find_correspondences_brute if brute else lib().oracle_find_correspondences
    c = fn(_f(src), n, _f(tgt), tgt.shape[0], np.float32(d), _i(iq), _i(im), _f(sd))
    return iq[:c].copy(), im[:c].copy(), sd[:c].copy()


def transform_cloud(xyz, nrm, T):
    xyz = _c32(xyz); nrm = _c32(nrm)
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float32)[:3, :4])
    oxyz = np.empty_like(xyz); onrm = np.empty_like(nrm)
    bmin = np.zeros(3, np.float32); bmax = np.zeros(3, np.float32)
    lib().oracle_transform_cloud(_f(xyz), _f(nrm), xyz.shape[0], _f(T), _f(oxyz), _f(onrm), _f(bmin), _f(bmax))
    return oxyz, onrm, bmin, bmax


def pair_system(sxyz, snrm, txyz, tnrm, iq, im, sq, st, tq, tt):
    sxyz = _c32(sxyz); snrm = _c32(snrm); txyz = _c32(txyz); tnrm = _c32(tnrm)
    iq = np.ascontiguousarray(iq, np.int32); im = np.ascontiguousarray(im, np.int32)
    sq = np.ascontiguousarray(sq, np.float32); st = np.ascontiguousarray(st, np.float32)
    tq = np.ascontiguousarray(tq, np.float32); tt = np.ascontiguousarray(tt, np.float32)
    H = np.zeros((12, 12)); b = np.zeros(12); cost = C.c_double(0)
    lib().oracle_icp_pair_system(_f(sxyz), _f(snrm), _f(txyz), _f(tnrm), _i(iq), _i(im), iq.shape[0],
                                 _f(sq), _f(st), _f(tq), _f(tt), _d(H), _d(b), C.byref(cost))
    return H, b, cost.value


def se3_update(x, q, t):
    x = np.ascontiguousarray(x, np.float64)
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    qo = np.zeros(4, np.float32); to = np.zeros(3, np.float32)
    lib().oracle_se3_update(_d(x), _f(q), _f(t), _f(qo), _f(to))
    return qo, to


def quat_to_R(q):
    q = np.ascontiguousarray(q, np.float32)
    R = np.zeros((3, 3), np.float32)
    lib().oracle_quat_to_R(_f(q), _f(R))
    return R


def ldlt_solve_upper(A, b):
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b)
    lib().oracle_ldlt_solve_upper(_d(A), A.shape[0], _d(b), _d(x))
    return x


def normals(xyz, k=0, radius=-1.0, viewpoint=(0, 0, 0), return_knn=False):
    xyz = _c32(xyz)
    n = xyz.shape[0]
    vp = np.ascontiguousarray(viewpoint, np.float32)
    on = np.zeros((n, 3), np.float32); oc = np.zeros(n, np.float32)
    knn = np.zeros((n, max(k, 1)), np.int32) if (return_knn and k > 0) else None
    r = lib().oracle_normals(_f(xyz), n, int(k), np.float32(radius), _f(vp), _f(on), _f(oc),
                             _i(knn) if knn is not None else None)
    if r != 0:
        raise ValueError("oracle_normals: need k>0 or radius>0")
    return (on, oc, knn) if return_knn else (on, oc)


def normals_sample(xyz, sample, k=0, radius=-1.0, viewpoint=(0, 0, 0)):
    """Normals / curvature / kNN lists of the points `sample` (indices) of the cloud, searched in the WHOLE cloud."""
    xyz = _c32(xyz)
    sample = np.ascontiguousarray(sample, np.int64)
    m = sample.shape[0]
    vp = np.ascontiguousarray(viewpoint, np.float32)
    on = np.zeros((m, 3), np.float32); oc = np.zeros(m, np.float32)
    knn = np.zeros((m, max(k, 1)), np.int32)
    f = lib().oracle_normals_sample
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    r = f(xyz.ctypes.data, xyz.shape[0], int(k), float(radius), vp.ctypes.data, sample.ctypes.data, m, on.ctypes.data, oc.ctypes.data,
          knn.ctypes.data if k > 0 else None)
    if r != 0:
        raise ValueError("oracle_normals_sample: need k>0 or radius>0")
    return on, oc, knn


def libm_eval(fn, x, y=None):
    """include/e3d_libm.h evaluated on the host; fn in {"atanf", "atan2f", "sinf", "cosf", "tanf", "log2f"} (atan2f(x, y): x is the
    first argument, y the second)."""
    code = {"atanf": 0, "atan2f": 1, "sinf": 2, "cosf": 3, "tanf": 4, "log2f": 5}[fn]
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.zeros_like(x)
    f = lib().oracle_libm_eval
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    f(code, x.ctypes.data, y.ctypes.data, x.size, out.ctypes.data)
    return out


def point_normal(xyz, indices):
    xyz = _c32(xyz); indices = np.ascontiguousarray(indices, np.int32)
    plane = np.zeros(4, np.float32); curv = C.c_float(0)
    lib().oracle_point_normal(_f(xyz), _i(indices), indices.shape[0], _f(plane), C.byref(curv))
    return plane, curv.value


def knn(xyz, queries, k):
    xyz = _c32(xyz); queries = _c32(queries)
    idx = np.zeros((queries.shape[0], k), np.int32); dist = np.zeros((queries.shape[0], k), np.float32)
    lib().oracle_knn(_f(xyz), xyz.shape[0], _f(queries), queries.shape[0], k, _i(idx), _f(dist))
    return idx, dist


def local_outlier_removal(xyz, mean_k, factor, negative=False):
    """LocalStatisticalOutlierRemoval on finite points -> (inlier mask, first-pass mean distances)."""
    xyz = _c32(xyz)
    n = xyz.shape[0]
    inl = np.zeros(n, np.uint8); md = np.zeros(n, np.float32)
    f = lib().oracle_local_outlier_removal
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    if f(xyz.ctypes.data, n, int(mean_k), float(factor), int(bool(negative)), inl.ctypes.data, md.ctypes.data) != 0:
        raise ValueError("oracle_local_outlier_removal: too few points")
    return inl.astype(bool), md
