"""ctypes binding of libe3dhip.so -- one Python method per C-ABI entry point of include/e3d_hip.h.

Array arguments may be numpy arrays (host) or torch CUDA tensors (device pointers are passed through;
the library detects them).  There is deliberately NO fallback path: if the shared library is missing the
import of any entry point fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class E3DError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "lib", "libe3dhip.so")


class PairRecord(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("src", C.c_int32), ("tgt", C.c_int32),
                ("count", C.c_int64), ("distance_sum", C.c_double)]


class IterRecord(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("inner_iterations", C.c_int32), ("full_passes", C.c_int32),
                ("cost_passes", C.c_int32), ("multi_cost_passes", C.c_int32), ("multi_cost_poses", C.c_int32),
                ("correspondences", C.c_int64), ("queries", C.c_int64),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("t_transform_ms", C.c_double), ("t_nn_ms", C.c_double), ("t_lm_ms", C.c_double),
                ("t_lm_kernel_ms", C.c_double), ("t_nn_query_ms", C.c_double), ("t_lm_full_kernel_ms", C.c_double),
                ("t_nn_certify_ms", C.c_double), ("t_nn_bounded_ms", C.c_double), ("t_nn_search_ms", C.c_double),
                ("nn_certify_queries", C.c_int64), ("nn_bounded_queries", C.c_int64), ("nn_search_queries", C.c_int64),
                ("nn_certify_launches", C.c_int32), ("nn_bounded_launches", C.c_int32), ("nn_search_launches", C.c_int32),
                ("lm_passes_skipped", C.c_int32), ("t_nn_sort_ms", C.c_double), ("t_nn_scan_ms", C.c_double), ("t_nn_compact_ms", C.c_double),
                ("corr_rows_rewritten", C.c_int64), ("corr_rows_walked", C.c_int64),
                ("nn_update_launches", C.c_int32), ("nn_kernel_launches", C.c_int32), ("nn_batches", C.c_int32), ("nn_sort_calls", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_void_p)
ALLREDUCE_DEVICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# name -> (restype, argtypes); also the list of symbols include/e3d_hip.h declares
SIGNATURES = {
    "e3d_abi_version": (C.c_int, []),
    "e3d_init": (C.c_int, [C.c_int]),
    "e3d_last_error": (C.c_char_p, []),
    "e3d_set_nn_mode": (C.c_int, [C.c_int]),
    "e3d_icp_create": (C.c_void_p, []),
    "e3d_icp_destroy": (None, [C.c_void_p]),
    "e3d_icp_add_cloud": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "e3d_icp_run": (C.c_int, [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int]),
    "e3d_icp_get_pose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "e3d_icp_set_max_inner_iterations": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_icp_set_sequential_distance_sum": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_icp_set_resident_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_icp_num_pair_records": (C.c_size_t, [C.c_void_p]),
    "e3d_icp_pair_records": (C.POINTER(PairRecord), [C.c_void_p]),
    "e3d_icp_num_iter_records": (C.c_size_t, [C.c_void_p]),
    "e3d_icp_iter_records": (C.POINTER(IterRecord), [C.c_void_p]),
    "e3d_icp_clear_records": (None, [C.c_void_p]),
    "e3d_icp_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]),
    "e3d_comm_unique_id": (C.c_int, [C.c_void_p]),
    "e3d_comm_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "e3d_comm_create_all": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_comm_destroy": (None, [C.c_void_p]),
    "e3d_comm_abort": (C.c_int, [C.c_void_p]),
    "e3d_comm_get_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "e3d_comm_rank": (C.c_int, [C.c_void_p]),
    "e3d_comm_world_size": (C.c_int, [C.c_void_p]),
    "e3d_icp_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "e3d_reg_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "e3d_reg_kernel_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "e3d_reg_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    "e3d_find_correspondences": (C.c_int64, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_float,
                                             C.c_void_p, C.c_void_p]),
    "e3d_transform_cloud": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "e3d_icp_pair_system": (C.c_int, [C.c_void_p] * 6 + [C.c_int64] + [C.c_void_p] * 7),
    "e3d_libm_eval": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "e3d_normals_knn": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_normals_radius": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_release_workspaces": (C.c_int, []),
    "e3d_local_outlier_removal": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    # (B) image registration kernels
    "e3d_reg_create": (C.c_void_p, [C.c_void_p]),
    "e3d_reg_destroy": (None, [C.c_void_p]),
    "e3d_reg_set_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "e3d_reg_set_point_scale": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_variable_descriptors": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_get_variable_descriptors": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_camera_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "e3d_reg_set_depth_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "e3d_reg_depth_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_depth_cost": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_intrinsics": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "e3d_reg_get_intrinsics_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_image": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_image_pose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_splat_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "e3d_reg_add_occlusion_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]),
    "e3d_reg_clear_occlusion_meshes": (C.c_int, [C.c_void_p]),
    "e3d_reg_set_occlusion_options": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_int]),
    "e3d_reg_occlusion_edge_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "e3d_reg_render_depth": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "e3d_reg_observe": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "e3d_reg_get_observations": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_observations": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_pass1": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_cost": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_color_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_reg_color_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "e3d_reg_color_finish": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_reg_get_image_pose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_update_observations": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_reg_set_scan_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "e3d_reg_count_scan_observations": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "e3d_reg_get_scan_observation_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "e3d_reg_set_scan_observation_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "e3d_reg_ground_truth_depth": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_scan_rendering": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "e3d_reg_set_cache_observations": (C.c_int, [C.c_void_p, C.c_int]),
    "e3d_reg_determine_observed_indices": (C.c_int, [C.c_void_p]),
    "e3d_reg_get_observed_indices": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "e3d_reg_set_observed_indices": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "e3d_reg_color_update": (C.c_int, [C.c_void_p]),
    "e3d_reg_compute_cost": (C.c_int, [C.c_void_p, C.c_void_p]),
    "e3d_reg_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "e3d_reg_run_on_current_scale": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_set_rig": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_get_rig": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "e3d_reg_add_rig_images": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "e3d_determine_point_neighbors": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "e3d_reg_point_radius_minmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "e3d_merge_close_points": (C.c_int64, [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "e3d_reg_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, ALLREDUCE_DEVICE_FN, C.c_void_p]),
    "e3d_reg_image_owner": (C.c_int, [C.c_void_p, C.c_int]),
}


ABI_VERSION = 5          # E3D_ABI_VERSION of include/e3d_hip.h


def lib():
    """Load libe3dhip.so (raises E3DError if it has not been built -- no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise E3DError("HIP extension missing: %s (run `python dataset-pipeline_amd/build.py`)" % p)
        # One HIP runtime per process: the ROCm torch wheel carries its own libamdhip64.  If torch is imported AFTER this
        # library has pulled in the system runtime, torch.cuda sees no device; imported first, both bind to the same copy
        # (measured on the MI355X box, tools/probe_hip_runtime.py).  torch is optional -- only tests / bench.py / dist.py use it.
        if os.environ.get("E3D_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:  # noqa: BLE001
                pass
        L = C.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.e3d_abi_version() != ABI_VERSION:      # the structures below mirror include/e3d_hip.h of that version
            raise E3DError("libe3dhip.so has ABI version %d, this binding expects %d: rebuild (python dataset-pipeline_amd/build.py)"
                           % (L.e3d_abi_version(), ABI_VERSION))
        _LIB = L
    return _LIB


def _err(prefix, code=None):
    msg = lib().e3d_last_error()
    raise E3DError("%s: %s%s" % (prefix, msg.decode() if msg else "unknown error",
                                 "" if code is None else " (code %d)" % code))


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a, dtype, keep):
    """Pointer of a numpy array / torch tensor with the given numpy dtype (contiguous); None passes through."""
    if a is None:
        return None
    if _is_torch(a):
        import torch
        want = {np.float32: torch.float32, np.int32: torch.int32, np.float64: torch.float64}[dtype]
        t = a.contiguous()
        if t.dtype != want:
            t = t.to(want)
        keep.append(t)
        if t.is_cuda:
            # the library reads device memory on its own (non-blocking) stream: wait for the work torch queued on the
            # tensor's device before handing the pointer over
            torch.cuda.current_stream(t.device).synchronize()
        return C.c_void_p(t.data_ptr())
    arr = np.ascontiguousarray(a, dtype=dtype)
    keep.append(arr)
    return C.c_void_p(arr.ctypes.data)


def _pose12(T):
    T = np.asarray(T, dtype=np.float32)
    return np.ascontiguousarray(T[:3, :4])


class PointToPlaneICP:
    """Host-side mirror of icp::PointToPlaneICP (src/icp/icp_point_to_plane.h:39-80) on the HIP library."""

    def __init__(self, device=None):
        L = lib()
        if device is not None and L.e3d_init(int(device)) < 0:
            _err("e3d_init")
        self._h = L.e3d_icp_create()
        if not self._h:
            _err("e3d_icp_create")
        self._cb = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().e3d_icp_destroy(h)
            self._h = None

    # int AddPointCloud(cloud, global_T_cloud, fixed)
    def add_point_cloud(self, xyz, normals, global_T_cloud, fixed):
        keep = []
        n = int(xyz.shape[0])
        T = _pose12(global_T_cloud)
        r = lib().e3d_icp_add_cloud(self._h, _ptr(xyz, np.float32, keep), _ptr(normals, np.float32, keep), n,
                                    C.c_void_p(T.ctypes.data), int(bool(fixed)))
        if r < -1:
            _err("e3d_icp_add_cloud", r)
        return r

    # bool Run(max_correspondence_distance, initial_iteration, max_num_iterations, thr, print_progress)
    def run(self, max_correspondence_distance, initial_iteration, max_num_iterations,
            convergence_threshold_max_movement, print_progress=False):
        r = lib().e3d_icp_run(self._h, float(np.float32(max_correspondence_distance)), int(initial_iteration),
                              int(max_num_iterations), float(np.float32(convergence_threshold_max_movement)),
                              int(bool(print_progress)))
        if r < 0:
            _err("e3d_icp_run", r)
        return bool(r)

    # Eigen::Affine3f GetResultGlobalTCloud(int)
    def get_result_global_T_cloud(self, cloud_index):
        T = np.zeros((3, 4), np.float32)
        r = lib().e3d_icp_get_pose(self._h, int(cloud_index), C.c_void_p(T.ctypes.data))
        if r < 0:
            if r == -5:
                raise IndexError(cloud_index)
            _err("e3d_icp_get_pose", r)
        out = np.eye(4, dtype=np.float32)
        out[:3] = T
        return out

    def set_sequential_distance_sum(self, enable):
        if lib().e3d_icp_set_sequential_distance_sum(self._h, int(bool(enable))) < 0:
            _err("e3d_icp_set_sequential_distance_sum")

    def set_resident_rows(self, enable):
        """Resident correspondence rows (default) or rows compacted afresh every outer iteration (include/e3d_hip.h)."""
        if lib().e3d_icp_set_resident_rows(self._h, int(bool(enable))) < 0:
            _err("e3d_icp_set_resident_rows")

    def set_max_inner_iterations(self, n):
        if lib().e3d_icp_set_max_inner_iterations(self._h, int(n)) < 0:
            _err("e3d_icp_set_max_inner_iterations")

    def set_comm(self, comm):
        """Native multi-GPU mode: `comm` is a Comm (the library's RCCL communicator); replaces set_shard."""
        self._comm = comm
        if lib().e3d_icp_set_comm(self._h, comm.handle if comm is not None else None) < 0:
            _err("e3d_icp_set_comm")

    def set_shard(self, rank, world_size, allreduce=None):
        """allreduce(np.ndarray float64, in place) -> None; called from inside run()."""
        if allreduce is not None:
            def _cb(buf, count, _user):
                try:
                    arr = np.ctypeslib.as_array(buf, shape=(count,))
                    allreduce(arr)
                    return 0
                except Exception:  # noqa: BLE001 -- must not propagate through C
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLREDUCE_FN(_cb)
        else:
            self._cb = ALLREDUCE_FN()
        if lib().e3d_icp_set_shard(self._h, int(rank), int(world_size), self._cb, None) < 0:
            _err("e3d_icp_set_shard")

    def pair_records(self):
        n = lib().e3d_icp_num_pair_records(self._h)
        p = lib().e3d_icp_pair_records(self._h)
        return [(p[i].iteration, p[i].src, p[i].tgt, p[i].count, p[i].distance_sum) for i in range(n)]

    def iter_records(self):
        n = lib().e3d_icp_num_iter_records(self._h)
        p = lib().e3d_icp_iter_records(self._h)
        names = [f[0] for f in IterRecord._fields_]
        return [{k: getattr(p[i], k) for k in names} for i in range(n)]

    def clear_records(self):
        lib().e3d_icp_clear_records(self._h)


def find_correspondences(source_xyz, target_xyz, max_correspondence_distance):
    """FindCorrespondencesFast: returns (match_index[n_src] int32 (-1 = none), sq_distance[n_src], count)."""
    keep = []
    ns, nt = int(source_xyz.shape[0]), int(target_xyz.shape[0])
    idx = np.full(max(ns, 1), -1, np.int32)
    d2 = np.zeros(max(ns, 1), np.float32)
    c = lib().e3d_find_correspondences(_ptr(source_xyz, np.float32, keep), ns, _ptr(target_xyz, np.float32, keep), nt,
                                       float(np.float32(max_correspondence_distance)),
                                       C.c_void_p(idx.ctypes.data), C.c_void_p(d2.ctypes.data))
    if c < 0:
        _err("e3d_find_correspondences", c)
    return idx[:ns], d2[:ns], int(c)


def transform_cloud(xyz, normals, T):
    keep = []
    n = int(xyz.shape[0])
    T = _pose12(T)
    oxyz = np.zeros((n, 3), np.float32)
    onrm = np.zeros((n, 3), np.float32)
    bmin = np.zeros(3, np.float32)
    bmax = np.zeros(3, np.float32)
    r = lib().e3d_transform_cloud(_ptr(xyz, np.float32, keep), _ptr(normals, np.float32, keep), n,
                                  C.c_void_p(T.ctypes.data), C.c_void_p(oxyz.ctypes.data),
                                  C.c_void_p(onrm.ctypes.data), C.c_void_p(bmin.ctypes.data),
                                  C.c_void_p(bmax.ctypes.data))
    if r < 0:
        _err("e3d_transform_cloud", r)
    return oxyz, onrm, bmin, bmax


def icp_pair_system(sxyz, snrm, txyz, tnrm, iq, im, sq, st, tq, tt):
    keep = []
    H = np.zeros((12, 12)); b = np.zeros(12); cost = np.zeros(1)
    r = lib().e3d_icp_pair_system(
        _ptr(sxyz, np.float32, keep), _ptr(snrm, np.float32, keep), _ptr(txyz, np.float32, keep),
        _ptr(tnrm, np.float32, keep), _ptr(iq, np.int32, keep), _ptr(im, np.int32, keep), int(len(iq)),
        _ptr(sq, np.float32, keep), _ptr(st, np.float32, keep), _ptr(tq, np.float32, keep),
        _ptr(tt, np.float32, keep), C.c_void_p(H.ctypes.data), C.c_void_p(b.ctypes.data),
        C.c_void_p(cost.ctypes.data))
    if r < 0:
        _err("e3d_icp_pair_system", r)
    return H, b, float(cost[0])


class Comm:
    """The library's RCCL communicator (include/e3d_hip.h: e3d_comm_*).  One rank per GPU.

    Processes: rank 0 calls Comm.unique_id(), ships the 128 bytes through the launcher's rendezvous, every rank builds
    Comm(id_bytes, rank, world, device).  Threads of one process: Comm.create_all(n_devices)."""

    def __init__(self, id_bytes=None, rank=0, world_size=1, device=0, _handle=None):
        if _handle is not None:
            self.handle = _handle
        else:
            if id_bytes is None:
                id_bytes = Comm.unique_id()
            buf = C.create_string_buffer(bytes(id_bytes), 128)
            self.handle = lib().e3d_comm_create(buf, int(rank), int(world_size), int(device))
            if not self.handle:
                _err("e3d_comm_create")
        self.rank = lib().e3d_comm_rank(self.handle)
        self.world_size = lib().e3d_comm_world_size(self.handle)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        if lib().e3d_comm_unique_id(buf) < 0:
            _err("e3d_comm_unique_id")
        return buf.raw

    @staticmethod
    def create_all(n_devices, devices=None):
        out = (C.c_void_p * n_devices)()
        dev = (C.c_int * n_devices)(*(devices if devices is not None else range(n_devices)))
        if lib().e3d_comm_create_all(int(n_devices), dev, out) < 0:
            _err("e3d_comm_create_all")
        return [Comm(_handle=out[i]) for i in range(n_devices)]

    def stats(self, reset=False):
        """(all-reduce milliseconds by HIP events, calls, payload bytes) of this rank since creation / the last reset."""
        ms, calls, nbytes = C.c_double(0), C.c_int64(0), C.c_int64(0)
        if lib().e3d_comm_get_stats(self.handle, C.byref(ms), C.byref(calls), C.byref(nbytes), int(bool(reset))) < 0:
            _err("e3d_comm_get_stats")
        return ms.value, calls.value, nbytes.value

    def destroy(self):
        if self.handle:
            lib().e3d_comm_destroy(self.handle)
            self.handle = None


def release_workspaces():
    """Free the device workspaces e3d_normals_knn / e3d_local_outlier_removal parked for their next call."""
    lib().e3d_release_workspaces()


def libm_eval(fn, x, y=None):
    """include/e3d_libm.h evaluated by a HIP kernel; fn in {"atanf", "atan2f", "sinf", "cosf", "tanf", "log2f"}."""
    code = {"atanf": 0, "atan2f": 1, "sinf": 2, "cosf": 3, "tanf": 4, "log2f": 5}[fn]
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.zeros_like(x)
    r = lib().e3d_libm_eval(code, C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), x.size, C.c_void_p(out.ctypes.data))
    if r < 0:
        _err("e3d_libm_eval", r)
    return out


def normals_knn(xyz, k, viewpoint=(0.0, 0.0, 0.0), return_knn=False):
    """NormalEstimationTwoPassOMP with setKSearch(k): returns (normals[n,3], curvature[n][, knn_idx[n,k]])."""
    keep = []
    n = int(xyz.shape[0])
    vp = np.ascontiguousarray(viewpoint, np.float32)
    on = np.zeros((n, 3), np.float32)
    oc = np.zeros(n, np.float32)
    knn = np.zeros((n, k), np.int32) if return_knn else None
    r = lib().e3d_normals_knn(_ptr(xyz, np.float32, keep), n, int(k), C.c_void_p(vp.ctypes.data),
                              C.c_void_p(on.ctypes.data), C.c_void_p(oc.ctypes.data),
                              C.c_void_p(knn.ctypes.data) if knn is not None else None)
    if r < 0:
        _err("e3d_normals_knn", r)
    return (on, oc, knn) if return_knn else (on, oc)


def normals_radius(xyz, radius, viewpoint=(0.0, 0.0, 0.0), return_counts=False):
    """NormalEstimationTwoPassOMP with setRadiusSearch(radius): (normals[n,3], curvature[n][, neighbour counts[n]])."""
    keep = []
    n = int(xyz.shape[0])
    vp = np.ascontiguousarray(viewpoint, np.float32)
    on = np.zeros((n, 3), np.float32)
    oc = np.zeros(n, np.float32)
    cnt = np.zeros(n, np.int32) if return_counts else None
    r = lib().e3d_normals_radius(_ptr(xyz, np.float32, keep), n, float(radius), C.c_void_p(vp.ctypes.data), C.c_void_p(on.ctypes.data),
                                 C.c_void_p(oc.ctypes.data), C.c_void_p(cnt.ctypes.data) if cnt is not None else None)
    if r < 0:
        _err("e3d_normals_radius", r)
    return (on, oc, cnt) if return_counts else (on, oc)


def local_outlier_removal(xyz, mean_k, distance_factor_threshold, negative=False, return_distances=False):
    """pcl::LocalStatisticalOutlierRemoval: bool mask of the points the filter keeps [, first-pass mean neighbour distances]."""
    keep = []
    n = int(xyz.shape[0])
    inl = np.zeros(n, np.uint8)
    md = np.zeros(n, np.float32) if return_distances else None
    r = lib().e3d_local_outlier_removal(_ptr(xyz, np.float32, keep), n, int(mean_k), float(distance_factor_threshold), int(bool(negative)),
                                        C.c_void_p(inl.ctypes.data), C.c_void_p(md.ctypes.data) if md is not None else None)
    if r < 0:
        _err("e3d_local_outlier_removal", r)
    return (inl.astype(bool), md) if return_distances else inl.astype(bool)


# ---- (B) image registration kernels ------------------------------------------------------------------------------------
CAMERA_PINHOLE, CAMERA_OPENCV, CAMERA_THIN_PRISM_FISHEYE, CAMERA_OPENCV_FISHEYE, CAMERA_FOV = 0, 1, 2, 3, 4


def determine_point_neighbors(xyz, neighbor_count, candidate_count, scan_indices=None, scan_count=1):
    """Problem::DeterminePointNeighbors -> (n, neighbor_count) uint32."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    out = np.zeros((xyz.shape[0], neighbor_count), np.uint32)
    si = np.ascontiguousarray(scan_indices, np.uint8) if scan_indices is not None else None
    r = lib().e3d_determine_point_neighbors(C.c_void_p(xyz.ctypes.data), xyz.shape[0], C.c_void_p(si.ctypes.data) if si is not None else None,
                                            scan_count, 1 if si is not None else 0, neighbor_count, candidate_count, C.c_void_p(out.ctypes.data))
    if r < 0:
        _err("e3d_determine_point_neighbors", r)
    return out


def merge_close_points(merge_distance, num_scans, xyz, colors, scan_indices, max_radius):
    """MergeClosePoints -> (xyz, colours, scan indices, max_radius) of the merged points, in centre order."""
    xyz = np.ascontiguousarray(xyz, np.float32); colors = np.ascontiguousarray(colors, np.float32)
    scan_indices = np.ascontiguousarray(scan_indices, np.uint8); max_radius = np.ascontiguousarray(max_radius, np.float32)
    n = xyz.shape[0]
    ox = np.zeros((max(n, 1), 3), np.float32); oc = np.zeros(max(n, 1), np.float32); osc = np.zeros(max(n, 1), np.uint8); om = np.zeros(max(n, 1), np.float32)
    m = lib().e3d_merge_close_points(float(merge_distance), int(num_scans), *[C.c_void_p(a.ctypes.data) for a in (xyz, colors, scan_indices, max_radius)], n,
                                     *[C.c_void_p(a.ctypes.data) for a in (ox, oc, osc, om)])
    if m < 0:
        _err("e3d_merge_close_points", m)
    return ox[:m].copy(), oc[:m].copy(), osc[:m].copy(), om[:m].copy()


class RegParams(C.Structure):
    """e3d_reg_params -- the opt::Parameters fields the device code reads (src/opt/parameters.h:40-68)."""
    _fields_ = [("point_neighbor_count", C.c_int32), ("robust_weighting_type", C.c_int32),
                ("robust_weighting_parameter", C.c_float), ("fixed_residuals_weight", C.c_float),
                ("variable_residuals_weight", C.c_float), ("maximum_valid_intensity", C.c_float),
                ("occlusion_depth_threshold", C.c_float), ("splat_radius", C.c_float),
                ("current_image_scale", C.c_int32), ("image_scale_count", C.c_int32),
                ("depth_residuals_weight", C.c_float), ("depth_robust_weighting_type", C.c_int32),
                ("depth_robust_weighting_parameter", C.c_float)]


def default_reg_params(**kw):
    p = RegParams(5, 1, float(np.float32(30 * np.sqrt(5) / np.sqrt(2))), 1.0, 1.0, 252.0, 0.01, 0.03, 0, 2, 0.0, 2, 0.02)     # parameters.h:40-68
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class RegProblem:
    """Device-resident mirror of the parts of opt::Problem the ImageRegistrator hot loops read, plus the kernel-level
    operators (render depth / observe / accumulate / cost / colour update).  One method per C-ABI entry point."""

    def __init__(self, params=None):
        self.params = params if params is not None else default_reg_params()
        self._h = lib().e3d_reg_create(C.byref(self.params))
        if not self._h:
            _err("e3d_reg_create")
        self._levels = {}
        self._nparams = {}
        self._dependent = set()
        self._image_intr = {}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().e3d_reg_destroy(h)
            self._h = None

    def _chk(self, r, what):
        if r < 0:
            _err(what, r)
        return r

    def set_params(self, params):
        self.params = params
        self._chk(lib().e3d_reg_set_params(self._h, C.byref(params)), "e3d_reg_set_params")

    def set_point_scale(self, scale, xyz, radius, neighbor_indices, fixed_descriptors=None):
        keep = []
        xyz = np.ascontiguousarray(xyz, np.float32)
        nbr = np.ascontiguousarray(neighbor_indices, np.uint32)
        fd = np.ascontiguousarray(fixed_descriptors, np.float32) if fixed_descriptors is not None else None
        self._chk(lib().e3d_reg_set_point_scale(self._h, scale, C.c_void_p(xyz.ctypes.data), xyz.shape[0], float(radius),
                                                C.c_void_p(nbr.ctypes.data), C.c_void_p(fd.ctypes.data) if fd is not None else None),
                  "e3d_reg_set_point_scale")

    def set_variable_descriptors(self, scale, descriptors, counts):
        d = np.ascontiguousarray(descriptors, np.float32); c = np.ascontiguousarray(counts, np.int32)
        self._chk(lib().e3d_reg_set_variable_descriptors(self._h, scale, C.c_void_p(d.ctypes.data), C.c_void_p(c.ctypes.data)), "set_variable_descriptors")

    def get_variable_descriptors(self, scale, n):
        d = np.zeros((n, self.params.point_neighbor_count), np.float32); c = np.zeros(n, np.int32)
        self._chk(lib().e3d_reg_get_variable_descriptors(self._h, scale, C.c_void_p(d.ctypes.data), C.c_void_p(c.ctypes.data)), "get_variable_descriptors")
        return d, c

    def set_intrinsics(self, intrinsics_id, width, height, parameters, min_image_scale, n_levels, camera_type=0):
        p = np.ascontiguousarray(parameters, np.float32)
        self._chk(lib().e3d_reg_set_intrinsics(self._h, intrinsics_id, camera_type, width, height, C.c_void_p(p.ctypes.data), len(p),
                                               min_image_scale, n_levels), "e3d_reg_set_intrinsics")
        self._levels[intrinsics_id] = n_levels
        self._nparams[intrinsics_id] = len(p)

    def intrinsics_level(self, intrinsics_id, level):
        w = C.c_int(); h = C.c_int(); p = np.zeros(self._nparams[intrinsics_id], np.float32); c = C.c_float()
        self._chk(lib().e3d_reg_get_intrinsics_level(self._h, intrinsics_id, level, C.byref(w), C.byref(h), C.c_void_p(p.ctypes.data), C.byref(c)), "get_intrinsics_level")
        return w.value, h.value, p, c.value

    def set_camera_mask(self, intrinsics_id, masks):
        """Intrinsics::camera_mask: one u8 mask per pyramid level (or None per level); masks=None removes it."""
        if masks is None:
            self._chk(lib().e3d_reg_set_camera_mask(self._h, intrinsics_id, None), "e3d_reg_set_camera_mask")
            return
        keep = [np.ascontiguousarray(m, np.uint8) if m is not None else None for m in masks]
        arr = (C.c_void_p * len(keep))(*[(m.ctypes.data if m is not None else None) for m in keep])
        self._chk(lib().e3d_reg_set_camera_mask(self._h, intrinsics_id, arr), "e3d_reg_set_camera_mask")

    def set_depth_maps(self, image_id, levels):
        """Problem::SetFixedDepthMaps for one image: one f32 map per pyramid level (None removes them)."""
        if levels is None:
            self._chk(lib().e3d_reg_set_depth_maps(self._h, image_id, None), "e3d_reg_set_depth_maps")
            return
        keep = [np.ascontiguousarray(l, np.float32) for l in levels]
        arr = (C.c_void_p * len(keep))(*[l.ctypes.data for l in keep])
        self._chk(lib().e3d_reg_set_depth_maps(self._h, image_id, arr), "e3d_reg_set_depth_maps")

    def depth_accumulate(self, image_id, point_scale):
        """(H, b, sum of robust residuals, count) of the depth residuals of one (image, point scale); V = I + 6."""
        V = self._nparams[self._image_intr[image_id]] + 6
        H = np.zeros((V, V), np.float64); b = np.zeros(V, np.float64)
        sm = C.c_double(0); cn = C.c_int64(0)
        self._chk(lib().e3d_reg_depth_accumulate(self._h, image_id, point_scale, C.c_void_p(H.ctypes.data), C.c_void_p(b.ctypes.data),
                                                 C.byref(sm), C.byref(cn)), "e3d_reg_depth_accumulate")
        return H, b, sm.value, cn.value

    def depth_cost(self, image_id, point_scale):
        sm = C.c_double(0); cn = C.c_int64(0)
        self._chk(lib().e3d_reg_depth_cost(self._h, image_id, point_scale, C.byref(sm), C.byref(cn)), "e3d_reg_depth_cost")
        return sm.value, cn.value

    def set_image(self, image_id, intrinsics_id, levels, masks=None):
        if levels is None:                       # an image owned by another rank: id, intrinsics and pose only
            self._chk(lib().e3d_reg_set_image(self._h, image_id, intrinsics_id, None, None), "e3d_reg_set_image")
            self._image_intr[image_id] = intrinsics_id
            return
        keep = [np.ascontiguousarray(l, np.uint8) for l in levels]
        arr = (C.c_void_p * len(keep))(*[l.ctypes.data for l in keep])
        marr = None
        if masks is not None:
            keepm = [np.ascontiguousarray(m, np.uint8) if m is not None else None for m in masks]
            marr = (C.c_void_p * len(keepm))(*[(m.ctypes.data if m is not None else None) for m in keepm])
        self._chk(lib().e3d_reg_set_image(self._h, image_id, intrinsics_id, arr, marr), "e3d_reg_set_image")
        self._image_intr[image_id] = intrinsics_id

    def set_image_pose(self, image_id, q, t):
        """image_T_global as unit quaternion (w, x, y, z) + translation."""
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        assert q.shape == (4,) and t.shape == (3,)
        self._chk(lib().e3d_reg_set_image_pose(self._h, image_id, C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data)), "e3d_reg_set_image_pose")

    def get_image_pose(self, image_id):
        q = np.zeros(4, np.float32); t = np.zeros(3, np.float32)
        self._chk(lib().e3d_reg_get_image_pose(self._h, image_id, C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data)), "e3d_reg_get_image_pose")
        return q, t

    def update_observations(self, border_size=1):
        self._chk(lib().e3d_reg_update_observations(self._h, border_size), "e3d_reg_update_observations")

    # ---- GroundTruthCreator (f4) -------------------------------------------------------------------------------------
    def set_scan_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32)
        self._n_scan = xyz.shape[0]
        self._chk(lib().e3d_reg_set_scan_points(self._h, C.c_void_p(xyz.ctypes.data), xyz.shape[0]), "e3d_reg_set_scan_points")

    def count_scan_observations(self, image_id, mask=None, excluded_flag=2):
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        self._chk(lib().e3d_reg_count_scan_observations(self._h, image_id, C.c_void_p(m.ctypes.data) if m is not None else None, excluded_flag),
                  "e3d_reg_count_scan_observations")

    def scan_observation_counts(self):
        out = np.zeros(self._n_scan, np.int32)
        self._chk(lib().e3d_reg_get_scan_observation_counts(self._h, C.c_void_p(out.ctypes.data)), "e3d_reg_get_scan_observation_counts")
        return out

    def set_scan_observation_counts(self, counts):
        counts = np.ascontiguousarray(counts, np.int32)
        assert counts.shape[0] == self._n_scan
        self._chk(lib().e3d_reg_set_scan_observation_counts(self._h, C.c_void_p(counts.ctypes.data)), "e3d_reg_set_scan_observation_counts")

    def ground_truth_depth(self, image_id, width, height, mask=None, excluded_flag=2, min_count=2):
        """-> (gt_depth, occlusion_depth), both (height, width) float32."""
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        gt = np.zeros((height, width), np.float32); occ = np.zeros((height, width), np.float32)
        self._chk(lib().e3d_reg_ground_truth_depth(self._h, image_id, C.c_void_p(m.ctypes.data) if m is not None else None, excluded_flag,
                                                   min_count, C.c_void_p(gt.ctypes.data), C.c_void_p(occ.ctypes.data)), "e3d_reg_ground_truth_depth")
        return gt, occ

    def scan_rendering(self, image_id, width, height, point_radius, mask=None, excluded_flag=2, min_count=2):
        """-> (height, width) uint32: index + 1 of the last scan point whose square covers the pixel, 0 where none does."""
        win = np.zeros((height, width), np.uint32)
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        self._chk(lib().e3d_reg_scan_rendering(self._h, image_id, C.c_void_p(m.ctypes.data) if m is not None else None, excluded_flag,
                                               min_count, point_radius, C.c_void_p(win.ctypes.data)), "e3d_reg_scan_rendering")
        return win

    def set_cache_observations(self, enabled):
        """Optimizer::set_cache_observations: update_observations re-projects the cached point index lists."""
        self._chk(lib().e3d_reg_set_cache_observations(self._h, int(bool(enabled))), "e3d_reg_set_cache_observations")

    def determine_observed_indices(self):
        """ObservationsCache::DetermineAndSaveObservedPointIndices without the files (full visibility pass at image scale 0)."""
        self._chk(lib().e3d_reg_determine_observed_indices(self._h), "e3d_reg_determine_observed_indices")

    def get_observed_indices(self, image_id, point_scale):
        n = lib().e3d_reg_get_observed_indices(self._h, image_id, point_scale, None)
        if n < 0:
            self._chk(-1, "e3d_reg_get_observed_indices")
        out = np.zeros(n, np.uint64)
        if n:
            self._chk(int(min(0, lib().e3d_reg_get_observed_indices(self._h, image_id, point_scale, C.c_void_p(out.ctypes.data)))),
                      "e3d_reg_get_observed_indices")
        return out

    def set_observed_indices(self, image_id, point_scale, indices):
        indices = np.ascontiguousarray(indices, np.uint64)
        self._chk(lib().e3d_reg_set_observed_indices(self._h, image_id, point_scale, C.c_void_p(indices.ctypes.data), indices.size),
                  "e3d_reg_set_observed_indices")

    def color_update(self):
        self._chk(lib().e3d_reg_color_update(self._h), "e3d_reg_color_update")

    def compute_cost(self):
        c = C.c_double()
        self._chk(lib().e3d_reg_compute_cost(self._h, C.byref(c)), "e3d_reg_compute_cost")
        return c.value

    def apply(self, lam, print_progress=False):
        """IntrinsicsAndPoseOptimizer::Apply -> (applied_update, lambda, max_change)."""
        a = C.c_int(); l = C.c_float(lam); m = C.c_float()
        self._chk(lib().e3d_reg_apply(self._h, int(print_progress), C.byref(a), C.byref(l), C.byref(m)), "e3d_reg_apply")
        return bool(a.value), l.value, m.value

    def kernel_times(self, reset=True):
        """(pass 1 ms, pass 2 ms, observations, calls) of the accumulate kernels since the last reset (HIP events)."""
        out = (C.c_double * 4)()
        self._chk(lib().e3d_reg_kernel_times(self._h, out, int(bool(reset))), "e3d_reg_kernel_times")
        return tuple(out)

    def profile(self, enable):
        """Switches the phase profile of run_on_current_scale on / off; returns {phase: ms} recorded so far (e3d_reg_profile)."""
        buf = C.create_string_buffer(16384)
        self._chk(lib().e3d_reg_profile(self._h, int(bool(enable)), buf, len(buf)), "e3d_reg_profile")
        out = {}
        self.kernel_groups = {}      # {group: (HIP-event ms, launches, work units)} of the same record ("k:" entries)
        for item in buf.value.decode().split(";"):
            if "=" in item:
                k, v = item.rsplit("=", 1)
                if k.startswith("k:"):
                    ms, calls, units = v.split(",")
                    self.kernel_groups[k[2:]] = (float(ms), int(calls), float(units))
                else:
                    out[k] = float(v)
        return out

    def set_comm(self, comm):
        """Image sharding with the library's own RCCL communicator (a Comm); call before the images are set."""
        self._comm = comm
        self._chk(lib().e3d_reg_set_comm(self._h, comm.handle if comm is not None else None), "e3d_reg_set_comm")

    def set_shard(self, rank, world_size, allreduce=None, allreduce_device=None):
        """Image sharding over ranks (image id mod world_size).  allreduce(np.ndarray float64, in place);
        allreduce_device(ptr, count, dtype) sums a DEVICE buffer in place (dtype 0 = f32, 1 = i32) -- see dist.py.
        Must be called before the images are set."""
        def _wrap_host(buf, count, _user):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:  # noqa: BLE001 -- must not propagate through C
                import traceback
                traceback.print_exc()
                return 1

        def _wrap_dev(ptr, count, dtype, _user):
            try:
                allreduce_device(ptr, count, dtype)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1
        self._cb_host = ALLREDUCE_FN(_wrap_host) if allreduce is not None else ALLREDUCE_FN()
        self._cb_dev = ALLREDUCE_DEVICE_FN(_wrap_dev) if allreduce_device is not None else ALLREDUCE_DEVICE_FN()
        self._chk(lib().e3d_reg_set_shard(self._h, int(rank), int(world_size), self._cb_host, self._cb_dev, None), "e3d_reg_set_shard")
        self._rank, self._world = int(rank), int(world_size)

    def image_owner(self, image_id):
        return self._chk(lib().e3d_reg_image_owner(self._h, int(image_id)), "e3d_reg_image_owner")

    def run_on_current_scale(self, max_num_iterations, max_change_convergence_threshold=0.0,
                             iterations_without_new_optimum_threshold=15, print_progress=False):
        """Optimizer::RunOnCurrentScale -> (converged, optimum_cost, iterations)."""
        c = C.c_double(); it = C.c_int()
        r = self._chk(lib().e3d_reg_run_on_current_scale(self._h, max_num_iterations, max_change_convergence_threshold,
                                                         iterations_without_new_optimum_threshold, int(print_progress),
                                                         C.byref(c), C.byref(it)), "e3d_reg_run_on_current_scale")
        return bool(r), c.value, it.value

    def add_occlusion_mesh(self, vertices, triangles, compute_edges=True):
        v = np.ascontiguousarray(vertices, np.float32); t = np.ascontiguousarray(triangles, np.uint32)
        return self._chk(lib().e3d_reg_add_occlusion_mesh(self._h, C.c_void_p(v.ctypes.data), v.shape[0], C.c_void_p(t.ctypes.data), t.shape[0],
                                                           1 if compute_edges else 0), "e3d_reg_add_occlusion_mesh")

    def clear_occlusion_meshes(self):
        self._chk(lib().e3d_reg_clear_occlusion_meshes(self._h), "e3d_reg_clear_occlusion_meshes")

    def set_occlusion_options(self, min_depth=0.05, max_depth=100.0, mask_occlusion_boundaries=True):
        self._chk(lib().e3d_reg_set_occlusion_options(self._h, min_depth, max_depth, 1 if mask_occlusion_boundaries else 0), "e3d_reg_set_occlusion_options")

    def occlusion_edge_count(self, mesh_index):
        return self._chk(lib().e3d_reg_occlusion_edge_count(self._h, mesh_index), "e3d_reg_occlusion_edge_count")

    def set_splat_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32)
        self._chk(lib().e3d_reg_set_splat_points(self._h, C.c_void_p(xyz.ctypes.data), xyz.shape[0]), "e3d_reg_set_splat_points")

    def render_depth(self, image_id, image_scale, shape=None):
        out = np.zeros(shape, np.float32) if shape is not None else None
        self._chk(lib().e3d_reg_render_depth(self._h, image_id, image_scale, C.c_void_p(out.ctypes.data) if out is not None else None), "e3d_reg_render_depth")
        return out

    def observe(self, image_id, point_scale, image_scale, border_size, indices=None):
        if indices is None:
            n = lib().e3d_reg_observe(self._h, image_id, point_scale, image_scale, border_size, None, 0)
        else:
            idx = np.ascontiguousarray(indices, np.uint32)
            n = lib().e3d_reg_observe(self._h, image_id, point_scale, image_scale, border_size, C.c_void_p(idx.ctypes.data), len(idx))
        return int(self._chk(n, "e3d_reg_observe"))

    def get_observations(self, image_id, point_scale, n):
        idx = np.zeros(n, np.uint32); x = np.zeros(n, np.float32); y = np.zeros(n, np.float32); s = np.zeros(n, np.float32); f = np.zeros(n, np.uint8)
        self._chk(lib().e3d_reg_get_observations(self._h, image_id, point_scale, *[C.c_void_p(a.ctypes.data) for a in (idx, x, y, s, f)]), "get_observations")
        return idx, x, y, s, f

    def set_observations(self, image_id, point_scale, idx, x, y, s):
        idx = np.ascontiguousarray(idx, np.uint32); x, y, s = [np.ascontiguousarray(a, np.float32) for a in (x, y, s)]
        self._chk(lib().e3d_reg_set_observations(self._h, image_id, point_scale, len(idx), *[C.c_void_p(a.ctypes.data) for a in (idx, x, y, s)]), "set_observations")

    def point_radius_minmax(self, xyz):
        """ComputeMinMaxPointRadius over the images of this problem -> (min_radius, max_radius)."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        mn = np.zeros(xyz.shape[0], np.float32); mx = np.zeros(xyz.shape[0], np.float32)
        self._chk(lib().e3d_reg_point_radius_minmax(self._h, C.c_void_p(xyz.ctypes.data), xyz.shape[0], C.c_void_p(mn.ctypes.data),
                                                    C.c_void_p(mx.ctypes.data)), "e3d_reg_point_radius_minmax")
        return mn, mx

    def set_rig(self, rig_id, image_T_rig):
        """image_T_rig: list of (q wxyz, t) per camera, camera 0 = reference."""
        q = np.ascontiguousarray([p[0] for p in image_T_rig], np.float32); t = np.ascontiguousarray([p[1] for p in image_T_rig], np.float32)
        self._chk(lib().e3d_reg_set_rig(self._h, rig_id, len(image_T_rig), C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data)), "e3d_reg_set_rig")

    def get_rig(self, rig_id, camera):
        q = np.zeros(4, np.float32); t = np.zeros(3, np.float32)
        self._chk(lib().e3d_reg_get_rig(self._h, rig_id, camera, C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data)), "e3d_reg_get_rig")
        return q, t

    def add_rig_images(self, rig_id, image_ids):
        ids = np.ascontiguousarray(image_ids, np.int32)
        self._chk(lib().e3d_reg_add_rig_images(self._h, rig_id, C.c_void_p(ids.ctypes.data), len(ids)), "e3d_reg_add_rig_images")
        for i in image_ids[1:]:
            self._dependent.add(int(i))

    def param_count(self, image_id):
        return self._nparams[self._image_intr[image_id]]

    def local_unknowns(self, image_id):
        return self.param_count(image_id) + (12 if image_id in self._dependent else 6)

    def pass1(self, image_id, point_scale, n):
        I = np.zeros(n, np.float32); ji = np.zeros((n, self.param_count(image_id)), np.float32); jp = np.zeros((n, 6), np.float32)
        self._chk(lib().e3d_reg_pass1(self._h, image_id, point_scale, *[C.c_void_p(a.ctypes.data) for a in (I, ji, jp)]), "e3d_reg_pass1")
        return I, ji, jp

    def accumulate(self, image_id, point_scale):
        V = self.local_unknowns(image_id)
        H = np.zeros((V, V)); b = np.zeros(V); sums = np.zeros(2); counts = np.zeros(2, np.int64)
        self._chk(lib().e3d_reg_accumulate(self._h, image_id, point_scale, *[C.c_void_p(a.ctypes.data) for a in (H, b, sums, counts)]), "e3d_reg_accumulate")
        return H, b, sums, counts

    def cost(self, image_id, point_scale):
        sums = np.zeros(2); counts = np.zeros(2, np.int64)
        self._chk(lib().e3d_reg_cost(self._h, image_id, point_scale, C.c_void_p(sums.ctypes.data), C.c_void_p(counts.ctypes.data)), "e3d_reg_cost")
        return sums, counts

    def color_begin(self, point_scale):
        self._chk(lib().e3d_reg_color_begin(self._h, point_scale), "e3d_reg_color_begin")

    def color_accumulate(self, image_id, point_scale):
        self._chk(lib().e3d_reg_color_accumulate(self._h, image_id, point_scale), "e3d_reg_color_accumulate")

    def color_finish(self, point_scale):
        self._chk(lib().e3d_reg_color_finish(self._h, point_scale), "e3d_reg_color_finish")
