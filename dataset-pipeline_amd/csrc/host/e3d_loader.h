// e3d_loader.h -- thin run-time loader of the C-ABI (include/e3d_hip.h) for the C++ host tools: dlopen()s
// libe3dhip.so and resolves the entry points.  There is no fallback: if the library or a symbol is missing the
// tools stop with an error message.
#pragma once

#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../../include/e3d_hip.h"

namespace e3d_host {

struct Api {
  void* handle = nullptr;
  int device_count = 0;
#define E3D_FN(name) decltype(&::name) name = nullptr;
  E3D_FN(e3d_abi_version) E3D_FN(e3d_init) E3D_FN(e3d_last_error) E3D_FN(e3d_icp_create) E3D_FN(e3d_icp_destroy)
  E3D_FN(e3d_icp_add_cloud) E3D_FN(e3d_icp_run) E3D_FN(e3d_icp_get_pose) E3D_FN(e3d_transform_cloud)
  E3D_FN(e3d_normals_knn) E3D_FN(e3d_normals_radius) E3D_FN(e3d_find_correspondences)
  E3D_FN(e3d_reg_create) E3D_FN(e3d_reg_destroy) E3D_FN(e3d_reg_set_params) E3D_FN(e3d_reg_set_point_scale) E3D_FN(e3d_reg_set_intrinsics) E3D_FN(e3d_reg_set_camera_mask) E3D_FN(e3d_reg_get_intrinsics_level) E3D_FN(e3d_reg_set_image) E3D_FN(e3d_reg_set_image_pose) E3D_FN(e3d_reg_get_image_pose) E3D_FN(e3d_reg_set_rig) E3D_FN(e3d_reg_get_rig) E3D_FN(e3d_reg_add_rig_images) E3D_FN(e3d_reg_set_splat_points) E3D_FN(e3d_reg_run_on_current_scale) E3D_FN(e3d_reg_compute_cost) E3D_FN(e3d_determine_point_neighbors) E3D_FN(e3d_reg_point_radius_minmax) E3D_FN(e3d_merge_close_points) E3D_FN(e3d_reg_add_occlusion_mesh) E3D_FN(e3d_reg_set_occlusion_options) E3D_FN(e3d_reg_set_cache_observations) E3D_FN(e3d_reg_determine_observed_indices) E3D_FN(e3d_reg_get_observed_indices) E3D_FN(e3d_reg_set_observed_indices) E3D_FN(e3d_local_outlier_removal) E3D_FN(e3d_reg_set_scan_points) E3D_FN(e3d_reg_count_scan_observations) E3D_FN(e3d_reg_get_scan_observation_counts) E3D_FN(e3d_reg_ground_truth_depth) E3D_FN(e3d_reg_scan_rendering) E3D_FN(e3d_comm_create_all) E3D_FN(e3d_comm_destroy) E3D_FN(e3d_comm_abort) E3D_FN(e3d_icp_set_comm) E3D_FN(e3d_reg_set_comm)
#undef E3D_FN
};

inline std::string exe_dir() {
  char buf[4096];
  const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
  if (n <= 0) return ".";
  buf[n] = 0;
  return std::string(dirname(buf));
}

inline Api& api() {
  static Api a;
  if (a.handle) return a;
  std::string tried;
  const char* env = getenv("E3D_HIP_LIBRARY");
  const std::string candidates[] = {env ? std::string(env) : std::string(), exe_dir() + "/../lib/libe3dhip.so",
                                    exe_dir() + "/libe3dhip.so", "libe3dhip.so"};
  for (const std::string& c : candidates) {
    if (c.empty()) continue;
    a.handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (a.handle) break;
    tried += "\n  " + c + ": " + dlerror();
  }
  if (!a.handle) {
    fprintf(stderr, "FATAL: cannot load the HIP library libe3dhip.so (set E3D_HIP_LIBRARY); tried:%s\n", tried.c_str());
    exit(EXIT_FAILURE);
  }
#define E3D_LOAD(name)                                                             \
  a.name = reinterpret_cast<decltype(a.name)>(dlsym(a.handle, #name));             \
  if (!a.name) { fprintf(stderr, "FATAL: libe3dhip.so lacks symbol %s\n", #name); exit(EXIT_FAILURE); }
  E3D_LOAD(e3d_abi_version) E3D_LOAD(e3d_init) E3D_LOAD(e3d_last_error) E3D_LOAD(e3d_icp_create)
  E3D_LOAD(e3d_icp_destroy) E3D_LOAD(e3d_icp_add_cloud) E3D_LOAD(e3d_icp_run) E3D_LOAD(e3d_icp_get_pose)
  E3D_LOAD(e3d_transform_cloud) E3D_LOAD(e3d_normals_knn) E3D_LOAD(e3d_normals_radius) E3D_LOAD(e3d_find_correspondences)
  E3D_LOAD(e3d_reg_create) E3D_LOAD(e3d_reg_destroy) E3D_LOAD(e3d_reg_set_params) E3D_LOAD(e3d_reg_set_point_scale) E3D_LOAD(e3d_reg_set_intrinsics) E3D_LOAD(e3d_reg_set_camera_mask) E3D_LOAD(e3d_reg_get_intrinsics_level) E3D_LOAD(e3d_reg_set_image) E3D_LOAD(e3d_reg_set_image_pose) E3D_LOAD(e3d_reg_get_image_pose) E3D_LOAD(e3d_reg_set_rig) E3D_LOAD(e3d_reg_get_rig) E3D_LOAD(e3d_reg_add_rig_images) E3D_LOAD(e3d_reg_set_splat_points) E3D_LOAD(e3d_reg_run_on_current_scale) E3D_LOAD(e3d_reg_compute_cost) E3D_LOAD(e3d_determine_point_neighbors) E3D_LOAD(e3d_reg_point_radius_minmax) E3D_LOAD(e3d_merge_close_points) E3D_LOAD(e3d_reg_add_occlusion_mesh) E3D_LOAD(e3d_reg_set_occlusion_options) E3D_LOAD(e3d_reg_set_cache_observations) E3D_LOAD(e3d_reg_determine_observed_indices) E3D_LOAD(e3d_reg_get_observed_indices) E3D_LOAD(e3d_reg_set_observed_indices) E3D_LOAD(e3d_local_outlier_removal) E3D_LOAD(e3d_reg_set_scan_points) E3D_LOAD(e3d_reg_count_scan_observations) E3D_LOAD(e3d_reg_get_scan_observation_counts) E3D_LOAD(e3d_reg_ground_truth_depth) E3D_LOAD(e3d_reg_scan_rendering) E3D_LOAD(e3d_comm_create_all) E3D_LOAD(e3d_comm_destroy) E3D_LOAD(e3d_comm_abort) E3D_LOAD(e3d_icp_set_comm) E3D_LOAD(e3d_reg_set_comm)
#undef E3D_LOAD
  if (a.e3d_abi_version() != E3D_ABI_VERSION) {
    fprintf(stderr, "FATAL: libe3dhip.so ABI version %d, expected %d\n", a.e3d_abi_version(), E3D_ABI_VERSION);
    exit(EXIT_FAILURE);
  }
  const char* dev = getenv("E3D_DEVICE");
  a.device_count = a.e3d_init(dev ? atoi(dev) : 0);
  if (a.device_count < 1) {
    fprintf(stderr, "FATAL: %s\n", a.e3d_last_error());
    exit(EXIT_FAILURE);
  }
  return a;
}

// --gpus N of the tools: one host thread per GPU inside the process, the library's own RCCL communicator between them
// (e3d_comm_create_all).  0 / 1 = single GPU.
inline int& gpu_count_setting() { static int n = 1; return n; }
inline bool set_gpu_count(int n) {
  if (n < 1) n = 1;
  if (n > api().device_count) {
    fprintf(stderr, "--gpus %d: only %d HIP device(s) visible\n", n, api().device_count);
    return false;
  }
  gpu_count_setting() = n;
  return true;
}

}  // namespace e3d_host
