// opt_problem.h -- host-side data model of ImageRegistrator above the C-ABI: the parts of opt::Problem,
// opt::Parameters, opt::Intrinsics, opt::Image, opt::Rig and io::*Colmap* that set the optimisation problem up and write
// its state back.  Every numeric hot loop is in libe3dhip.so; this file is I/O and bookkeeping.
//
//   opt::Parameters                         src/opt/parameters.h:40-239
//   io::InitializeStateFromColmapModel      src/io/colmap_model.cc:788-868
//   opt::AssignRigs                         src/opt/rig.cc:25-271
//   Problem::InitializeImages / LoadImages  src/opt/problem.cc:476-505, Image::LoadImageData src/opt/image.cc:40-72
//   Problem::Load/SaveMultiResPointCloud    src/opt/problem.cc:62-159,364-411
//   Problem::ComputeMultiResPointCloud      src/opt/problem.cc:160-362, CreateMultiScalePointCloud
//                                           src/opt/multi_scale_point_cloud.cc:182-369 (radius ranges and merging on the GPU)
//   fixed descriptors                       src/opt/problem.cc:549-572
//   io::ExportProblemToColmap / ExportRigs  src/io/colmap_model.cc:286-516
#pragma once

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "e3d_loader.h"
#include "io_colmap.h"
#include "io_image.h"
#include "io_ply.h"
#include "io_scans.h"
#include "util.h"

namespace e3d_host {

// ---- opt::Parameters ----------------------------------------------------------------------------------------------------------
struct Parameters {
  int point_neighbor_count = 5;
  int point_neighbor_candidate_count = 25;
  float min_mean_intensity_difference_for_points = 5;
  int robust_weighting_type = 1;                                   // RobustWeighting::Type: 0 none, 1 huber, 2 tukey
  float robust_weighting_parameter = (float)(30 * std::sqrt(5) / std::sqrt(2));
  int max_initial_image_area_in_pixels = 200 * 160;
  float fixed_residuals_weight = 1.f;
  float variable_residuals_weight = 1.f;
  int depth_robust_weighting_type = 2;
  float depth_robust_weighting_parameter = 0.02f;
  float depth_residuals_weight = 0;
  int maximum_valid_intensity = 252;
  int min_occlusion_check_image_scale = 0;
  float occlusion_depth_threshold = 0.01f;
  float min_occlusion_depth = 0.05f;
  float max_occlusion_depth = 100.f;
  float splat_radius = 0.03f;
  float min_radius_bias = 1.05f;
  float merge_distance_factor = 4.0f;

  static bool parse_type(int argc, char** argv, const char* name, int* type) {
    std::string v;
    parse_argument(argc, argv, name, v);
    if (v.empty()) return true;
    if (v == "none") *type = 0; else if (v == "huber") *type = 1; else if (v == "tukey") *type = 2;
    else { std::cerr << "Value of " << name << " parameter not recognized" << std::endl; return false; }
    return true;
  }
  bool SetFromArguments(int argc, char** argv) {
    parse_argument(argc, argv, "--point_neighbor_count", point_neighbor_count);
    parse_argument(argc, argv, "--point_neighbor_candidate_count", point_neighbor_candidate_count);
    parse_argument(argc, argv, "--min_mean_intensity_difference_for_points", min_mean_intensity_difference_for_points);
    if (!parse_type(argc, argv, "--robust_weighting_type", &robust_weighting_type)) return false;
    parse_argument(argc, argv, "--robust_weighting_parameter", robust_weighting_parameter);
    parse_argument(argc, argv, "--max_initial_image_area_in_pixels", max_initial_image_area_in_pixels);
    parse_argument(argc, argv, "--fixed_residuals_weight", fixed_residuals_weight);
    parse_argument(argc, argv, "--variable_residuals_weight", variable_residuals_weight);
    if (!parse_type(argc, argv, "--depth_robust_weighting_type", &depth_robust_weighting_type)) return false;
    parse_argument(argc, argv, "--depth_robust_weighting_parameter", depth_robust_weighting_parameter);
    parse_argument(argc, argv, "--depth_residuals_weight", depth_residuals_weight);
    parse_argument(argc, argv, "--maximum_valid_intensity", maximum_valid_intensity);
    parse_argument(argc, argv, "--min_occlusion_check_image_scale", min_occlusion_check_image_scale);
    parse_argument(argc, argv, "--occlusion_depth_threshold", occlusion_depth_threshold);
    parse_argument(argc, argv, "--max_occlusion_depth", max_occlusion_depth);
    parse_argument(argc, argv, "--min_occlusion_depth", min_occlusion_depth);
    parse_argument(argc, argv, "--splat_radius", splat_radius);
    parse_argument(argc, argv, "--scale_factor", global_scale_factor());
    parse_argument(argc, argv, "--min_radius_bias", min_radius_bias);
    parse_argument(argc, argv, "--merge_distance_factor", merge_distance_factor);
    return true;
  }
  // same lines as the reference, including its missing separators (parameters.h:120-123)
  void OutputValues(std::ostream& s) const {
    s << "point_neighbor_count " << point_neighbor_count << std::endl;
    s << "point_neighbor_candidate_count " << point_neighbor_candidate_count << std::endl;
    s << "min_mean_intensity_difference_for_points " << min_mean_intensity_difference_for_points << std::endl;
    s << "robust_weighting_type " << robust_weighting_type << std::endl;
    s << "robust_weighting_parameter " << robust_weighting_parameter << std::endl;
    s << "max_initial_image_area_in_pixels " << max_initial_image_area_in_pixels << std::endl;
    s << "fixed_residuals_weight " << fixed_residuals_weight << std::endl;
    s << "variable_residuals_weight " << variable_residuals_weight << std::endl;
    s << "depth_robust_weighting_type " << depth_robust_weighting_type << std::endl;
    s << "depth_robust_weighting_parameter " << depth_robust_weighting_parameter << std::endl;
    s << "depth_residuals_weight " << depth_residuals_weight << std::endl;
    s << "maximum_valid_intensity " << maximum_valid_intensity << std::endl;
    s << "min_occlusion_check_image_scale " << min_occlusion_check_image_scale << std::endl;
    s << "occlusion_depth_threshold " << occlusion_depth_threshold << std::endl;
    s << "max_occlusion_depth" << max_occlusion_depth << std::endl;
    s << "min_occlusion_depth" << min_occlusion_depth << std::endl;
    s << "splat_radius" << splat_radius << std::endl;
    s << "scale_factor" << global_scale_factor() << std::endl;
    s << "min_radius_bias " << min_radius_bias << std::endl;
    s << "merge_distance_factor " << merge_distance_factor << std::endl;
  }
};

// ---- small SE3f helpers on (w x y z | t), f32 like Sophus::SE3f; only used in set-up code -----------------------------------
struct Pose7 { float q[4] = {1, 0, 0, 0}; float t[3] = {0, 0, 0}; };

inline void quat_rotate(const float* q, const float* p, float* o) {     // Sophus SO3 action
  const float* v = q + 1;
  float uv[3] = {v[1] * p[2] - v[2] * p[1], v[2] * p[0] - v[0] * p[2], v[0] * p[1] - v[1] * p[0]};
  for (int i = 0; i < 3; ++i) uv[i] += uv[i];
  const float c[3] = {v[1] * uv[2] - v[2] * uv[1], v[2] * uv[0] - v[0] * uv[2], v[0] * uv[1] - v[1] * uv[0]};
  for (int i = 0; i < 3; ++i) o[i] = p[i] + q[0] * uv[i] + c[i];
}
inline Pose7 pose_mul(const Pose7& a, const Pose7& b) {
  Pose7 r;
  r.q[0] = a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2] - a.q[3] * b.q[3];
  r.q[1] = a.q[0] * b.q[1] + a.q[1] * b.q[0] + a.q[2] * b.q[3] - a.q[3] * b.q[2];
  r.q[2] = a.q[0] * b.q[2] + a.q[2] * b.q[0] + a.q[3] * b.q[1] - a.q[1] * b.q[3];
  r.q[3] = a.q[0] * b.q[3] + a.q[3] * b.q[0] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
  const float n = std::sqrt(r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3]);
  for (float& v : r.q) v /= n;
  float rt[3];
  quat_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = rt[i] + a.t[i];
  return r;
}
inline Pose7 pose_inverse(const Pose7& a) {
  Pose7 r;
  r.q[0] = a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = -a.q[3];
  const float nt[3] = {a.t[0] * -1, a.t[1] * -1, a.t[2] * -1};
  quat_rotate(r.q, nt, r.t);
  return r;
}
inline void pose_rotation(const Pose7& a, double* R /*row-major 3x3*/) {
  const double w = a.q[0], x = a.q[1], y = a.q[2], z = a.q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
// U V^T of the SVD of M (the rotation closest to a sum of rotations, rig.cc:148-153,196-199): Newton iteration of the
// polar decomposition, X <- (X + X^-T) / 2
inline void nearest_rotation(const double* M, double* R) {
  double X[9];
  for (int i = 0; i < 9; ++i) X[i] = M[i];
  for (int it = 0; it < 100; ++it) {
    const double det = X[0] * (X[4] * X[8] - X[5] * X[7]) - X[1] * (X[3] * X[8] - X[5] * X[6]) + X[2] * (X[3] * X[7] - X[4] * X[6]);
    // inverse transpose = cofactor matrix / det
    const double C[9] = {X[4] * X[8] - X[5] * X[7], X[5] * X[6] - X[3] * X[8], X[3] * X[7] - X[4] * X[6],
                         X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                         X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
    double change = 0;
    for (int i = 0; i < 9; ++i) { const double n = 0.5 * (X[i] + C[i] / det); change = std::max(change, std::fabs(n - X[i])); X[i] = n; }
    if (change < 1e-15) break;
  }
  for (int i = 0; i < 9; ++i) R[i] = X[i];
}
inline void rotation_to_quat(const double* m, float* q /*w x y z*/) {     // Eigen's matrix -> quaternion conversion
  double tr = m[0] + m[4] + m[8], w, v[3];
  if (tr > 0) {
    tr = std::sqrt(tr + 1.0); w = 0.5 * tr; tr = 0.5 / tr;
    v[0] = (m[7] - m[5]) * tr; v[1] = (m[2] - m[6]) * tr; v[2] = (m[3] - m[1]) * tr;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    tr = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    v[i] = 0.5 * tr; tr = 0.5 / tr;
    w = (m[3 * k + j] - m[3 * j + k]) * tr;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * tr;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * tr;
  }
  q[0] = (float)w; q[1] = (float)v[0]; q[2] = (float)v[1]; q[3] = (float)v[2];
}

inline std::string path_parent(const std::string& p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? std::string() : p.substr(0, s); }
inline std::string path_filename(const std::string& p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? p : p.substr(s + 1); }
inline std::string replace_extension(const std::string& p, const std::string& ext) {
  const size_t slash = p.find_last_of('/'), dot = p.find_last_of('.');
  const std::string stem = (dot == std::string::npos || (slash != std::string::npos && dot < slash)) ? p : p.substr(0, dot);
  return stem + "." + ext;
}
inline bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
inline void create_directories(const std::string& p) {
  std::string cur;
  std::istringstream s(p);
  std::string part;
  if (!p.empty() && p[0] == '/') cur = "/";
  while (std::getline(s, part, '/')) {
    if (part.empty()) continue;
    cur += part + "/";
    mkdir(cur.c_str(), 0777);
  }
}
// util::RelativePath (src/base/util.cc:36-66)
inline std::string relative_path(const std::string& from, const std::string& to) {
  auto split = [](const std::string& p) {
    std::vector<std::string> v;
    if (!p.empty() && p[0] == '/') v.push_back("/");
    std::istringstream s(p); std::string part;
    while (std::getline(s, part, '/')) if (!part.empty()) v.push_back(part);
    return v;
  };
  const std::vector<std::string> a = split(from), b = split(to);
  size_t i = 0;
  while (i < a.size() && i < b.size() && a[i] == b[i]) ++i;
  std::string r;
  for (size_t k = i; k < a.size(); ++k) r += (r.empty() ? "" : "/") + std::string("..");
  for (size_t k = i; k < b.size(); ++k) r += (r.empty() ? "" : "/") + b[k];
  return r;
}

// ---- problem state --------------------------------------------------------------------------------------------------------------
struct HostIntrinsics {
  int intrinsics_id = 0, model = 0, width = 0, height = 0, n_params = 0, min_image_scale = 0;
  float params[12] = {0};
  std::string model_name;
  bool camera_mask_checked = false, camera_mask_uploaded = false;
  std::vector<GrayImage> camera_mask;       // per pyramid level; empty = none
};
struct HostImage {
  int image_id = 0, intrinsics_id = 0, rig_images_id = -1;
  Pose7 image_T_global;
  std::string file_path;
};
struct HostRig { int rig_id = 0; std::vector<std::string> folder_names; std::vector<Pose7> image_T_rig; };
struct HostRigImages { int rig_images_id = 0, rig_id = 0; std::vector<int> image_ids; };

inline int camera_model_from_name(const std::string& name) {
  if (name == "PINHOLE") return E3D_CAMERA_PINHOLE;
  if (name == "OPENCV") return E3D_CAMERA_OPENCV;
  if (name == "THIN_PRISM_FISHEYE") return E3D_CAMERA_THIN_PRISM_FISHEYE;
  if (name == "OPENCV_FISHEYE") return E3D_CAMERA_OPENCV_FISHEYE;
  if (name == "FOV") return E3D_CAMERA_FOV;
  if (name == "SIMPLE_PINHOLE") return E3D_CAMERA_SIMPLE_PINHOLE;
  if (name == "POLYNOMIAL_3") return E3D_CAMERA_POLYNOMIAL_3;
  if (name == "FISHEYE_POLYNOMIAL_2_TANGENTIAL_2") return E3D_CAMERA_FISHEYE_POLYNOMIAL_2_TANGENTIAL_2;
  // [QUIRK] camera_base.cc:73-74: the factory entries of RADIAL_FISHEYE / SIMPLE_RADIAL_FISHEYE construct RadialCamera /
  // SimpleRadialCamera, whose constructors set the types kRadial / kSimpleRadial -- the fisheye names behave like the plain ones
  // and are written back as RADIAL / SIMPLE_RADIAL (camera_model_name below)
  if (name == "RADIAL" || name == "RADIAL_FISHEYE") return E3D_CAMERA_RADIAL;
  if (name == "SIMPLE_RADIAL" || name == "SIMPLE_RADIAL_FISHEYE") return E3D_CAMERA_SIMPLE_RADIAL;
  return -1;
}
inline const char* camera_model_name(int model) {
  static const char* const names[] = {"PINHOLE", "OPENCV", "THIN_PRISM_FISHEYE", "OPENCV_FISHEYE", "FOV", "SIMPLE_PINHOLE", "SIMPLE_RADIAL",
                                      "RADIAL", "POLYNOMIAL_3", "FISHEYE_POLYNOMIAL_2_TANGENTIAL_2"};
  return (model >= 0 && model < 10) ? names[model] : "INVALID";
}
inline bool camera_unique_focal(int model) {
  return model == E3D_CAMERA_SIMPLE_PINHOLE || model == E3D_CAMERA_SIMPLE_RADIAL || model == E3D_CAMERA_RADIAL;
}
inline int camera_param_count(int model) {
  static const int counts[] = {4, 8, 12, 8, 5, 3, 4, 5, 7, 8};
  return (model >= 0 && model < 10) ? counts[model] : 0;
}

class Problem {
 public:
  Parameters prm;
  std::vector<HostIntrinsics> intrinsics_list;
  std::map<int, HostImage> images;              // ascending image id (the reference iterates an unordered_map)
  std::vector<HostRig> rigs;
  std::vector<HostRigImages> rig_images;
  int image_scale_count = 1;
  std::vector<float> point_radii;
  std::vector<std::vector<float>> points;       // per point scale: n x 3
  std::vector<std::vector<float>> colors;       // multi_res_colors
  std::vector<std::vector<uint32_t>> neighbors; // neighbor_point_indices_ (n * K)
  e3d_reg_t* reg = nullptr;                     // rank 0's device problem (the only one without --gpus)
  // --gpus N: one device problem per GPU, images sharded (image id mod N), the library's RCCL communicator between them
  std::vector<e3d_reg_t*> regs;
  std::vector<e3d_comm_t*> comms;
  std::string occlusion_mesh_path, occlusion_splats_path;       // empty: splats of the scan points

  ~Problem() {
    for (e3d_reg_t* r : regs) if (r) api().e3d_reg_destroy(r);
    for (e3d_comm_t* c : comms) if (c) api().e3d_comm_destroy(c);
  }
  int world() const { return (int)std::max<size_t>(regs.size(), 1); }
  e3d_reg_t* owner(int image_id) const { const int w = world(); return regs[(size_t)(((image_id % w) + w) % w)]; }
  // the same call on every rank's device problem (state every rank keeps: intrinsics, poses, points, rigs, options)
  template <class F> bool all_regs(F f, const char* what) {
    for (e3d_reg_t* r : regs) if (f(r) < 0) return lib_fail(what);
    return true;
  }
  // a collective step: every rank from its own host thread
  template <class F> bool all_regs_parallel(F f, const char* what) {
    std::vector<int> res(regs.size(), 0);
    std::vector<std::string> err(regs.size());
    std::atomic<int> first_failed{-1};
    auto work = [&](size_t k) {
      res[k] = f(regs[k], (int)k);
      if (res[k] < 0) {   // ranks waiting in a collective for this one would block forever: abort all communicators
        err[k] = api().e3d_last_error();
        int none = -1;
        first_failed.compare_exchange_strong(none, (int)k);
        for (e3d_comm_t* c : comms) if (c) api().e3d_comm_abort(c);
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < regs.size(); ++k) th.emplace_back(work, k);
    work(0);
    for (std::thread& t : th) t.join();
    if (first_failed.load() >= 0) { std::cerr << what << ": " << err[(size_t)first_failed.load()] << std::endl; return false; }
    return true;
  }

  bool fail(const std::string& what) const { std::cerr << what << std::endl; return false; }
  bool lib_fail(const char* what) const { std::cerr << what << ": " << api().e3d_last_error() << std::endl; return false; }

  // io::InitializeStateFromColmapModel
  bool InitializeStateFromColmapModel(const std::string& model_path, const std::string& image_base_path, const std::unordered_set<int>& ignore) {
    std::map<int, ColmapCamera> cams;
    if (!ReadColmapCameras(model_path + "/cameras.txt", &cams)) return fail("Cannot read initial camera intrinsics from " + model_path + "/cameras.txt");
    std::map<int, ColmapImage> ims;
    if (!ReadColmapImages(model_path + "/images.txt", &ims)) return fail("Cannot read initial image poses from " + model_path + "/images.txt");
    std::map<int, int> camera_to_intrinsics;
    for (const auto& kv : cams) {
      const ColmapCamera& c = kv.second;
      if (ignore.count(c.camera_id)) { std::cout << "Ignoring camera id " << c.camera_id << "." << std::endl; continue; }
      HostIntrinsics in;
      in.intrinsics_id = (int)intrinsics_list.size();
      in.model = camera_model_from_name(c.model_name);
      if (in.model < 0) return fail("Unknown camera model " + c.model_name);
      in.model_name = camera_model_name(in.model);
      in.n_params = camera_param_count(in.model);
      if ((int)c.parameters.size() != in.n_params) return fail("Wrong parameter count for camera model " + c.model_name);
      in.width = c.width; in.height = c.height;
      for (int i = 0; i < in.n_params; ++i) in.params[i] = (float)c.parameters[i];
      // ShiftedBy(-0.5, -0.5): COLMAP's pixel-corner origin -> pixel centre; cx cy are parameters 1, 2 of the models with one
      // focal length (camera_base_impl.h:93-106)
      const int ci = camera_unique_focal(in.model) ? 1 : 2;
      in.params[ci] += -0.5f; in.params[ci + 1] += -0.5f;
      camera_to_intrinsics[c.camera_id] = in.intrinsics_id;
      intrinsics_list.push_back(in);
    }
    if (intrinsics_list.empty()) return fail("No cameras defined.");
    for (const auto& kv : ims) {
      const ColmapImage& ci = kv.second;
      if (ignore.count(ci.camera_id)) continue;
      if (!camera_to_intrinsics.count(ci.camera_id)) return fail("Image refers to an unknown camera id");
      HostImage im;
      im.image_id = (int)images.size();                          // Problem::AddImage: next free sequential id
      im.intrinsics_id = camera_to_intrinsics[ci.camera_id];
      for (int i = 0; i < 4; ++i) im.image_T_global.q[i] = ci.q[i];
      for (int i = 0; i < 3; ++i) im.image_T_global.t[i] = ci.t[i];
      im.file_path = (!ci.file_path.empty() && ci.file_path[0] == '/') ? ci.file_path : join_path(image_base_path, ci.file_path);
      images[im.image_id] = im;
    }
    if (images.size() < 2) return fail("Less than 2 images defined.");
    return true;
  }

  // opt::AssignRigs
  bool AssignRigs(const std::vector<ColmapRig>& rig_vector) {
    std::map<std::string, int> prefix_to_rig;
    for (const ColmapRig& r : rig_vector) {
      if (r.cameras.size() == 1) continue;
      HostRig rig;
      rig.rig_id = (int)rigs.size();
      for (const ColmapRigCamera& c : r.cameras) { prefix_to_rig[c.image_prefix] = rig.rig_id; rig.folder_names.push_back(c.image_prefix); }
      rig.image_T_rig.resize(rig.folder_names.size());
      rigs.push_back(rig);
    }
    std::vector<std::map<std::string, int>> filename_to_frame(rigs.size());
    for (auto& kv : images) {
      HostImage& im = kv.second;
      const std::string folder = path_filename(path_parent(im.file_path));
      auto it = prefix_to_rig.find(folder);
      if (it == prefix_to_rig.end()) { im.rig_images_id = -1; continue; }
      HostRig& rig = rigs[it->second];
      const std::string fn = path_filename(im.file_path);
      int fid;
      if (!filename_to_frame[rig.rig_id].count(fn)) {
        HostRigImages f;
        f.rig_images_id = (int)rig_images.size(); f.rig_id = rig.rig_id;
        f.image_ids.assign(rig.folder_names.size(), -1);
        rig_images.push_back(f);
        filename_to_frame[rig.rig_id][fn] = f.rig_images_id;
        fid = f.rig_images_id;
      } else fid = filename_to_frame[rig.rig_id][fn];
      int cam = -1;
      for (size_t i = 0; i < rig.folder_names.size(); ++i) if (rig.folder_names[i] == folder) { cam = (int)i; break; }
      rig_images[fid].image_ids[cam] = im.image_id;
      im.rig_images_id = fid;
    }
    for (HostRig& rig : rigs) {
      const int nc = (int)rig.folder_names.size();
      // folder and intrinsics of each rig camera, from any of its images: needed to add the images missing from a frame
      // (rig.cc:92-125; all images of one rig camera must share folder and intrinsics)
      std::vector<std::string> camera_folder(nc);
      std::vector<int> camera_intrinsics(nc, -1);
      for (const HostRigImages& f : rig_images) {
        if (f.rig_id != rig.rig_id) continue;
        for (int c = 0; c < nc; ++c) {
          if (f.image_ids[c] < 0) continue;
          const HostImage& im = images[f.image_ids[c]];
          const std::string folder = path_parent(im.file_path);
          if (!camera_folder[c].empty() && camera_folder[c] != folder) return fail("Images of one rig camera lie in different folders: " + folder);
          camera_folder[c] = folder;
          if (camera_intrinsics[c] >= 0 && camera_intrinsics[c] != im.intrinsics_id)
            return fail("A camera of a rig has images with different intrinsics IDs, i.e., the input state seems to be wrong. Aborting.");
          camera_intrinsics[c] = im.intrinsics_id;
        }
      }
      // average reference_T_other over all frames: rotations through the nearest rotation of their sum, translations by mean
      std::vector<std::vector<double>> Rsum(nc - 1, std::vector<double>(9, 0.0)), tsum(nc - 1, std::vector<double>(3, 0.0));
      std::vector<int> count(nc - 1, 0);
      for (const HostRigImages& f : rig_images) {
        if (f.rig_id != rig.rig_id || f.image_ids[0] < 0) continue;
        for (int k = 0; k < nc - 1; ++k) {
          if (f.image_ids[k + 1] < 0) continue;
          const Pose7 ref_T_other = pose_mul(images[f.image_ids[0]].image_T_global, pose_inverse(images[f.image_ids[k + 1]].image_T_global));
          double R[9];
          pose_rotation(ref_T_other, R);
          for (int i = 0; i < 9; ++i) Rsum[k][i] += R[i];
          for (int i = 0; i < 3; ++i) tsum[k][i] += ref_T_other.t[i];
          count[k] += 1;
        }
      }
      rig.image_T_rig[0] = Pose7();
      for (int k = 0; k < nc - 1; ++k) {
        if (!count[k]) return fail("A rig camera is never observed together with the reference camera");
        double R[9];
        nearest_rotation(Rsum[k].data(), R);
        Pose7 avg;
        rotation_to_quat(R, avg.q);
        for (int i = 0; i < 3; ++i) avg.t[i] = (float)(tsum[k][i] / count[k]);
        rig.image_T_rig[k + 1] = pose_inverse(avg);
      }
      // every frame: average global_T_rig over its images, then derive all image poses from it
      for (HostRigImages& f : rig_images) {
        if (f.rig_id != rig.rig_id) continue;
        double Rs[9] = {0}, ts[3] = {0};
        int n = 0;
        std::string frame_file_name;
        for (int c = 0; c < nc; ++c) {
          if (f.image_ids[c] < 0) continue;                    // added below, at the pose the rig gives it (rig.cc:236-252)
          frame_file_name = path_filename(images[f.image_ids[c]].file_path);
          const Pose7 est = pose_mul(pose_inverse(images[f.image_ids[c]].image_T_global), rig.image_T_rig[c]);
          double R[9];
          pose_rotation(est, R);
          for (int i = 0; i < 9; ++i) Rs[i] += R[i];
          for (int i = 0; i < 3; ++i) ts[i] += est.t[i];
          ++n;
        }
        double R[9];
        nearest_rotation(Rs, R);
        Pose7 global_T_rig;
        rotation_to_quat(R, global_T_rig.q);
        for (int i = 0; i < 3; ++i) global_T_rig.t[i] = (float)(ts[i] / n);
        for (int c = 0; c < nc; ++c) {
          if (f.image_ids[c] < 0) {
            if (camera_folder[c].empty()) return fail("Attempting to add a missing image to a rig_images, but no image of that camera has been observed.");
            HostImage added;
            added.image_id = (int)images.size();                                     // Problem::AddImage (problem.cc:451-462)
            while (images.count(added.image_id)) ++added.image_id;
            added.intrinsics_id = camera_intrinsics[c];
            added.file_path = join_path(camera_folder[c], frame_file_name);
            added.rig_images_id = f.rig_images_id;
            images[added.image_id] = added;
            f.image_ids[c] = added.image_id;
          }
          images[f.image_ids[c]].image_T_global = pose_inverse(pose_mul(global_T_rig, pose_inverse(rig.image_T_rig[c])));
        }
      }
    }
    size_t assigned = 0;
    for (const auto& kv : images) if (kv.second.rig_images_id >= 0) ++assigned;
    std::cout << "AssignRigs(): assigned " << assigned << " out of " << images.size() << " images to rig(s)" << std::endl;
    return true;
  }

  // Intrinsics::ComputeImageScaleCount (intrinsics.h:82-86)
  int ComputeImageScaleCount(const HostIntrinsics& in) const {
    const int max_pixel_count = in.width * in.height;
    const double area_factor = max_pixel_count * 1.0 / prm.max_initial_image_area_in_pixels;
    return std::max<int>(2, (int)(1 + std::ceil(std::log(area_factor) / std::log(4))));
  }

  bool LoadMultiResPointCloud(const std::string& dir) {
    std::ifstream f(dir + "/metadata.txt");
    if (!f) return false;
    std::string name, version;
    f >> name >> version;
    if (version != "1") return fail("Unsupported multi-res point cloud format version: " + version);
    int candidates = 0, nbr = 0, scales = 0;
    f >> name >> candidates;
    if (name != "neighbor_candidate_count" || candidates != prm.point_neighbor_candidate_count)
      return fail("LoadMultiResPointCloud(): neighbor_candidate_count from file does not fit to the point_neighbor_candidate_count setting. Delete the saved multi-res point cloud to re-generate it.");
    f >> name >> nbr;
    if (name != "neighbor_count" || nbr != prm.point_neighbor_count)
      return fail("LoadMultiResPointCloud(): neighbor_count from file does not fit to the point_neighbor_count setting. Delete the saved multi-res point cloud to re-generate it.");
    f >> name >> scales;
    if (name != "point_scale_count") return fail("LoadMultiResPointCloud(): Reading error.");
    point_radii.resize(scales); points.resize(scales); colors.resize(scales); neighbors.resize(scales);
    for (int i = 0; i < scales; ++i) { f >> name >> point_radii[i]; if (name != "point_radius") return fail("LoadMultiResPointCloud(): Reading error."); }
    for (int s = 0; s < scales; ++s) {
      PointCloud c;
      const std::string p = dir + "/points_of_scale_" + std::to_string(s) + ".ply";
      if (loadPLYFile(p, c) < 0 || c.intensity.size() != c.size()) return fail("LoadMultiResPointCloud(): Cannot read " + p);
      points[s].swap(c.xyz); colors[s].swap(c.intensity);
    }
    FILE* nf = fopen((dir + "/neighbor_point_indices").c_str(), "rb");
    if (!nf) return fail("Cannot open " + dir + "/neighbor_point_indices for reading");
    for (int s = 0; s < scales; ++s) {
      const size_t cnt = (size_t)prm.point_neighbor_count * (points[s].size() / 3);
      std::vector<uint64_t> raw(cnt);                            // std::size_t of the reference build (64 bit)
      if (fread(raw.data(), sizeof(uint64_t), cnt, nf) < cnt) { fclose(nf); return fail("Unexpected EOF in " + dir + "/neighbor_point_indices"); }
      neighbors[s].resize(cnt);
      for (size_t i = 0; i < cnt; ++i) neighbors[s][i] = (uint32_t)raw[i];
    }
    fclose(nf);
    return true;
  }

  // OcclusionGeometry::AddMesh / AddSplats (occlusion_geometry.cc:64-139): a .ply mesh (scaled by the global scale_factor) or a
  // MeshLab project of meshes (each with its own Sim3 pose)
  bool AddOcclusionMesh(const std::string& path, bool compute_edges, const float* left = nullptr) {
    auto add = [&](const std::string& ply, const float* T_mesh) {
      float T[12];
      for (int i = 0; i < 12; ++i) T[i] = T_mesh[i];
      if (left)        // transformation * global_T_mesh (occlusion_geometry.cc:84,113), as a product of the two 3x4 matrices
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c)
            T[4 * r + c] = left[4 * r] * T_mesh[c] + left[4 * r + 1] * T_mesh[4 + c] + left[4 * r + 2] * T_mesh[8 + c] + (c == 3 ? left[4 * r + 3] : 0.f);
      std::cout << "adding mesh " << ply << std::endl;
      std::vector<float> xyz, global;
      std::vector<uint32_t> tris;
      if (loadPLYMesh(ply, xyz, tris) < 0) { std::cerr << "Cannot read file: " << ply << std::endl; return false; }
      global.resize(xyz.size());
      float bmin[3], bmax[3];
      if (api().e3d_transform_cloud(xyz.data(), nullptr, xyz.size() / 3, T, global.data(), nullptr, bmin, bmax) < 0) return lib_fail("e3d_transform_cloud");
      if (compute_edges) std::cout << "computing edges" << std::endl;
      if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_add_occlusion_mesh(r, global.data(), global.size() / 3, tris.data(), tris.size() / 3, compute_edges ? 1 : 0); }, "e3d_reg_add_occlusion_mesh")) return false;
      return true;
    };
    const std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : std::string();
    if (ext == ".mlp") {
      std::vector<MeshInfo> infos;
      if (!ReadMeshLabProject(path, &infos)) return fail("Cannot read mesh poses from " + path);
      for (const MeshInfo& info : infos) {
        float T[12];
        info.global_T_mesh.matrix3x4(T);
        const std::string file = (!info.filename.empty() && info.filename[0] == '/') ? info.filename : join_path(parent_path(path), info.filename);
        if (!add(file, T)) return false;
      }
      std::cout << "Done." << std::endl;
      return true;
    }
    if (ext == ".ply") {
      const float sc = global_scale_factor();                 // Sim3f(Identity * scale): uniform scaling
      const float T[12] = {sc, 0, 0, 0, 0, sc, 0, 0, 0, 0, sc, 0};
      return add(path, T);
    }
    return fail("Mesh file format must be either .mlp or .ply, got " + path);
  }

  // Problem::SaveMultiResPointCloud (modify_colors_for_display = false)
  bool SaveMultiResPointCloud(const std::string& dir) const {
    create_directories(dir);
    std::ofstream f(dir + "/metadata.txt");
    f << "version 1" << std::endl;
    f << "neighbor_candidate_count " << prm.point_neighbor_candidate_count << std::endl;
    f << "neighbor_count " << prm.point_neighbor_count << std::endl;
    f << "point_scale_count " << point_radii.size() << std::endl;
    for (float r : point_radii) f << "point_radius " << r << std::endl;
    f.close();
    for (size_t s = 0; s < points.size(); ++s)
      if (savePLYFileBinaryXYZI(dir + "/points_of_scale_" + std::to_string(s) + ".ply", points[s], colors[s]) < 0) return fail("Cannot write the multi-res point cloud to " + dir);
    FILE* nf = fopen((dir + "/neighbor_point_indices").c_str(), "wb");
    if (!nf) return fail("Cannot open " + dir + "/neighbor_point_indices for writing");
    for (size_t s = 0; s < neighbors.size(); ++s) {
      std::vector<uint64_t> raw(neighbors[s].begin(), neighbors[s].end());
      fwrite(raw.data(), sizeof(uint64_t), raw.size(), nf);
    }
    fclose(nf);
    return true;
  }

  struct ScaleCloud { double radius; std::vector<float> pts, col, max_radius; std::vector<uint8_t> scan; };

  // Problem::ComputeMultiResPointCloud.  scans: global-frame points + RGB per scan (PreprocessScans :182-211).
  bool ComputeMultiResPointCloud(const std::vector<PointCloud::Ptr>& scans) {
    const bool use_fixed = prm.fixed_residuals_weight > 0;
    const int num_scans = (int)scans.size();
    std::cout << "ComputeMultiResPointCloud(): Pre-processing scans ..." << std::endl;
    std::vector<float> pts, col;
    std::vector<uint8_t> scan_idx;
    for (int si = 0; si < num_scans; ++si) {
      const PointCloud& c = *scans[si];
      if (c.rgb.size() != c.xyz.size()) return fail("ComputeMultiResPointCloud(): the scans need colours (red green blue)");
      pts.insert(pts.end(), c.xyz.begin(), c.xyz.end());
      for (size_t i = 0; i < c.size(); ++i) {
        col.push_back((float)(0.299 * c.rgb[3 * i] + 0.587 * c.rgb[3 * i + 1] + 0.114 * c.rgb[3 * i + 2]));
        scan_idx.push_back((uint8_t)si);
      }
    }
    const size_t n = pts.size() / 3;
    std::cout << "ComputeMultiResPointCloud(): Creating multi-res point cloud ..." << std::endl;
    std::vector<float> min_radius(n), max_radius(n);
    if (regs.size() > 1) return fail("the multi-resolution point cloud is built on one GPU (every image on one device): run once without --gpus, or let the tool do it (it does when the directory is missing)");
    if (api().e3d_reg_point_radius_minmax(reg, pts.data(), n, min_radius.data(), max_radius.data()) < 0) return lib_fail("e3d_reg_point_radius_minmax");
    float min_radius_value = INFINITY, max_radius_value = -INFINITY;
    for (size_t i = 0; i < n; ++i) { min_radius_value = std::min(min_radius_value, min_radius[i]); max_radius_value = std::max(max_radius_value, max_radius[i]); }
    if (!std::isfinite(min_radius_value)) return fail("ComputeMultiResPointCloud(): no scan point is visible in any image");
    // CreateMultiScalePointCloud :264-369
    const float min_point_radius = min_radius_value * prm.min_radius_bias;
    double radius = min_point_radius;
    ScaleCloud last;
    for (size_t i = 0; i < n; ++i)
      if (radius >= min_radius[i]) {
        last.pts.insert(last.pts.end(), pts.begin() + 3 * i, pts.begin() + 3 * i + 3);
        last.col.push_back(col[i]); last.scan.push_back(scan_idx[i]); last.max_radius.push_back(max_radius[i]);
      }
    std::vector<ScaleCloud> out;
    float last_radius = -1;
    while (true) {
      if (last_radius > 0) {
        ScaleCloud next;
        for (size_t i = 0; i < last.col.size(); ++i)
          if (radius <= last.max_radius[i]) {
            next.pts.insert(next.pts.end(), last.pts.begin() + 3 * i, last.pts.begin() + 3 * i + 3);
            next.col.push_back(last.col[i]); next.scan.push_back(last.scan[i]); next.max_radius.push_back(last.max_radius[i]);
          }
        for (size_t i = 0; i < n; ++i)
          if (last_radius < min_radius[i] && radius >= min_radius[i]) {
            next.pts.insert(next.pts.end(), pts.begin() + 3 * i, pts.begin() + 3 * i + 3);
            next.col.push_back(col[i]); next.scan.push_back(scan_idx[i]); next.max_radius.push_back(max_radius[i]);
          }
        last = std::move(next);
      }
      ScaleCloud merged;
      merged.radius = radius;
      const size_t m = last.col.size();
      merged.pts.resize(3 * std::max<size_t>(m, 1)); merged.col.resize(std::max<size_t>(m, 1)); merged.scan.resize(std::max<size_t>(m, 1)); merged.max_radius.resize(std::max<size_t>(m, 1));
      const int64_t k = m ? api().e3d_merge_close_points((float)(prm.merge_distance_factor * radius), num_scans, last.pts.data(), last.col.data(),
                                                       last.scan.data(), last.max_radius.data(), m, merged.pts.data(), merged.col.data(),
                                                       merged.scan.data(), merged.max_radius.data())
                            : 0;
      if (k < 0) return lib_fail("e3d_merge_close_points");
      merged.pts.resize(3 * (size_t)k); merged.col.resize(k); merged.scan.resize(k); merged.max_radius.resize(k);
      out.push_back(merged);
      last_radius = (float)radius;
      radius *= 2;
      if (radius >= max_radius_value * 0.99f) break;
      last = merged;
    }
    std::cout << "ComputeMultiResPointCloud(): #Initial point scales: " << out.size() << std::endl;
    const int need = prm.point_neighbor_candidate_count + 1;
    auto sufficient = [&](const ScaleCloud& c) {
      if (!use_fixed) return (int)c.col.size() >= need;
      std::vector<int> per(num_scans, 0);
      for (uint8_t s2 : c.scan) per[s2]++;
      for (int v : per) if (v < need) return false;
      return true;
    };
    auto filter_scales = [&](const char* kept, std::vector<ScaleCloud>& v) {
      std::vector<ScaleCloud> r;
      for (size_t i = 0; i < v.size(); ++i) {
        std::cout << (sufficient(v[i]) ? kept : "Deleting") << " point_scale " << i << ": " << v[i].col.size() << " points" << std::endl;
        if (sufficient(v[i])) r.push_back(std::move(v[i]));
      }
      v.swap(r);
    };
    std::cout << "ComputeMultiResPointCloud(): Filtering out scales with insufficient point count ..." << std::endl;
    filter_scales("Remaining", out);
    const int K = prm.point_neighbor_count;
    auto neighbors_of = [&](const ScaleCloud& c, std::vector<uint32_t>* nb) {
      nb->resize(c.col.size() * (size_t)K);
      return api().e3d_determine_point_neighbors(c.pts.data(), c.col.size(), c.scan.data(), num_scans, use_fixed ? 1 : 0, K,
                                                 prm.point_neighbor_candidate_count, nb->data()) >= 0;
    };
    std::cout << "ComputeMultiResPointCloud(): Determining neighbors ..." << std::endl;
    std::cout << "ComputeMultiResPointCloud(): Filtering out points with small intensity differences ..." << std::endl;
    for (ScaleCloud& c : out) {
      std::vector<uint32_t> nb;
      if (!neighbors_of(c, &nb)) return lib_fail("e3d_determine_point_neighbors");
      const size_t m = c.col.size();
      std::vector<char> del(m), del2(m, 1);
      for (size_t p = 0; p < m; ++p) {
        float sum = 0;
        for (int k = 0; k < K; ++k) sum += std::fabs(c.col[nb[p * K + k]] - c.col[p]);
        del[p] = (sum / K) < prm.min_mean_intensity_difference_for_points;
      }
      for (size_t p = 0; p < m; ++p)
        if (!del[p]) { del2[p] = 0; for (int k = 0; k < K; ++k) del2[nb[p * K + k]] = 0; }     // keep the neighbours of kept points too
      size_t o = 0;
      for (size_t p = 0; p < m; ++p) {
        if (del2[p]) continue;
        for (int a = 0; a < 3; ++a) c.pts[3 * o + a] = c.pts[3 * p + a];
        c.col[o] = c.col[p]; c.scan[o] = c.scan[p];
        ++o;
      }
      c.pts.resize(3 * o); c.col.resize(o); c.scan.resize(o);
    }
    std::cout << "ComputeMultiResPointCloud(): Filtering out scales with insufficient point count ..." << std::endl;
    filter_scales("Final", out);
    std::cout << "ComputeMultiResPointCloud(): Determining neighbors ..." << std::endl;
    point_radii.clear(); points.clear(); colors.clear(); neighbors.clear();
    for (ScaleCloud& c : out) {
      std::vector<uint32_t> nb;
      if (!neighbors_of(c, &nb)) return lib_fail("e3d_determine_point_neighbors");
      point_radii.push_back((float)c.radius); points.push_back(std::move(c.pts)); colors.push_back(std::move(c.col)); neighbors.push_back(std::move(nb));
    }
    if (point_radii.empty()) return fail("ComputeMultiResPointCloud(): no point scale has enough points");
    return true;
  }

  // Problem::SetScanGeometryAndInitialize: image scales, pyramids, point scales, fixed descriptors -> device
  bool SetScanGeometryAndInitialize(const std::vector<PointCloud::Ptr>& scans, const std::vector<float>& occlusion_points,
                                    const std::string& multi_res_dir) {
    if (!InitializeImages()) return false;
    if (!SetOcclusionGeometry(occlusion_points, nullptr)) return false;
    return SetMultiResGeometry(scans, multi_res_dir);
  }

  // Problem::InitializeImages (problem.cc): image scale count, camera pyramids, image + mask pyramids -> device problem
  bool InitializeImages() {
    image_scale_count = 1;
    for (const HostIntrinsics& in : intrinsics_list) image_scale_count = std::max(image_scale_count, ComputeImageScaleCount(in));
    std::cout << "#Image scales: " << image_scale_count << std::endl;
    for (HostIntrinsics& in : intrinsics_list) in.min_image_scale = image_scale_count - ComputeImageScaleCount(in);

    e3d_reg_params rp{};
    rp.point_neighbor_count = prm.point_neighbor_count;
    rp.robust_weighting_type = prm.robust_weighting_type;
    rp.robust_weighting_parameter = prm.robust_weighting_parameter;
    rp.fixed_residuals_weight = prm.fixed_residuals_weight;
    rp.variable_residuals_weight = prm.variable_residuals_weight;
    rp.maximum_valid_intensity = (float)prm.maximum_valid_intensity;
    rp.occlusion_depth_threshold = prm.occlusion_depth_threshold;
    rp.splat_radius = prm.splat_radius;
    rp.current_image_scale = 0;
    rp.image_scale_count = image_scale_count;
    const int n_gpus = gpu_count_setting();
    if (n_gpus > 1) {
      comms.assign((size_t)n_gpus, nullptr);
      if (api().e3d_comm_create_all(n_gpus, nullptr, comms.data()) < 0) return lib_fail("e3d_comm_create_all");
    }
    for (int r = 0; r < n_gpus; ++r) {
      if (n_gpus > 1 && api().e3d_init(r) < 1) return lib_fail("e3d_init");
      e3d_reg_t* h = api().e3d_reg_create(&rp);
      if (!h) return lib_fail("e3d_reg_create");
      regs.push_back(h);
      if (n_gpus > 1 && api().e3d_reg_set_comm(h, comms[(size_t)r]) < 0) return lib_fail("e3d_reg_set_comm");
    }
    if (n_gpus > 1) api().e3d_init(0);
    reg = regs[0];
    reg_params = rp;

    for (const HostIntrinsics& in : intrinsics_list)
      if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_intrinsics(r, in.intrinsics_id, in.model, in.width, in.height, in.params, in.n_params,
                                                                             in.min_image_scale, image_scale_count - in.min_image_scale); },
                    "e3d_reg_set_intrinsics"))
        return false;

    std::cout << "LoadImages(): Reading image data ..." << std::endl;
    // Decoding (PNG inflate / JPEG Huffman + IDCT, 0.4 - 0.6 s per 24 MP image) and the pyramids run on up to 16 host threads, a batch
    // of 32 images at a time; checks, messages and uploads then happen in image order as before.
    struct Decoded { GrayImage g; std::vector<GrayImage> pyr, mask; std::string err, mask_path; bool mask_size_bad = false, mask_value_bad = false; };
    std::vector<HostImage*> order;
    for (auto& kv : images) order.push_back(&kv.second);
    const size_t kBatch = 32;
    const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    for (size_t b0 = 0; b0 < order.size(); b0 += kBatch) {
    const size_t nb = std::min(kBatch, order.size() - b0);
    std::vector<Decoded> decoded(nb);
    {
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (;;) {
          const size_t j = next++;
          if (j >= nb) break;
          const HostImage& wim = *order[b0 + j];
          const HostIntrinsics& win = intrinsics_list[wim.intrinsics_id];
          const int wlevels = image_scale_count - win.min_image_scale;
          Decoded& d = decoded[j];
          d.g = imread_gray(wim.file_path, &d.err);
          if (d.g.empty() || d.g.width != win.width || d.g.height != win.height) continue;
          d.pyr = build_image_pyramid(d.g, wlevels);
          const std::string wdir = path_parent(wim.file_path);
          d.mask_path = path_parent(wdir) + "/masks_for_images/" + path_filename(wdir) + "/" + replace_extension(path_filename(wim.file_path), "png");
          if (file_exists(d.mask_path)) {
            std::string merr;
            GrayImage m = imread_gray(d.mask_path, &merr);
            if (m.width != d.g.width || m.height != d.g.height) { d.mask_size_bad = true; continue; }
            for (uint8_t v : m.data) if (v != 0 && v != 1 && v != 2) { d.mask_value_bad = true; break; }
            if (!d.mask_value_bad) d.mask = build_mask_pyramid(m, wlevels);
          }
        }
      };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < n_threads && t < nb; ++t) pool.emplace_back(work);
      work();
      for (std::thread& t : pool) t.join();
    }
    for (size_t bj = 0; bj < nb; ++bj) {
      HostImage& im = *order[b0 + bj];
      HostIntrinsics& in = intrinsics_list[im.intrinsics_id];
      const int levels = image_scale_count - in.min_image_scale;
      std::string err = decoded[bj].err;
      GrayImage& g = decoded[bj].g;
      if (g.empty()) return fail("Cannot read image: " + im.file_path + " (" + err + ")");
      if (g.width != in.width || g.height != in.height) return fail("Image size differs from its camera: " + im.file_path);
      std::vector<GrayImage>& pyr = decoded[bj].pyr;
      std::vector<GrayImage>& mask = decoded[bj].mask;
      const std::string image_dir = path_parent(im.file_path), dataset_dir = path_parent(image_dir);
      const std::string& mask_path = decoded[bj].mask_path;
      if (decoded[bj].mask_size_bad) return fail("Image and mask_ sizes differ! " + mask_path);
      if (decoded[bj].mask_value_bad) return fail("Unknown mask_ value in " + mask_path);
      if (!in.camera_mask_checked) {
        in.camera_mask_checked = true;
        const std::string cam_mask_path = replace_extension(dataset_dir + "/masks_for_cameras/" + path_filename(image_dir), "png");
        if (file_exists(cam_mask_path)) {
          GrayImage m = imread_gray(cam_mask_path, &err);
          if (m.width != g.width || m.height != g.height) return fail("Image and mask_ sizes differ! " + cam_mask_path);
          in.camera_mask = build_mask_pyramid(m, levels);
        }
      }
      bool upload_camera_mask = false;
      if (!in.camera_mask.empty() && !in.camera_mask_uploaded) { in.camera_mask_uploaded = true; upload_camera_mask = true; }
      // level sizes: the camera pyramid rounds (ScaledBy, int(0.5 w + 0.5)), the image pyramid truncates (int(0.5 cols)); they
      // agree for the even sizes of real pyramids, otherwise the image level is edge-padded to the camera level's size
      std::vector<const uint8_t*> lp(levels), lm(levels, nullptr);
      for (int l = 0; l < levels; ++l) {
        int w = 0, h = 0;
        if (api().e3d_reg_get_intrinsics_level(reg, in.intrinsics_id, l, &w, &h, nullptr, nullptr) < 0) return lib_fail("e3d_reg_get_intrinsics_level");
        auto fit = [&](GrayImage& a) {
          if (a.width == w && a.height == h) return;
          GrayImage o; o.width = w; o.height = h; o.data.resize((size_t)w * h);
          for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) o.data[(size_t)y * w + x] = a.data[(size_t)std::min(y, a.height - 1) * a.width + std::min(x, a.width - 1)];
          a = o;
        };
        fit(pyr[l]); lp[l] = pyr[l].data.data();
        if (!mask.empty()) { fit(mask[l]); lm[l] = mask[l].data.data(); }
        if (upload_camera_mask) fit(in.camera_mask[l]);
      }
      // Intrinsics::camera_mask: once per camera, to every rank (observations are created by the image's owner, but the point
      // radius computation of the multi-resolution cloud runs wherever the image lives)
      if (upload_camera_mask) {
        std::vector<const uint8_t*> cm(levels);
        for (int l = 0; l < levels; ++l) cm[l] = in.camera_mask[l].data.data();
        for (e3d_reg_t* r : regs)
          if (api().e3d_reg_set_camera_mask(r, in.intrinsics_id, cm.data()) < 0) return lib_fail("e3d_reg_set_camera_mask");
      }
      // pixels go to the image's owner only; every rank knows the image and its pose
      for (e3d_reg_t* r : regs) {
        const bool mine = r == owner(im.image_id);
        if (api().e3d_reg_set_image(r, im.image_id, im.intrinsics_id, mine ? lp.data() : nullptr, (mine && !mask.empty()) ? lm.data() : nullptr) < 0) return lib_fail("e3d_reg_set_image");
        if (api().e3d_reg_set_image_pose(r, im.image_id, im.image_T_global.q, im.image_T_global.t) < 0) return lib_fail("e3d_reg_set_image_pose");
      }
    }
    }
    for (const HostRig& rig : rigs) {
      std::vector<float> q, t;
      for (const Pose7& p : rig.image_T_rig) { q.insert(q.end(), p.q, p.q + 4); t.insert(t.end(), p.t, p.t + 3); }
      if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_rig(r, rig.rig_id, (int)rig.image_T_rig.size(), q.data(), t.data()); }, "e3d_reg_set_rig")) return false;
    }
    for (const HostRigImages& f : rig_images)
      if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_add_rig_images(r, f.rig_id, f.image_ids.data(), (int)f.image_ids.size()); }, "e3d_reg_add_rig_images")) return false;

    return true;
  }

  // the occlusion geometry: meshes if given (optionally moved by `left`, a row-major 3x4 transform applied after each mesh's own
  // pose -- GroundTruthCreator's first_scan_up_transformation), else 2D splats of all scan points
  bool SetOcclusionGeometry(const std::vector<float>& occlusion_points, const float* left) {
    if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_occlusion_options(r, prm.min_occlusion_depth, prm.max_occlusion_depth, 1); }, "e3d_reg_set_occlusion_options")) return false;
    if (occlusion_mesh_path.empty() && occlusion_splats_path.empty()) {
      if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_splat_points(r, occlusion_points.data(), occlusion_points.size() / 3); }, "e3d_reg_set_splat_points")) return false;
    } else {
      if (!occlusion_mesh_path.empty() && !AddOcclusionMesh(occlusion_mesh_path, true, left)) return false;
      if (!occlusion_splats_path.empty() && !AddOcclusionMesh(occlusion_splats_path, false, left)) return false;
    }
    return true;
  }

  bool SetMultiResGeometry(const std::vector<PointCloud::Ptr>& scans, const std::string& multi_res_dir) {
    if (multi_res_dir.empty()) return fail("Please specify --multi_res_point_cloud_directory_path.");
    if (LoadMultiResPointCloud(multi_res_dir)) {
      std::cout << "SetScanGeometryAndInitialize(): Loaded existing multi-res point cloud." << std::endl;
    } else {
      if (!ComputeMultiResPointCloud(scans)) return false;
      // saved for faster loading next time (and to keep it constant while camera poses change)
      if (!SaveMultiResPointCloud(multi_res_dir)) return false;
    }
    const int K = prm.point_neighbor_count;
    const bool use_fixed = prm.fixed_residuals_weight > 0;
    if (use_fixed) std::cout << "SetScanGeometryAndInitialize(): Compute fixed point descriptors from colors ..." << std::endl;
    for (size_t s = 0; s < points.size(); ++s) {
      const size_t n = points[s].size() / 3;
      std::vector<float> desc;
      if (use_fixed) {
        desc.resize(n * K);
        for (size_t p = 0; p < n; ++p)
          for (int k = 0; k < K; ++k) desc[p * K + k] = colors[s][neighbors[s][p * K + k]] - colors[s][p];      // ComputeDescriptor
      }
      if (!all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_point_scale(r, (int)s, points[s].data(), n, point_radii[s], neighbors[s].data(), use_fixed ? desc.data() : nullptr); }, "e3d_reg_set_point_scale")) return false;
    }
    return true;
  }

  int max_image_scale() const { return image_scale_count - 1; }
  e3d_reg_params reg_params{};

  // ObservationsCache::ObservationsCache (src/opt/observations_cache.cc:39-50): if the folder exists the per-image
  // `<folder>/<image dir name>/<image file name>.observed_indices` files are loaded (:70-102; a missing file or a different
  // point scale count is fatal), otherwise the lists are determined on the device at image scale 0 and written (:104-158).
  // File layout: int point_scale_count, then per point scale std::size_t n + std::size_t[n].
  bool PrepareObservationsCache(const std::string& path) {
    auto file_of = [&](const HostImage& im) {
      return join_path(join_path(path, path_filename(path_parent(im.file_path))), path_filename(im.file_path) + ".observed_indices");
    };
    const int scale_count = (int)point_radii.size();
    if (file_exists(path)) {
      for (auto& kv : images) {
        const std::string fn = file_of(kv.second);
        FILE* f = fopen(fn.c_str(), "rb");
        if (!f) return fail("Missing file for observed point indices: " + fn + ". Delete the observed point indices directory to re-generate the files.");
        int file_scale_count = 0;
        bool ok = fread(&file_scale_count, sizeof(int), 1, f) == 1;
        if (ok && file_scale_count != scale_count) {
          fclose(f);
          return fail("Point scale count differs between observed points file and current setting. Delete observed_point_indices directory to re-generate the files with the new setting.");
        }
        for (int ps = 0; ok && ps < scale_count; ++ps) {
          uint64_t n = 0;
          ok = fread(&n, sizeof(uint64_t), 1, f) == 1;
          std::vector<uint64_t> list(ok ? n : 0);
          ok = ok && fread(list.data(), sizeof(uint64_t), n, f) == n;
          if (ok && api().e3d_reg_set_observed_indices(owner(kv.first), kv.first, ps, list.data(), list.size()) < 0) { fclose(f); return lib_fail("e3d_reg_set_observed_indices"); }
        }
        fclose(f);
        if (!ok) return fail("Cannot read observed point indices: " + fn + " (file corrupted?)");
      }
      return true;
    }
    if (!all_regs_parallel([&](e3d_reg_t* r, int) { return api().e3d_reg_determine_observed_indices(r); }, "e3d_reg_determine_observed_indices")) return false;
    for (auto& kv : images) {
      const std::string fn = file_of(kv.second);
      create_directories(path_parent(fn));
      FILE* f = fopen(fn.c_str(), "wb");
      if (!f) return fail("Cannot write " + fn);
      fwrite(&scale_count, sizeof(int), 1, f);
      for (int ps = 0; ps < scale_count; ++ps) {
        const int64_t n = api().e3d_reg_get_observed_indices(owner(kv.first), kv.first, ps, nullptr);
        if (n < 0) { fclose(f); return lib_fail("e3d_reg_get_observed_indices"); }
        std::vector<uint64_t> list((size_t)n);
        if (n && api().e3d_reg_get_observed_indices(owner(kv.first), kv.first, ps, list.data()) < 0) { fclose(f); return lib_fail("e3d_reg_get_observed_indices"); }
        const uint64_t count = (uint64_t)n;
        fwrite(&count, sizeof(uint64_t), 1, f);
        fwrite(list.data(), sizeof(uint64_t), list.size(), f);
      }
      fclose(f);
    }
    return true;
  }

  // pulls the optimised intrinsics, poses and rig extrinsics back from the device
  bool ReadBackState() {
    for (HostIntrinsics& in : intrinsics_list)
      if (api().e3d_reg_get_intrinsics_level(reg, in.intrinsics_id, 0, nullptr, nullptr, in.params, nullptr) < 0) return lib_fail("e3d_reg_get_intrinsics_level");
    for (auto& kv : images)
      if (api().e3d_reg_get_image_pose(reg, kv.first, kv.second.image_T_global.q, kv.second.image_T_global.t) < 0) return lib_fail("e3d_reg_get_image_pose");
    for (HostRig& rig : rigs)
      for (size_t c = 0; c < rig.image_T_rig.size(); ++c)
        if (api().e3d_reg_get_rig(reg, rig.rig_id, (int)c, rig.image_T_rig[c].q, rig.image_T_rig[c].t) < 0) return lib_fail("e3d_reg_get_rig");
    return true;
  }

  // io::ExportProblemToColmap(problem, image_base_path, write_points = false, write_images = false, write_project = false, dir)
  bool ExportToColmap(const std::string& image_base_path, const std::string& dir) const {
    create_directories(dir);
    std::ofstream cf(dir + "/cameras.txt");
    cf << "# Camera list with one line of data per camera:" << std::endl;
    cf << "#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]" << std::endl;
    cf << "# Number of cameras: " << intrinsics_list.size() << std::endl;
    for (size_t i = 0; i < intrinsics_list.size(); ++i) {
      const HostIntrinsics& in = intrinsics_list[i];
      cf << i << " " << in.model_name << " " << in.width << " " << in.height;
      for (int p = 0; p < in.n_params; ++p) {
        float v = in.params[p];
        const int ci = camera_unique_focal(in.model) ? 1 : 2;
        if (p == ci || p == ci + 1) v += 0.5f;                    // ShiftedBy(0.5, 0.5)
        cf << " " << v;
      }
      cf << std::endl;
    }
    std::ofstream f(dir + "/images.txt");
    f << "# Image list with two lines of data per image:" << std::endl;
    f << "#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME" << std::endl;
    f << "#   POINTS2D[] as (X, Y, POINT3D_ID)" << std::endl;
    f << "# Number of images: " << images.size() << std::endl;
    for (const auto& kv : images) {
      const HostImage& im = kv.second;
      f << im.image_id << " " << im.image_T_global.q[0] << " " << im.image_T_global.q[1] << " " << im.image_T_global.q[2] << " "
        << im.image_T_global.q[3] << " " << im.image_T_global.t[0] << " " << im.image_T_global.t[1] << " " << im.image_T_global.t[2] << " "
        << im.intrinsics_id << " " << relative_path(image_base_path, im.file_path) << std::endl;
      f << std::endl;
    }
    std::ofstream pf(dir + "/points3D.txt");
    pf << "";
    return (bool)f && (bool)cf;
  }

  // io::ExportRigs
  bool ExportRigs(const std::string& dir) const {
    std::vector<ColmapRig> out;
    for (const HostRig& rig : rigs) {
      ColmapRig r;
      bool found = false;
      for (const HostRigImages& fr : rig_images) {
        if (fr.rig_id != rig.rig_id) continue;
        for (size_t c = 0; c < rig.folder_names.size(); ++c) {
          ColmapRigCamera rc;
          rc.camera_id = images.at(fr.image_ids[c]).intrinsics_id;
          rc.image_prefix = rig.folder_names[c];
          r.cameras.push_back(rc);
        }
        found = true;
        break;
      }
      if (!found) return fail("ExportRigs(): a rig without images");
      r.ref_camera_id = r.cameras.front().camera_id;
      out.push_back(r);
    }
    return WriteColmapRigs(dir + "/rigs.json", out);
  }
};

}  // namespace e3d_host
