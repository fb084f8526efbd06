/*
 * oracle/e3d_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's hot path, used as the parity
 * checker by tests/, __graft_entry__.smoke() and as bench.py's cpu_baseline
 * ("kind": "port").  Nothing in the product library links or loads this.
 *
 * Parity status: the reference itself cannot be compiled in this image (PCL,
 * FLANN, Eigen, OpenCV, glog, Boost are absent), so this
 * oracle is pinned against
 *   - the reference's own known-answer tests restated in tests/
 *     (src/opt/test/test_icp.cc:39-172 PlaneCase / IdenticalCloud properties),
 *   - brute-force nearest-neighbour cross checks and scipy cKDTree,
 * and is "parity unpinned" where the reference has no test: NN tie-breaking,
 * per-pair correspondence counts, normal estimation (pcl::eigen33), f32
 * operation order inside PCL/Eigen (recalled, see oracle_math.h).
 */
#ifndef E3D_ORACLE_H
#define E3D_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (A) point-to-plane ICP: restates icp::PointToPlaneICP
 * (src/icp/icp_point_to_plane.h:39-80, .cc:109-342) and PointToPlaneICPImpl
 * (src/icp/icp_point_to_plane_impl.h:43-318). ------------------------------ */
typedef struct oracle_icp oracle_icp;

typedef struct {
  int32_t iteration;     /* outer iteration number passed to Run */
  int32_t src, tgt;      /* impl cloud indices as printed by the reference; -1 = "fixed clouds" */
  int64_t count;         /* number of correspondences */
  float   distance_sum;  /* f32 sum of squared NN distances in correspondence order */
} oracle_icp_pair_record;

typedef struct {
  int32_t iteration;
  int32_t inner_iterations;   /* LM iterations executed (<=150) */
  int32_t accumulate_passes;  /* full H/b/cost passes over all correspondences */
  int32_t cost_passes;        /* cost-only passes (LM tries) */
  int64_t correspondences;    /* total over all pairs */
  double  initial_cost, final_cost;
  double  t_transform_s, t_nn_s, t_lm_s;  /* wall seconds */
} oracle_icp_iter_record;

oracle_icp* oracle_icp_create(void);
void oracle_icp_destroy(oracle_icp*);
/* xyz,nrm: n x 3 f32 (copied).  T: row-major 3x4 affine global_T_cloud. Returns
 * the cloud index (>=0) or -1 for fixed clouds (icp_point_to_plane.cc:109-135). */
int oracle_icp_add_cloud(oracle_icp*, const float* xyz, const float* nrm, size_t n,
                         const float T[12], int fixed);
/* icp_point_to_plane.cc:137-163.  Returns 1 when converged, 0 otherwise, <0 on error. */
int oracle_icp_run(oracle_icp*, float max_correspondence_distance, int initial_iteration,
                   int max_num_iterations, float convergence_threshold_max_movement,
                   int print_progress);
int oracle_icp_get_pose(oracle_icp*, int cloud_index, float T[12]);
/* inner LM iteration cap (reference: 150, icp_point_to_plane.cc:312); exposed so
 * that bounded CPU-baseline samples can be timed. */
void oracle_icp_set_max_inner_iterations(oracle_icp*, int n);
/* threads for the NN phase: 0 = reference-faithful (parallel over the M*M pair
 * loop only, icp_point_to_plane.cc:208), 1 = all-core (queries parallel). */
void oracle_icp_set_all_core(oracle_icp*, int on);

size_t oracle_icp_num_pair_records(const oracle_icp*);
const oracle_icp_pair_record* oracle_icp_pair_records(const oracle_icp*);
size_t oracle_icp_num_iter_records(const oracle_icp*);
const oracle_icp_iter_record* oracle_icp_iter_records(const oracle_icp*);

/* Stand-alone pieces (used by unit parity tests of individual kernels). */
/* FindCorrespondencesFast (icp_point_to_plane.cc:42-105): out arrays sized n_src. */
int64_t oracle_find_correspondences(const float* src_xyz, size_t n_src,
                                    const float* tgt_xyz, size_t n_tgt,
                                    float max_correspondence_distance,
                                    int32_t* idx_query, int32_t* idx_match, float* sqdist);
/* brute-force version of the same (O(n_src*n_tgt)), to validate the kd-tree */
int64_t oracle_find_correspondences_brute(const float* src_xyz, size_t n_src,
                                          const float* tgt_xyz, size_t n_tgt,
                                          float max_correspondence_distance,
                                          int32_t* idx_query, int32_t* idx_match, float* sqdist);
/* pcl::transformPointCloudWithNormals + bbox (icp_point_to_plane.cc:189-205). */
void oracle_transform_cloud(const float* xyz, const float* nrm, size_t n, const float T[12],
                            float* out_xyz, float* out_nrm, float bbox_min[3], float bbox_max[3]);
/* One accumulate pass (icp_point_to_plane_impl.h:119-211) for one directed pair
 * with both clouds at inner pose {q(wxyz), t}; writes the 12x12 pair system
 * (row-major, [src(6), tgt(6)], all entries, no triangular masking), 12-vector
 * b and the cost. */
void oracle_icp_pair_system(const float* src_xyz, const float* src_nrm,
                            const float* tgt_xyz, const float* tgt_nrm,
                            const int32_t* idx_query, const int32_t* idx_match, int64_t n_corr,
                            const float src_q[4], const float src_t[3],
                            const float tgt_q[4], const float tgt_t[3],
                            double H[144], double b[12], double* cost);
/* SE3d::exp(-x).cast<float>() * pose  (icp_point_to_plane_impl.h:235). */
void oracle_se3_update(const double x[6], const float q_in[4], const float t_in[3],
                       float q_out[4], float t_out[3]);
void oracle_quat_to_R(const float q[4], float R[9]);
/* dense symmetric solve reading the upper triangle only */
void oracle_ldlt_solve_upper(const double* A, int n, const double* b, double* x);

/* ---- (A') normal estimation: restates pcl::NormalEstimationTwoPassOMP
 * (src/geometry/two_pass_normal_3d_omp.hpp:48-119) with kNN (k>0) or radius
 * search; out_normal n x 3, out_curvature n.  Returns 0 on success.
 * knn_idx_out (optional, n*k int32) receives the neighbour lists (k search). */
int oracle_normals(const float* xyz, size_t n, int k, float radius,
                   const float viewpoint[3], float* out_normal, float* out_curvature,
                   int32_t* knn_idx_out);
/* two-pass mean+covariance (two_pass_centroid.hpp:155-259) + solvePlaneParameters
 * for one explicit neighbour list. */
/* test hook: include/e3d_libm.h on the host; fn: 0 atanf(x) 1 atan2f(x, y) 2 sinf 3 cosf 4 tanf 5 log2f */
void oracle_libm_eval(int fn, const float* x, const float* y, size_t n, float* out);

void oracle_point_normal(const float* xyz, const int32_t* indices, int count,
                         float plane[4], float* curvature);
/* kNN alone: idx/dist n_q x k, sorted by (dist, index). */
void oracle_knn(const float* xyz, size_t n, const float* queries, size_t n_q, int k,
                int32_t* idx, float* dist);

/* ---- (B) dense photometric image registration: per-observation arithmetic of ImageRegistrator (oracle_reg.c).
 * PINHOLE / OPENCV / THIN_PRISM_FISHEYE cameras, non-rig images, colour residuals (fixed + variable descriptors).
 * Jacobian / system sizes follow the camera: I = n_params, V = I + 6; j_intr is n_obs x I, H is V x V row-major. ------- */
/* type: 0 PINHOLE (4 parameters), 1 OPENCV (8), 2 THIN_PRISM_FISHEYE (12); see oracle_camera.h */
typedef struct {
  int type; int width, height; int n_params;
  float p[12];
  float cutoff2;          /* CameraBaseImpl::radius_cutoff_squared_ of the outermost model */
  float inner_cutoff2;    /* that of the inner non-fisheye model (type 2) */
  float fx_inv, fy_inv, cx_inv, cy_inv;
} oreg_camera;

/* a non-reference image of a rig frame: image_T_rig of its camera and the pose of the frame's reference image */
typedef struct { float q_image_T_rig[4]; float q_rig_T_global[4]; float t_rig_T_global[3]; } oreg_rig_link;

void oracle_reg_camera_init(oreg_camera* c, int type, int w, int h, const float* params);
void oracle_reg_camera_scaled(const oreg_camera* in, float factor, oreg_camera* out);
/* single-point entry points of the camera functions (unit tests restating src/camera/test/test_camera.cc) */
void oracle_reg_camera_distort(const oreg_camera* c, float nx, float ny, float out[2]);
void oracle_reg_camera_undistort(const oreg_camera* c, float dx, float dy, float out[2], int* converged);
void oracle_reg_camera_project(const oreg_camera* c, const float P[3], float out[2]);
void oracle_reg_camera_deriv_by_world(const oreg_camera* c, const float P[3], float d[6]);
void oracle_reg_camera_deriv_by_intrinsics(const oreg_camera* c, const float P[3], float* d /* 2 x n_params */);
void oracle_interp_trilinear_u8(const uint8_t* img0, int w0, const uint8_t* img1, int w1, float x0, float y0, float z, float* value);
void oracle_interp_trilinear_d_u8(const uint8_t* img0, int w0, const uint8_t* img1, int w1, float x0, float y0, float z,
                                  float* value, float* dx, float* dy, float* dz);
void oracle_interp_trilinear_f32(const float* img0, int w0, const float* img1, int w1, float x0, float y0, float z, float* value);
void oracle_interp_trilinear_d_f32(const float* img0, int w0, const float* img1, int w1, float x0, float y0, float z,
                                   float* value, float* dx, float* dy, float* dz);
float oracle_reg_robust_residual(int type, float param, float r);
float oracle_reg_robust_weight(int type, float param, float r);
void oracle_reg_splat_depth(const float* pts, size_t n, const float R[9], const float t[3], const oreg_camera* cam,
                            float point_radius, float* depth);
size_t oracle_reg_observe(const float* pts, size_t n_pts, float point_radius, const uint32_t* indices, size_t n_idx,
                          const float R[9], const float t[3], const oreg_camera* levels, int min_image_scale,
                          int n_levels, const uint8_t* const* images, const uint8_t* const* masks,
                          const float* occlusion, int image_scale, int border, int current_image_scale,
                          int image_scale_count, float occlusion_threshold, float max_valid_intensity,
                          uint32_t* out_idx, float* out_x, float* out_y, float* out_scale);
void oracle_reg_neighbors_observed(size_t n_pts, const uint32_t* obs_idx, size_t n_obs, const uint32_t* nbr, int K,
                                   uint8_t* flags);
void oracle_reg_pass1(const float* pts, float point_radius, const oreg_camera* cam_min, int min_image_scale,
                      const uint8_t* const* images, const int* widths, const float R[9], const float t[3],
                      const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale, size_t n_obs,
                      float* intensities, float* j_intr, float* j_pose);
void oracle_reg_accumulate(const float* pts, size_t n_pts, float point_radius, const uint32_t* nbr, int K,
                           const float* fixed_desc, const float* var_desc, const int32_t* obs_counts,
                           const oreg_camera* cam_min, int min_image_scale, const uint8_t* const* images, const int* widths,
                           const float R[9], const float t[3], const uint32_t* obs_idx, const float* obs_x,
                           const float* obs_y, const float* obs_scale, const uint8_t* flags, size_t n_obs, int robust_type,
                           float robust_param, float fixed_weight, float var_weight, double* H, double* b,
                           double sums[2], int64_t counts[2]);
/* rig variants: rig == NULL is the plain image; else j_rig is n_obs x 6 and the local system has V = I + 12 unknowns
 * [intrinsics, rig extrinsics, pose of the rig frame's reference image] */
void oracle_reg_pass1_rig(const float* pts, float point_radius, const oreg_camera* cam_min, int min_image_scale,
                          const uint8_t* const* images, const int* widths, const float R[9], const float t[3],
                          const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale, size_t n_obs,
                          const oreg_rig_link* rig, float* intensities, float* j_intr, float* j_pose, float* j_rig);
void oracle_reg_accumulate_rig(const float* pts, size_t n_pts, float point_radius, const uint32_t* nbr, int K,
                               const float* fixed_desc, const float* var_desc, const int32_t* obs_counts,
                               const oreg_camera* cam_min, int min_image_scale, const uint8_t* const* images, const int* widths,
                               const float R[9], const float t[3], const uint32_t* obs_idx, const float* obs_x,
                               const float* obs_y, const float* obs_scale, const uint8_t* flags, size_t n_obs, int robust_type,
                               float robust_param, float fixed_weight, float var_weight, const oreg_rig_link* rig,
                               double* H, double* b, double sums[2], int64_t counts[2]);
void oracle_se3_mul(const float qa[4], const float ta[3], const float qb[4], const float tb[3], float q[4], float t[3]);
void oracle_reg_cost(size_t n_pts, const uint32_t* nbr, int K, const float* fixed_desc, const float* var_desc,
                     const int32_t* obs_counts, int min_image_scale, const uint8_t* const* images, const int* widths,
                     const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale,
                     const uint8_t* flags, size_t n_obs, int robust_type, float robust_param, float fixed_weight,
                     float var_weight, double sums[2], int64_t counts[2]);
void oracle_reg_color_accumulate(size_t n_pts, const uint32_t* nbr, int K, int min_image_scale, const uint8_t* const* images,
                                 const int* widths, const uint32_t* obs_idx, const float* obs_x, const float* obs_y,
                                 const float* obs_scale, const uint8_t* flags, size_t n_obs, float* descriptors,
                                 int32_t* obs_counts);
void oracle_reg_color_finish(size_t n_pts, int K, float* descriptors, const int32_t* obs_counts);

/* ---- (f1) multi-resolution point cloud construction (oracle_multires.c, oracle_shuffle.cc) ------------------------------- */
void oracle_undistortion_lookup(const oreg_camera* c, float* out /* height x width x 2 */);
void oracle_image_to_normalized(const oreg_camera* c, const float* lookup, float px, float py, float out[2]);
void oracle_point_radius_minmax(const float* pts, size_t n, const float q[4], const float t[3], const oreg_camera* cam,
                                int image_scale, int min_image_scale, const oreg_camera* cam_min, const float* lookup_min,
                                const uint8_t* image_level, const uint8_t* mask_level, const float* occlusion,
                                float occlusion_threshold, float max_valid_intensity, double min_scaling_factor,
                                float* min_radius, float* max_radius);
size_t oracle_merge_close_points(float merge_distance, int num_scans, const float* pts, const float* colors,
                                 const uint8_t* scan_idx, const float* max_radius, size_t n, float* out_pts,
                                 float* out_colors, uint8_t* out_scan, float* out_max_radius);
int oracle_determine_point_neighbors(const float* xyz, size_t n, const uint8_t* scan_indices, int scan_count, int limit_to_same_scan,
                                     int neighbor_count, int candidate_count, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
