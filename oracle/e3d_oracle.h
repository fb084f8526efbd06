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
void oracle_point_normal(const float* xyz, const int32_t* indices, int count,
                         float plane[4], float* curvature);
/* kNN alone: idx/dist n_q x k, sorted by (dist, index). */
void oracle_knn(const float* xyz, size_t n, const float* queries, size_t n_q, int k,
                int32_t* idx, float* dist);

#ifdef __cplusplus
}
#endif
#endif
