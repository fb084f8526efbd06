/*
 * include/e3d_hip.h -- C-ABI of libe3dhip.so (MI355X / gfx950 HIP implementation of the
 * ETH3D dataset-pipeline scan-alignment hot path).
 *
 * The reference has no FFI: its drop-in boundary is the C++ class surface that the tool
 * mains and gtest binaries call.  Every entry point below names the
 * reference interface it replaces (file:line relative to the reference tree).  A thin C++
 * shim with the reference's class and method names sits on top of this ABI
 * (dataset-pipeline_amd/csrc/host/icp_point_to_plane.h); INTEGRATION.md shows the binding a
 * maintainer would add to the reference.
 *
 * Conventions
 *   - all functions return int status: >= 0 success (value documented per function),
 *     E3D_ERR_* (< -1) on failure; e3d_last_error() returns a thread-local message.  Nothing
 *     aborts the process (the reference CHECK()s / throws instead).
 *   - point data is n x 3 float32, row-major ("xyz xyz ...").  Pointers may be host pointers
 *     or HIP device pointers (detected with hipPointerGetAttributes); the library copies what
 *     it needs into its own device buffers (SoA, sorted by grid cell), so callers keep
 *     ownership and may free their buffers after the call returns.
 *   - poses are row-major 3x4 float32 affine matrices (Eigen::Affine3f rows 0..2).
 *   - one host thread per handle; each handle owns one HIP stream on the device that was
 *     current when it was created.
 */
#ifndef E3D_HIP_H
#define E3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: e3d_reg_params grew by the three depth-residual fields; the ICP iteration record by the NN phase times (round 2)
 * 3: e3d_icp_iter_record grew by t_nn_sort_ms / t_nn_scan_ms / t_nn_compact_ms; e3d_comm_abort, e3d_reg_profile,
 *    e3d_icp_set_sequential_distance_sum (round 3)
 * 4: e3d_icp_set_resident_rows, e3d_comm_get_stats; iteration record: multi_cost_poses, lm_passes_skipped in the two reserved
 *    words, corr_rows_rewritten / corr_rows_walked appended (round 4)
 * 5: iteration record: nn_update_launches, nn_kernel_launches, nn_batches, nn_sort_calls appended (round 5: a batch of directed pairs
 *    per kernel launch) */
#define E3D_ABI_VERSION 5

#define E3D_ERR_INVALID   (-2)   /* bad argument / bad handle state          */
#define E3D_ERR_HIP       (-3)   /* a HIP runtime call failed                */
#define E3D_ERR_NO_DEVICE (-4)   /* no gfx950 device visible                 */
#define E3D_ERR_INDEX     (-5)   /* cloud index out of range (reference: .at() throws) */

/* ---- library ------------------------------------------------------------------------- */
int e3d_abi_version(void);
/* Select the HIP device for subsequently created handles; returns the device count. */
int e3d_init(int device);
const char* e3d_last_error(void);
/* Nearest-neighbour kernel selection for handles created afterwards: 0 = automatic (by points per grid
 * cell and grid size), 1 = one thread per query (sparse data), 2 = queries sorted by target cell + LDS-staged
 * candidate buckets found through the hash table (huge sparse grids), 3 = the same with the dense cell-start
 * directory and whole row segments per wave (default for dense scans).  All three are exact and return
 * identical results; the switch only exists for tests and profiling. */
int e3d_set_nn_mode(int mode);

/* ---- (A) icp::PointToPlaneICP  (src/icp/icp_point_to_plane.h:39-80) --------------------- */
typedef struct e3d_icp e3d_icp_t;

/* PointToPlaneICP::PointToPlaneICP()  (icp_point_to_plane.cc:107) */
e3d_icp_t* e3d_icp_create(void);
void e3d_icp_destroy(e3d_icp_t* icp);

/* int PointToPlaneICP::AddPointCloud(cloud, global_T_cloud, fixed)
 * (icp_point_to_plane.h:46-48, .cc:109-135).  Returns the cloud index (>= 0) for movable
 * clouds and -1 for fixed clouds (which are transformed once and merged into one fixed
 * cloud), exactly like the reference. */
int e3d_icp_add_cloud(e3d_icp_t* icp, const float* xyz, const float* normals, size_t n,
                      const float global_T_cloud[12], int fixed);

/* bool PointToPlaneICP::Run(max_correspondence_distance, initial_iteration,
 *   max_num_iterations, convergence_threshold_max_movement, print_progress)
 * (icp_point_to_plane.h:50-54, .cc:137-163).  Returns 1 if converged, 0 if not.  With
 * print_progress the reference's stdout lines are printed (same text; "avg. distance" is
 * accumulated in f64 on the device instead of a sequential f32 sum). */
int e3d_icp_run(e3d_icp_t* icp, float max_correspondence_distance, int initial_iteration,
                int max_num_iterations, float convergence_threshold_max_movement,
                int print_progress);

/* Eigen::Affine3f PointToPlaneICP::GetResultGlobalTCloud(int)  (icp_point_to_plane.h:56,
 * .cc:165-167). */
int e3d_icp_get_pose(e3d_icp_t* icp, int cloud_index, float global_T_cloud[12]);

/* Inner LM iteration cap of PointToPlaneICPImpl (reference constant 150,
 * icp_point_to_plane.cc:312); exposed for bounded benchmarks, default 150. */
int e3d_icp_set_max_inner_iterations(e3d_icp_t* icp, int n);
/* The "avg. distance" of the progress line and the pair records' distance_sum from the reference's own sum: f32, sequential, in
 * original source order (icp_point_to_plane.cc:226-229) -- byte-identical stdout of a single-rank run, at the price of one device -> host copy of the
 * distances and a host loop per pair and outer iteration.  Default off (environment E3D_ICP_SEQUENTIAL_DISTANCE_SUM=1 switches it
 * on for the tools): the f64 sum on the device, which does not stagnate at 2^24 times the typical term.  Ignored when sharded. */
int e3d_icp_set_sequential_distance_sum(e3d_icp_t* icp, int enable);
/* Correspondence rows of the LM passes.  1 (default; environment E3D_ICP_RESIDENT=0 flips it): every directed pair of the
 * certificate search keeps one RESIDENT row per query (48 B, source order, movable clouds in their local frame) for as long as
 * its grids live; an outer iteration rewrites only the rows whose partner changed and the LM passes apply the outer pose
 * (pcl::transformPointCloudWithNormals' operation order, icp_point_to_plane.cc:192-195) in front of the inner one -- the same
 * f32 numbers as rows written in the global frame.  0: all correspondences are gathered, transformed and compacted into fresh
 * rows every outer iteration (the reference's own data flow, icp_point_to_plane.cc:183-309; used automatically when the resident
 * rows do not fit HBM).  Same correspondences, counts and residuals either way; only the order of the f64 sums differs. */
int e3d_icp_set_resident_rows(e3d_icp_t* icp, int enable);

/* Per-pair correspondence report of every AlignMeshes call since creation -- the numbers the
 * reference only prints (icp_point_to_plane.cc:226-237).  src/tgt are the impl cloud indices
 * the reference prints; -1 stands for "fixed clouds". */
typedef struct {
  int32_t iteration;
  int32_t src, tgt;
  int64_t count;
  double  distance_sum;      /* sum of squared NN distances (f64 accumulation) */
} e3d_icp_pair_record;

typedef struct {
  int32_t iteration;
  int32_t inner_iterations;  /* LM iterations executed (<= 150)                            */
  int32_t full_passes;       /* fused H/b/cost passes over all correspondences             */
  int32_t cost_passes;       /* cost-only passes (one pose set)                            */
  int32_t multi_cost_passes; /* cost-only passes evaluating LM tries (0 or 1)..9 at once   */
  int32_t multi_cost_poses;  /* ABI 4: distinct new pose sets those passes evaluated (<= 10 each; tries whose f32 poses equal the
                                current ones or an earlier try's are not evaluated again)  */
  int64_t correspondences;   /* total over all directed pairs of this rank                 */
  int64_t queries;           /* NN queries issued by this rank                             */
  double  initial_cost, final_cost;
  double  t_transform_ms, t_nn_ms, t_lm_ms;   /* HIP-event times on the handle's stream   */
  double  t_lm_kernel_ms;    /* sum of LM pass kernel durations (HIP events)               */
  double  t_nn_query_ms;     /* sum of NN query kernel durations (HIP events)              */
  double  t_lm_full_kernel_ms; /* ... of the fused H/b/cost passes alone (k_lm_pass<1..3>)   */
  double  t_nn_certify_ms;   /* k_nn_certify launches (HIP events)                         */
  double  t_nn_bounded_ms;   /* k_nn_bounded launches                                      */
  double  t_nn_search_ms;    /* k_nn_rows / k_nn_cells / k_nn_query / k_nn_mfma launches   */
  int64_t nn_certify_queries, nn_bounded_queries, nn_search_queries;   /* queries those launches covered (nn_search_queries: the
                                  queries k_nn_rows visited -- those the key kernel settled because no target point lies in their
                                  27 cells are not among them; nn_certify_queries is 0 for a pair whose certificates were not
                                  tested in this iteration) */
  int32_t nn_certify_launches, nn_bounded_launches, nn_search_launches;
  int32_t lm_passes_skipped; /* ABI 4: LM passes not launched because every pose asked for had been evaluated already */
  /* ABI 3: the rest of the NN phase, so that the per-kernel times add up to the step (HIP events) */
  double  t_nn_sort_ms;      /* query keys (incl. settling the queries with an empty 27-cell block) + radix sort of the rest */
  double  t_nn_scan_ms;      /* match counts + scans (order-preserving compaction, first stage) */
  double  t_nn_compact_ms;   /* k_compact_corr / k_corr_update: the correspondence rows        */
  /* ABI 4 */
  int64_t corr_rows_rewritten; /* correspondence rows (48 B) written this iteration: every correspondence with compacted rows, the
                                  rows whose partner changed with resident rows                                        */
  int64_t corr_rows_walked;    /* rows one LM pass reads: the correspondences, or 64 x the active row groups of resident rows */
  /* ABI 5: launches of the NN phase (the all-pairs job's per-pair launches are what does not shrink with the number of GPUs) */
  int32_t nn_update_launches;  /* k_corr_update / k_compact_corr launches                                                */
  int32_t nn_kernel_launches;  /* all kernels the library itself launched in the NN phase (search, keys, row update, totals;
                                  the launches inside rocPRIM's radix sort are not counted: nn_sort_calls sorts)          */
  int32_t nn_batches;          /* batches of directed pairs that ran with one launch per kernel (0: pair by pair)          */
  int32_t nn_sort_calls;       /* radix sorts of query lists (k_nn_rows path)                                             */
} e3d_icp_iter_record;

size_t e3d_icp_num_pair_records(const e3d_icp_t* icp);
const e3d_icp_pair_record* e3d_icp_pair_records(const e3d_icp_t* icp);
size_t e3d_icp_num_iter_records(const e3d_icp_t* icp);
const e3d_icp_iter_record* e3d_icp_iter_records(const e3d_icp_t* icp);
void e3d_icp_clear_records(e3d_icp_t* icp);

/* Multi-GPU (one process per GPU): every rank holds all clouds and handles the slice
 * [n*rank/world, n*(rank+1)/world) of every directed pair's source cloud (in grid-cell order), so
 * any number of pairs -- including the 2 pairs of a 2-scan job -- shards evenly.  `allreduce` must sum `count` doubles in
 * place across ranks (RCCL/gloo through the host language; buffer is HOST memory) and leave
 * the identical result on every rank.  The reference has no equivalent (single process).
 * world_size == 1 with a callback is a tap: every buffer still passes through the callback (the sum over one rank is the
 * identity), which lets a caller record the reduced sums of a single-GPU run -- bench.py replays them to a handle that works as
 * rank 0 of a world of 8 on the same GPU to measure what one rank of an 8-GPU job does (its `scale_model`). */
typedef int (*e3d_allreduce_fn)(double* buffer, size_t count, void* user);
int e3d_icp_set_shard(e3d_icp_t* icp, int rank, int world_size,
                      e3d_allreduce_fn allreduce, void* user);

/* Native collectives: an RCCL communicator owned by the library (one rank per GPU, xGMI inside a node).  The per-pair
 * normal-equation blocks of every LM pass are reduced on the GPU and all-reduced in place on the handle's stream, the
 * counts once per outer iteration; no host hop and no callback.  Ranks are either processes (rank 0: e3d_comm_unique_id,
 * the 128 bytes travel through the launcher's rendezvous, every rank: e3d_comm_create) or host threads of one process
 * (e3d_comm_create_all: out[i] is the communicator of devices[i], devices == NULL means 0..n-1).  world_size = 1 is valid
 * (the collectives then run with a single rank).  e3d_icp_set_comm replaces e3d_icp_set_shard; the communicator must
 * outlive the handle and live on the handle's device. */
#define E3D_COMM_ID_BYTES 128
typedef struct e3d_comm e3d_comm_t;
int e3d_comm_unique_id(char id[E3D_COMM_ID_BYTES]);
e3d_comm_t* e3d_comm_create(const char id[E3D_COMM_ID_BYTES], int rank, int world_size, int device);
int e3d_comm_create_all(int n_devices, const int* devices, e3d_comm_t** out);
void e3d_comm_destroy(e3d_comm_t* comm);
/* Aborts the communicator (ncclCommAbort): an enqueue or a collective of THIS communicator that waits for a rank that will never
 * arrive returns with an error instead of blocking forever.  Callable from any host thread, also while the communicator's own
 * thread is inside a collective (no lock is shared with the enqueue); every later collective on `comm` fails; the object is still
 * released with e3d_comm_destroy.  Aborting one communicator does not release its peers: when a rank's step fails the tools abort
 * ALL local communicators (--gpus N, csrc/host/icp_point_to_plane.h). */
int e3d_comm_abort(e3d_comm_t* comm);
/* HIP-event time, number and payload of the all-reduces enqueued on `comm` since creation (or the last reset): what the bench
 * line reports per rank at N > 1 (ABI 4).  Waits for the collectives enqueued so far.  To be called from the host thread that
 * enqueues on `comm` (or after joining it): the counters are not synchronised against a running enqueue. */
int e3d_comm_get_stats(e3d_comm_t* comm, double* allreduce_ms, int64_t* allreduce_calls, int64_t* allreduce_bytes, int reset);
int e3d_comm_rank(const e3d_comm_t* comm);
int e3d_comm_world_size(const e3d_comm_t* comm);
int e3d_icp_set_comm(e3d_icp_t* icp, e3d_comm_t* comm);

/* ---- stand-alone kernels behind the same arithmetic (parity tests, other callers) ------ */

/* FindCorrespondencesFast(source, target, max_correspondence_distance)
 * (icp_point_to_plane.cc:42-105): for every source point the exact nearest target point with
 * squared distance < (float)(d*d), lowest target index on ties.  match_index[i] = target index
 * or -1; sq_distance[i] valid where matched.  Returns the number of correspondences. */
int64_t e3d_find_correspondences(const float* source_xyz, size_t n_source,
                                 const float* target_xyz, size_t n_target,
                                 float max_correspondence_distance,
                                 int32_t* match_index, float* sq_distance);

/* pcl::transformPointCloudWithNormals + AlignedBox extend (icp_point_to_plane.cc:189-205). */
int e3d_transform_cloud(const float* xyz, const float* normals, size_t n, const float T[12],
                        float* out_xyz, float* out_normals, float bbox_min[3], float bbox_max[3]);

/* One accumulate pass of PointToPlaneICPImpl::compute (icp_point_to_plane_impl.h:119-211) for
 * one directed pair at inner poses {q = w,x,y,z ; t}.  Outputs the 12x12 pair system over
 * [source(6), target(6)] (row-major, upper triangle filled, lower mirrored), b(12), cost. */
int e3d_icp_pair_system(const float* src_xyz, const float* src_normals,
                        const float* tgt_xyz, const float* tgt_normals,
                        const int32_t* index_query, const int32_t* index_match, int64_t n_corr,
                        const float src_q[4], const float src_t[3],
                        const float tgt_q[4], const float tgt_t[3],
                        double H[144], double b[12], double* cost);

/* ---- (A') pcl::NormalEstimationTwoPassOMP (src/geometry/two_pass_normal_3d_omp.h:53-99) - */
/* setInputCloud + setKSearch(k) + setViewPoint + compute  (call sites
 * src/exe/icp_scan_aligner.cc:323-330, src/exe/normal_estimator.cc:177-194).
 * out_normals n x 3, out_curvature n.  knn_indices (optional, n*k int32) receives each point's
 * neighbour list sorted by (squared distance, index).  Pointers may be host or device memory: a
 * cloud in device memory is read in place, and results for device out_normals / out_curvature are
 * written by the kernels directly (no staging copies; the call still returns after its stream
 * has finished).  The library works on its own stream: device inputs must be complete (the
 * caller's stream work that produces them finished) when the call is made. */
int e3d_normals_knn(const float* xyz, size_t n, int k, const float viewpoint[3],
                    float* out_normals, float* out_curvature, int32_t* knn_indices);

/* Test hook: the bit-defined elementary functions of include/e3d_libm.h evaluated by a HIP kernel (n values, host or device
 * pointers).  fn: 0 atanf(x), 1 atan2f(x, y), 2 sinf(x), 3 cosf(x), 4 tanf(x), 5 log2f(x).  The reference calls the C library
 * at these places (pcl::eigen33 for src/geometry/two_pass_normal_3d.h:92-109, src/camera/camera_base_impl_fisheye.h:66-153,
 * src/opt/visibility_estimator.cc:437); kernels, host code and oracle all use this one implementation instead. */
int e3d_libm_eval(int fn, const float* x, const float* y, size_t n, float* out);

/* e3d_normals_knn / e3d_local_outlier_removal keep their device workspace (the sort, grid and list buffers of the last call, per
 * device) for the next call instead of paying hipMalloc / hipFree every time; this frees what is parked.  Workspaces larger than
 * E3D_WORKSPACE_KEEP_GB (environment, default 32) are never kept.  No counterpart in the reference (PCL allocates per call). */
int e3d_release_workspaces(void);

/* The same estimator with setRadiusSearch(radius) instead of setKSearch: every point strictly within the radius
 * (squared distance < (float)((double)radius * radius)) takes part; fewer than 3 -> NaN.  neighbor_counts (optional, n)
 * receives the number of points found, the query itself included. */
int e3d_normals_radius(const float* xyz, size_t n, float radius, const float viewpoint[3],
                       float* out_normals, float* out_curvature, int32_t* neighbor_counts);

/* pcl::LocalStatisticalOutlierRemoval<PointT>::applyFilterIndices (src/geometry/local_statistical_outlier_removal.hpp:71-172),
 * the filter of PointCloudCleaner (src/exe/point_cloud_cleaner.cc:80-107) and of the multi-resolution pipeline's callers.
 * First pass: per point the mean distance to its mean_k nearest neighbours (exact kNN with k = mean_k + 1, entry 0 being
 * the point itself; f64 sum of the f32 roots of FLANN's squared f32 distances, stored as f32).  Second pass: a point is
 * removed if its own value exceeds distance_factor_threshold x the f64 mean of its neighbours' (positive) values
 * (`negative` inverts the test like setNegative).  inlier[i] = 1 for points the filter keeps; non-finite points are never
 * kept.  mean_distances (optional, n floats) receives the first-pass values.  Neighbours at exactly equal distance are
 * ordered by index (FLANN's order there is unpinned). */
int e3d_local_outlier_removal(const float* xyz, size_t n, int mean_k, double distance_factor_threshold, int negative,
                              uint8_t* inlier, float* mean_distances);

/* ---- (B) ImageRegistrator: dense photometric residual / Jacobian kernels -------------------------------------
 * Device-resident mirror of the parts of opt::Problem the hot loops read (src/opt/problem.h:300-388) and the inner
 * operator surfaces of the optimizer:
 *   OcclusionGeometry::RenderDepthMap (CPU-splat path)      src/opt/occlusion_geometry.cc:404-464      -> e3d_reg_render_depth
 *   VisibilityEstimator::AppendObservationsFor...           src/opt/visibility_estimator.cc:258-295,366-532 -> e3d_reg_observe
 *   VisibilityEstimator::DetermineIfAllNeighborsAreObserved src/opt/visibility_estimator.cc:199-256    -> (inside e3d_reg_observe)
 *   IntrinsicsAndPoseOptimizer::AccumulateHAndBAndResidualsForObservations
 *                                                           src/opt/intrinsics_and_pose_optimizer.cc:624-1296 -> e3d_reg_accumulate
 *   CostCalculator::AccumulateResidualsForObservations      src/opt/cost_calculator.cc:102-271         -> e3d_reg_cost
 *   ColorOptimizer::Apply                                   src/opt/color_optimizer.cc:40-123          -> e3d_reg_color_*
 * Coverage: every camera model of the reference's factory, non-rig and rig images, colour residuals (fixed + variable descriptors),
 * image and camera masks, depth-map residuals (off by default, as in the reference; not for the dependent images of a rig, which the
 * reference aborts on). */
typedef struct e3d_reg e3d_reg_t;

typedef struct {
  int32_t point_neighbor_count;        /* K, opt::Parameters::point_neighbor_count (parameters.h:42)   */
  int32_t robust_weighting_type;       /* 0 none, 1 Huber, 2 Tukey (robust_weighting.h:40-44)           */
  float   robust_weighting_parameter;
  float   fixed_residuals_weight;
  float   variable_residuals_weight;
  float   maximum_valid_intensity;     /* 252                                                           */
  float   occlusion_depth_threshold;   /* 0.01                                                          */
  float   splat_radius;                /* 0.03                                                          */
  int32_t current_image_scale;         /* Problem::current_image_scale()                                */
  int32_t image_scale_count;           /* Problem::image_scale_count()                                  */
  /* depth-based residuals (parameters.h:53-55,165-172: "not used in ETH3D pipeline"; 0 = disabled, the reference's default) */
  float   depth_residuals_weight;
  int32_t depth_robust_weighting_type; /* 2 (Tukey)                                                     */
  float   depth_robust_weighting_parameter;   /* 0.02                                                   */
} e3d_reg_params;

/* camera models (COLMAP names, src/camera/camera_base.cc:66-77) and their parameter counts:
 * PINHOLE fx fy cx cy (camera_pinhole.h:40-86); OPENCV + k1 k2 p1 p2 (camera_polynomial_tangential.h:41-159);
 * THIN_PRISM_FISHEYE + k1 k2 p1 p2 k3 k4 sx1 sy1 (camera_benchmark.h:44-52); OPENCV_FISHEYE + k1 k2 k3 k4
 * (camera_fisheye_polynomial_4.h:42-50 over camera_polynomial_4.h:43-135); FOV + omega (camera_fisheye_fov.h:44-176);
 * SIMPLE_PINHOLE f cx cy (camera_simple_pinhole.h:41-88); SIMPLE_RADIAL f cx cy k (camera_simple_radial.h:43-110); RADIAL f cx cy k1 k2
 * (camera_radial.h:43-123); POLYNOMIAL_3 fx fy cx cy k1 k2 k3 (camera_polynomial.h:43-127); FISHEYE_POLYNOMIAL_2_TANGENTIAL_2 fx fy cx cy
 * k1 k2 p1 p2 (camera_fisheye_polynomial_tangential.h).  Parameter vectors, Jacobian columns and e3d_reg_get_intrinsics_level follow the
 * class's GetParameters order. */
#define E3D_CAMERA_PINHOLE 0             /* I = 4  */
#define E3D_CAMERA_OPENCV 1              /* I = 8  */
#define E3D_CAMERA_THIN_PRISM_FISHEYE 2  /* I = 12 */
#define E3D_CAMERA_OPENCV_FISHEYE 3      /* I = 8  */
#define E3D_CAMERA_FOV 4                 /* I = 5  */
#define E3D_CAMERA_SIMPLE_PINHOLE 5      /* I = 3: f cx cy                  (one focal length: parameter order as in COLMAP) */
#define E3D_CAMERA_SIMPLE_RADIAL 6       /* I = 4: f cx cy k                (also the name SIMPLE_RADIAL_FISHEYE, camera_base.cc:74) */
#define E3D_CAMERA_RADIAL 7              /* I = 5: f cx cy k1 k2            (also the name RADIAL_FISHEYE, camera_base.cc:73) */
#define E3D_CAMERA_POLYNOMIAL_3 8        /* I = 7: fx fy cx cy k1 k2 k3 */
#define E3D_CAMERA_FISHEYE_POLYNOMIAL_2_TANGENTIAL_2 9   /* I = 8: fx fy cx cy k1 k2 p1 p2 */
/* the three classes of src/camera that the reference's factory never creates (camera_base.cc:66-77): no camera name reaches them */
#define E3D_CAMERA_FULL_OPENCV 10        /* I = 12: fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6   (camera_full_opencv.h:41-196) */
#define E3D_CAMERA_RADIAL_FISHEYE_CLASS 11         /* I = 5: f cx cy k1 k2   RadialFisheyeCamera = FisheyeBase over RadialCamera */
#define E3D_CAMERA_SIMPLE_RADIAL_FISHEYE_CLASS 12  /* I = 4: f cx cy k       SimpleRadialFisheyeCamera = FisheyeBase over SimpleRadialCamera */

e3d_reg_t* e3d_reg_create(const e3d_reg_params* params);
void e3d_reg_destroy(e3d_reg_t* reg);
int e3d_reg_set_params(e3d_reg_t* reg, const e3d_reg_params* params);

/* One point scale of the multi-resolution cloud: points()[s], point radius, neighbor_point_indices_[s] (n*K, u32),
 * fixed_descriptors()[s] (n*K or NULL).  Variable descriptors start at 0 and observation counts at 99999 when fixed
 * descriptors are given (problem.cc:549-572), else 0. */
int e3d_reg_set_point_scale(e3d_reg_t* reg, int point_scale, const float* xyz, size_t n, float point_radius,
                            const uint32_t* neighbor_indices, const float* fixed_descriptors);
int e3d_reg_set_variable_descriptors(e3d_reg_t* reg, int point_scale, const float* descriptors,
                                     const int32_t* observation_counts);
int e3d_reg_get_variable_descriptors(e3d_reg_t* reg, int point_scale, float* descriptors, int32_t* observation_counts);

/* opt::Intrinsics: the model of the best available scale + its ScaledBy(0.5) pyramid of n_levels models
 * (intrinsics.cc:46-51, camera_base_impl.h:70-89) and radius cut-offs (camera_base_impl.h:410-463). */
int e3d_reg_set_intrinsics(e3d_reg_t* reg, int intrinsics_id, int camera_type, int width, int height,
                           const float* parameters, int n_parameters, int min_image_scale, int n_levels);

/* Intrinsics::camera_mask (src/opt/intrinsics.h:104; loaded once per camera by Image::LoadImageData, src/opt/image.cc:62-72): one u8
 * mask per pyramid level of the camera (level_masks[l]: width_l x height_l bytes, host or device memory, or NULL), shared by all
 * images of these intrinsics.  An observation is dropped where the image's own mask OR the camera mask is non-zero
 * (src/opt/visibility_estimator.cc:335-345, 482-503).  level_masks == NULL removes the mask.  Call after e3d_reg_set_intrinsics;
 * parameter updates of the optimiser keep it. */
int e3d_reg_set_camera_mask(e3d_reg_t* reg, int intrinsics_id, const uint8_t* const* level_masks);

/* Depth residuals (src/opt/intrinsics_and_pose_optimizer.cc:747-757, 1150-1214; cost_calculator.cc:221-245; problem.cc:593-631):
 * residual = 1 / (depth map interpolated at the observation) - 1 / (depth of the point in the image frame), one per observation,
 * weighted by e3d_reg_params.depth_residuals_weight with its own robust weighting.  Off by default and unused by the reference's
 * tools (no tool loads depth maps; its alignment test does, test_alignment.cc:469-500).  As in the reference the non-reference
 * images of a rig are not supported with depth residuals (:1199-1207 aborts) -- here that is an error return.
 *   e3d_reg_set_depth_maps: Problem::SetFixedDepthMaps for one image, one f32 map per pyramid level of its camera (caller-built
 *     pyramid, like the test's cv::resize INTER_AREA chain); NULL removes them.  Required for every image once the weight is > 0.
 *   e3d_reg_depth_accumulate / e3d_reg_depth_cost: the depth part of AccumulateHAndBForImage / ComputeResidualsForImage for one
 *     (image, point scale); e3d_reg_apply, e3d_reg_compute_cost and e3d_reg_run_on_current_scale include them by themselves. */
int e3d_reg_set_depth_maps(e3d_reg_t* reg, int image_id, const float* const* level_depths);
int e3d_reg_depth_accumulate(e3d_reg_t* reg, int image_id, int point_scale, double* H, double* b, double* sum, int64_t* count);
int e3d_reg_depth_cost(e3d_reg_t* reg, int image_id, int point_scale, double* sum, int64_t* count);
/* queries one level of the pyramid the library built: size, parameters (n_parameters floats) and the radius cut-off
 * (+inf for PINHOLE; for the fisheye models the cut-off of the inner non-fisheye model, which is the one projection tests) */
int e3d_reg_get_intrinsics_level(e3d_reg_t* reg, int intrinsics_id, int level, int* width, int* height,
                                 float* parameters, float* radius_cutoff_squared);
/* opt::Image: u8 pyramid (level l has the size of intrinsics level l) and optional masks; pose image_T_global as the
 * Sophus::SE3f state: unit quaternion {w, x, y, z} + translation (what COLMAP images.txt stores). */
int e3d_reg_set_image(e3d_reg_t* reg, int image_id, int intrinsics_id, const uint8_t* const* level_pixels,
                      const uint8_t* const* level_masks);
int e3d_reg_set_image_pose(e3d_reg_t* reg, int image_id, const float q[4], const float t[3]);
int e3d_reg_get_image_pose(e3d_reg_t* reg, int image_id, float q[4], float t[3]);
/* Camera rigs (src/opt/rig.h:41-76, RigImages in src/opt/problem.h): image_T_rig of every camera (n_cameras x {w,x,y,z},
 * n_cameras x 3; camera 0 is the reference), and frames = one image id per camera, image_ids[0] being the reference
 * image whose pose is the rig pose.  The other images' poses are derived: image_T_rig[c] * image_T_global(reference)
 * (intrinsics_and_pose_optimizer.cc:539-548); they contribute the 6 extrinsics unknowns of their camera and the 6 pose
 * unknowns of the reference image (:140-153, Rig::Update rig.cc:9-23). */
int e3d_reg_set_rig(e3d_reg_t* reg, int rig_id, int n_cameras, const float* q, const float* t);
int e3d_reg_get_rig(e3d_reg_t* reg, int rig_id, int camera_index, float q[4], float t[3]);
int e3d_reg_add_rig_images(e3d_reg_t* reg, int rig_id, const int* image_ids, int n_cameras);
/* OcclusionGeometry::SetSplatPoints */
int e3d_reg_set_splat_points(e3d_reg_t* reg, const float* xyz, size_t n);

/* OcclusionGeometry::AddMesh / AddSplats (src/opt/occlusion_geometry.cc:64-182): occlusion geometry as triangle meshes,
 * used when no splat points are set (OcclusionGeometry::RenderDepthMap :211-271).  The reference renders them with OpenGL
 * (src/opengl/renderer.cc); here a software rasteriser with the same conventions (vertex-shader distortion per camera
 * model, depth = camera-space z, nearest fragment, 0 where there is no geometry, near / far planes min / max occlusion
 * depth) followed by MaskOutOcclusionBoundaries (:284-402) over the edges extracted when compute_edges != 0
 * (ComputeEdgeNormalsList / FilterEdgeList :488-645).  Vertices are global-frame xyz, triangles 3 x u32.  Returns the
 * number of meshes held. */
int e3d_reg_add_occlusion_mesh(e3d_reg_t* reg, const float* vertices, size_t n_vertices, const uint32_t* triangles,
                               size_t n_triangles, int compute_edges);
int e3d_reg_clear_occlusion_meshes(e3d_reg_t* reg);
int e3d_reg_set_occlusion_options(e3d_reg_t* reg, float min_depth, float max_depth, int mask_occlusion_boundaries);
int64_t e3d_reg_occlusion_edge_count(e3d_reg_t* reg, int mesh_index);

/* Measurement: HIP-event durations of the two kernels behind e3d_reg_accumulate, summed over its calls since the last reset:
 * out = {pass 1 ms (k_reg_pass1: per-observation intensity + Jacobian rows), pass 2 ms (k_reg_pass2 / k_reg_pass2_mfma: residuals and
 * normal equations), observations processed, calls}. */
int e3d_reg_kernel_times(e3d_reg_t* reg, double out[4], int reset);
/* Wall-clock split of e3d_reg_run_on_current_scale by phase (accumulate, host solve, trial states, re-projection, costs, occlusion
 * depth maps, visibility, colour update).  enable = 1 switches it on (the library then synchronises its stream at every phase
 * boundary: the run gets slower, the split adds up), 0 off; `out` (may be NULL) receives "phase=milliseconds;..." of the phases
 * recorded so far, followed (ABI 4) by one "k:group=milliseconds,launches,units;" entry per kernel group of the iteration (HIP events
 * on the handle's stream, no synchronisation of their own; units = the points, observations, pixels or triangles the launches
 * covered -- what bench.py prices a group's algorithmic bytes with).  Switching clears the record.  Measurement only -- no effect on
 * results. */
int e3d_reg_profile(e3d_reg_t* reg, int enable, char* out, size_t capacity);

/* Renders the occlusion depth map of an image at an image scale (kept on the device for e3d_reg_observe);
 * depth_out (optional) receives height x width floats. */
int e3d_reg_render_depth(e3d_reg_t* reg, int image_id, int image_scale, float* depth_out);
/* Creates the observations of a point scale in an image (indices == NULL: every point, with occlusion, mask and
 * over-saturation tests against the last rendered depth map of this image and scale; indices != NULL: the given
 * visibility list without those tests) in point order, and their all-neighbours-observed flags.  Returns the count. */
int64_t e3d_reg_observe(e3d_reg_t* reg, int image_id, int point_scale, int image_scale, int border_size,
                        const uint32_t* indices, size_t n_indices);
int e3d_reg_get_observations(e3d_reg_t* reg, int image_id, int point_scale, uint32_t* point_index, float* x, float* y,
                             float* image_scale, uint8_t* all_neighbors_observed);
int e3d_reg_set_observations(e3d_reg_t* reg, int image_id, int point_scale, size_t n, const uint32_t* point_index,
                             const float* x, const float* y, const float* image_scale);
/* ComputePointIntensityAndJacobians for every observation (n x 1, n x I, n x 6); mainly for tests. */
int e3d_reg_pass1(e3d_reg_t* reg, int image_id, int point_scale, float* intensities, float* j_intrinsics, float* j_pose);
/* H: V x V row-major, upper triangle filled; b: V.  V = I + 6 with variables [intrinsics(I), pose(6)], or V = I + 12 with
 * [intrinsics(I), rig extrinsics(6), pose of the rig frame's reference image(6)] for a non-reference rig image;
 * sums / counts: [fixed, variable] robust residual sums and residual counts. */
int e3d_reg_accumulate(e3d_reg_t* reg, int image_id, int point_scale, double* H, double* b, double sums[2],
                       int64_t counts[2]);
int e3d_reg_cost(e3d_reg_t* reg, int image_id, int point_scale, double sums[2], int64_t counts[2]);
int e3d_reg_color_begin(e3d_reg_t* reg, int point_scale);
int e3d_reg_color_accumulate(e3d_reg_t* reg, int image_id, int point_scale);
int e3d_reg_color_finish(e3d_reg_t* reg, int point_scale);

/* Whole-problem steps of opt::Optimizer::RunOnCurrentScale (src/opt/optimizer.cc:49-182) on the device-resident state.
 * Images are visited in ascending image id. */
/* VisibilityEstimator::CreateObservationsForAllImages + DetermineIfAllNeighborsAreObserved (optimizer.cc:119-128); with
 * e3d_reg_set_cache_observations(1) it is ObservationsCache::GetObservations (observations_cache.cc:52-68) instead. */
int e3d_reg_update_observations(e3d_reg_t* reg, int border_size);
/* GroundTruthCreator (src/exe/ground_truth_creator.cc): visibility of the full-resolution scan points in the registered images.
 *   e3d_reg_set_scan_points: the (global-frame) scan points whose observations are counted; counts start at 0.
 *   e3d_reg_count_scan_observations: AccumulateScanObservationsForImage (:45-86) for one image -- occlusion depth map at the
 *     highest available resolution (RenderDepthMap at intrinsics.min_image_scale), then for every scan point: z > 0, rounded
 *     pixel inside the image, occlusion_image + occlusion_depth_threshold >= z, and mask(iy, ix) != excluded_flag (pass the
 *     level-0 image mask and opt::MaskType::kEvalObs = 2; NULL = no mask) -> count += 1.
 *   e3d_reg_get/set_scan_observation_counts: the counters (one int per scan point, input order).  With image sharding every
 *     rank counts its own images; the caller adds the vectors.
 *   e3d_reg_ground_truth_depth: the depth-map part of CreateGroundTruthForImage (:104-117, :146-189): gt_depth (w x h floats,
 *     +inf where nothing was seen) = min z over the visible points with count >= min_count (the tool uses 2); occlusion_depth
 *     (optional) receives the occlusion depth map the tool writes to occlusion_depth/. */
int e3d_reg_set_scan_points(e3d_reg_t* reg, const float* xyz, size_t n);
int e3d_reg_count_scan_observations(e3d_reg_t* reg, int image_id, const uint8_t* mask, int excluded_flag);
int e3d_reg_get_scan_observation_counts(e3d_reg_t* reg, int32_t* counts);
int e3d_reg_set_scan_observation_counts(e3d_reg_t* reg, const int32_t* counts);
int e3d_reg_ground_truth_depth(e3d_reg_t* reg, int image_id, const uint8_t* mask, int excluded_flag, int min_count, float* gt_depth,
                               float* occlusion_depth);
/* CreateGroundTruthForImage, scan rendering part (src/exe/ground_truth_creator.cc:149,175-187): the reference paints a square of
 * 2 * point_radius + 1 pixels over the image for every visible scan point seen in >= min_count images, in point order.  The result
 * per pixel is the LAST point covering it: winner[y * width + x] = point index + 1 (order of e3d_reg_set_scan_points), 0 = untouched. */
int e3d_reg_scan_rendering(e3d_reg_t* reg, int image_id, const uint8_t* mask, int excluded_flag, int min_count, int point_radius,
                           uint32_t* winner);
/* Observations cache (src/opt/observations_cache.{h,cc}; Optimizer::set_cache_observations, optimizer.h).  When enabled, the
 * observation update re-projects a fixed per-image list of point indices with the current state and applies only the
 * scale-fit and border tests (VisibilityEstimator::AppendObservationsForIndexedPointsVisibleInImage,
 * visibility_estimator.cc:140-168, :367-403) -- no occlusion rendering, masks or over-saturation test.
 *   e3d_reg_determine_observed_indices: ObservationsCache::DetermineAndSaveObservedPointIndices (observations_cache.cc:
 *     104-125) minus the files -- a full visibility pass at image scale 0 with the current state fills the lists (of the
 *     images this rank owns).  e3d_reg_run_on_current_scale calls it by itself when caching is on and lists are missing.
 *   e3d_reg_get_observed_indices: returns the length of the list of (image, point scale) and, if `indices` is not NULL,
 *     copies it out as the std::size_t values the `.observed_indices` files hold (observations_cache.cc:146-156).
 *   e3d_reg_set_observed_indices: installs a list read from such a file (observations_cache.cc:70-102). */
int e3d_reg_set_cache_observations(e3d_reg_t* reg, int enabled);
int e3d_reg_determine_observed_indices(e3d_reg_t* reg);
int64_t e3d_reg_get_observed_indices(e3d_reg_t* reg, int image_id, int point_scale, uint64_t* indices);
int e3d_reg_set_observed_indices(e3d_reg_t* reg, int image_id, int point_scale, const uint64_t* indices, size_t count);
/* ColorOptimizer::Apply (color_optimizer.cc:40-123) */
int e3d_reg_color_update(e3d_reg_t* reg);
/* CostCalculator::ComputeCost (cost_calculator.cc:44-100) */
int e3d_reg_compute_cost(e3d_reg_t* reg, double* cost);
/* IntrinsicsAndPoseOptimizer::Apply (intrinsics_and_pose_optimizer.cc:48-259): one LM step with <= 10 tries */
int e3d_reg_apply(e3d_reg_t* reg, int print_progress, int* applied_update, float* lambda, float* max_change);
/* bool Optimizer::RunOnCurrentScale(max_num_iterations, max_change_convergence_threshold,
 *   iterations_without_new_optimum_threshold, <observations_cache_path: host side>, print_progress, &optimum_cost); returns 1 if
 * converged, 0 if not.  The state (intrinsics, poses) is left at the optimum, like the reference. */
int e3d_reg_run_on_current_scale(e3d_reg_t* reg, int max_num_iterations, float max_change_convergence_threshold,
                                 int iterations_without_new_optimum_threshold, int print_progress, double* optimum_cost,
                                 int* iterations_done);

/* Problem::DeterminePointNeighbors (src/opt/problem.cc:706-786): neighbor_indices[p * neighbor_count + j] = the j-th of the
 * shuffled (candidate_count + 1)-nearest-neighbour candidates of point p (the point itself excluded), searched within p's
 * own scan when limit_to_same_scan != 0 (fixed scan colours).  Same libstdc++ std::shuffle / std::mt19937(0) stream as the
 * reference. */
int e3d_determine_point_neighbors(const float* xyz, size_t n, const uint8_t* scan_indices, int scan_count,
                                  int limit_to_same_scan, int neighbor_count, int candidate_count,
                                  uint32_t* neighbor_indices);

/* ComputeMinMaxPointRadius over all images (src/opt/multi_scale_point_cloud.cc:126-180,236-262): for every point the smallest
 * radius that projects to half a pixel in some image (min_radius, +inf if never observed) and the largest such radius
 * divided by the minimum scaling factor 2^-(image_scale_count - 1) (max_radius, -inf if never observed).  Uses the images,
 * intrinsics, splat points and parameters already set on the handle (current_image_scale = 0 at set-up time). */
int e3d_reg_point_radius_minmax(e3d_reg_t* reg, const float* xyz, size_t n, float* min_radius, float* max_radius);

/* MergeClosePoints (src/opt/multi_scale_point_cloud.cc:44-124): greedy merge in point order -- every point not yet absorbed
 * becomes a centre and absorbs all points strictly within merge_distance -- computed in parallel (the centres are the
 * lexicographically first maximal independent set).  Outputs (capacity n each) in centre order: mean position, mean colour of
 * the scan with most merged points, that scan's index, maximum of max_radius.  Returns the number of output points. */
int64_t e3d_merge_close_points(float merge_distance, int num_scans, const float* xyz, const float* colors,
                               const uint8_t* scan_indices, const float* max_radius, size_t n, float* out_xyz,
                               float* out_colors, uint8_t* out_scan_indices, float* out_max_radius);

/* Multi-GPU (one process per GPU): images are sharded, image `id` belongs to rank `id mod world_size`
 * (e3d_reg_image_owner).  Every rank declares every intrinsics block, point scale and image (ids and poses) so that the
 * variable layout is global, but only the owner of an image uploads its pyramid -- e3d_reg_set_image accepts
 * level_pixels == NULL for images of other ranks -- and only the owner creates observations and residuals for it.
 * Exchange steps: the dense H, b and the residual sums / counts once per Apply and 4 doubles per LM try and
 * per cost evaluation (`allreduce`, HOST f64 buffer); the variable descriptors (K*N f32) and observation counts (N i32)
 * of every point scale once per colour update (`allreduce_device`: DEVICE buffer, dtype 0 = f32, 1 = i32, in-place sum;
 * the library has synchronised its stream before the call and the callback must return with the result complete).
 * Both callbacks must leave the identical result on every rank; all ranks then take the same LM decisions. */
typedef int (*e3d_allreduce_device_fn)(void* device_buffer, size_t count, int dtype, void* user);
int e3d_reg_set_shard(e3d_reg_t* reg, int rank, int world_size, e3d_allreduce_fn allreduce,
                      e3d_allreduce_device_fn allreduce_device, void* user);
/* The same sharding with the library's own RCCL communicator (see e3d_comm_create): the block-sparse normal equations, the
 * residual sums and the variable descriptors are all-reduced on the handle's stream; call before the images are set. */
int e3d_reg_set_comm(e3d_reg_t* reg, e3d_comm_t* comm);
int e3d_reg_image_owner(e3d_reg_t* reg, int image_id);

#ifdef __cplusplus
}
#endif
#endif  /* E3D_HIP_H */
