/*
 * oracle/oracle_normals.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, path (A')).
 *
 * Restates
 *   pcl::NormalEstimationTwoPassOMP::computeFeature   src/geometry/two_pass_normal_3d_omp.hpp:48-119
 *   pcl::computePointNormalTwoPass (indices)          src/geometry/two_pass_normal_3d.h:92-109
 *   pcl::computeMeanAndCovarianceMatrixTwoPass        src/geometry/two_pass_centroid.hpp:155-259
 * and the PCL 1.10 pieces they call that are NOT in /root/reference (recalled;
 * "parity unpinned" -- the reference has no test that touches src/geometry):
 *   pcl::solvePlaneParameters / pcl::eigen33 / computeRoots / computeRoots2
 *   pcl::flipNormalTowardsViewpoint
 *   pcl::Feature::searchForNeighbors -> KdTreeFLANN::nearestKSearch / radiusSearch
 * Neighbour order = ascending (f32 squared distance, index); the query point
 * itself (distance 0) is part of its neighbourhood.
 */
#include "e3d_oracle.h"
#include "oracle_kdtree.h"
#include "../include/e3d_libm.h"   /* bit-defined atan2f / cosf / sinf shared with the HIP kernels */

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_libm_select.h"

static void compute_roots2(float b, float c, float* roots) {
  roots[0] = 0.f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0) d = 0.0f;
  float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

static void compute_roots(const float* m /*row-major 3x3*/, float* roots) {
#define M(i, j) m[3 * (i) + (j)]
  float c0 = M(0, 0) * M(1, 1) * M(2, 2) + 2.f * M(0, 1) * M(0, 2) * M(1, 2) -
             M(0, 0) * M(1, 2) * M(1, 2) - M(1, 1) * M(0, 2) * M(0, 2) - M(2, 2) * M(0, 1) * M(0, 1);
  float c1 = M(0, 0) * M(1, 1) - M(0, 1) * M(0, 1) + M(0, 0) * M(2, 2) - M(0, 2) * M(0, 2) +
             M(1, 1) * M(2, 2) - M(1, 2) * M(1, 2);
  float c2 = M(0, 0) + M(1, 1) + M(2, 2);
#undef M
  if (fabsf(c0) < FLT_EPSILON) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = sqrtf(3.0f);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  float rho = sqrtf(-a_over_3);
  float theta = om_atan2f(sqrtf(-q), half_b) * s_inv3;
  float cos_theta = om_cosf(theta);
  float sin_theta = om_sinf(theta);
  roots[0] = c2_over_3 + 2.f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  if (roots[1] >= roots[2]) {
    t = roots[1]; roots[1] = roots[2]; roots[2] = t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  }
  if (roots[0] <= 0.f) compute_roots2(c2, c1, roots);
}

static inline void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline float sqnorm3(const float* a) {
  float e0 = a[0] * a[0], e1 = a[1] * a[1], e2 = a[2] * a[2];
  return e0 + (e1 + e2);
}

/* pcl::eigen33(mat, eigenvalue, eigenvector): smallest eigenpair */
static void eigen33(const float* cov, float* eigenvalue, float* v) {
  float scale = 0.f;
  for (int i = 0; i < 9; ++i) { float a = fabsf(cov[i]); if (a > scale) scale = a; }
  if (scale <= FLT_MIN) scale = 1.0f;
  float s[9];
  for (int i = 0; i < 9; ++i) s[i] = cov[i] / scale;
  float roots[3];
  compute_roots(s, roots);
  *eigenvalue = roots[0] * scale;
  s[0] -= roots[0]; s[4] -= roots[0]; s[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  cross3(s + 0, s + 3, v1);
  cross3(s + 0, s + 6, v2);
  cross3(s + 3, s + 6, v3);
  float l1 = sqnorm3(v1), l2 = sqnorm3(v2), l3 = sqnorm3(v3);
  const float* best; float len;
  if (l1 >= l2 && l1 >= l3) { best = v1; len = l1; }
  else if (l2 >= l1 && l2 >= l3) { best = v2; len = l2; }
  else { best = v3; len = l3; }
  float sl = sqrtf(len);
  v[0] = best[0] / sl; v[1] = best[1] / sl; v[2] = best[2] / sl;
}

void oracle_point_normal(const float* xyz, const int32_t* indices, int count, float plane[4], float* curvature) {
  if (count < 3) {
    plane[0] = plane[1] = plane[2] = plane[3] = NAN; *curvature = NAN;
    return;
  }
  /* two_pass_centroid.hpp:162-192 (dense branch) */
  float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < count; ++i) {
    const float* p = xyz + 3 * (size_t)indices[i];
    accu[6] += p[0]; accu[7] += p[1]; accu[8] += p[2];
  }
  float fc = (float)count;
  for (int i = 0; i < 9; ++i) accu[i] = accu[i] / fc;
  for (int i = 0; i < count; ++i) {
    const float* p = xyz + 3 * (size_t)indices[i];
    accu[0] += (p[0] - accu[6]) * (p[0] - accu[6]);
    accu[1] += (p[0] - accu[6]) * (p[1] - accu[7]);
    accu[2] += (p[0] - accu[6]) * (p[2] - accu[8]);
    accu[3] += (p[1] - accu[7]) * (p[1] - accu[7]);
    accu[4] += (p[1] - accu[7]) * (p[2] - accu[8]);
    accu[5] += (p[2] - accu[8]) * (p[2] - accu[8]);
  }
  float cov[9];
  cov[0] = accu[0] / fc; cov[1] = accu[1] / fc; cov[2] = accu[2] / fc;
  cov[4] = accu[3] / fc; cov[5] = accu[4] / fc; cov[8] = accu[5] / fc;
  cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
  /* solvePlaneParameters */
  float ev, n[3];
  eigen33(cov, &ev, n);
  plane[0] = n[0]; plane[1] = n[1]; plane[2] = n[2];
  /* plane_parameters[3] = -1 * plane_parameters.dot(point), 4-vector SSE dot with w terms 0*1 */
  {
    float e0 = n[0] * accu[6], e1 = n[1] * accu[7], e2 = n[2] * accu[8], e3 = 0.f * 1.f;
    plane[3] = -1.f * ((e0 + e2) + (e1 + e3));
  }
  float eig_sum = cov[0] + cov[4] + cov[8];
  if (eig_sum != 0) *curvature = fabsf(ev / eig_sum);
  else *curvature = 0;
}

static inline void flip_towards(const float* p, const float* vp, float* n) {
  float vx = vp[0] - p[0], vy = vp[1] - p[1], vz = vp[2] - p[2];
  float cos_theta = (vx * n[0] + vy * n[1] + vz * n[2]);
  if (cos_theta < 0) { n[0] *= -1; n[1] *= -1; n[2] *= -1; }
}

/* sample == NULL: every point is a query (outputs indexed by point); else the n_sample listed points are (outputs indexed by
 * position in the list) -- the same tree over the WHOLE cloud either way, so a sample of a 20 M point scan is checked against
 * exactly what the full run would return for those points */
static int normals_impl(const float* xyz, size_t n, int k, float radius, const float vp[3], const int64_t* sample, size_t n_sample,
                        float* out_normal, float* out_curv, int32_t* knn_idx_out) {
  int use_radius = (radius > 0.f);
  float r2 = 0.f;
  if (use_radius) { double r = (double)radius; r2 = (float)(r * r); }
  if (!use_radius && k <= 0) return -1;
  okd_tree* tree = okd_build(xyz, n);
  const long long nq = sample ? (long long)n_sample : (long long)n;
#pragma omp parallel
  {
    int cap = use_radius ? 4096 : k;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    float* dist = (float*)malloc(sizeof(float) * (size_t)cap);
#pragma omp for schedule(dynamic, 1024)
    for (long long i = 0; i < nq; ++i) {
      const float* q = xyz + 3 * (size_t)(sample ? sample[i] : i);
      int cnt;
      if (use_radius) {
        cnt = okd_radius(tree, q, r2, cap, idx, dist);
        if (cnt > cap) {
          cap = cnt;
          idx = (int32_t*)realloc(idx, sizeof(int32_t) * (size_t)cap);
          dist = (float*)realloc(dist, sizeof(float) * (size_t)cap);
          cnt = okd_radius(tree, q, r2, cap, idx, dist);
        }
      } else {
        cnt = okd_knn(tree, q, k, idx, dist);
        if (knn_idx_out) {
          for (int j = 0; j < k; ++j) knn_idx_out[(size_t)i * k + j] = (j < cnt) ? idx[j] : -1;
        }
      }
      float* on = out_normal + 3 * (size_t)i;
      if (cnt == 0) { on[0] = on[1] = on[2] = NAN; out_curv[i] = NAN; continue; }
      float plane[4];
      oracle_point_normal(xyz, idx, cnt, plane, &out_curv[i]);
      on[0] = plane[0]; on[1] = plane[1]; on[2] = plane[2];
      flip_towards(q, vp, on);
    }
    free(idx); free(dist);
  }
  okd_free(tree);
  return 0;
}

/* NormalEstimationTwoPassOMP::computeFeature (src/geometry/two_pass_normal_3d_omp.hpp:48-119) */
int oracle_normals(const float* xyz, size_t n, int k, float radius, const float vp[3],
                   float* out_normal, float* out_curv, int32_t* knn_idx_out) {
  return normals_impl(xyz, n, k, radius, vp, NULL, 0, out_normal, out_curv, knn_idx_out);
}
int oracle_normals_sample(const float* xyz, size_t n, int k, float radius, const float vp[3], const int64_t* sample, size_t n_sample,
                          float* out_normal, float* out_curv, int32_t* knn_idx_out) {
  return normals_impl(xyz, n, k, radius, vp, sample, n_sample, out_normal, out_curv, knn_idx_out);
}

void oracle_knn(const float* xyz, size_t n, const float* queries, size_t n_q, int k, int32_t* idx, float* dist) {
  okd_tree* tree = okd_build(xyz, n);
#pragma omp parallel for schedule(dynamic, 1024)
  for (long long i = 0; i < (long long)n_q; ++i) {
    int cnt = okd_knn(tree, queries + 3 * (size_t)i, k, idx + (size_t)i * k, dist + (size_t)i * k);
    for (int j = cnt; j < k; ++j) { idx[(size_t)i * k + j] = -1; dist[(size_t)i * k + j] = INFINITY; }
  }
  okd_free(tree);
}

/* pcl::LocalStatisticalOutlierRemoval<PointT>::applyFilterIndices (src/geometry/local_statistical_outlier_removal.hpp:71-172)
 * for a cloud of finite points with indices_ = the whole cloud.  `sqrt (nn_dists[k])` on a float resolves to the float
 * overload when <math.h> is visible (recalled; unpinned -- the reference has no test of this filter). */
int oracle_local_outlier_removal(const float* xyz, size_t n, int mean_k, double factor, int negative, uint8_t* inlier,
                                 float* distances) {
  /* fewer than mean_k + 1 points: pcl::KdTreeFLANN::nearestKSearch clamps k to the cloud size and shrinks nn_indices / nn_dists;
   * the first pass then reads nn_dists[k] beyond the shrunk size (stale zeros of the vector's initial contents -- restated as 0
   * here), the second pass walks nn_indices.size () entries (:129) */
  if (n == 0) return 0;
  const int k = mean_k + 1;
  const int found = (n < (size_t)k) ? (int)n : k;
  okd_tree* tree = okd_build(xyz, n);
  int32_t* nn = (int32_t*)malloc(sizeof(int32_t) * n * (size_t)k);
  /* first pass (:85-110) */
#pragma omp parallel
  {
    float* nd = (float*)malloc(sizeof(float) * (size_t)k);
#pragma omp for schedule(dynamic, 1024)
    for (long long i = 0; i < (long long)n; ++i) {
      okd_knn(tree, xyz + 3 * (size_t)i, found, nn + (size_t)i * k, nd);
      double dist_sum = 0.0;
      for (int j = 1; j < found; ++j) dist_sum += sqrtf(nd[j]);
      distances[i] = (float)(dist_sum / mean_k);
    }
    free(nd);
  }
  /* second pass (:113-160) */
  for (size_t i = 0; i < n; ++i) {
    int valid = 0;
    double sum = 0;
    for (int j = 1; j < found; ++j) {
      const double d = distances[nn[i * k + j]];
      if (d > 0) { ++valid; sum += d; }
    }
    const double mean = sum / (double)valid;
    const double threshold = mean * factor;
    const int removed = (!negative && distances[i] > threshold) || (negative && distances[i] <= threshold);
    inlier[i] = removed ? 0 : 1;
  }
  free(nn);
  okd_free(tree);
  return 0;
}

/* ---- test hook: the shared bit-defined elementary functions (include/e3d_libm.h) evaluated on the host ---------------- */
/* fn: 0 atanf(x), 1 atan2f(y = x[i], x = y[i]), 2 sinf, 3 cosf, 4 tanf, 5 log2f */
void oracle_libm_eval(int fn, const float* x, const float* y, size_t n, float* out) {
  for (size_t i = 0; i < n; ++i) {
    switch (fn) {
      case 0: out[i] = e3d_atanf(x[i]); break;
      case 1: out[i] = e3d_atan2f(x[i], y[i]); break;
      case 2: out[i] = e3d_sinf(x[i]); break;
      case 3: out[i] = e3d_cosf(x[i]); break;
      case 4: out[i] = e3d_tanf(x[i]); break;
      default: out[i] = e3d_log2f(x[i]); break;
    }
  }
}
