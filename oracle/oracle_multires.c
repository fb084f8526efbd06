/*
 * oracle/oracle_multires.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the multi-resolution point cloud construction of ImageRegistrator:
 *   CameraBaseImpl::InitializeUndistortionLookup / ImageToNormalized   src/camera/camera_base_impl.h:188-211,252-268
 *   ComputeMinMaxPointRadius                                           src/opt/multi_scale_point_cloud.cc:126-180
 *   VisibilityEstimator::_AppendObservationsForImageNoScale            src/opt/visibility_estimator.cc:297-364
 *   MergeClosePoints                                                   src/opt/multi_scale_point_cloud.cc:44-124
 * The multi-scale loop and the filters of Problem::ComputeMultiResPointCloud are orchestrated in oracle/multires.py.
 *
 * Parity unpinned (no reference test; FLANN's unsorted radius-search order decides the f32 summation order and the
 * majority-scan tie in MergeClosePoints): this oracle visits merged neighbours in increasing point index.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "e3d_oracle.h"
#include "oracle_kdtree.h"
#include "oracle_math.h"

static inline int f2i_(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (-2147483647 - 1); }

/* one entry per pixel: Undistort(f_inv * (x, y) + c_inv), row-major, interleaved x y */
void oracle_undistortion_lookup(const oreg_camera* c, float* out) {
  for (int y = 0; y < c->height; ++y)
    for (int x = 0; x < c->width; ++x) {
      float u[2];
      oracle_reg_camera_undistort(c, c->fx_inv * x + c->cx_inv, c->fy_inv * y + c->cy_inv, u, NULL);
      out[2 * ((size_t)y * c->width + x)] = u[0];
      out[2 * ((size_t)y * c->width + x) + 1] = u[1];
    }
}

/* ImageToNormalized(Vector2f) with the lookup table: bilinear.  The reference clamps y to height - 1.00f and then reads row
 * y + 1 (one row past the table when y == height - 1, with weight 0); that row index is clamped here. */
void oracle_image_to_normalized(const oreg_camera* c, const float* lookup, float px, float py, float out[2]) {
  if (c->type == 4) {       /* FisheyeFOVCamera overrides ImageToNormalized: Undistort(ImageToDistorted(p)), closed form (camera_fisheye_fov.h:65-74) */
    oracle_reg_camera_undistort(c, c->fx_inv * px + c->cx_inv, c->fy_inv * py + c->cy_inv, out, NULL);
    return;
  }
  if (c->type == 0 || c->type == 5) {     /* PinholeCamera / SimplePinholeCamera: ImageToDistorted, no table, no clamp (camera_pinhole.h:55-63, camera_simple_pinhole.h:66-74) */
    out[0] = c->fx_inv * px + c->cx_inv; out[1] = c->fy_inv * py + c->cy_inv;
    return;
  }
  float cx = px < c->width - 1.001f ? px : c->width - 1.001f;
  float cy = py < c->height - 1.00f ? py : c->height - 1.00f;
  if (!(cx > 0.f)) cx = 0.f;      /* cwiseMax(0) */
  if (!(cy > 0.f)) cy = 0.f;
  const int ix = (int)cx, iy = (int)cy;
  const float fx = cx - (float)ix, fy = cy - (float)iy;
  const int iy1 = iy + 1 < c->height ? iy + 1 : c->height - 1;
  const float* tl = lookup + 2 * ((size_t)iy * c->width + ix);
  const float* tr = tl + 2;
  const float* bl = lookup + 2 * ((size_t)iy1 * c->width + ix);
  const float* br = bl + 2;
  for (int k = 0; k < 2; ++k)
    out[k] = (1 - fy) * ((1 - fx) * tl[k] + fx * tr[k]) + fy * ((1 - fx) * bl[k] + fx * br[k]);
}

/* ComputeMinMaxPointRadius for one image: observations without scale test at `image_scale` (camera `cam`, occlusion map,
 * mask / saturation of that pyramid level), then the radius that projects to half a pixel at min_image_scale. */
void oracle_point_radius_minmax(const float* pts, size_t n, const float q[4], const float t[3], const oreg_camera* cam,
                                int image_scale, int min_image_scale, const oreg_camera* cam_min, const float* lookup_min,
                                const uint8_t* image_level, const uint8_t* mask_level, const float* occlusion,
                                float occlusion_threshold, float max_valid_intensity, double min_scaling_factor,
                                float* min_radius, float* max_radius) {
  float R[9];
  om_quat_to_R_f(q, R);
  for (size_t i = 0; i < n; ++i) {
    const float* p = pts + 3 * i;
    float pp[3];
    for (int k = 0; k < 3; ++k) pp[k] = (R[3 * k] * p[0] + (R[3 * k + 1] * p[1] + R[3 * k + 2] * p[2])) + t[k];
    if (!(pp[2] > 0.f)) continue;
    const float P3[3] = {pp[0], pp[1], pp[2]};
    float ixy[2];
    oracle_reg_camera_project(cam, P3, ixy);
    const int ix = f2i_(ixy[0] + 0.5f), iy = f2i_(ixy[1] + 0.5f);
    if (!(ixy[0] + 0.5f >= 0 && ixy[1] + 0.5f >= 0 && ix >= 0 && iy >= 0 && ix < cam->width && iy < cam->height)) continue;
    if (!(occlusion[(size_t)iy * cam->width + ix] + occlusion_threshold >= pp[2])) continue;
    if (mask_level && mask_level[(size_t)iy * cam->width + ix] != 0) continue;
    if (image_level[(size_t)iy * cam->width + ix] > max_valid_intensity) continue;
    float returned_scale = image_scale - 1e-6f;
    float ox = ixy[0], oy = ixy[1];
    if (returned_scale < 0.f) {
      returned_scale = 0.f;
      ox = 0.5f * (ox + 0.5f) - 0.5f;
      oy = 0.5f * (oy + 0.5f) - 0.5f;
    }
    /* PointObservation::image_x_at_scale(min_image_scale) */
    const int smaller_scale = (int)returned_scale + 1;
    const float up = (float)pow(2, smaller_scale - min_image_scale);
    const float mx = up * (ox + 0.5f) - 0.5f, my = up * (oy + 0.5f) - 0.5f;
    /* pp again, this time as Sophus::SE3f * point (multi_scale_point_cloud.cc:149) */
    float uv[3], cr[3], g[3];
    om_cross_f(q + 1, p, uv);
    for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
    om_cross_f(q + 1, uv, cr);
    for (int k = 0; k < 3; ++k) g[k] = ((p[k] + q[0] * uv[k]) + cr[k]) + t[k];
    if (!(g[2] > 0.f)) continue;
    const float offx = (mx - 0.5f < 0) ? (mx + 0.5f) : (mx - 0.5f);
    float nxy[2];
    oracle_image_to_normalized(cam_min, lookup_min, offx, my, nxy);
    const float o3[3] = {g[2] * nxy[0], g[2] * nxy[1], g[2] * 1.f};
    const float d0 = g[0] - o3[0], d1 = g[1] - o3[1], d2 = g[2] - o3[2];
    const float point_radius = sqrtf(d0 * d0 + (d1 * d1 + d2 * d2));      /* Eigen Vector3f::norm() */
    if (point_radius < min_radius[i]) min_radius[i] = point_radius;
    const float mr = (float)((double)point_radius / min_scaling_factor);
    if (mr > max_radius[i]) max_radius[i] = mr;
  }
}

static int cmp_i32(const void* a, const void* b) { const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

/* MergeClosePoints: greedy in point order; every not-yet-merged point becomes a centre and absorbs ALL points within the
 * merge distance (merged ones included).  Returns the number of output points. */
size_t oracle_merge_close_points(float merge_distance, int num_scans, const float* pts, const float* colors,
                                 const uint8_t* scan_idx, const float* max_radius, size_t n, float* out_pts,
                                 float* out_colors, uint8_t* out_scan, float* out_max_radius) {
  if (n == 0) return 0;
  okd_tree* tree = okd_build(pts, n);
  const double md = (double)merge_distance;
  const float r2 = (float)(md * md);
  uint8_t* done = (uint8_t*)calloc(n, 1);
  int* count = (int*)malloc(sizeof(int) * num_scans);
  float* csum = (float*)malloc(sizeof(float) * num_scans);
  int cap = 1024;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * cap);
  float* dist = (float*)malloc(sizeof(float) * cap);
  size_t n_out = 0;
  for (size_t i = 0; i < n; ++i) {
    if (done[i]) continue;
    int cnt = okd_radius(tree, pts + 3 * i, r2, cap, idx, dist);
    if (cnt > cap) {
      cap = cnt * 2;
      idx = (int32_t*)realloc(idx, sizeof(int32_t) * cap); dist = (float*)realloc(dist, sizeof(float) * cap);
      cnt = okd_radius(tree, pts + 3 * i, r2, cap, idx, dist);
    }
    qsort(idx, cnt, sizeof(int32_t), cmp_i32);                 /* canonical visiting order: increasing index */
    for (int s = 0; s < num_scans; ++s) { count[s] = 0; csum[s] = 0.f; }
    float avg[3] = {0.f, 0.f, 0.f};
    int total = 0, best_scan = -1, best_count = 0;
    float mr = -1.f;
    for (int k = 0; k < cnt; ++k) {
      const int32_t j = idx[k];
      const int s = scan_idx[j];
      for (int a = 0; a < 3; ++a) avg[a] += pts[3 * (size_t)j + a];
      csum[s] += colors[j];
      if (max_radius[j] > mr) mr = max_radius[j];
      count[s] += 1;
      if (count[s] > best_count) { best_count = count[s]; best_scan = s; }
      total += 1;
      done[j] = 1;
    }
    for (int a = 0; a < 3; ++a) out_pts[3 * n_out + a] = avg[a] / (float)total;
    out_colors[n_out] = csum[best_scan] / (float)count[best_scan];
    out_scan[n_out] = (uint8_t)best_scan;
    out_max_radius[n_out] = mr;
    ++n_out;
  }
  free(done); free(count); free(csum); free(idx); free(dist);
  okd_free(tree);
  return n_out;
}
