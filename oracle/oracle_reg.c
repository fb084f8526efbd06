/*
 * oracle/oracle_reg.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, path (B): dense photometric image registration).
 *
 * Restates the per-observation arithmetic of the reference's ImageRegistrator hot path:
 *   InterpolateBilinear/Trilinear(WithDerivatives)NoCheck   src/opt/interpolate_bilinear.h:36-74, interpolate_trilinear.h:44-87
 *   RobustWeighting                                        src/opt/robust_weighting.h:61-107
 *   camera device functions (PINHOLE)                      src/camera/camera_base_impl.h:155-164,333-408,410-463, camera_pinhole.h:50-85
 *   OcclusionGeometry::_RenderDepthMapWithSplatsCPU         src/opt/occlusion_geometry.cc:404-464
 *   VisibilityEstimator::_AppendObservationsForImage /
 *     _AppendObservationsForIndexedPointsVisibleInImage /
 *     CreateObservationIfScaleFits /
 *     DetermineIfAllNeighborsAreObserved                    src/opt/visibility_estimator.cc:258-295,366-403,405-532,199-256
 *   IntrinsicsAndPoseOptimizer::AccumulateHAndBAndResidualsForObservations,
 *     ComputePointIntensityAndJacobians, AccumulateHAndBAndResidualForColorObservation, AccumulateOnHAndB
 *                                                          src/opt/intrinsics_and_pose_optimizer.cc:624-837,932-1217,839-930,1219-1296
 *   CostCalculator::AccumulateResidualsForObservations / ComputePointColorResidual   src/opt/cost_calculator.cc:102-271
 *   ColorOptimizer::Apply                                   src/opt/color_optimizer.cc:40-123
 * Also: rig images (J_rig), depth residuals (oracle_reg_depth_*), the ten camera models of the factory (oracle_camera.h).
 * Pin: RobustWeighting and the descriptor are checked bit for bit against the reference's own headers compiled into oracle/_ref
 * (tests/test_oracle_ref.py); the rest against the reference's known-answer tests and finite differences (tests/test_oracle_*.py).
 * float -> int conversions follow x86 cvttss2si (out-of-range / NaN -> INT_MIN), which is what the reference's
 * `int ix = v + 0.5f;` compiles to.
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "e3d_oracle.h"
#include "oracle_math.h"

static inline int f2i(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : INT_MIN; }
static inline int d2i(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN; }

/* ---- cameras (oracle_camera.h) ------------------------------------------------------------------------------- */
#include "oracle_camera.h"

void oracle_reg_camera_init(oreg_camera* c, int type, int w, int h, const float* params) { ocam_init(c, type, w, h, params); }

/* CameraBaseImpl::ScaledBy (camera_base_impl.h:70-89): constructs a new camera => re-runs InitCutoff */
void oracle_reg_camera_scaled(const oreg_camera* in, float factor, oreg_camera* out) {
  float p[12];
  if (ocam_unique_focal(in->type)) {       /* GetParameters: [f cx cy q...]; parameters[0] *= factor, [1], [2] the centre (:82-86) */
    p[0] = in->p[0] * factor;
    p[1] = factor * (in->p[2] + 0.5f) - 0.5f; p[2] = factor * (in->p[3] + 0.5f) - 0.5f;
    for (int i = 3; i < in->n_params; ++i) p[i] = in->p[i + 1];
  } else {
    for (int i = 0; i < in->n_params; ++i) p[i] = in->p[i];
    p[0] = in->p[0] * factor; p[1] = in->p[1] * factor;
    p[2] = factor * (in->p[2] + 0.5f) - 0.5f; p[3] = factor * (in->p[3] + 0.5f) - 0.5f;
  }
  ocam_init(out, in->type, (int)(factor * in->width + 0.5f), (int)(factor * in->height + 0.5f), p);
}
void oracle_reg_camera_distort(const oreg_camera* c, float nx, float ny, float out[2]) { ocam_distort(c, nx, ny, &out[0], &out[1]); }
/* Child::Undistort: IterativeUndistort from the distorted point itself; the fisheye wrapper un-warps the inner model's
 * solution with om_tanf(r)/r (camera_base_impl_fisheye.h:81-92) */
void oracle_reg_camera_undistort(const oreg_camera* c, float dx, float dy, float out[2], int* converged) {
  float ux, uy; int conv;
  if (c->type == 4) {
    ocam_fov_undistort(c, dx, dy, &ux, &uy);
    out[0] = ux; out[1] = uy;
    if (converged) *converged = isfinite(ux) && isfinite(uy);
    return;
  }
  ocam_iterative_undistort(c, dx, dy, dx, dy, &ux, &uy, &conv);
  if (ocam_is_fisheye(c->type)) {
    const float r = sqrtf(ux * ux + uy * uy);
    const float factor = (r < OCAM_FISHEYE_EPS) ? 1.f : ((r > (float)(M_PI / 2.f)) ? INFINITY : om_tanf(r) / r);
    ux = factor * ux; uy = factor * uy;
  }
  out[0] = ux; out[1] = uy;
  if (converged) *converged = conv;
}
void oracle_reg_camera_project(const oreg_camera* c, const float P[3], float out[2]) {
  cam_normalized_to_image(c, P[0] / P[2], P[1] / P[2], &out[0], &out[1]);
}
void oracle_reg_camera_deriv_by_world(const oreg_camera* c, const float P[3], float d[6]) { cam_image_deriv_by_world(c, P, d); }
void oracle_reg_camera_deriv_by_intrinsics(const oreg_camera* c, const float P[3], float* d) { cam_image_deriv_by_intrinsics(c, P, d); }

/* ---- interpolation ------------------------------------------------------------------------------------------ */
#define DEF_INTERP(SUFFIX, T)                                                                                   \
  static inline float bilinear_##SUFFIX(const T* img, int w, float x, float y, int ix, int iy) {                 \
    const float fx = x - ix, fxi = 1.f - fx, fy = y - iy, fyi = 1.f - fy;                                        \
    const T* r0 = img + (size_t)iy * w; const T* r1 = img + (size_t)(iy + 1) * w;                                \
    return fyi * (fxi * r0[ix] + fx * r0[ix + 1]) + fy * (fxi * r1[ix] + fx * r1[ix + 1]);                       \
  }                                                                                                              \
  static inline void bilinear_d_##SUFFIX(const T* img, int w, float x, float y, int ix, int iy, float* v,        \
                                         float* dx, float* dy) {                                                 \
    const T* r0 = img + (size_t)iy * w; const T* r1 = img + (size_t)(iy + 1) * w;                                \
    const T tl = r0[ix], tr = r0[ix + 1], bl = r1[ix], br = r1[ix + 1];                                          \
    const float fx = x - ix, fxi = 1.f - fx, fy = y - iy, fyi = 1.f - fy;                                        \
    const float top = fxi * tl + fx * tr, bottom = fxi * bl + fx * br;                                           \
    *v = fyi * top + fy * bottom;                                                                                \
    *dx = fy * (br - bl) + fyi * (tr - tl);                                                                      \
    *dy = bottom - top;                                                                                          \
  }                                                                                                              \
  void oracle_interp_trilinear_##SUFFIX(const T* img0, int w0, const T* img1, int w1, float x0, float y0,        \
                                        float z, float* value) {                                                 \
    const int ix0 = (int)x0, iy0 = (int)y0;                                                                      \
    const float v0 = bilinear_##SUFFIX(img0, w0, x0, y0, ix0, iy0);                                              \
    const float x1 = 2 * (x0 + 0.5f) - 0.5f, y1 = 2 * (y0 + 0.5f) - 0.5f;                                        \
    const float v1 = bilinear_##SUFFIX(img1, w1, x1, y1, (int)x1, (int)y1);                                      \
    *value = (1 - z) * v0 + z * v1;                                                                              \
  }                                                                                                              \
  void oracle_interp_trilinear_d_##SUFFIX(const T* img0, int w0, const T* img1, int w1, float x0, float y0,      \
                                          float z, float* value, float* dx, float* dy, float* dz) {              \
    float v0, dx0, dy0, v1, dx1, dy1;                                                                            \
    bilinear_d_##SUFFIX(img0, w0, x0, y0, (int)x0, (int)y0, &v0, &dx0, &dy0);                                    \
    const float x1 = 2 * (x0 + 0.5f) - 0.5f, y1 = 2 * (y0 + 0.5f) - 0.5f;                                        \
    bilinear_d_##SUFFIX(img1, w1, x1, y1, (int)x1, (int)y1, &v1, &dx1, &dy1);                                    \
    *value = (1 - z) * v0 + z * v1;                                                                              \
    *dx = (1 - z) * dx0 + z * 2 * dx1;                                                                           \
    *dy = (1 - z) * dy0 + z * 2 * dy1;                                                                           \
    *dz = v1 - v0;                                                                                               \
  }
DEF_INTERP(u8, uint8_t)
DEF_INTERP(f32, float)

/* ---- robust weighting ----------------------------------------------------------------------------------------- */
static inline float robust_residual(int type, float param, float r) {
  if (type == 1) { const float a = fabsf(r); return (a < param) ? 0.5f * r * r : param * (a - 0.5f * param); }
  if (type == 2) {
    const float a = fabsf(r);
    if (a < param) { const float q = r / param; const float t = 1.f - q * q; return (1 / 6.f) * param * param * (1 - t * t * t); }
    return (1 / 6.f) * param * param;
  }
  return 0.5f * r * r;
}
static inline float robust_weight(int type, float param, float r) {
  if (type == 1) { const float a = fabsf(r); return (a < param) ? 1.f : (param / a); }
  if (type == 2) { const float a = fabsf(r); if (a < param) { const float q = r / param; const float t = 1.f - q * q; return t * t; } return 0.f; }
  return 1.f;
}
float oracle_reg_robust_residual(int type, float param, float r) { return robust_residual(type, param, r); }
float oracle_reg_robust_weight(int type, float param, float r) { return robust_weight(type, param, r); }

static inline float dot3e(const float* a, const float* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
static inline void rt(const float* R, const float* t, const float* p, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = dot3e(R + 3 * i, p) + t[i];
}

/* ---- a24: splat depth map ---------------------------------------------------------------------------------------- */
void oracle_reg_splat_depth(const float* pts, size_t n, const float R[9], const float t[3], const oreg_camera* cam,
                            float point_radius, float* depth) {
  const int W = cam->width, H = cam->height;
  for (size_t i = 0; i < (size_t)W * H; ++i) depth[i] = INFINITY;
  const float max_splat_radius = 10;
  for (size_t i = 0; i < n; ++i) {
    float pp[3];
    rt(R, t, pts + 3 * i, pp);
    if (!(pp[2] > 0.f)) continue;
    float px, py, d[6];
    cam_normalized_to_image(cam, pp[0] / pp[2], pp[1] / pp[2], &px, &py);
    cam_image_deriv_by_world(cam, pp, d);
    float rx = sqrtf(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])) * point_radius;
    float ry = sqrtf(d[3] * d[3] + (d[4] * d[4] + d[5] * d[5])) * point_radius;
    if (max_splat_radius < rx) rx = max_splat_radius;
    if (max_splat_radius < ry) ry = max_splat_radius;
    const int ix = f2i(px + 0.5f), iy = f2i(py + 0.5f);
    int min_x = d2i((double)((float)ix - rx) + 0.5), min_y = d2i((double)((float)iy - ry) + 0.5);
    int end_x = d2i((double)((float)ix + rx) + 1.5), end_y = d2i((double)((float)iy + ry) + 1.5);
    if (min_x < 0) min_x = 0;
    if (min_y < 0) min_y = 0;
    if (end_x > W) end_x = W;
    if (end_y > H) end_y = H;
    if (min_y < end_y && min_x < end_x)
      for (int y = min_y; y < end_y; ++y)
        for (int x = min_x; x < end_x; ++x)
          if (depth[(size_t)y * W + x] > pp[2]) depth[(size_t)y * W + x] = pp[2];
  }
}

/* ---- a20 / a21: observations -------------------------------------------------------------------------------------- */
size_t oracle_reg_observe(const float* pts, size_t n_pts, float point_radius, const uint32_t* indices, size_t n_idx,
                          const float R[9], const float t[3], const oreg_camera* levels, int min_image_scale,
                          int n_levels, const uint8_t* const* images, const uint8_t* const* masks,
                          const float* occlusion, int image_scale, int border, int current_image_scale,
                          int image_scale_count, float occlusion_threshold, float max_valid_intensity,
                          uint32_t* out_idx, float* out_x, float* out_y, float* out_scale) {
  const int lvl = image_scale - min_image_scale < 0 ? 0 : image_scale - min_image_scale;
  const oreg_camera* cam = &levels[lvl];
  const int check_masks = (indices == NULL);
  const size_t count = indices ? n_idx : n_pts;
  size_t n_out = 0;
  for (size_t k = 0; k < count; ++k) {
    const size_t pi = indices ? indices[k] : k;
    float pp[3];
    rt(R, t, pts + 3 * pi, pp);
    if (!(pp[2] > 0.f)) continue;
    float ixf, iyf;
    cam_normalized_to_image(cam, pp[0] / pp[2], pp[1] / pp[2], &ixf, &iyf);
    int ix = f2i(ixf + 0.5f), iy = f2i(iyf + 0.5f);
    if (!(ix >= 0 && iy >= 0 && ix < cam->width && iy < cam->height)) continue;
    if (!indices && !(occlusion[(size_t)iy * cam->width + ix] + occlusion_threshold >= pp[2])) continue;
    /* CreateObservationIfScaleFits */
    const float pr[3] = {pp[0] + point_radius, pp[1] + 0.f, pp[2] + 0.f};
    float rxf, ryf;
    cam_normalized_to_image(cam, pr[0] / pr[2], pr[1] / pr[2], &rxf, &ryf);
    const float dx = rxf - ixf, dy = ryf - iyf;
    const float radius_pixels = sqrtf(dx * dx + dy * dy);
    const float observation_scale = image_scale + om_log2f(2 * radius_pixels);
    const int lo = min_image_scale > current_image_scale ? min_image_scale : current_image_scale;
    if (!(observation_scale >= lo && f2i(observation_scale) < image_scale_count - 1)) continue;
    const int small_scale = f2i(observation_scale) + 1;
    int li = small_scale - min_image_scale; if (li < 0) li = 0;
    if (li >= n_levels) continue;   /* cannot happen for consistent inputs (image_scale_count bounds it) */
    const oreg_camera* ic = &levels[li];
    const float nx = cam->fx_inv * ixf + cam->cx_inv, ny = cam->fy_inv * iyf + cam->cy_inv;
    const float jx = ic->p[0] * nx + ic->p[2], jy = ic->p[1] * ny + ic->p[3];
    ix = f2i(jx + 0.5f); iy = f2i(jy + 0.5f);
    if (!(jx + 0.5f >= border && jy + 0.5f >= border && ix >= border && iy >= border && ix < ic->width - border &&
          iy < ic->height - border))
      continue;
    if (check_masks) {
      const int pl = small_scale - min_image_scale;
      if (masks && masks[pl] && masks[pl][(size_t)iy * ic->width + ix] != 0) continue;
      if (images[pl][(size_t)iy * ic->width + ix] > max_valid_intensity) continue;
    }
    out_idx[n_out] = (uint32_t)pi; out_x[n_out] = jx; out_y[n_out] = jy; out_scale[n_out] = observation_scale;
    ++n_out;
  }
  return n_out;
}

/* ---- a22 ---------------------------------------------------------------------------------------------------------- */
void oracle_reg_neighbors_observed(size_t n_pts, const uint32_t* obs_idx, size_t n_obs, const uint32_t* nbr, int K,
                                   uint8_t* flags) {
  uint8_t* seen = (uint8_t*)calloc(n_pts ? n_pts : 1, 1);
  for (size_t i = 0; i < n_obs; ++i) seen[obs_idx[i]] = 1;
  for (size_t i = 0; i < n_obs; ++i) {
    uint8_t all = 1;
    for (int k = 0; k < K; ++k) if (!seen[nbr[(size_t)obs_idx[i] * K + k]]) { all = 0; break; }
    flags[i] = all;
  }
  free(seen);
}

/* ---- a16: intensity + Jacobian rows of one observation (I = cam_min->n_params, non-rig) --------------------------------- */
static void point_intensity_and_jacobians(const float* point, float point_radius, const oreg_camera* cam_min,
                                          int min_image_scale, const uint8_t* const* images, const int* widths,
                                          const float R[9], const float t[3], float ox, float oy, float oscale,
                                          float* intensity, float* j_intr /*I*/, float* j_pose /*6*/,
                                          const oreg_rig_link* rig /* NULL unless a non-reference rig image */, float* j_rig /*6*/) {
  const int I = cam_min->n_params;
  float T[3];
  rt(R, t, point, T);
  const int small_scale = f2i(oscale) + 1, large_scale = f2i(oscale);
  float ji[3];
  oracle_interp_trilinear_d_u8(images[small_scale - min_image_scale], widths[small_scale - min_image_scale],
                               images[large_scale - min_image_scale], widths[large_scale - min_image_scale], ox, oy,
                               1 - (oscale - (float)f2i(oscale)), intensity, &ji[0], &ji[1], &ji[2]);
  ji[2] = -1 * ji[2];
  const float scale_factor = (float)pow(2, min_image_scale - small_scale);
  const float inv_scale_factor = 1.f / scale_factor;
  ji[0] *= scale_factor; ji[1] *= scale_factor;
  const float mx = inv_scale_factor * (ox + 0.5f) - 0.5f, my = inv_scale_factor * (oy + 0.5f) - 0.5f;
  const float To[3] = {T[0] + point_radius, T[1], T[2]};
  float offx, offy;
  cam_normalized_to_image(cam_min, To[0] / To[2], To[1] / To[2], &offx, &offy);
  const float rdx = offx - mx, rdy = offy - my;
  float denom = 0.693147180559945f * (rdx * rdx + rdy * rdy);
  if (denom < 1e-6f) denom = 1e-6f;
  float P[36], Po[24];      /* 3 x I and 2 x I */
  cam_image_deriv_by_intrinsics(cam_min, T, P);
  cam_image_deriv_by_intrinsics(cam_min, To, Po);
  for (int i = 0; i < I; ++i) P[2 * I + i] = ((Po[i] - P[i]) * rdx + (Po[I + i] - P[I + i]) * rdy) / denom;
  for (int i = 0; i < I; ++i) j_intr[i] = ji[0] * P[i] + (ji[1] * P[I + i] + ji[2] * P[2 * I + i]);
  float W[9], Wo[6];        /* 3 x 3 and 2 x 3 */
  cam_image_deriv_by_world(cam_min, T, W);
  cam_image_deriv_by_world(cam_min, To, Wo);
  for (int i = 0; i < 3; ++i) W[6 + i] = ((Wo[i] - W[i]) * rdx + (Wo[3 + i] - W[3 + i]) * rdy) / denom;
  float a[3];
  for (int i = 0; i < 3; ++i) a[i] = ji[0] * W[i] + (ji[1] * W[3 + i] + ji[2] * W[6 + i]);
  /* [I3 | 0 z -y ; -z 0 x ; y -x 0] */
  const float C[18] = {1, 0, 0, 0, T[2], -1 * T[1], 0, 1, 0, -1 * T[2], 0, T[0], 0, 0, 1, T[1], -1 * T[0], 0};
  if (!rig) {
    for (int j = 0; j < 6; ++j) j_pose[j] = a[0] * C[j] + (a[1] * C[6 + j] + a[2] * C[12 + j]);
    return;
  }
  /* non-reference rig image (intrinsics_and_pose_optimizer.cc:1107-1150): the pose block differentiates the rig pose,
   * j_pose = ((j * P) * image_T_rig.rotationMatrix()) * [I3 | -[rig_T_global * point]x], and the extrinsics block is the
   * ordinary pose formula at the transformed point. */
  for (int j = 0; j < 6; ++j) j_rig[j] = a[0] * C[j] + (a[1] * C[6 + j] + a[2] * C[12 + j]);
  float Rir[9];
  om_quat_to_R_f(rig->q_image_T_rig, Rir);
  float ar[3];
  for (int i = 0; i < 3; ++i) ar[i] = a[0] * Rir[i] + (a[1] * Rir[3 + i] + a[2] * Rir[6 + i]);
  /* rig_point = rig_T_global * point: Sophus SO3 action p + w*uv + v x uv with uv = 2 (v x p), then + translation */
  const float* v = rig->q_rig_T_global + 1; const float w = rig->q_rig_T_global[0];
  float uv[3], cr[3], G[3];
  om_cross_f(v, point, uv);
  for (int i = 0; i < 3; ++i) uv[i] = uv[i] + uv[i];
  om_cross_f(v, uv, cr);
  for (int i = 0; i < 3; ++i) G[i] = ((point[i] + w * uv[i]) + cr[i]) + rig->t_rig_T_global[i];
  const float D[18] = {1, 0, 0, 0, G[2], -1 * G[1], 0, 1, 0, -1 * G[2], 0, G[0], 0, 0, 1, G[1], -1 * G[0], 0};
  for (int j = 0; j < 6; ++j) j_pose[j] = ar[0] * D[j] + (ar[1] * D[6 + j] + ar[2] * D[12 + j]);
}

void oracle_reg_pass1_rig(const float* pts, float point_radius, const oreg_camera* cam_min, int min_image_scale,
                          const uint8_t* const* images, const int* widths, const float R[9], const float t[3],
                          const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale, size_t n_obs,
                          const oreg_rig_link* rig, float* intensities, float* j_intr, float* j_pose, float* j_rig) {
  float dummy[6];
  for (size_t i = 0; i < n_obs; ++i)
    point_intensity_and_jacobians(pts + 3 * (size_t)obs_idx[i], point_radius, cam_min, min_image_scale, images, widths, R,
                                  t, obs_x[i], obs_y[i], obs_scale[i], &intensities[i], j_intr + (size_t)cam_min->n_params * i,
                                  j_pose + 6 * i, rig, (rig && j_rig) ? j_rig + 6 * i : dummy);
}
void oracle_reg_pass1(const float* pts, float point_radius, const oreg_camera* cam_min, int min_image_scale,
                      const uint8_t* const* images, const int* widths, const float R[9], const float t[3],
                      const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale, size_t n_obs,
                      float* intensities, float* j_intr, float* j_pose) {
  oracle_reg_pass1_rig(pts, point_radius, cam_min, min_image_scale, images, widths, R, t, obs_idx, obs_x, obs_y, obs_scale, n_obs,
                       NULL, intensities, j_intr, j_pose, NULL);
}

/* a18: AccumulateOnHAndB on the local (I+6) x (I+6) block: products in f32, cast, add in f64 */
/* local variable order [intrinsics(I), rig extrinsics(6, dependent rig images only), pose(6)]: the global layout is
 * intrinsics < rigs < poses (CountAndIndexVariables :442-473), so the local upper triangle is what the reference's block
 * updates (:1246-1283) touch */
static void accumulate_on_h_and_b(int I, float weight, float residual, const float* ji, const float* jr, const float* jp, double* H,
                                  double* b) {
  if (weight == 0) return;
  const int V = I + (jr ? 12 : 6);
  float J[24];
  int n = 0;
  for (int i = 0; i < I; ++i) J[n++] = ji[i];
  if (jr) for (int i = 0; i < 6; ++i) J[n++] = jr[i];
  for (int i = 0; i < 6; ++i) J[n++] = jp[i];
  for (int i = 0; i < V; ++i)
    for (int j = i; j < V; ++j) {
      /* (weight * j^T) * j  in f32 (block-wise expressions of the reference evaluate to exactly this per entry) */
      const float wj = weight * J[i];
      H[i * V + j] += (double)(wj * J[j]);
    }
  const float wr = weight * residual;
  for (int i = 0; i < V; ++i) b[i] += (double)(wr * J[i]);
}

/* ---- a15 + a17 + a18 --------------------------------------------------------------------------------------------------- */
void oracle_reg_accumulate(const float* pts, size_t n_pts, float point_radius, const uint32_t* nbr, int K,
                           const float* fixed_desc, const float* var_desc, const int32_t* obs_counts,
                           const oreg_camera* cam_min, int min_image_scale, const uint8_t* const* images, const int* widths,
                           const float R[9], const float t[3], const uint32_t* obs_idx, const float* obs_x,
                           const float* obs_y, const float* obs_scale, const uint8_t* flags, size_t n_obs, int robust_type,
                           float robust_param, float fixed_weight, float var_weight, double* H /*VxV*/, double* b /*V*/,
                           double sums[2], int64_t counts[2]) {
  oracle_reg_accumulate_rig(pts, n_pts, point_radius, nbr, K, fixed_desc, var_desc, obs_counts, cam_min, min_image_scale, images,
                            widths, R, t, obs_idx, obs_x, obs_y, obs_scale, flags, n_obs, robust_type, robust_param, fixed_weight,
                            var_weight, NULL, H, b, sums, counts);
}

void oracle_reg_accumulate_rig(const float* pts, size_t n_pts, float point_radius, const uint32_t* nbr, int K,
                               const float* fixed_desc, const float* var_desc, const int32_t* obs_counts,
                               const oreg_camera* cam_min, int min_image_scale, const uint8_t* const* images, const int* widths,
                               const float R[9], const float t[3], const uint32_t* obs_idx, const float* obs_x,
                               const float* obs_y, const float* obs_scale, const uint8_t* flags, size_t n_obs, int robust_type,
                               float robust_param, float fixed_weight, float var_weight, const oreg_rig_link* rig,
                               double* H /*VxV*/, double* b /*V*/, double sums[2], int64_t counts[2]) {
  const int NI = cam_min->n_params, V = NI + (rig ? 12 : 6);
  float* JR = (float*)malloc(sizeof(float) * 6 * (n_obs + 1));
  float* I = (float*)malloc(sizeof(float) * (n_obs + 1));
  float* JI = (float*)malloc(sizeof(float) * NI * (n_obs + 1));
  float* JP = (float*)malloc(sizeof(float) * 6 * (n_obs + 1));
  int64_t* row = (int64_t*)malloc(sizeof(int64_t) * (n_pts + 1));
  for (size_t i = 0; i < n_pts; ++i) row[i] = -1;
  oracle_reg_pass1_rig(pts, point_radius, cam_min, min_image_scale, images, widths, R, t, obs_idx, obs_x, obs_y, obs_scale, n_obs,
                       rig, I, JI, JP, JR);
  for (size_t i = 0; i < n_obs; ++i) row[obs_idx[i]] = (int64_t)i;
  memset(H, 0, sizeof(double) * V * V); memset(b, 0, sizeof(double) * V);
  sums[0] = sums[1] = 0; counts[0] = counts[1] = 0;
  float comp[64];
  for (size_t i = 0; i < n_obs; ++i) {
    if (!flags[i]) continue;
    const size_t p = obs_idx[i];
    for (int kind = 0; kind < 2; ++kind) {
      const float sw = kind == 0 ? fixed_weight : var_weight;
      if (!(sw > 0)) continue;
      if (kind == 1 && !(obs_counts[p] >= 2)) continue;
      const float* desc = kind == 0 ? fixed_desc : var_desc;
      float pr = 0.f;
      for (int k = 0; k < K; ++k) {
        const int64_t nr = row[nbr[p * K + k]];
        const float image_descriptor = I[nr] - I[i];
        const float c = image_descriptor - desc[p * K + k];
        comp[k] = c;
        pr += c * c;
      }
      pr = sqrtf(pr);
      counts[kind]++;
      sums[kind] += robust_residual(robust_type, robust_param, pr);
      const float w = sw * robust_weight(robust_type, robust_param, pr);
      if (w != 0) {
        for (int k = 0; k < K; ++k) {
          const int64_t nr = row[nbr[p * K + k]];
          float ji[12], jp[6], jr[6];
          for (int q = 0; q < NI; ++q) ji[q] = JI[(size_t)NI * nr + q] - JI[(size_t)NI * i + q];
          for (int q = 0; q < 6; ++q) jp[q] = JP[6 * nr + q] - JP[6 * i + q];
          if (rig) for (int q = 0; q < 6; ++q) jr[q] = JR[6 * nr + q] - JR[6 * i + q];
          accumulate_on_h_and_b(NI, w, comp[k], ji, rig ? jr : NULL, jp, H, b);
        }
      }
    }
  }
  free(I); free(JI); free(JP); free(JR); free(row);
}

/* Sophus::SE3f product a * b (rig.image_T_rig[c] * first_image.image_T_global, intrinsics_and_pose_optimizer.cc:545-546) */
void oracle_se3_mul(const float qa[4], const float ta[3], const float qb[4], const float tb[3], float q[4], float t[3]) {
  om_se3f a, b, c;
  memcpy(a.q, qa, sizeof a.q); memcpy(a.t, ta, sizeof a.t); memcpy(b.q, qb, sizeof b.q); memcpy(b.t, tb, sizeof b.t);
  om_se3f_mul(&a, &b, &c);
  memcpy(q, c.q, sizeof c.q); memcpy(t, c.t, sizeof c.t);
}

/* ---- a19 --------------------------------------------------------------------------------------------------------------- */
/* ---- depth residuals (off by default in the reference: parameters.h:55,165 "not used in ETH3D pipeline") ------------------------------
 * ComputePointIntensityAndJacobians, depth part (intrinsics_and_pose_optimizer.cc:1150-1214), for images that are not dependent rig
 * images (the reference aborts with "Not implemented yet" for those, :1199-1207): residual = 1 / interpolated depth - 1 / point depth
 * with the point depth from Sophus' image_T_global * point (q = w x y z), Jacobian = the colour terms with the depth pyramid's
 * interpolation derivative times -1 / depth^2, minus (-1 / z^2) times the z row of d(camera point) / d(pose). */
void oracle_reg_depth_rows(const float* pts, float point_radius, const oreg_camera* cam_min, int min_image_scale,
                           const float* const* depth_maps, const int* widths, const float R[9], const float t[3], const float q[4],
                           const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale, size_t n_obs,
                           float* residuals, float* j_intr, float* j_pose) {
  const int I = cam_min->n_params;
  for (size_t o = 0; o < n_obs; ++o) {
    const float* point = pts + 3 * (size_t)obs_idx[o];
    const float ox = obs_x[o], oy = obs_y[o], oscale = obs_scale[o];
    float T[3];
    rt(R, t, point, T);
    const int small_scale = f2i(oscale) + 1, large_scale = f2i(oscale);
    float depth, jd[3];
    oracle_interp_trilinear_d_f32(depth_maps[small_scale - min_image_scale], widths[small_scale - min_image_scale],
                                  depth_maps[large_scale - min_image_scale], widths[large_scale - min_image_scale], ox, oy,
                                  1 - (oscale - (float)f2i(oscale)), &depth, &jd[0], &jd[1], &jd[2]);
    jd[2] = -1 * jd[2];
    const float scale_factor = (float)pow(2, min_image_scale - small_scale);
    const float inv_scale_factor = 1.f / scale_factor;
    jd[0] *= scale_factor; jd[1] *= scale_factor;
    const float inv_depth = (depth != 0) ? (1.f / depth) : 0.f;
    /* pp = image.image_T_global * point (Sophus SE3f): p + w uv + v x uv, uv = 2 (v x p), + translation */
    float uv[3], cr[3], pp[3];
    om_cross_f(q + 1, point, uv);
    for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
    om_cross_f(q + 1, uv, cr);
    for (int k = 0; k < 3; ++k) pp[k] = ((point[k] + q[0] * uv[k]) + cr[k]) + t[k];
    const float point_inv_depth = (pp[2] != 0.f) ? (1.f / pp[2]) : 0.f;
    residuals[o] = inv_depth - point_inv_depth;
    const float j_inv = -1 / (depth * depth);
    const float ji[3] = {j_inv * jd[0], j_inv * jd[1], j_inv * jd[2]};
    /* the projection terms, exactly as in point_intensity_and_jacobians */
    const float mx = inv_scale_factor * (ox + 0.5f) - 0.5f, my = inv_scale_factor * (oy + 0.5f) - 0.5f;
    const float To[3] = {T[0] + point_radius, T[1], T[2]};
    float offx, offy;
    cam_normalized_to_image(cam_min, To[0] / To[2], To[1] / To[2], &offx, &offy);
    const float rdx = offx - mx, rdy = offy - my;
    float denom = 0.693147180559945f * (rdx * rdx + rdy * rdy);
    if (denom < 1e-6f) denom = 1e-6f;
    float P[36], Po[24];
    cam_image_deriv_by_intrinsics(cam_min, T, P);
    cam_image_deriv_by_intrinsics(cam_min, To, Po);
    for (int i = 0; i < I; ++i) P[2 * I + i] = ((Po[i] - P[i]) * rdx + (Po[I + i] - P[I + i]) * rdy) / denom;
    for (int i = 0; i < I; ++i) j_intr[(size_t)I * o + i] = ji[0] * P[i] + (ji[1] * P[I + i] + ji[2] * P[2 * I + i]);
    float W[9], Wo[6];
    cam_image_deriv_by_world(cam_min, T, W);
    cam_image_deriv_by_world(cam_min, To, Wo);
    for (int i = 0; i < 3; ++i) W[6 + i] = ((Wo[i] - W[i]) * rdx + (Wo[3 + i] - W[3 + i]) * rdy) / denom;
    float a[3];
    for (int i = 0; i < 3; ++i) a[i] = ji[0] * W[i] + (ji[1] * W[3 + i] + ji[2] * W[6 + i]);
    const float C[18] = {1, 0, 0, 0, T[2], -1 * T[1], 0, 1, 0, -1 * T[2], 0, T[0], 0, 0, 1, T[1], -1 * T[0], 0};
    const float j_point_inv = -1 / (T[2] * T[2]);
    for (int j = 0; j < 6; ++j) {
      float v = a[0] * C[j] + (a[1] * C[6 + j] + a[2] * C[12 + j]);
      v -= j_point_inv * C[12 + j];                       /* -= j_point_depth_inversion * j_camera_space_point_wrt_pose.row(2) */
      j_pose[6 * o + j] = v;
    }
  }
}

/* AccumulateOnHAndB for the depth residuals of one image (:747-757, :1219-1296): weight = robust weight * depth_residuals_weight
 * (f32), H += ((weight J^T) J) cast to double (upper triangle, V = I + 6 local unknowns [intrinsics, pose], row-major V x V),
 * b += (weight residual) J; sum / count as in the cost (every observation counts). */
void oracle_reg_depth_accumulate(const float* residuals, const float* j_intr, const float* j_pose, size_t n_obs, int I,
                                 int robust_type, float robust_param, float depth_weight, double* H, double* b, double* sum,
                                 int64_t* count) {
  const int V = I + 6;
  for (int i = 0; i < V * V; ++i) H[i] = 0;
  for (int i = 0; i < V; ++i) b[i] = 0;
  *sum = 0; *count = 0;
  float J[32];
  for (size_t o = 0; o < n_obs; ++o) {
    const float r = residuals[o];
    ++*count;
    *sum += robust_residual(robust_type, robust_param, r);
    float w = robust_weight(robust_type, robust_param, r);
    w *= depth_weight;
    if (w == 0) continue;
    for (int i = 0; i < I; ++i) J[i] = j_intr[(size_t)I * o + i];
    for (int i = 0; i < 6; ++i) J[I + i] = j_pose[6 * o + i];
    for (int rr = 0; rr < V; ++rr) {
      const float wj = w * J[rr];
      for (int c = rr; c < V; ++c) H[(size_t)rr * V + c] += (double)(wj * J[c]);
    }
    const float wr = w * r;
    for (int i = 0; i < V; ++i) b[i] += (double)(wr * J[i]);
  }
}

/* CostCalculator, depth part (cost_calculator.cc:221-245) */
void oracle_reg_depth_cost(const float* pts, int min_image_scale, const float* const* depth_maps, const int* widths, const float q[4],
                           const float t[3], const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale,
                           size_t n_obs, int robust_type, float robust_param, double* sum, int64_t* count) {
  *sum = 0; *count = 0;
  for (size_t o = 0; o < n_obs; ++o) {
    const float* point = pts + 3 * (size_t)obs_idx[o];
    const float oscale = obs_scale[o];
    const int small_scale = f2i(oscale) + 1, large_scale = f2i(oscale);
    float depth;
    oracle_interp_trilinear_f32(depth_maps[small_scale - min_image_scale], widths[small_scale - min_image_scale],
                                depth_maps[large_scale - min_image_scale], widths[large_scale - min_image_scale], obs_x[o], obs_y[o],
                                1 - (oscale - (float)f2i(oscale)), &depth);
    const float inv_depth = (depth != 0) ? (1.f / depth) : 0.f;
    float uv[3], cr[3], ppz;
    om_cross_f(q + 1, point, uv);
    for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
    om_cross_f(q + 1, uv, cr);
    ppz = ((point[2] + q[0] * uv[2]) + cr[2]) + t[2];
    const float point_inv_depth = (ppz != 0.f) ? (1.f / ppz) : 0.f;
    const float r = inv_depth - point_inv_depth;
    ++*count;
    *sum += robust_residual(robust_type, robust_param, r);
  }
}

void oracle_reg_cost(size_t n_pts, const uint32_t* nbr, int K, const float* fixed_desc, const float* var_desc,
                     const int32_t* obs_counts, int min_image_scale, const uint8_t* const* images, const int* widths,
                     const uint32_t* obs_idx, const float* obs_x, const float* obs_y, const float* obs_scale,
                     const uint8_t* flags, size_t n_obs, int robust_type, float robust_param, float fixed_weight,
                     float var_weight, double sums[2], int64_t counts[2]) {
  float* I = (float*)malloc(sizeof(float) * (n_pts + 1));
  for (size_t i = 0; i < n_pts; ++i) I[i] = -1.f;
  for (size_t i = 0; i < n_obs; ++i) {
    const int s = f2i(obs_scale[i]);
    oracle_interp_trilinear_u8(images[s + 1 - min_image_scale], widths[s + 1 - min_image_scale], images[s - min_image_scale],
                               widths[s - min_image_scale], obs_x[i], obs_y[i], 1 - (obs_scale[i] - (float)s), &I[obs_idx[i]]);
  }
  sums[0] = sums[1] = 0; counts[0] = counts[1] = 0;
  for (size_t i = 0; i < n_obs; ++i) {
    if (!flags[i]) continue;
    const size_t p = obs_idx[i];
    for (int kind = 0; kind < 2; ++kind) {
      const float sw = kind == 0 ? fixed_weight : var_weight;
      if (!(sw > 0)) continue;
      if (kind == 1 && !(obs_counts[p] >= 2)) continue;
      const float* desc = kind == 0 ? fixed_desc : var_desc;
      float pr = 0.f;
      for (int k = 0; k < K; ++k) {
        const float c = (I[nbr[p * K + k]] - I[p]) - desc[p * K + k];
        pr += c * c;
      }
      pr = sqrtf(pr);
      sums[kind] += robust_residual(robust_type, robust_param, pr);
      counts[kind]++;
    }
  }
  free(I);
}

/* ---- a23: one image's contribution, and the final division ---------------------------------------------------------------- */
void oracle_reg_color_accumulate(size_t n_pts, const uint32_t* nbr, int K, int min_image_scale, const uint8_t* const* images,
                                 const int* widths, const uint32_t* obs_idx, const float* obs_x, const float* obs_y,
                                 const float* obs_scale, const uint8_t* flags, size_t n_obs, float* descriptors,
                                 int32_t* obs_counts) {
  float* I = (float*)malloc(sizeof(float) * (n_pts + 1));
  for (size_t i = 0; i < n_pts; ++i) I[i] = -1.f;
  for (size_t i = 0; i < n_obs; ++i) {
    const int s = f2i(obs_scale[i]);
    oracle_interp_trilinear_u8(images[s + 1 - min_image_scale], widths[s + 1 - min_image_scale], images[s - min_image_scale],
                               widths[s - min_image_scale], obs_x[i], obs_y[i], 1 - (obs_scale[i] - (float)s), &I[obs_idx[i]]);
  }
  for (size_t i = 0; i < n_obs; ++i) {
    if (!flags[i]) continue;
    const size_t p = obs_idx[i];
    obs_counts[p] += 1;
    for (int k = 0; k < K; ++k) descriptors[p * K + k] += I[nbr[p * K + k]] - I[p];
  }
  free(I);
}
void oracle_reg_color_finish(size_t n_pts, int K, float* descriptors, const int32_t* obs_counts) {
  for (size_t i = 0; i < n_pts; ++i)
    if (obs_counts[i] > 1)
      for (int k = 0; k < K; ++k) descriptors[i * K + k] /= obs_counts[i];
}

/* GroundTruthCreator visibility (src/exe/ground_truth_creator.cc:63-86 counting, :152-189 ground-truth depth).
 * mode 0: counts[i] += 1 for visible points; mode 1: gt_depth (pre-filled with +inf by the caller) = min z over visible
 * points with counts[i] >= min_count.  mask may be NULL; excluded = opt::MaskType::kEvalObs. */
void oracle_scan_visibility(const float* pts, size_t n, const float R[9], const float t[3], const oreg_camera* cam,
                            const float* occlusion, float occlusion_threshold, const uint8_t* mask, int excluded, int mode,
                            int min_count, int32_t* counts, float* gt_depth) {
  for (size_t i = 0; i < n; ++i) {
    if (mode == 1 && counts[i] < min_count) continue;
    float pp[3];
    rt(R, t, pts + 3 * i, pp);
    if (!(pp[2] > 0)) continue;
    float px, py;
    cam_normalized_to_image(cam, pp[0] / pp[2], pp[1] / pp[2], &px, &py);
    const int ix = f2i(px + 0.5f), iy = f2i(py + 0.5f);
    if (!(ix >= 0 && iy >= 0 && ix < cam->width && iy < cam->height)) continue;
    const size_t o = (size_t)iy * cam->width + ix;
    if (!(occlusion[o] + occlusion_threshold >= pp[2])) continue;
    if (mask && mask[o] == excluded) continue;
    if (mode == 0) counts[i] += 1;
    else if (pp[2] < gt_depth[o]) gt_depth[o] = pp[2];
  }
}

/* CreateGroundTruthForImage, scan rendering part (ground_truth_creator.cc:149,160-189): the image is painted in scan-point order with
 * squares of 2 * radius + 1 pixels; rendering[pixel] = index + 1 of the point painted last, 0 = the image's own colour. */
void oracle_scan_rendering(const float* pts, size_t n, const float R[9], const float t[3], const oreg_camera* cam,
                           const float* occlusion, float occlusion_threshold, const uint8_t* mask, int excluded, int min_count,
                           const int32_t* counts, int radius, uint32_t* rendering) {
  for (size_t i = 0; i < n; ++i) {
    if (counts[i] < min_count) continue;
    float pp[3];
    rt(R, t, pts + 3 * i, pp);
    if (!(pp[2] > 0)) continue;
    float px, py;
    cam_normalized_to_image(cam, pp[0] / pp[2], pp[1] / pp[2], &px, &py);
    const int ix = f2i(px + 0.5f), iy = f2i(py + 0.5f);
    if (!(ix >= 0 && iy >= 0 && ix < cam->width && iy < cam->height)) continue;
    const size_t o = (size_t)iy * cam->width + ix;
    if (!(occlusion[o] + occlusion_threshold >= pp[2])) continue;
    if (mask && mask[o] == excluded) continue;
    const int x0 = ix - radius < 0 ? 0 : ix - radius, y0 = iy - radius < 0 ? 0 : iy - radius;
    const int x1 = ix + radius + 1 > cam->width ? cam->width : ix + radius + 1, y1 = iy + radius + 1 > cam->height ? cam->height : iy + radius + 1;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) rendering[(size_t)y * cam->width + x] = (uint32_t)i + 1u;
  }
}
