/*
 * oracle/oracle_camera.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's camera models that ImageRegistrator is run with:
 *   type 0  PINHOLE             camera::PinholeCamera               src/camera/camera_pinhole.h:40-86      I = 4
 *   type 1  OPENCV              camera::PolynomialTangentialCamera  src/camera/camera_polynomial_tangential.h:41-159   I = 8
 *   type 2  THIN_PRISM_FISHEYE  camera::BenchmarkCamera = FisheyeBase<ThinPrismCamera>
 *                               src/camera/camera_benchmark.h:44-52, camera_base_impl_fisheye.h:43-162,
 *                               camera_thin_prism.h:43-162                                                  I = 12
 * and of the shared CRTP base (src/camera/camera_base_impl.h): NormalizedToImage :155-164, IterativeUndistort :216-250,
 * UndistortFromInside :278-328, DistortedDerivativeByWorld / ImageDerivativeByWorld :333-360, ImageDerivativeByIntrinsics
 * :369-408, InitCutoff :410-463, ScaledBy :70-89; CameraBase pixel mapping src/camera/camera_base.cc:81-86.
 *
 * All arithmetic is f32 in the reference's expression order (compile with -ffp-contract=off).  Eigen fixed-size products
 * of 2-vectors / 2x2 matrices evaluate each coefficient as a0*b0 + a1*b1.
 *
 * Cut-offs: PinholeCamera never calls InitCutoff (camera_pinhole.cc:35-43) => +inf.  PolynomialTangentialCamera calls it
 * in its constructor (camera_polynomial_tangential.cc:35-50).  BenchmarkCamera: the outer fisheye object never calls it
 * (+inf, camera_benchmark.cc:35-46); its inner ThinPrismCamera does (camera_thin_prism.cc:34-51) and the fisheye wrapper
 * tests atan(r)^2 against that inner value (camera_base_impl_fisheye.h:69-72).
 *
 * `atan2(r, 1.f)` (camera_base_impl_fisheye.h:68,109,140) is restated as atan2f: inside the opt/ translation units the float
 * overload is assumed visible (same assumption as log2 in visibility_estimator.cc:437).  Unpinned by any reference test
 * at the ulp level; the GPU parity tests for this model carry a tolerance for it.
 */
#ifndef E3D_ORACLE_CAMERA_H
#define E3D_ORACLE_CAMERA_H

#include <math.h>
#include <string.h>

#include "e3d_oracle.h"
#include "../include/e3d_libm.h"   /* bit-defined atanf / atan2f / tanf ... shared with the HIP kernels */
#include "oracle_libm_select.h"

/* 0 PINHOLE, 1 OPENCV, 2 THIN_PRISM_FISHEYE, 3 OPENCV_FISHEYE = FisheyeBase over Polynomial4Camera (camera_fisheye_polynomial_4.h,
 * camera_polynomial_4.h:43-135: radial factor 1 + r2 (k1 + r2 (k2 + r2 (k3 + r2 k4)))), 4 FOV = FisheyeFOVCamera
 * (camera_fisheye_fov.h:44-176; p[4] = omega, and -- derived, not parameters -- p[5] = two_tan_omega_half_, p[6] = image_radius_) */
/* 5 SIMPLE_PINHOLE (camera_simple_pinhole.h:41-88), 6 SIMPLE_RADIAL (camera_simple_radial.h:43-110), 7 RADIAL (camera_radial.h:43-123),
 * 8 POLYNOMIAL_3 = PolynomialCamera (camera_polynomial.h:43-127), 9 FISHEYE_POLYNOMIAL_2_TANGENTIAL_2 = FisheyeBase over
 * PolynomialTangentialCamera (camera_fisheye_polynomial_tangential.h).  5-7 have ONE focal length: parameter vector [f cx cy ...]
 * (UniqueFocalLength, camera_base_impl.h:65-67,394-407); INSIDE oreg_camera every model is stored as p = [fx fy cx cy q...] with
 * fx = fy = f for those (the classes are constructed that way, camera_radial.cc:43-46), n_params is the model's own count.
 * The names RADIAL_FISHEYE / SIMPLE_RADIAL_FISHEYE create types 7 / 6 (camera_base.cc:73-74 [QUIRK]). */
/* 10 FullOpenCVCamera (camera_full_opencv.h:41-196: [fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6], rational radial factor), 11 RadialFisheyeCamera =
 * FisheyeBase over RadialCamera (camera_radial_fisheye.cc:34-45), 12 SimpleRadialFisheyeCamera = FisheyeBase over SimpleRadialCamera
 * (camera_simple_radial_fisheye.cc:34-45): the three classes of src/camera that camera_base.cc:66-77 never creates. */
static inline int ocam_param_count(int type) {
  static const int counts[13] = {4, 8, 12, 8, 5, 3, 4, 5, 7, 8, 12, 5, 4};
  return (type >= 0 && type < 13) ? counts[type] : 0;
}
static inline int ocam_is_fisheye(int type) { return type == 2 || type == 3 || type == 9 || type == 11 || type == 12; }
static inline int ocam_unique_focal(int type) { return type == 5 || type == 6 || type == 7 || type == 11 || type == 12; }
/* the model whose polynomial the `_plain` functions evaluate: the camera inside the two radial fisheye wrappers */
static inline int ocam_plain_type(int type) { return type == 11 ? 7 : (type == 12 ? 6 : type); }
static inline int ocam_distortion_count(int type) { return ocam_param_count(type) - (ocam_unique_focal(type) ? 3 : 4); }
static inline int ocam_is_poly_tang(int type) { return type == 1 || type == 9; }

/* ---- the polynomial models' Distort on a point already past the (optional) fisheye pre-warp --------------------------- */
static inline void ocam_distort_plain(const oreg_camera* c, float nx, float ny, float* ox, float* oy) {
  const int t = ocam_plain_type(c->type);
  if (t == 0 || t == 5) { *ox = nx; *oy = ny; return; }
  const float* q = c->p + 4;
  if (t == 10) {                            /* camera_full_opencv.h:53-78 */
    const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3], k3 = q[4], k4 = q[5], k5 = q[6], k6 = q[7];
    const float x2 = nx * nx, xy = nx * ny, y2 = ny * ny;
    const float r2 = x2 + y2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float radial = (1.f + k1 * r2 + k2 * r4 + k3 * r6) / (1.f + k4 * r2 + k5 * r4 + k6 * r6);
    const float dx = 2.f * p1 * xy + p2 * (r2 + 2.f * x2);
    const float dy = 2.f * p2 * xy + p1 * (r2 + 2.f * y2);
    *ox = radial * nx + dx; *oy = radial * ny + dy;
    return;
  }
  if (t == 6 || t == 7 || t == 8) {   /* RadialBase::Distort (camera_base_impl_radial.h:52-56) */
    const float r2 = nx * nx + ny * ny;
    float f;
    if (t == 6) f = 1.0f + r2 * q[0];                                   /* camera_simple_radial.h:60-62 */
    else if (t == 7) f = 1.0f + r2 * (q[0] + r2 * q[1]);                /* camera_radial.h:60-65 */
    else f = 1.0f + r2 * (q[0] + r2 * (q[1] + r2 * q[2]));                    /* camera_polynomial.h:58-64 */
    *ox = nx * f; *oy = ny * f;
    return;
  }
  if (c->type == 4) {                       /* camera_fisheye_fov.h:55-63 */
    const float r = sqrtf(nx * nx + ny * ny);
    const float factor = (r < 1e-6f) ? 1.f : (om_atanf(r * q[1]) / (r * q[0]));
    *ox = nx * factor; *oy = ny * factor;
    return;
  }
  const float x2 = nx * nx, xy = nx * ny, y2 = ny * ny;
  const float r2 = x2 + y2;
  if (c->type == 3) {                       /* RadialBase::Distort: point * DistortionFactor(squaredNorm) */
    const float f = 1.0f + r2 * (q[0] + r2 * (q[1] + r2 * (q[2] + r2 * q[3])));
    *ox = nx * f; *oy = ny * f;
    return;
  }
  if (ocam_is_poly_tang(c->type)) {
    const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3];
    const float radial = 1 + r2 * (k1 + r2 * k2);
    const float dx = 2.f * p1 * xy + p2 * (r2 + 2.f * x2);
    const float dy = 2.f * p2 * xy + p1 * (r2 + 2.f * y2);
    *ox = nx * radial + dx; *oy = ny * radial + dy;
    return;
  }
  const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3], k3 = q[4], k4 = q[5], sx1 = q[6], sy1 = q[7];
  const float radial = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
  const float dx = 2.f * p1 * xy + p2 * (r2 + 2.f * x2) + sx1 * r2;
  const float dy = 2.f * p2 * xy + p1 * (r2 + 2.f * y2) + sy1 * r2;
  *ox = nx * radial + dx; *oy = ny * radial + dy;
}

/* DistortedDerivativeByNormalized of the polynomial part: J = [j0 j1; j2 j3] */
static inline void ocam_ddn_plain(const oreg_camera* c, float nx, float ny, float* J) {
  const int t = ocam_plain_type(c->type);
  if (t == 0 || t == 5) { J[0] = 1.f; J[1] = 0.f; J[2] = 0.f; J[3] = 1.f; return; }
  const float* q = c->p + 4;
  if (t == 10) {                            /* camera_full_opencv.h:126-171 */
    const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3], k3 = q[4], k4 = q[5], k5 = q[6], k6 = q[7];
    const float x2 = nx * nx, y2 = ny * ny, xy = nx * ny;
    const float r2 = x2 + y2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float radial_numerator = 1.f + k1 * r2 + k2 * r4 + k3 * r6;
    const float radial_denominator = 1.f + k4 * r2 + k5 * r4 + k6 * r6;
    const float radial = radial_numerator / radial_denominator;
    const float d_radial_numerator = 2 * k1 + 4 * k2 * r2 + 6 * k3 * r4;
    const float d_radial_denominator = 2 * k4 + 4 * k5 * r2 + 6 * k6 * r4;
    const float d_radial = (d_radial_numerator * radial_denominator - d_radial_denominator * radial_numerator) /
                           (radial_denominator * radial_denominator);
    const float d_tan_x_nx = 2 * ny * p1 + 6 * p2 * nx;
    const float d_tan_y_ny = 2 * nx * p2 + 6 * p1 * ny;
    const float d_tan_y_nx = 2 * ny * p2 + 2 * p1 * nx;
    const float d_tan_x_ny = 2 * nx * p1 + 2 * p2 * ny;
    J[0] = radial + x2 * d_radial + d_tan_x_nx;
    J[1] = xy * d_radial + d_tan_x_ny;
    J[2] = xy * d_radial + d_tan_y_nx;
    J[3] = radial + y2 * d_radial + d_tan_y_ny;
    return;
  }
  if (t == 6) {                             /* camera_simple_radial.h:75-88 */
    const float k1 = q[0];
    const float nxs = nx * nx, nys = ny * ny;
    const float ru2 = nxs + nys;
    J[0] = k1 * (ru2 + 2 * nxs) + 1;
    J[1] = 2 * nx * ny * k1;
    J[2] = J[1];
    J[3] = k1 * (ru2 + 2 * nys) + 1;
    return;
  }
  if (t == 7 || t == 8) {                   /* camera_radial.h:82-101, camera_polynomial.h:81-101 */
    const float nx2 = nx * nx, ny2 = ny * ny, nxny = nx * ny;
    const float r2 = nx2 + ny2;
    float term1, term2;
    if (t == 7) {
      const float k1 = q[0], k2 = q[1];
      term1 = 2 * k1 + r2 * (4 * k2);
      term2 = 1 + r2 * (k1 + r2 * (k2));
    } else {
      const float k1 = q[0], k2 = q[1], k3 = q[2];
      term1 = 2 * k1 + r2 * (4 * k2 + r2 * 6 * k3);
      term2 = 1 + r2 * (k1 + r2 * (k2 + r2 * k3));
    }
    J[0] = nx2 * term1 + term2;
    J[1] = nxny * term1;
    J[2] = J[1];
    J[3] = ny2 * term1 + term2;
    return;
  }
  if (c->type == 4) {                       /* camera_fisheye_fov.h:131-160 */
    const float omega = q[0], tt = q[1];
    const float nx_times_ny = nx * ny, nxs = nx * nx, nys = ny * ny;
    const float radius_square = nxs + nys;
    const float radius = sqrtf(radius_square);
    if (radius < 1e-6f) { J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1; return; }
    const float rdw = om_atanf(radius * tt);
    const float tts = tt * tt;
    const float part1 = omega * radius_square * radius;
    const float part2 = omega * (tts * radius_square + 1) * radius_square;
    const float part3 = rdw / (omega * radius);
    J[0] = part3 - (nxs * rdw) / part1 + (nxs * tt) / part2;
    J[1] = nx_times_ny * (tt / part2 - rdw / part1);
    J[2] = J[1];
    J[3] = part3 - (nys * rdw) / part1 + (nys * tt) / part2;
    return;
  }
  const float nx2 = nx * nx, ny2 = ny * ny;
  const float r2 = nx2 + ny2;
  if (c->type == 3) {                       /* camera_polynomial_4.h:78-98 */
    const float k1 = q[0], k2 = q[1], k3 = q[2], k4 = q[3];
    const float nxny = nx * ny;
    const float term1 = 2 * k1 + r2 * (4 * k2 + r2 * (6 * k3 + r2 * 8 * k4));
    const float term2 = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
    J[0] = nx2 * term1 + term2;
    J[1] = nxny * term1;
    J[2] = J[1];
    J[3] = ny2 * term1 + term2;
    return;
  }
  if (ocam_is_poly_tang(c->type)) {
    const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3];
    const float term1 = 2 * k1 + r2 * 4 * k2;
    const float term2 = 1 + r2 * (k1 + r2 * k2);
    J[0] = nx2 * term1 + term2 + 6 * p2 * nx + 2 * p1 * ny;
    J[1] = nx * ny * term1 + 2 * p1 * nx + 2 * p2 * ny;
    J[2] = J[1];
    J[3] = ny2 * term1 + term2 + 2 * p2 * nx + 6 * p1 * ny;
    return;
  }
  const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3], k3 = q[4], k4 = q[5], sx1 = q[6], sy1 = q[7];
  const float nx_ny = nx * ny;
  const float term1 = 2 * k1 + r2 * (4 * k2 + r2 * (6 * k3 + r2 * 8 * k4));
  const float term2 = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
  const float term3 = nx_ny * term1 + 2 * (p1 * nx + p2 * ny);
  J[0] = nx2 * term1 + term2 + 6 * p2 * nx + 2 * p1 * ny + 2 * sx1 * nx;
  J[1] = term3 + 2 * sx1 * ny;
  J[2] = term3 + 2 * sy1 * nx;
  J[3] = ny2 * term1 + term2 + 6 * p1 * ny + 2 * p2 * nx + 2 * sy1 * ny;
}

/* DistortedDerivativeByDistortionParameters of the polynomial part: 2 x (I-4), row-major with row stride `ld` */
static inline void ocam_ddp_plain(const oreg_camera* c, float nx, float ny, float* d0, float* d1) {
  const int t = ocam_plain_type(c->type);
  if (t == 0 || t == 5) return;
  if (t == 10) {                            /* camera_full_opencv.h:83-123: columns k1 k2 p1 p2 k3 k4 k5 k6 */
    const float* q = c->p + 4;
    const float k1 = q[0], k2 = q[1], k3 = q[4], k4 = q[5], k5 = q[6], k6 = q[7];
    const float x2 = nx * nx, y2 = ny * ny;
    const float r2 = x2 + y2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float radial_numerator = 1.f + k1 * r2 + k2 * r4 + k3 * r6;
    const float radial_denominator = 1.f + k4 * r2 + k5 * r4 + k6 * r6;
    const float radial = radial_numerator / radial_denominator;
    d0[0] = nx * r2 / radial_denominator; d0[1] = nx * r4 / radial_denominator; d0[2] = nx * 2.f * ny; d0[3] = (r2 + 2 * x2);
    d0[4] = nx * r6 / radial_denominator;
    d0[5] = -nx * r2 * radial / radial_denominator; d0[6] = -nx * r4 * radial / radial_denominator; d0[7] = -nx * r6 * radial / radial_denominator;
    d1[0] = ny * r2 / radial_denominator; d1[1] = ny * r4 / radial_denominator; d1[2] = (r2 + 2 * y2); d1[3] = ny * 2.f * nx;
    d1[4] = ny * r6 / radial_denominator;
    d1[5] = -ny * r2 * radial / radial_denominator; d1[6] = -ny * r4 * radial / radial_denominator; d1[7] = -ny * r6 * radial / radial_denominator;
    return;
  }
  if (t == 6 || t == 7 || t == 8) {   /* camera_simple_radial.h:67-72, camera_radial.h:70-79, camera_polynomial.h:69-79 */
    const float rs = nx * nx + ny * ny;
    d0[0] = nx * rs; d1[0] = ny * rs;
    if (t != 6) { d0[1] = d0[0] * rs; d1[1] = d1[0] * rs; }
    if (t == 8) { d0[2] = d0[1] * rs; d1[2] = d1[1] * rs; }
    return;
  }
  if (c->type == 4) {                       /* camera_fisheye_fov.h:94-118 */
    const float omega = c->p[4], tt = c->p[5];
    const float radius_square = nx * nx + ny * ny;
    const float radius = sqrtf(radius_square);
    const float four_tan_omega_half_square = tt * tt;
    const float tan_omega_half_square_plus_one = 0.25f * four_tan_omega_half_square + 1.f;
    const float denominator_1 = omega * (four_tan_omega_half_square * radius_square + 1.f);
    const float numerator_2 = om_atanf(tt * radius);
    const float denominator_2 = omega * omega * radius;
    d0[0] = (radius < 1e-6f) ? 0.f : ((nx * tan_omega_half_square_plus_one) / denominator_1 - (nx * numerator_2) / denominator_2);
    d1[0] = (radius < 1e-6f) ? 0.f : ((ny * tan_omega_half_square_plus_one) / denominator_1 - (ny * numerator_2) / denominator_2);
    return;
  }
  if (c->type == 3) {                       /* camera_polynomial_4.h:63-75; radius_square = squaredNorm */
    const float rs = nx * nx + ny * ny;
    d0[0] = nx * rs; d0[1] = d0[0] * rs; d0[2] = d0[1] * rs; d0[3] = d0[2] * rs;
    d1[0] = ny * rs; d1[1] = d1[0] * rs; d1[2] = d1[1] * rs; d1[3] = d1[2] * rs;
    return;
  }
  const float nx2 = nx * nx, ny2 = ny * ny;
  const float two_nx_ny = 2.f * nx * ny;
  const float r2 = nx2 + ny2;
  d0[0] = nx * r2; d0[1] = d0[0] * r2; d0[2] = two_nx_ny; d0[3] = (r2 + 2.f * nx2);
  d1[0] = ny * r2; d1[1] = d1[0] * r2; d1[2] = (r2 + 2.f * ny2); d1[3] = two_nx_ny;
  if (c->type == 2) {
    d0[4] = d0[1] * r2; d0[5] = d0[4] * r2; d0[6] = r2; d0[7] = 0;
    d1[4] = d1[1] * r2; d1[5] = d1[4] * r2; d1[6] = 0; d1[7] = r2;
  }
}

#define OCAM_FISHEYE_EPS 1e-6f

/* Child::Distort */
static inline void ocam_distort(const oreg_camera* c, float nx, float ny, float* ox, float* oy) {
  if (!ocam_is_fisheye(c->type)) { ocam_distort_plain(c, nx, ny, ox, oy); return; }
  const float r = sqrtf(nx * nx + ny * ny);
  if (r > OCAM_FISHEYE_EPS) {
    const float atan_r = om_atan2f(r, 1.f);
    if (atan_r * atan_r > c->inner_cutoff2) { *ox = nx * INFINITY; *oy = ny * INFINITY; return; }
    const float theta_by_r = atan_r / r;
    ocam_distort_plain(c, nx * theta_by_r, ny * theta_by_r, ox, oy);
  } else {
    ocam_distort_plain(c, nx, ny, ox, oy);
  }
}

/* Child::DistortedDerivativeByNormalized */
static inline void ocam_ddn(const oreg_camera* c, float nx, float ny, float* J) {
  if (!ocam_is_fisheye(c->type)) { ocam_ddn_plain(c, nx, ny, J); return; }
  const float nx_ny = nx * ny, nx2 = nx * nx, ny2 = ny * ny;
  const float r2 = nx2 + ny2;
  const float r = sqrtf(r2);
  if (r > OCAM_FISHEYE_EPS) {
    const float atan_r = om_atan2f(r, 1.f);
    if (atan_r * atan_r > c->inner_cutoff2) { J[0] = J[1] = J[2] = J[3] = 0.f; return; }
    const float theta_by_r = atan_r / r;
    const float term1 = r2 * (r2 + 1);
    const float term2 = theta_by_r / r2;
    const float f00 = ny2 * term2 + nx2 / term1;
    const float f01 = nx_ny / term1 - nx_ny * term2;
    const float f10 = f01;
    const float f11 = nx2 * term2 + ny2 / term1;
    float D[4];
    ocam_ddn_plain(c, theta_by_r * nx, theta_by_r * ny, D);
    J[0] = D[0] * f00 + D[1] * f10; J[1] = D[0] * f01 + D[1] * f11;
    J[2] = D[2] * f00 + D[3] * f10; J[3] = D[2] * f01 + D[3] * f11;
  } else {
    ocam_ddn_plain(c, nx, ny, J);
  }
}

/* Child::DistortedDerivativeByDistortionParameters */
static inline void ocam_ddp(const oreg_camera* c, float nx, float ny, float* d0, float* d1) {
  if (!ocam_is_fisheye(c->type)) { ocam_ddp_plain(c, nx, ny, d0, d1); return; }
  const float r = sqrtf(nx * nx + ny * ny);
  if (r > OCAM_FISHEYE_EPS) {
    const float atan_r = om_atan2f(r, 1.f);
    if (atan_r * atan_r > c->inner_cutoff2) { for (int i = 0; i < 8; ++i) d0[i] = d1[i] = 0.f; return; }
    const float theta_by_r = atan_r / r;
    ocam_ddp_plain(c, theta_by_r * nx, theta_by_r * ny, d0, d1);
  } else {
    ocam_ddp_plain(c, nx, ny, d0, d1);
  }
}

/* CameraBaseImpl::NormalizedToImage */
static inline void cam_normalized_to_image(const oreg_camera* c, float nx, float ny, float* ox, float* oy) {
  const float r2 = nx * nx + ny * ny;
  if (isinf(r2) || r2 > c->cutoff2) { *ox = nx * INFINITY; *oy = ny * INFINITY; return; }
  float dx, dy;
  ocam_distort(c, nx, ny, &dx, &dy);
  *ox = c->p[0] * dx + c->p[2];
  *oy = c->p[1] * dy + c->p[3];
}

/* ImageDerivativeByWorld: 2x3 row-major */
static inline void cam_image_deriv_by_world(const oreg_camera* c, const float* P, float* d) {
  const float nx = P[0] / P[2], ny = P[1] / P[2];
  if (nx * nx + ny * ny < c->cutoff2) {
    const float zi = 1.f / P[2];
    float J[4];
    ocam_ddn(c, nx, ny, J);
    /* normalize_deriv = [zi 0 -nx*zi ; 0 zi -ny*zi] */
    const float n02 = (-1.f * nx) * zi, n12 = (-1.f * ny) * zi;
    d[0] = J[0] * zi + J[1] * 0.f; d[1] = J[0] * 0.f + J[1] * zi; d[2] = J[0] * n02 + J[1] * n12;
    d[3] = J[2] * zi + J[3] * 0.f; d[4] = J[2] * 0.f + J[3] * zi; d[5] = J[2] * n02 + J[3] * n12;
  } else {
    for (int i = 0; i < 6; ++i) d[i] = 0.f;
  }
  for (int i = 0; i < 3; ++i) { d[i] = c->p[0] * d[i]; d[3 + i] = c->p[1] * d[3 + i]; }
}

/* ImageDerivativeByIntrinsics: 2 x I row-major (row stride I) */
static inline void cam_image_deriv_by_intrinsics(const oreg_camera* c, const float* P, float* d) {
  const int I = c->n_params;
  const float nx = P[0] / P[2], ny = P[1] / P[2];
  if (nx * nx + ny * ny > c->cutoff2) { for (int i = 0; i < 2 * I; ++i) d[i] = 0.f; return; }
  float dx, dy;
  ocam_distort(c, nx, ny, &dx, &dy);
  if (!ocam_unique_focal(c->type)) {
    d[0] = dx; d[1] = 0.f; d[2] = 1.f; d[3] = 0.f;
    d[I + 0] = 0.f; d[I + 1] = dy; d[I + 2] = 0.f; d[I + 3] = 1.f;
    if (I > 4) {
      ocam_ddp(c, nx, ny, d + 4, d + I + 4);
      for (int i = 4; i < I; ++i) { d[i] = c->p[0] * d[i]; d[I + i] = c->p[1] * d[I + i]; }
    }
  } else {                                  /* [f, cx, cy, distortion...] (camera_base_impl.h:394-407) */
    d[0] = dx; d[1] = 1.f; d[2] = 0.f;
    d[I + 0] = dy; d[I + 1] = 0.f; d[I + 2] = 1.f;
    if (I > 3) {
      ocam_ddp(c, nx, ny, d + 3, d + I + 3);
      for (int i = 3; i < I; ++i) { d[i] = c->p[0] * d[i]; d[I + i] = c->p[1] * d[I + i]; }
    }
  }
}

/* IterativeUndistort of a non-fisheye model (camera_base_impl.h:216-250) */
static inline void ocam_iterative_undistort(const oreg_camera* c, float dx, float dy, float sx, float sy, float* ux, float* uy,
                                            int* converged) {
  *converged = 0;
  float x = sx, y = sy;
  for (int i = 0; i < 100; ++i) {
    float cx, cy;
    ocam_distort_plain(c, x, y, &cx, &cy);
    const float ex = cx - dx, ey = cy - dy;
    if (ex * ex + ey * ey < 1e-10f) { *converged = 1; break; }
    float J[4];
    ocam_ddn_plain(c, x, y, J);
    /* Jd2 = Jd^T Jd */
    const float a = J[0] * J[0] + J[2] * J[2], b = J[0] * J[1] + J[2] * J[3];
    const float cc = J[1] * J[0] + J[3] * J[2], d = J[1] * J[1] + J[3] * J[3];
    /* Eigen 2x2 inverse: invdet = 1 / det; [d -b; -c a] * invdet */
    const float invdet = 1.f / (a * d - cc * b);
    const float i00 = d * invdet, i01 = -b * invdet, i10 = -cc * invdet, i11 = a * invdet;
    /* M = inverse * Jd */
    const float m00 = i00 * J[0] + i01 * J[2], m01 = i00 * J[1] + i01 * J[3];
    const float m10 = i10 * J[0] + i11 * J[2], m11 = i10 * J[1] + i11 * J[3];
    x -= m00 * ex + m01 * ey;
    y -= m10 * ex + m11 * ey;
  }
  *ux = x; *uy = y;
}

/* UndistortFromInside (camera_base_impl.h:278-328): squared radius of the innermost solution and of the second best one
 * (the latter only meaningful when *second_available) */
static inline void ocam_undistort_from_inside(const oreg_camera* c, float dx, float dy, int* converged, float* best_r2,
                                              int* second_available, float* second_r2) {
  *converged = 0; *second_available = 0;
  float best_radius = INFINITY, second_best_radius = INFINITY;
  float bx = 0.f, by = 0.f, sbx = 0.f, sby = 0.f;
  for (int yi = 0; yi < 10; ++yi) {
    const float iy = dy + 1.5f * (yi - 0.5f * 10) / (0.5f * 10);
    for (int xi = 0; xi < 10; ++xi) {
      const float ix = dx + 1.5f * (xi - 0.5f * 10) / (0.5f * 10);
      float rx, ry; int conv;
      ocam_iterative_undistort(c, dx, dy, ix, iy, &rx, &ry, &conv);
      if (!conv) continue;
      const float radius = sqrtf(rx * rx + ry * ry);
      if (radius < 0.99f * best_radius) {
        second_best_radius = best_radius;
        sbx = bx; sby = by;
        *second_available = *converged;
        best_radius = radius; bx = rx; by = ry;
        *converged = 1;
      } else if (radius > 1 / 0.99f * best_radius && radius < 0.99f * second_best_radius) {
        second_best_radius = radius;
        sbx = rx; sby = ry;
        *second_available = 1;
      }
    }
  }
  *best_r2 = bx * bx + by * by;
  *second_r2 = sbx * sbx + sby * sby;
}

/* InitCutoff of a non-fisheye distorted model (camera_base_impl.h:410-463) */
static inline float ocam_init_cutoff(const oreg_camera* c) {
  float min_candidate = 0.f, max_candidate = INFINITY;
  for (int pass = 0; pass < 2; ++pass) {
    const int cnt = pass == 0 ? c->width : c->height;
    for (int i = 0; i < cnt; ++i)
      for (int e = 0; e < 2; ++e) {
        const float px = pass == 0 ? (float)i : (e == 0 ? 0.f : (float)(c->width - 1));
        const float py = pass == 0 ? (e == 0 ? 0.f : (float)(c->height - 1)) : (float)i;
        const float ddx = c->fx_inv * px + c->cx_inv, ddy = c->fy_inv * py + c->cy_inv;
        int conv, second; float r2, s2 = 0.f;
        ocam_undistort_from_inside(c, ddx, ddy, &conv, &r2, &second, &s2);
        if (conv) {
          if (r2 > min_candidate) min_candidate = r2;          /* std::max(r2, min_candidate) */
          if (second && s2 < max_candidate) max_candidate = s2;
        }
      }
  }
  const float a = min_candidate * 1.01f;
  return (max_candidate < a) ? max_candidate : a;              /* std::min(a, max_candidate) */
}

/* FisheyeFOVCamera::Undistort (camera_fisheye_fov.h:76-86), closed form; also its ImageToNormalized (no lookup table, no clamp) */
static inline void ocam_fov_undistort(const oreg_camera* c, float dx, float dy, float* ux, float* uy) {
  const float r = sqrtf(dx * dx + dy * dy);
  const float factor = (r < 1e-6f) ? 1.f : ((r > c->p[6]) ? INFINITY : (om_tanf(r * c->p[4]) / (r * c->p[5])));
  *ux = factor * dx; *uy = factor * dy;
}

/* RadialBase (camera_base_impl_radial.h): 1-D Gauss-Newton on the radius (:59-88), UndistortFromInside over 10 start radii
 * (:104-140), InitCutoff from the farthest image corner (:142-170).  `q` = k1..k4 of Polynomial4Camera. */
static int ocam_radial_type_;   /* the RadialBase child whose DistortionFactor is meant: 3 (Polynomial4), 7 (Radial), 8 (Polynomial) */
static inline float ocam_radial_factor(const float* q, float r2) {
  if (ocam_radial_type_ == 7) return 1.0f + r2 * (q[0] + r2 * q[1]);                              /* camera_radial.h:60-65 */
  if (ocam_radial_type_ == 8) return 1.0f + r2 * (q[0] + r2 * (q[1] + r2 * q[2]));                /* camera_polynomial.h:58-64 */
  return 1.0f + r2 * (q[0] + r2 * (q[1] + r2 * (q[2] + r2 * q[3])));
}
static inline float ocam_radial_dfactor(const float* q, float r2) {      /* DistortedDerivativeByNormalized(r2) of the child, literal expression order */
  if (ocam_radial_type_ == 7) return 1.f + r2 * (3.f * q[0] + r2 * 5.f * q[1]);                   /* camera_radial.h:103-107 */
  if (ocam_radial_type_ == 8) return 1.0f + r2 * (3.0f * q[0] + r2 * (5.0f * q[1] + r2 * 7.0f * q[2]));   /* camera_polynomial.h:103-107 */
  return 1.0f + r2 * (3.0f * q[0] + r2 * (5.0f * q[1] + r2 * (7.0f * q[2] + r2 * (9.0f * q[3]))));      /* camera_polynomial_4.h:100-110 */
}
static inline float ocam_radial_iterative_undistort(const float* q, float distorted_r, float starting_r, int* converged) {
  *converged = 0;
  float ur = starting_r, ur2 = starting_r * starting_r;
  for (int i = 0; i < 100; ++i) {
    const float r_candidate = ur * ocam_radial_factor(q, ur2);
    const float delta_r = r_candidate - distorted_r;
    if (delta_r * delta_r < 1e-10f) { *converged = 1; break; }
    const float deriv = ocam_radial_dfactor(q, ur2);
    const float step = delta_r / deriv;
    ur -= step;
    ur2 = ur * ur;
  }
  return ur;
}
static inline float ocam_radial_init_cutoff(const oreg_camera* c) {
  const float* q = c->p + 4;
  ocam_radial_type_ = ocam_plain_type(c->type);
  /* ImageToDistorted of the four corners (0,0) (0,H) (W,0) (W,H): k_inv applied, Eigen norm = sqrt(x*x + y*y) */
  float test_r = 0.f;
  for (int k = 0; k < 4; ++k) {
    const float px = (k & 2) ? (float)c->width : 0.f, py = (k & 1) ? (float)c->height : 0.f;
    const float x = c->fx_inv * px + c->cx_inv, y = c->fy_inv * py + c->cy_inv;
    const float r = sqrtf(x * x + y * y);
    if (k == 0 || r > test_r) test_r = r;                       /* std::max(test, r), starting from the first corner */
  }
  int converged = 0, second_available = 0;
  float best = INFINITY, second = INFINITY;
  for (int i = 0; i < 10; ++i) {
    /* distorted_radius + kGridHalfExtent * (i - 0.5 * kNumGridSteps) / (0.5f * kNumGridSteps): float * double / float, + float -> float */
    const float init_radius = (float)((double)test_r + (double)1.5f * ((double)i - 0.5 * 10) / (double)(0.5f * 10));
    int tc;
    const float result = ocam_radial_iterative_undistort(q, test_r, init_radius, &tc);
    if (tc) {
      if (result < 0.99f * best) {
        second = best; second_available = converged;
        best = result; converged = 1;
      } else if (result > 1 / 0.99f * best && result < 0.99f * second) {
        second = result; second_available = 1;
      }
    }
  }
  if (converged && best > 0) {
    if (second_available && second > 0) {
      const float a = best * best * 1.01f, b = second * second;
      return (b < a) ? b : a;                                   /* std::min(a, b) */
    }
    return best * best * 1.01f;
  }
  return INFINITY;
}

static inline void ocam_init(oreg_camera* c, int type, int w, int h, const float* params) {
  memset(c, 0, sizeof *c);
  c->type = type; c->width = w; c->height = h; c->n_params = ocam_param_count(type);
  if (ocam_unique_focal(type)) {            /* [f cx cy q...] -> fx = fy = f (camera_simple_radial.cc:44-49) */
    c->p[0] = params[0]; c->p[1] = params[0]; c->p[2] = params[1]; c->p[3] = params[2];
    for (int i = 3; i < c->n_params; ++i) c->p[i + 1] = params[i];
  } else {
    for (int i = 0; i < c->n_params; ++i) c->p[i] = params[i];
  }
  /* CameraBase (camera_base.cc:81-86): k_inv_ = (1.0/fx, 1.0/fy, -1.0*cx/fx, -1.0*cy/fy), double expressions -> float */
  c->fx_inv = (float)(1.0 / (double)c->p[0]); c->fy_inv = (float)(1.0 / (double)c->p[1]);
  c->cx_inv = (float)(-1.0 * (double)c->p[2] / (double)c->p[0]); c->cy_inv = (float)(-1.0 * (double)c->p[3] / (double)c->p[1]);
  c->cutoff2 = INFINITY; c->inner_cutoff2 = INFINITY;
  if (type == 1 || type == 10) c->cutoff2 = ocam_init_cutoff(c);      /* the general InitCutoff in the constructor (camera_full_opencv.cc:41,53) */
  else if (type == 11) c->inner_cutoff2 = ocam_radial_init_cutoff(c); /* the RadialCamera inside (its constructor calls RadialBase::InitCutoff) */
  else if (type == 12) { if (c->p[4] < 0) c->inner_cutoff2 = -1.f / (3 * c->p[4]); }   /* the SimpleRadialCamera inside (camera_simple_radial.cc:51-57) */
  else if (type == 2) {
    /* the inner ThinPrismCamera: a non-fisheye camera with the same parameters */
    oreg_camera inner = *c;
    inner.cutoff2 = INFINITY;
    c->inner_cutoff2 = ocam_init_cutoff(&inner);
  } else if (type == 3) {
    c->inner_cutoff2 = ocam_radial_init_cutoff(c);              /* the inner Polynomial4Camera (its constructor calls InitCutoff) */
  } else if (type == 7 || type == 8) {
    c->cutoff2 = ocam_radial_init_cutoff(c);                    /* RadialBase::InitCutoff in the constructor (camera_radial.cc:40, camera_polynomial.cc:40) */
  } else if (type == 6) {
    if (c->p[4] < 0) c->cutoff2 = -1.f / (3 * c->p[4]);         /* SimpleRadialCamera::InitCutoff (camera_simple_radial.cc:51-57) */
  } else if (type == 9) {
    /* the inner PolynomialTangentialCamera: the general InitCutoff of a camera with the same parameters */
    oreg_camera inner = *c;
    inner.type = 1; inner.cutoff2 = INFINITY;
    c->inner_cutoff2 = ocam_init_cutoff(&inner);
  } else if (type == 4) {
    /* camera_fisheye_fov.cc:37-51: two_tan_omega_half_(2.0f * tan(0.5f * omega_)), image_radius_(M_PI / (2 * omega_)); no cut-off.
     * with g++/libstdc++ <math.h> puts std::tan(float) / std::atan(float) into the global namespace (checked here with a
     * static_assert on decltype(tan(1.0f))), so these are tanf / atanf */
    c->p[5] = 2.0f * om_tanf(0.5f * c->p[4]);
    c->p[6] = (float)(M_PI / (double)(2 * c->p[4]));
  }
}

#endif
