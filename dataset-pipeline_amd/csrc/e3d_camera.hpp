// Camera models of path (B) as device functions.  f32 throughout, reference expression order, no contraction.
//
//   kPinhole          camera::PinholeCamera               src/camera/camera_pinhole.h:40-86                     I = 4
//   kOpenCV           camera::PolynomialTangentialCamera  src/camera/camera_polynomial_tangential.h:41-159      I = 8
//   kThinPrismFisheye camera::BenchmarkCamera             src/camera/camera_benchmark.h:44-52 =
//                     FisheyeBase (camera_base_impl_fisheye.h:43-162) over ThinPrismCamera (camera_thin_prism.h:43-162)   I = 12
//   kOpenCVFisheye    camera::FisheyePolynomial4Camera    src/camera/camera_fisheye_polynomial_4.h:42-50 =
//                     FisheyeBase over Polynomial4Camera (camera_polynomial_4.h:43-135, RadialBase)                        I = 8
//   kFov              camera::FisheyeFOVCamera            src/camera/camera_fisheye_fov.h:44-176 (Devernay-Faugeras)      I = 5
//                     q[0] = omega; q[1] = two_tan_omega_half_, q[2] = image_radius_ (derived in camera_fisheye_fov.cc:37-51,
//                     not parameters); closed-form Undistort, no cut-off, no lookup table
//   kSimplePinhole    camera::SimplePinholeCamera         src/camera/camera_simple_pinhole.h:41-88    [f cx cy]          I = 3
//   kSimpleRadial     camera::SimpleRadialCamera          src/camera/camera_simple_radial.h:43-110 (RadialBase) [f cx cy k]   I = 4
//   kRadial           camera::RadialCamera                src/camera/camera_radial.h:43-123 (RadialBase) [f cx cy k1 k2]      I = 5
//   kPolynomial3      camera::PolynomialCamera            src/camera/camera_polynomial.h:43-127 (RadialBase) [fx fy cx cy k1 k2 k3]  I = 7
//   kFisheyePolyTang  camera::FisheyePolynomialTangentialCamera  src/camera/camera_fisheye_polynomial_tangential.h =
//                     FisheyeBase over PolynomialTangentialCamera (the kOpenCV polynomial)                          I = 8
//   kFullOpenCV       camera::FullOpenCVCamera            src/camera/camera_full_opencv.h:41-196 [fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6] I = 12
//   kRadialFisheye    camera::RadialFisheyeCamera         src/camera/camera_radial_fisheye.h = FisheyeBase over RadialCamera   [f cx cy k1 k2]  I = 5
//   kSimpleRadialFisheye  camera::SimpleRadialFisheyeCamera  src/camera/camera_simple_radial_fisheye.h = FisheyeBase over SimpleRadialCamera  I = 4
//                     (the three classes of src/camera the reference's factory never creates, camera_base.cc:66-77: reachable through
//                     the C-ABI's camera_type only)
//   The models with ONE focal length (UniqueFocalLength(), camera_base_impl.h:65-67) keep fx = fy = f in CamLevel; their
//   parameter vector, and with it the column order of the intrinsics Jacobian, is [f, cx, cy, distortion...] (:394-407).
//   The COLMAP names RADIAL_FISHEYE / SIMPLE_RADIAL_FISHEYE construct RadialCamera / SimpleRadialCamera in the reference's
//   factory (camera_base.cc:73-74) [QUIRK]: they are aliases of kRadial / kSimpleRadial here.
//
// Shared CRTP base, src/camera/camera_base_impl.h: NormalizedToImage :155-164, IterativeUndistort :216-250,
// UndistortFromInside :278-328, ImageDerivativeByWorld :333-360, ImageDerivativeByIntrinsics :369-408, InitCutoff :410-463.
//
// The model is a template parameter of every function (and of the kernels that call them): the Jacobian widths are
// compile-time constants, so rows stay in registers, and a launch costs no per-thread model dispatch.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/e3d_libm.h"   // bit-defined atanf / atan2f / tanf: the same bits on host and device

namespace e3d {

enum : int { kPinhole = 0, kOpenCV = 1, kThinPrismFisheye = 2, kOpenCVFisheye = 3, kFov = 4, kSimplePinhole = 5, kSimpleRadial = 6,
             kRadial = 7, kPolynomial3 = 8, kFisheyePolyTang = 9, kFullOpenCV = 10, kRadialFisheye = 11, kSimpleRadialFisheye = 12,
             kNumCameraModels = 13 };

__host__ __device__ constexpr int cam_param_count(int model) {
  return model == kPinhole ? 4 : model == kFov ? 5 : (model == kOpenCV || model == kOpenCVFisheye || model == kFisheyePolyTang) ? 8 :
         model == kSimplePinhole ? 3 : (model == kSimpleRadial || model == kSimpleRadialFisheye) ? 4 :
         (model == kRadial || model == kRadialFisheye) ? 5 : model == kPolynomial3 ? 7 : 12;
}
__host__ __device__ constexpr bool cam_is_fisheye(int model) {
  return model == kThinPrismFisheye || model == kOpenCVFisheye || model == kFisheyePolyTang || model == kRadialFisheye || model == kSimpleRadialFisheye;
}
__host__ __device__ constexpr bool cam_unique_focal(int model) {
  return model == kSimplePinhole || model == kSimpleRadial || model == kRadial || model == kRadialFisheye || model == kSimpleRadialFisheye;
}
__host__ __device__ constexpr int cam_distortion_count(int model) { return cam_param_count(model) - (cam_unique_focal(model) ? 3 : 4); }
// the polynomial inside is PolynomialTangentialCamera's (k1 k2 p1 p2)
__host__ __device__ constexpr bool cam_is_poly_tang(int model) { return model == kOpenCV || model == kFisheyePolyTang; }
// RadialBase children (for the fisheye wrappers: the model inside): Distort = point * DistortionFactor(squaredNorm)
__host__ __device__ constexpr bool cam_is_radial(int model) {
  return model == kSimpleRadial || model == kRadial || model == kPolynomial3 || model == kRadialFisheye || model == kSimpleRadialFisheye;
}
// the polynomial inside is SimpleRadialCamera's (k) / RadialCamera's (k1 k2)
__host__ __device__ constexpr bool cam_is_simple_radial(int model) { return model == kSimpleRadial || model == kSimpleRadialFisheye; }
__host__ __device__ constexpr bool cam_is_radial2(int model) { return model == kRadial || model == kRadialFisheye; }

struct CamLevel {
  int model;
  int width, height;
  float fx, fy, cx, cy;
  float q[8];                                   // distortion parameters in the reference's GetParameters order
  float fx_inv, fy_inv, cx_inv, cy_inv;         // CameraBase::k_inv_ (camera_base.cc:84)
  float cutoff2;                                // radius_cutoff_squared_ of the outermost model
  float inner_cutoff2;                          // ... of the non-fisheye model inside a FisheyeBase
};

#define E3D_CAM_INF __uint_as_float(0x7f800000u)

// ---- polynomial part --------------------------------------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ void cam_distort_plain(const CamLevel& c, float nx, float ny, float& ox, float& oy) {
  if constexpr (M == kPinhole || M == kSimplePinhole) {
    ox = nx; oy = ny;
  } else if constexpr (cam_is_radial(M)) {         // RadialBase::Distort (camera_base_impl_radial.h:52-56) with the child's DistortionFactor
    const float r2 = nx * nx + ny * ny;
    float f;
    if constexpr (cam_is_simple_radial(M)) f = 1.0f + r2 * c.q[0];                              // camera_simple_radial.h:60-62
    else if constexpr (cam_is_radial2(M)) f = 1.0f + r2 * (c.q[0] + r2 * c.q[1]);               // camera_radial.h:60-65
    else f = 1.0f + r2 * (c.q[0] + r2 * (c.q[1] + r2 * c.q[2]));                                // camera_polynomial.h:58-64
    ox = nx * f; oy = ny * f;
  } else if constexpr (M == kFov) {                // camera_fisheye_fov.h:55-63
    const float r = sqrtf(nx * nx + ny * ny);
    const float factor = (r < 1e-6f) ? 1.f : (e3d_atanf(r * c.q[1]) / (r * c.q[0]));
    ox = nx * factor; oy = ny * factor;
  } else if constexpr (M == kOpenCVFisheye) {      // RadialBase::Distort: point * DistortionFactor(squaredNorm)
    const float r2 = nx * nx + ny * ny;
    const float f = 1.0f + r2 * (c.q[0] + r2 * (c.q[1] + r2 * (c.q[2] + r2 * c.q[3])));
    ox = nx * f; oy = ny * f;
  } else if constexpr (M == kFullOpenCV) {         // camera_full_opencv.h:53-78; q = [k1 k2 p1 p2 k3 k4 k5 k6]
    const float k1 = c.q[0], k2 = c.q[1], p1 = c.q[2], p2 = c.q[3], k3 = c.q[4], k4 = c.q[5], k5 = c.q[6], k6 = c.q[7];
    const float x2 = nx * nx, xy = nx * ny, y2 = ny * ny;
    const float r2 = x2 + y2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float radial = (((1.f + k1 * r2) + k2 * r4) + k3 * r6) / (((1.f + k4 * r2) + k5 * r4) + k6 * r6);
    const float dx = 2.f * p1 * xy + p2 * (r2 + 2.f * x2);
    const float dy = 2.f * p2 * xy + p1 * (r2 + 2.f * y2);
    ox = radial * nx + dx; oy = radial * ny + dy;
  } else {
    const float x2 = nx * nx, xy = nx * ny, y2 = ny * ny;
    const float r2 = x2 + y2;
    const float k1 = c.q[0], k2 = c.q[1], p1 = c.q[2], p2 = c.q[3];
    if constexpr (cam_is_poly_tang(M)) {
      const float radial = 1 + r2 * (k1 + r2 * k2);
      const float dx = 2.f * p1 * xy + p2 * (r2 + 2.f * x2);
      const float dy = 2.f * p2 * xy + p1 * (r2 + 2.f * y2);
      ox = nx * radial + dx; oy = ny * radial + dy;
    } else {
      const float k3 = c.q[4], k4 = c.q[5], sx1 = c.q[6], sy1 = c.q[7];
      const float radial = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
      const float dx = 2.f * p1 * xy + p2 * (r2 + 2.f * x2) + sx1 * r2;
      const float dy = 2.f * p2 * xy + p1 * (r2 + 2.f * y2) + sy1 * r2;
      ox = nx * radial + dx; oy = ny * radial + dy;
    }
  }
}

// DistortedDerivativeByNormalized, J = [J0 J1; J2 J3]
template <int M>
__device__ __forceinline__ void cam_ddn_plain(const CamLevel& c, float nx, float ny, float* J) {
  if constexpr (M == kPinhole || M == kSimplePinhole) {
    J[0] = 1.f; J[1] = 0.f; J[2] = 0.f; J[3] = 1.f;
  } else if constexpr (cam_is_simple_radial(M)) {  // camera_simple_radial.h:75-88
    const float k1 = c.q[0];
    const float nxs = nx * nx, nys = ny * ny;
    const float ru2 = nxs + nys;
    J[0] = k1 * (ru2 + 2 * nxs) + 1;
    J[1] = 2 * nx * ny * k1;
    J[2] = J[1];
    J[3] = k1 * (ru2 + 2 * nys) + 1;
  } else if constexpr (cam_is_radial2(M) || M == kPolynomial3) {     // camera_radial.h:82-101, camera_polynomial.h:81-101
    const float nx2 = nx * nx, ny2 = ny * ny, nxny = nx * ny;
    const float r2 = nx2 + ny2;
    float term1, term2;
    if constexpr (cam_is_radial2(M)) {
      const float k1 = c.q[0], k2 = c.q[1];
      term1 = 2 * k1 + r2 * (4 * k2);
      term2 = 1 + r2 * (k1 + r2 * (k2));
    } else {
      const float k1 = c.q[0], k2 = c.q[1], k3 = c.q[2];
      term1 = 2 * k1 + r2 * (4 * k2 + r2 * 6 * k3);
      term2 = 1 + r2 * (k1 + r2 * (k2 + r2 * k3));
    }
    J[0] = nx2 * term1 + term2;
    J[1] = nxny * term1;
    J[2] = J[1];
    J[3] = ny2 * term1 + term2;
  } else if constexpr (M == kFov) {                // camera_fisheye_fov.h:131-160
    const float omega = c.q[0], tt = c.q[1];
    const float nx_times_ny = nx * ny, nxs = nx * nx, nys = ny * ny;
    const float radius_square = nxs + nys;
    const float radius = sqrtf(radius_square);
    if (radius < 1e-6f) { J[0] = 1.f; J[1] = 0.f; J[2] = 0.f; J[3] = 1.f; return; }
    const float rdw = e3d_atanf(radius * tt);
    const float tts = tt * tt;
    const float part1 = omega * radius_square * radius;
    const float part2 = omega * (tts * radius_square + 1) * radius_square;
    const float part3 = rdw / (omega * radius);
    J[0] = part3 - (nxs * rdw) / part1 + (nxs * tt) / part2;
    J[1] = nx_times_ny * (tt / part2 - rdw / part1);
    J[2] = J[1];
    J[3] = part3 - (nys * rdw) / part1 + (nys * tt) / part2;
  } else if constexpr (M == kFullOpenCV) {         // camera_full_opencv.h:126-171
    const float k1 = c.q[0], k2 = c.q[1], p1 = c.q[2], p2 = c.q[3], k3 = c.q[4], k4 = c.q[5], k5 = c.q[6], k6 = c.q[7];
    const float x2 = nx * nx, y2 = ny * ny, xy = nx * ny;
    const float r2 = x2 + y2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float radial_numerator = ((1.f + k1 * r2) + k2 * r4) + k3 * r6;
    const float radial_denominator = ((1.f + k4 * r2) + k5 * r4) + k6 * r6;
    const float radial = radial_numerator / radial_denominator;
    const float d_radial_numerator = (2 * k1 + 4 * k2 * r2) + 6 * k3 * r4;
    const float d_radial_denominator = (2 * k4 + 4 * k5 * r2) + 6 * k6 * r4;
    const float d_radial = (d_radial_numerator * radial_denominator - d_radial_denominator * radial_numerator) /
                           (radial_denominator * radial_denominator);
    const float d_tan_x_nx = 2 * ny * p1 + 6 * p2 * nx;
    const float d_tan_y_ny = 2 * nx * p2 + 6 * p1 * ny;
    const float d_tan_y_nx = 2 * ny * p2 + 2 * p1 * nx;
    const float d_tan_x_ny = 2 * nx * p1 + 2 * p2 * ny;
    J[0] = (radial + x2 * d_radial) + d_tan_x_nx;
    J[1] = xy * d_radial + d_tan_x_ny;
    J[2] = xy * d_radial + d_tan_y_nx;
    J[3] = (radial + y2 * d_radial) + d_tan_y_ny;
  } else if constexpr (M == kOpenCVFisheye) {      // camera_polynomial_4.h:78-98
    const float nx2 = nx * nx, ny2 = ny * ny, nxny = nx * ny;
    const float r2 = nx2 + ny2;
    const float k1 = c.q[0], k2 = c.q[1], k3 = c.q[2], k4 = c.q[3];
    const float term1 = 2 * k1 + r2 * (4 * k2 + r2 * (6 * k3 + r2 * 8 * k4));
    const float term2 = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
    J[0] = nx2 * term1 + term2;
    J[1] = nxny * term1;
    J[2] = J[1];
    J[3] = ny2 * term1 + term2;
  } else {
    const float nx2 = nx * nx, ny2 = ny * ny;
    const float r2 = nx2 + ny2;
    const float k1 = c.q[0], k2 = c.q[1], p1 = c.q[2], p2 = c.q[3];
    if constexpr (cam_is_poly_tang(M)) {
      const float term1 = 2 * k1 + r2 * 4 * k2;
      const float term2 = 1 + r2 * (k1 + r2 * k2);
      J[0] = nx2 * term1 + term2 + 6 * p2 * nx + 2 * p1 * ny;
      J[1] = nx * ny * term1 + 2 * p1 * nx + 2 * p2 * ny;
      J[2] = J[1];
      J[3] = ny2 * term1 + term2 + 2 * p2 * nx + 6 * p1 * ny;
    } else {
      const float k3 = c.q[4], k4 = c.q[5], sx1 = c.q[6], sy1 = c.q[7];
      const float nx_ny = nx * ny;
      const float term1 = 2 * k1 + r2 * (4 * k2 + r2 * (6 * k3 + r2 * 8 * k4));
      const float term2 = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
      const float term3 = nx_ny * term1 + 2 * (p1 * nx + p2 * ny);
      J[0] = nx2 * term1 + term2 + 6 * p2 * nx + 2 * p1 * ny + 2 * sx1 * nx;
      J[1] = term3 + 2 * sx1 * ny;
      J[2] = term3 + 2 * sy1 * nx;
      J[3] = ny2 * term1 + term2 + 6 * p1 * ny + 2 * p2 * nx + 2 * sy1 * ny;
    }
  }
}

// DistortedDerivativeByDistortionParameters: rows d0, d1 of I - 4 entries
template <int M>
__device__ __forceinline__ void cam_ddp_plain(const CamLevel& c, float nx, float ny, float* d0, float* d1) {
  if constexpr (M == kFov) {                       // camera_fisheye_fov.h:94-118, one column (omega)
    const float omega = c.q[0], tt = c.q[1];
    const float radius_square = nx * nx + ny * ny;
    const float radius = sqrtf(radius_square);
    const float four_tan_omega_half_square = tt * tt;
    const float tan_omega_half_square_plus_one = 0.25f * four_tan_omega_half_square + 1.f;
    const float denominator_1 = omega * (four_tan_omega_half_square * radius_square + 1.f);
    const float numerator_2 = e3d_atanf(tt * radius);
    const float denominator_2 = omega * omega * radius;
    d0[0] = (radius < 1e-6f) ? 0.f : ((nx * tan_omega_half_square_plus_one) / denominator_1 - (nx * numerator_2) / denominator_2);
    d1[0] = (radius < 1e-6f) ? 0.f : ((ny * tan_omega_half_square_plus_one) / denominator_1 - (ny * numerator_2) / denominator_2);
  } else if constexpr (M == kOpenCVFisheye) {      // camera_polynomial_4.h:63-75
    const float rs = nx * nx + ny * ny;
    d0[0] = nx * rs; d0[1] = d0[0] * rs; d0[2] = d0[1] * rs; d0[3] = d0[2] * rs;
    d1[0] = ny * rs; d1[1] = d1[0] * rs; d1[2] = d1[1] * rs; d1[3] = d1[2] * rs;
  } else if constexpr (M == kFullOpenCV) {         // camera_full_opencv.h:83-123: columns k1 k2 p1 p2 k3 k4 k5 k6
    const float k1 = c.q[0], k2 = c.q[1], k3 = c.q[4], k4 = c.q[5], k5 = c.q[6], k6 = c.q[7];
    const float x2 = nx * nx, y2 = ny * ny;
    const float r2 = x2 + y2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float radial_numerator = ((1.f + k1 * r2) + k2 * r4) + k3 * r6;
    const float radial_denominator = ((1.f + k4 * r2) + k5 * r4) + k6 * r6;
    const float radial = radial_numerator / radial_denominator;
    d0[0] = nx * r2 / radial_denominator; d0[1] = nx * r4 / radial_denominator; d0[2] = nx * 2.f * ny; d0[3] = (r2 + 2 * x2);
    d0[4] = nx * r6 / radial_denominator;
    d0[5] = -nx * r2 * radial / radial_denominator; d0[6] = -nx * r4 * radial / radial_denominator; d0[7] = -nx * r6 * radial / radial_denominator;
    d1[0] = ny * r2 / radial_denominator; d1[1] = ny * r4 / radial_denominator; d1[2] = (r2 + 2 * y2); d1[3] = ny * 2.f * nx;
    d1[4] = ny * r6 / radial_denominator;
    d1[5] = -ny * r2 * radial / radial_denominator; d1[6] = -ny * r4 * radial / radial_denominator; d1[7] = -ny * r6 * radial / radial_denominator;
  } else if constexpr (cam_is_radial(M)) {         // camera_simple_radial.h:67-72, camera_radial.h:70-79, camera_polynomial.h:69-79
    const float rs = nx * nx + ny * ny;
    d0[0] = nx * rs; d1[0] = ny * rs;
    if constexpr (!cam_is_simple_radial(M)) { d0[1] = d0[0] * rs; d1[1] = d1[0] * rs; }
    if constexpr (M == kPolynomial3) { d0[2] = d0[1] * rs; d1[2] = d1[1] * rs; }
  } else if constexpr (M != kPinhole && M != kSimplePinhole) {
    const float nx2 = nx * nx, ny2 = ny * ny;
    const float two_nx_ny = 2.f * nx * ny;
    const float r2 = nx2 + ny2;
    d0[0] = nx * r2; d0[1] = d0[0] * r2; d0[2] = two_nx_ny; d0[3] = (r2 + 2.f * nx2);
    d1[0] = ny * r2; d1[1] = d1[0] * r2; d1[2] = (r2 + 2.f * ny2); d1[3] = two_nx_ny;
    if constexpr (M == kThinPrismFisheye) {
      d0[4] = d0[1] * r2; d0[5] = d0[4] * r2; d0[6] = r2; d0[7] = 0;
      d1[4] = d1[1] * r2; d1[5] = d1[4] * r2; d1[6] = 0; d1[7] = r2;
    }
  }
}

constexpr float kFisheyeEpsilon = 1e-6f;

// The fisheye models' Distort and both derivative functions each start from atan(r) of the same normalized point
// (camera_base_impl_fisheye.h:66-153): one number, worth ~150 binary64 instructions in e3d_libm.h.  A kernel that calls several of
// them for one point evaluates it once (cam_theta) and hands it down; the three-argument forms below compute it themselves.
// Same operands, same operations: the same bits either way.
struct CamTheta { float r, atan_r; };           // atan_r is set iff r > kFisheyeEpsilon
template <int M>
__device__ __forceinline__ CamTheta cam_theta(float nx, float ny) {
  CamTheta t{0.f, 0.f};
  if constexpr (cam_is_fisheye(M)) {
    t.r = sqrtf(nx * nx + ny * ny);
    if (t.r > kFisheyeEpsilon) t.atan_r = e3d_atan2f(t.r, 1.f);
  }
  return t;
}

// ---- Child::Distort / DistortedDerivativeByNormalized / ...ByDistortionParameters ---------------------------------------------
template <int M>
__device__ __forceinline__ void cam_distort(const CamLevel& c, float nx, float ny, const CamTheta& th, float& ox, float& oy) {
  if constexpr (!cam_is_fisheye(M)) {
    cam_distort_plain<M>(c, nx, ny, ox, oy);
  } else {
    const float r = th.r;
    if (r > kFisheyeEpsilon) {
      const float atan_r = th.atan_r;
      if (atan_r * atan_r > c.inner_cutoff2) { ox = nx * E3D_CAM_INF; oy = ny * E3D_CAM_INF; return; }
      const float theta_by_r = atan_r / r;
      cam_distort_plain<M>(c, nx * theta_by_r, ny * theta_by_r, ox, oy);
    } else {
      cam_distort_plain<M>(c, nx, ny, ox, oy);
    }
  }
}

template <int M>
__device__ __forceinline__ void cam_distort(const CamLevel& c, float nx, float ny, float& ox, float& oy) {
  cam_distort<M>(c, nx, ny, cam_theta<M>(nx, ny), ox, oy);
}

template <int M>
__device__ __forceinline__ void cam_ddn(const CamLevel& c, float nx, float ny, const CamTheta& th, float* J) {
  if constexpr (!cam_is_fisheye(M)) {
    cam_ddn_plain<M>(c, nx, ny, J);
  } else {
    const float nx_ny = nx * ny, nx2 = nx * nx, ny2 = ny * ny;
    const float r2 = nx2 + ny2;
    const float r = th.r;                       // = sqrtf(r2)
    if (r > kFisheyeEpsilon) {
      const float atan_r = th.atan_r;
      if (atan_r * atan_r > c.inner_cutoff2) { J[0] = J[1] = J[2] = J[3] = 0.f; return; }
      const float theta_by_r = atan_r / r;
      const float term1 = r2 * (r2 + 1);
      const float term2 = theta_by_r / r2;
      const float f00 = ny2 * term2 + nx2 / term1;
      const float f01 = nx_ny / term1 - nx_ny * term2;
      const float f10 = f01;
      const float f11 = nx2 * term2 + ny2 / term1;
      float D[4];
      cam_ddn_plain<M>(c, theta_by_r * nx, theta_by_r * ny, D);
      J[0] = D[0] * f00 + D[1] * f10; J[1] = D[0] * f01 + D[1] * f11;
      J[2] = D[2] * f00 + D[3] * f10; J[3] = D[2] * f01 + D[3] * f11;
    } else {
      cam_ddn_plain<M>(c, nx, ny, J);
    }
  }
}

template <int M>
__device__ __forceinline__ void cam_ddn(const CamLevel& c, float nx, float ny, float* J) { cam_ddn<M>(c, nx, ny, cam_theta<M>(nx, ny), J); }

template <int M>
__device__ __forceinline__ void cam_ddp(const CamLevel& c, float nx, float ny, const CamTheta& th, float* d0, float* d1) {
  if constexpr (!cam_is_fisheye(M)) {
    cam_ddp_plain<M>(c, nx, ny, d0, d1);
  } else {
    const float r = th.r;
    if (r > kFisheyeEpsilon) {
      const float atan_r = th.atan_r;
      if (atan_r * atan_r > c.inner_cutoff2) {
#pragma unroll
        for (int i = 0; i < cam_distortion_count(M); ++i) { d0[i] = 0.f; d1[i] = 0.f; }
        return;
      }
      const float theta_by_r = atan_r / r;
      cam_ddp_plain<M>(c, theta_by_r * nx, theta_by_r * ny, d0, d1);
    } else {
      cam_ddp_plain<M>(c, nx, ny, d0, d1);
    }
  }
}

template <int M>
__device__ __forceinline__ void cam_ddp(const CamLevel& c, float nx, float ny, float* d0, float* d1) {
  cam_ddp<M>(c, nx, ny, cam_theta<M>(nx, ny), d0, d1);
}

// ---- CameraBaseImpl ---------------------------------------------------------------------------------------------------------------
// (th: cam_theta of the SAME normalized point the function forms itself, X / Z and Y / Z)
template <int M>
__device__ __forceinline__ void cam_normalized_to_image(const CamLevel& c, float nx, float ny, const CamTheta& th, float& ox, float& oy) {
  const float r2 = nx * nx + ny * ny;
  if (isinf(r2) || r2 > c.cutoff2) { ox = nx * E3D_CAM_INF; oy = ny * E3D_CAM_INF; return; }
  float dx, dy;
  cam_distort<M>(c, nx, ny, th, dx, dy);
  ox = c.fx * dx + c.cx;
  oy = c.fy * dy + c.cy;
}
template <int M>
__device__ __forceinline__ void cam_normalized_to_image(const CamLevel& c, float nx, float ny, float& ox, float& oy) {
  const float r2 = nx * nx + ny * ny;
  if (isinf(r2) || r2 > c.cutoff2) { ox = nx * E3D_CAM_INF; oy = ny * E3D_CAM_INF; return; }
  cam_normalized_to_image<M>(c, nx, ny, cam_theta<M>(nx, ny), ox, oy);
}

// 2 x 3 row-major
template <int M>
__device__ __forceinline__ void cam_image_deriv_by_world(const CamLevel& c, float X, float Y, float Z, const CamTheta& th, float* d) {
  const float nx = X / Z, ny = Y / Z;
  if (nx * nx + ny * ny < c.cutoff2) {
    const float zi = 1.f / Z;
    float J[4];
    cam_ddn<M>(c, nx, ny, th, J);
    const float n02 = (-1.f * nx) * zi, n12 = (-1.f * ny) * zi;      // normalize_deriv = [zi 0 n02; 0 zi n12]
    d[0] = J[0] * zi + J[1] * 0.f; d[1] = J[0] * 0.f + J[1] * zi; d[2] = J[0] * n02 + J[1] * n12;
    d[3] = J[2] * zi + J[3] * 0.f; d[4] = J[2] * 0.f + J[3] * zi; d[5] = J[2] * n02 + J[3] * n12;
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { d[i] = c.fx * d[i]; d[3 + i] = c.fy * d[3 + i]; }
}

template <int M>
__device__ __forceinline__ void cam_image_deriv_by_world(const CamLevel& c, float X, float Y, float Z, float* d) {
  cam_image_deriv_by_world<M>(c, X, Y, Z, cam_theta<M>(X / Z, Y / Z), d);
}

// 2 x I row-major (row stride I)
template <int M>
__device__ __forceinline__ void cam_image_deriv_by_intrinsics(const CamLevel& c, float X, float Y, float Z, const CamTheta& th, float* d) {
  constexpr int I = cam_param_count(M);
  const float nx = X / Z, ny = Y / Z;
  if (nx * nx + ny * ny > c.cutoff2) {
#pragma unroll
    for (int i = 0; i < 2 * I; ++i) d[i] = 0.f;
    return;
  }
  float dx, dy;
  cam_distort<M>(c, nx, ny, th, dx, dy);
  if constexpr (!cam_unique_focal(M)) {
    d[0] = dx; d[1] = 0.f; d[2] = 1.f; d[3] = 0.f;
    d[I + 0] = 0.f; d[I + 1] = dy; d[I + 2] = 0.f; d[I + 3] = 1.f;
    if constexpr (I > 4) {
      cam_ddp<M>(c, nx, ny, th, d + 4, d + I + 4);
#pragma unroll
      for (int i = 4; i < I; ++i) { d[i] = c.fx * d[i]; d[I + i] = c.fy * d[I + i]; }
    }
  } else {                                        // [f, cx, cy, distortion...] (camera_base_impl.h:394-407)
    d[0] = dx; d[1] = 1.f; d[2] = 0.f;
    d[I + 0] = dy; d[I + 1] = 0.f; d[I + 2] = 1.f;
    if constexpr (I > 3) {
      cam_ddp<M>(c, nx, ny, th, d + 3, d + I + 3);
#pragma unroll
      for (int i = 3; i < I; ++i) { d[i] = c.fx * d[i]; d[I + i] = c.fy * d[I + i]; }
    }
  }
}

template <int M>
__device__ __forceinline__ void cam_image_deriv_by_intrinsics(const CamLevel& c, float X, float Y, float Z, float* d) {
  cam_image_deriv_by_intrinsics<M>(c, X, Y, Z, cam_theta<M>(X / Z, Y / Z), d);
}

// FisheyeFOVCamera::Undistort (camera_fisheye_fov.h:76-86): closed form, infinity past image_radius_
__device__ __forceinline__ void cam_fov_undistort(const CamLevel& c, float dx, float dy, float& ux, float& uy) {
  const float r = sqrtf(dx * dx + dy * dy);
  const float factor = (r < 1e-6f) ? 1.f : ((r > c.q[2]) ? E3D_CAM_INF : (e3d_tanf(r * c.q[0]) / (r * c.q[1])));
  ux = factor * dx; uy = factor * dy;
}

// IterativeUndistort of the non-fisheye model M (Gauss-Newton, <= 100 iterations, |delta|^2 < 1e-10)
template <int M>
__device__ __forceinline__ bool cam_iterative_undistort(const CamLevel& c, float dx, float dy, float sx, float sy, float& ux, float& uy) {
  bool converged = false;
  float x = sx, y = sy;
  for (int i = 0; i < 100; ++i) {
    float cx, cy;
    cam_distort_plain<M>(c, x, y, cx, cy);
    const float ex = cx - dx, ey = cy - dy;
    if (ex * ex + ey * ey < 1e-10f) { converged = true; break; }
    float J[4];
    cam_ddn_plain<M>(c, x, y, J);
    const float a = J[0] * J[0] + J[2] * J[2], b = J[0] * J[1] + J[2] * J[3];        // Jd^T Jd
    const float cc = J[1] * J[0] + J[3] * J[2], d = J[1] * J[1] + J[3] * J[3];
    const float invdet = 1.f / (a * d - cc * b);                                      // Eigen's 2x2 inverse
    const float i00 = d * invdet, i01 = -b * invdet, i10 = -cc * invdet, i11 = a * invdet;
    const float m00 = i00 * J[0] + i01 * J[2], m01 = i00 * J[1] + i01 * J[3];
    const float m10 = i10 * J[0] + i11 * J[2], m11 = i10 * J[1] + i11 * J[3];
    x -= m00 * ex + m01 * ey;
    y -= m10 * ex + m11 * ey;
  }
  ux = x; uy = y;
  return converged;
}

}  // namespace e3d
