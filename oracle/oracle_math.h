/*
 * oracle/oracle_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Scalar restatement of the small-matrix / Lie-group arithmetic used by the
 * reference's ICP path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link this; the product library never does.
 *
 * Canonical float operation orders (see DESIGN.md "f32 operation orders"):
 *   dot3 / row-times-vector : a0*b0 + (a1*b1 + a2*b2)     (Eigen 3.3 unrolled
 *       redux of a 3-vector, recalled; reference call sites
 *       src/icp/icp_point_to_plane_impl.h:146-151,158,185)
 *   pcl::transformPointCloudWithNormals (PCL 1.10 SSE Transformer, recalled):
 *       p' = x*c0 + (y*c1 + (z*c2 + c3)),  n' = x*c0 + (y*c1 + z*c2)
 *       (reference call sites src/icp/icp_point_to_plane.cc:120-123,192-195)
 *   L2_Simple squared distance (FLANN 1.9.1, recalled): (dx*dx + dy*dy) + dz*dz
 * Everything is compiled with -ffp-contract=off (reference flags are
 * -O2 -msse2 -msse3, CMakeLists.txt:82 => no FMA contraction).
 */
#ifndef E3D_ORACLE_MATH_H
#define E3D_ORACLE_MATH_H

#include <math.h>
#include <stddef.h>
#include <string.h>

/* ---- f32 helpers ------------------------------------------------------- */

static inline float om_dot3f(const float* a, const float* b) {
  float e0 = a[0] * b[0];
  float e1 = a[1] * b[1];
  float e2 = a[2] * b[2];
  return e0 + (e1 + e2);
}

/* R row-major 3x3, y = R*p + t with Eigen's expression order. */
static inline void om_rot_trans_f(const float* R, const float* t,
                                  const float* p, float* y) {
  for (int i = 0; i < 3; ++i) {
    float s = om_dot3f(R + 3 * i, p);
    y[i] = s + t[i];
  }
}

static inline void om_rot_f(const float* R, const float* p, float* y) {
  for (int i = 0; i < 3; ++i) y[i] = om_dot3f(R + 3 * i, p);
}

/* PCL Transformer::se3 / so3 order; T is a row-major 3x4 affine. */
static inline void om_pcl_se3(const float* T, const float* p, float* y) {
  for (int i = 0; i < 3; ++i) {
    float p0 = p[0] * T[4 * i + 0];
    float p1 = p[1] * T[4 * i + 1];
    float p2 = p[2] * T[4 * i + 2];
    y[i] = p0 + (p1 + (p2 + T[4 * i + 3]));
  }
}
static inline void om_pcl_so3(const float* T, const float* p, float* y) {
  for (int i = 0; i < 3; ++i) {
    float p0 = p[0] * T[4 * i + 0];
    float p1 = p[1] * T[4 * i + 1];
    float p2 = p[2] * T[4 * i + 2];
    y[i] = p0 + (p1 + p2);
  }
}

static inline float om_sqdist3f(const float* a, const float* b) {
  float dx = a[0] - b[0];
  float dy = a[1] - b[1];
  float dz = a[2] - b[2];
  float acc = dx * dx;
  acc = acc + dy * dy;
  acc = acc + dz * dz;
  return acc;
}

/* ---- SE3 (Sophus restatement) ------------------------------------------
 * thirdparty/sophus/so3.hpp:585-621 (expAndTheta), se3.hpp:763-785 (exp),
 * so3.hpp:167-169 (cast -> renormalise), so3.hpp:329-343 (product ->
 * renormalise), so3.hpp:362-370 (point action), se3.hpp:308-312 (product).
 * Quaternion stored as {w,x,y,z}. */

typedef struct { float q[4]; float t[3]; } om_se3f;   /* q = w,x,y,z */
typedef struct { double q[4]; double t[3]; } om_se3d;

static inline void om_se3f_identity(om_se3f* s) {
  s->q[0] = 1.f; s->q[1] = s->q[2] = s->q[3] = 0.f;
  s->t[0] = s->t[1] = s->t[2] = 0.f;
}

/* Eigen Quaternion::toRotationMatrix (recalled from Eigen 3.3.7
 * Geometry/Quaternion.h), row-major output. */
static inline void om_quat_to_R_f(const float* q, float* R) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}
static inline void om_quat_to_R_d(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2. * x, ty = 2. * y, tz = 2. * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1. - (tyy + tzz); R[1] = txy - twz;        R[2] = txz + twy;
  R[3] = txy + twz;        R[4] = 1. - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;        R[7] = tyz + twx;        R[8] = 1. - (txx + tyy);
}

/* SO3::normalize (so3.hpp:297-303): coeffs /= coeffs.norm().  Eigen's SSE
 * horizontal add of the 4 squared coeffs {x,y,z,w}: (x2+z2)+(y2+w2) (recalled). */
static inline void om_quat_normalize_f(float* q) {
  float w2 = q[0] * q[0], x2 = q[1] * q[1], y2 = q[2] * q[2], z2 = q[3] * q[3];
  float len = sqrtf((x2 + z2) + (y2 + w2));
  q[0] = q[0] / len; q[1] = q[1] / len; q[2] = q[2] / len; q[3] = q[3] / len;
}

/* SE3d::exp(a), a = [upsilon(3), omega(3)]. */
static inline void om_se3d_exp(const double* a, om_se3d* out) {
  const double eps = 1e-10;               /* Sophus::Constants<double>::epsilon */
  const double* ups = a;
  const double* om = a + 3;
  double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double theta, imag, real;
  if (theta_sq < eps * eps) {
    theta = 0.0;
    double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    theta = sqrt(theta_sq);
    double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  out->q[0] = real; out->q[1] = imag * om[0]; out->q[2] = imag * om[1];
  out->q[3] = imag * om[2];

  double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double Om2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += Om[3 * i + k] * Om[3 * k + j];
      Om2[3 * i + j] = s;
    }
  double V[9];
  if (theta < eps) {
    om_quat_to_R_d(out->q, V);
  } else {
    double tsq = theta * theta;
    double c1 = (1.0 - cos(theta)) / tsq;
    double c2 = (theta - sin(theta)) / (tsq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
  }
  for (int i = 0; i < 3; ++i)
    out->t[i] = V[3 * i] * ups[0] + V[3 * i + 1] * ups[1] + V[3 * i + 2] * ups[2];
}

/* SE3d -> SE3f cast (se3.hpp:128-131): quaternion cast then renormalised in f32. */
static inline void om_se3_cast_f(const om_se3d* d, om_se3f* f) {
  for (int i = 0; i < 4; ++i) f->q[i] = (float)d->q[i];
  om_quat_normalize_f(f->q);
  for (int i = 0; i < 3; ++i) f->t[i] = (float)d->t[i];
}

static inline void om_cross_f(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* c = a * b for SE3f (se3.hpp:308-312). */
static inline void om_se3f_mul(const om_se3f* a, const om_se3f* b, om_se3f* c) {
  const float aw = a->q[0], ax = a->q[1], ay = a->q[2], az = a->q[3];
  const float bw = b->q[0], bx = b->q[1], by = b->q[2], bz = b->q[3];
  om_se3f r;
  r.q[0] = aw * bw - ax * bx - ay * by - az * bz;
  r.q[1] = aw * bx + ax * bw + ay * bz - az * by;
  r.q[2] = aw * by + ay * bw + az * bx - ax * bz;
  r.q[3] = aw * bz + az * bw + ax * by - ay * bx;
  om_quat_normalize_f(r.q);
  /* translation = a.t + a.so3 * b.t  with  so3*p = p + w*uv + qv x uv, uv = 2 (qv x p) */
  float uv[3], c2[3];
  om_cross_f(a->q + 1, b->t, uv);
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  om_cross_f(a->q + 1, uv, c2);
  for (int i = 0; i < 3; ++i) {
    float rp = (b->t[i] + aw * uv[i]) + c2[i];
    r.t[i] = a->t[i] + rp;
  }
  *c = r;
}

/* ---- dense symmetric solve (Eigen selfadjointView<Upper>().ldlt().solve) ----
 * LDL^T with diagonal pivoting on the upper triangle of A (n x n, row-major,
 * only entries j>=i read).  Work buffers are caller-provided:
 * W: n*n doubles, perm: n ints.  Returns x in-place in b. */
static inline void om_ldlt_solve_upper(const double* A, int n, double* b,
                                       double* W, int* perm) {
  /* build full symmetric copy */
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) { W[i * n + j] = A[i * n + j]; W[j * n + i] = A[i * n + j]; }
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    /* pivot: largest |diagonal| in the trailing block */
    int p = k; double best = fabs(W[k * n + k]);
    for (int i = k + 1; i < n; ++i) { double v = fabs(W[i * n + i]); if (v > best) { best = v; p = i; } }
    if (p != k) {
      for (int j = 0; j < n; ++j) { double t = W[k * n + j]; W[k * n + j] = W[p * n + j]; W[p * n + j] = t; }
      for (int i = 0; i < n; ++i) { double t = W[i * n + k]; W[i * n + k] = W[i * n + p]; W[i * n + p] = t; }
      int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
    }
    double d = W[k * n + k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < n; ++i) {
      double l = W[i * n + k] / d;
      /* row k (upper part) still holds the un-eliminated symmetric values a_kj */
      for (int j = k + 1; j <= i; ++j) W[i * n + j] -= l * W[k * n + j];
      W[i * n + k] = l;
    }
    for (int i = k + 1; i < n; ++i)
      for (int j = i + 1; j < n; ++j) W[i * n + j] = W[j * n + i];
  }
  /* solve: P A P^T = L D L^T */
  double tmp_stack[64];
  double* tmp = (n <= 64) ? tmp_stack : (double*)__builtin_alloca(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) tmp[i] = b[perm[i]];
  for (int i = 0; i < n; ++i) { double s = tmp[i]; for (int j = 0; j < i; ++j) s -= W[i * n + j] * tmp[j]; tmp[i] = s; }
  for (int i = 0; i < n; ++i) { double d = W[i * n + i]; tmp[i] = (d != 0.0) ? tmp[i] / d : 0.0; }
  for (int i = n - 1; i >= 0; --i) { double s = tmp[i]; for (int j = i + 1; j < n; ++j) s -= W[j * n + i] * tmp[j]; tmp[i] = s; }
  for (int i = 0; i < n; ++i) b[perm[i]] = tmp[i];
}

#endif  /* E3D_ORACLE_MATH_H */
