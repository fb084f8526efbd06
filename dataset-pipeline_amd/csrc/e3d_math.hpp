// e3d_math.hpp -- host-side pose arithmetic of the ICP driver: Sophus SE3 semantics in the
// reference's precisions, and the dense symmetric solve.  Compiled with -ffp-contract=off.
//
// Reference semantics followed (thirdparty/sophus is header-only and needs Eigen, which this
// image lacks, so the arithmetic is restated; f32 operation orders are documented in DESIGN.md):
//   SO3::expAndTheta  thirdparty/sophus/so3.hpp:585-621
//   SE3::exp          thirdparty/sophus/se3.hpp:763-785
//   SO3 product + renormalisation  so3.hpp:329-343, 483-489, 297-303
//   SO3 * point       so3.hpp:362-370
//   SE3 product       se3.hpp:308-312 ; cast se3.hpp:128-131
//   pose update       src/icp/icp_point_to_plane_impl.h:235
#pragma once

#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <vector>

namespace e3d {

struct Quatf { float w = 1.f, x = 0.f, y = 0.f, z = 0.f; };
struct SE3f { Quatf q; float t[3] = {0.f, 0.f, 0.f}; };
struct SE3d { double w = 1, x = 0, y = 0, z = 0; double t[3] = {0, 0, 0}; };

// Eigen::Quaternion::toRotationMatrix, row-major 3x3.
template <typename S>
inline void quat_to_matrix(S w, S x, S y, S z, S* R) {
  const S tx = S(2) * x, ty = S(2) * y, tz = S(2) * z;
  const S twx = tx * w, twy = ty * w, twz = tz * w;
  const S txx = tx * x, txy = ty * x, txz = tz * x;
  const S tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = S(1) - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = S(1) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = S(1) - (txx + tyy);
}

inline void normalize(Quatf& q) {
  // Eigen packet reduction of {x,y,z,w}^2: (x2 + z2) + (y2 + w2)
  const float x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
  const float len = std::sqrt((x2 + z2) + (y2 + w2));
  q.w = q.w / len; q.x = q.x / len; q.y = q.y / len; q.z = q.z / len;
}

inline SE3d se3_exp(const double* a) {
  constexpr double kEps = 1e-10;
  const double ox = a[3], oy = a[4], oz = a[5];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  double theta, imag, real;
  if (theta_sq < kEps * kEps) {
    theta = 0.0;
    const double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    theta = std::sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    imag = std::sin(half_theta) / theta;
    real = std::cos(half_theta);
  }
  SE3d r;
  r.w = real; r.x = imag * ox; r.y = imag * oy; r.z = imag * oz;
  const double Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  double Om2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += Om[3 * i + k] * Om[3 * k + j];
      Om2[3 * i + j] = s;
    }
  double V[9];
  if (theta < kEps) {
    quat_to_matrix<double>(r.w, r.x, r.y, r.z, V);
  } else {
    const double tsq = theta * theta;
    const double c1 = (1.0 - std::cos(theta)) / tsq;
    const double c2 = (theta - std::sin(theta)) / (tsq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
  }
  for (int i = 0; i < 3; ++i) r.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
  return r;
}

inline SE3f se3_cast(const SE3d& d) {
  SE3f f;
  f.q.w = (float)d.w; f.q.x = (float)d.x; f.q.y = (float)d.y; f.q.z = (float)d.z;
  normalize(f.q);
  for (int i = 0; i < 3; ++i) f.t[i] = (float)d.t[i];
  return f;
}

inline SE3f se3_mul(const SE3f& a, const SE3f& b) {
  SE3f r;
  r.q.w = a.q.w * b.q.w - a.q.x * b.q.x - a.q.y * b.q.y - a.q.z * b.q.z;
  r.q.x = a.q.w * b.q.x + a.q.x * b.q.w + a.q.y * b.q.z - a.q.z * b.q.y;
  r.q.y = a.q.w * b.q.y + a.q.y * b.q.w + a.q.z * b.q.x - a.q.x * b.q.z;
  r.q.z = a.q.w * b.q.z + a.q.z * b.q.w + a.q.x * b.q.y - a.q.y * b.q.x;
  normalize(r.q);
  const float v[3] = {a.q.x, a.q.y, a.q.z};
  const float* p = b.t;
  float uv[3] = {v[1] * p[2] - v[2] * p[1], v[2] * p[0] - v[0] * p[2], v[0] * p[1] - v[1] * p[0]};
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  const float c[3] = {v[1] * uv[2] - v[2] * uv[1], v[2] * uv[0] - v[0] * uv[2], v[0] * uv[1] - v[1] * uv[0]};
  for (int i = 0; i < 3; ++i) {
    const float rp = (p[i] + a.q.w * uv[i]) + c[i];
    r.t[i] = a.t[i] + rp;
  }
  return r;
}

// pose' = SE3d::exp(-x).cast<float>() * pose
inline SE3f se3_apply_update(const double* x, const SE3f& pose) {
  double mx[6];
  for (int i = 0; i < 6; ++i) mx[i] = -x[i];
  return se3_mul(se3_cast(se3_exp(mx)), pose);
}

// 3-term f32 inner product in Eigen's unrolled order e0 + (e1 + e2).
inline float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
  const float e0 = a0 * b0, e1 = a1 * b1, e2 = a2 * b2;
  return e0 + (e1 + e2);
}

// x = A^{-1} b reading only the upper triangle of the symmetric n x n matrix A (row-major):
// LDL^T with diagonal pivoting (Eigen: A.selfadjointView<Upper>().ldlt().solve(b),
// icp_point_to_plane_impl.h:226).
inline void ldlt_solve_upper(const double* A, int n, const double* b, double* x,
                             std::vector<double>& W, std::vector<int>& perm) {
  // Only the LOWER triangle of W is kept (W[i][j], j <= i): the symmetric exchange of a pivot touches the lower triangle alone, and
  // the column of step k is copied out before its entries are overwritten by the multipliers.  Operation for operation the
  // arithmetic of the plain full-matrix form (every l = W[i][k] / d, every W[i][j] -= l * W[j][k] in the same order; round 6 --
  // the full-matrix form re-mirrored the trailing block after every step, n^3 / 3 strided copies: 0.13 ms of a 90-unknown solve,
  // thirty of them per outer iteration of a 16-scan job); tests/cpp/ldlt_test.cc compares the two bit for bit.
  W.resize((size_t)n * n + (size_t)2 * n);
  perm.resize(n);
  double* const col = W.data() + (size_t)n * n;        // column k below the diagonal, before the division
  double* const y = col + n;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) W[(size_t)i * n + j] = A[(size_t)j * n + i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = std::fabs(W[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(W[(size_t)i * n + i]);
      if (v > best) { best = v; p = i; }
    }
    if (p != k) {                                         // rows / columns k <-> p of the symmetric matrix, lower triangle only
      for (int j = 0; j < k; ++j) std::swap(W[(size_t)k * n + j], W[(size_t)p * n + j]);
      std::swap(W[(size_t)k * n + k], W[(size_t)p * n + p]);
      for (int m = k + 1; m < p; ++m) std::swap(W[(size_t)m * n + k], W[(size_t)p * n + m]);
      for (int m = p + 1; m < n; ++m) std::swap(W[(size_t)m * n + k], W[(size_t)m * n + p]);
      std::swap(perm[k], perm[p]);
    }
    const double d = W[(size_t)k * n + k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < n; ++i) col[i] = W[(size_t)i * n + k];
    for (int i = k + 1; i < n; ++i) {
      double* const Wi = W.data() + (size_t)i * n;
      const double l = col[i] / d;
      for (int j = k + 1; j <= i; ++j) Wi[j] -= l * col[j];
      Wi[k] = l;
    }
  }
  for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
  for (int i = 0; i < n; ++i) {
    double s = y[i];
    for (int j = 0; j < i; ++j) s -= W[(size_t)i * n + j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < n; ++i) {
    const double d = W[(size_t)i * n + i];
    y[i] = (d != 0.0) ? y[i] / d : 0.0;
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int j = i + 1; j < n; ++j) s -= W[(size_t)j * n + i] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}

// The normal equations of IntrinsicsAndPoseOptimizer have arrow structure: a dense block of `s` shared unknowns (all intrinsics,
// then all rig extrinsics) and one 6 x 6 block per pose that couples to the shared block only (AccumulateOnHAndB,
// intrinsics_and_pose_optimizer.cc:841-938, never writes a pose-pose block of two different poses).  The reference stores and
// solves H densely (Eigen LDLT on V x V); for hundreds of images that is O(V^3) host work and V^2 doubles per exchange.  ArrowSystem
// keeps only the non-zero blocks: s^2 + np (6 s + 36) numbers, solved through the Schur complement on the shared block in
// O(np (s^2 + s)) -- the same solution up to f64 rounding.
struct ArrowSystem {
  int s = 0, np = 0;                 // V = s + 6 np
  std::vector<double> A;             // s x s, row-major, upper triangle used
  std::vector<double> B;             // np x 6 x s: B[(6 p + r) s + c] = H[c][s + 6 p + r]
  std::vector<double> D;             // np x 6 x 6, upper triangle used
  std::vector<double> b;             // V
  void reset(int shared, int poses) {
    s = shared; np = poses;
    A.assign((size_t)s * s, 0.0); B.assign((size_t)np * 6 * s, 0.0); D.assign((size_t)np * 36, 0.0); b.assign((size_t)s + 6 * (size_t)np, 0.0);
  }
  int size() const { return s + 6 * np; }
  // H[gr][gc] += v for gr <= gc; returns false for an entry outside the arrow pattern (two different poses)
  bool add(int gr, int gc, double v) {
    if (gc < s) { A[(size_t)gr * s + gc] += v; return true; }
    const int p = (gc - s) / 6, r = (gc - s) % 6;
    if (gr < s) { B[((size_t)6 * p + r) * s + gr] += v; return true; }
    if ((gr - s) / 6 != p) return false;
    D[(size_t)p * 36 + (size_t)((gr - s) % 6) * 6 + r] += v;
    return true;
  }
  size_t packed_size() const { return A.size() + B.size() + D.size() + b.size(); }
  void pack(double* out) const {
    size_t o = 0;
    for (const std::vector<double>* v : {&A, &B, &D, &b}) { std::copy(v->begin(), v->end(), out + o); o += v->size(); }
  }
  void unpack(const double* in) {
    size_t o = 0;
    for (std::vector<double>* v : {&A, &B, &D, &b}) { std::copy(in + o, in + o + v->size(), v->begin()); o += v->size(); }
  }
  // dense V x V upper triangle (row-major), for the reference-order LDLT on small systems
  void to_dense(std::vector<double>& H) const {
    const int V = size();
    H.assign((size_t)V * V, 0.0);
    for (int i = 0; i < s; ++i) for (int j = i; j < s; ++j) H[(size_t)i * V + j] = A[(size_t)i * s + j];
    for (int p = 0; p < np; ++p)
      for (int r = 0; r < 6; ++r) {
        const int g = s + 6 * p + r;
        for (int c = 0; c < s; ++c) H[(size_t)c * V + g] = B[((size_t)6 * p + r) * s + c];
        for (int c = r; c < 6; ++c) H[(size_t)g * V + (s + 6 * p + c)] = D[(size_t)p * 36 + (size_t)r * 6 + c];
      }
  }
  // x = (H with its diagonal scaled by `damping`)^-1 b
  void solve(double damping, double* x) const {
    std::vector<double> W, S(A), rhs(b.begin(), b.begin() + s), col(6), sol(6), Dp(36);
    std::vector<int> perm;
    std::vector<double> DinvB((size_t)np * 6 * s), Dinvb((size_t)np * 6);
    for (int i = 0; i < s; ++i) S[(size_t)i * s + i] *= damping;
    for (int p = 0; p < np; ++p) {
      for (int i = 0; i < 36; ++i) Dp[i] = D[(size_t)p * 36 + i];
      for (int i = 0; i < 6; ++i) Dp[i * 6 + i] *= damping;
      const double* Bp = &B[(size_t)p * 6 * s];
      // D_p^-1 applied to the coupling columns and to b_p (6 x 6 pivoted LDLT per right-hand side; a block without observations
      // is all zero and yields zeros, like the dense solver's zero pivots)
      for (int c = 0; c <= s; ++c) {
        for (int r = 0; r < 6; ++r) col[r] = (c < s) ? Bp[(size_t)r * s + c] : b[(size_t)s + 6 * p + r];
        ldlt_solve_upper(Dp.data(), 6, col.data(), sol.data(), W, perm);
        for (int r = 0; r < 6; ++r) { if (c < s) DinvB[((size_t)6 * p + r) * s + c] = sol[r]; else Dinvb[(size_t)6 * p + r] = sol[r]; }
      }
      for (int i = 0; i < s; ++i) {
        double acc = 0;
        for (int r = 0; r < 6; ++r) acc += Bp[(size_t)r * s + i] * Dinvb[(size_t)6 * p + r];
        rhs[i] -= acc;
        for (int j = i; j < s; ++j) {
          double a2 = 0;
          for (int r = 0; r < 6; ++r) a2 += Bp[(size_t)r * s + i] * DinvB[((size_t)6 * p + r) * s + j];
          S[(size_t)i * s + j] -= a2;
        }
      }
    }
    std::vector<double> xs(s > 0 ? s : 1);
    if (s > 0) ldlt_solve_upper(S.data(), s, rhs.data(), xs.data(), W, perm);
    for (int i = 0; i < s; ++i) x[i] = xs[i];
    for (int p = 0; p < np; ++p)
      for (int r = 0; r < 6; ++r) {
        double v = Dinvb[(size_t)6 * p + r];
        for (int c = 0; c < s; ++c) v -= DinvB[((size_t)6 * p + r) * s + c] * xs[c];
        x[s + 6 * p + r] = v;
      }
  }
};

// Smallest / largest singular value of the 3x3 linear part of a row-major 3x4 affine (f64, Jacobi on L^T L).
inline void singular_value_range_3x3(const float* T, double* smin, double* smax) {
  double A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)T[4 * k + i] * (double)T[4 * k + j];
      A[3 * i + j] = s;
    }
  for (int sweep = 0; sweep < 32; ++sweep) {
    double off = std::fabs(A[1]) + std::fabs(A[2]) + std::fabs(A[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double app = A[3 * p + p], aqq = A[3 * q + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
      }
  }
  double m = A[0], M = A[0];
  if (A[4] < m) m = A[4];
  if (A[8] < m) m = A[8];
  if (A[4] > M) M = A[4];
  if (A[8] > M) M = A[8];
  *smin = m > 0 ? std::sqrt(m) : 0.0;
  *smax = M > 0 ? std::sqrt(M) : 0.0;
}
inline double min_singular_value_3x3(const float* T) { double a, b; singular_value_range_3x3(T, &a, &b); return a; }
inline double max_singular_value_3x3(const float* T) { double a, b; singular_value_range_3x3(T, &a, &b); return b; }

// Inverse of the 3x3 linear part (f64 adjugate), row-major out[9]; returns false if singular.
inline bool invert_3x3(const float* T, double* out) {
  const double m[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                     m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (std::fabs(det) < 1e-300) return false;
  const double id = 1.0 / det;
  out[0] = (m[4] * m[8] - m[5] * m[7]) * id; out[1] = (m[2] * m[7] - m[1] * m[8]) * id; out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  out[3] = (m[5] * m[6] - m[3] * m[8]) * id; out[4] = (m[0] * m[8] - m[2] * m[6]) * id; out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  out[6] = (m[3] * m[7] - m[4] * m[6]) * id; out[7] = (m[1] * m[6] - m[0] * m[7]) * id; out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return true;
}

}  // namespace e3d
