// Host unit test of e3d::ldlt_solve_upper (dataset-pipeline_amd/csrc/e3d_math.hpp): the lower-triangle form against the plain
// full-matrix form of the same pivoted LDL^T (the order of Eigen's selfadjointView<Upper>().ldlt().solve(b),
// icp_point_to_plane_impl.h:226) -- bit for bit, on positive definite, indefinite, rank-deficient and damped systems.
// Usage: ldlt_test <n> <seed> <kind 0 spd | 1 indefinite | 2 singular | 3 zero rows> [repeat] ; prints "differing_words max_abs_x [us_new us_full]".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../dataset-pipeline_amd/csrc/e3d_math.hpp"

// the full-matrix form (both triangles kept, trailing block mirrored after every elimination step)
static void ldlt_full(const double* A, int n, const double* b, double* x, std::vector<double>& W, std::vector<int>& perm) {
  W.assign((size_t)n * n, 0.0);
  perm.resize(n);
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) W[(size_t)i * n + j] = W[(size_t)j * n + i] = A[(size_t)i * n + j];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = std::fabs(W[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(W[(size_t)i * n + i]);
      if (v > best) { best = v; p = i; }
    }
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(W[(size_t)k * n + j], W[(size_t)p * n + j]);
      for (int i = 0; i < n; ++i) std::swap(W[(size_t)i * n + k], W[(size_t)i * n + p]);
      std::swap(perm[k], perm[p]);
    }
    const double d = W[(size_t)k * n + k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < n; ++i) {
      const double l = W[(size_t)i * n + k] / d;
      for (int j = k + 1; j <= i; ++j) W[(size_t)i * n + j] -= l * W[(size_t)k * n + j];
      W[(size_t)i * n + k] = l;
    }
    for (int i = k + 1; i < n; ++i)
      for (int j = i + 1; j < n; ++j) W[(size_t)i * n + j] = W[(size_t)j * n + i];
  }
  std::vector<double> y(n);
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

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 90, seed = argc > 2 ? atoi(argv[2]) : 1, kind = argc > 3 ? atoi(argv[3]) : 0;
  const int repeat = argc > 4 ? atoi(argv[4]) : 0;
  std::mt19937 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> H((size_t)n * n, 0.0), b(n);
  const int rows = (kind == 2) ? n / 2 : 3 * n;            // kind 2: rank n / 2
  for (int r = 0; r < rows; ++r) {
    std::vector<double> J(n);
    for (int i = 0; i < n; ++i) J[i] = nd(gen) * (1.0 + 100.0 * (i % 7 == 0));
    if (kind == 3) for (int i = 0; i < n; i += 5) J[i] = 0.0;   // unknowns without any residual: zero rows and columns
    const double sgn = (kind == 1 && (r & 1)) ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i)
      for (int j = i; j < n; ++j) H[(size_t)i * n + j] += sgn * J[i] * J[j];
  }
  for (int i = 0; i < n; ++i) b[i] = nd(gen);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) H[(size_t)i * n + j] = -12345.678;     // the lower triangle must never be read
  std::vector<double> x1(n), x2(n), W1, W2;
  std::vector<int> p1, p2;
  e3d::ldlt_solve_upper(H.data(), n, b.data(), x1.data(), W1, p1);
  ldlt_full(H.data(), n, b.data(), x2.data(), W2, p2);
  int differ = 0;
  double mx = 0;
  for (int i = 0; i < n; ++i) {
    if (std::memcmp(&x1[i], &x2[i], sizeof(double)) != 0) ++differ;
    if (p1[i] != p2[i]) ++differ;
    if (std::isfinite(x2[i])) mx = std::max(mx, std::fabs(x2[i]));
  }
  printf("%d %.3e", differ, mx);
  if (repeat > 0) {
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < repeat; ++r) e3d::ldlt_solve_upper(H.data(), n, b.data(), x1.data(), W1, p1);
    auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < repeat; ++r) ldlt_full(H.data(), n, b.data(), x2.data(), W2, p2);
    auto t2 = std::chrono::steady_clock::now();
    printf(" %.1f %.1f", std::chrono::duration<double, std::micro>(t1 - t0).count() / repeat, std::chrono::duration<double, std::micro>(t2 - t1).count() / repeat);
  }
  printf("\n");
  return 0;
}
