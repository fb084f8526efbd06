// Host unit test of e3d::ArrowSystem (dataset-pipeline_amd/csrc/e3d_math.hpp): the Schur-complement solve of the arrow-structured
// normal equations of IntrinsicsAndPoseOptimizer against the reference-order dense pivoted LDLT on the same matrix.
// Usage: arrow_system_test <n_shared> <n_poses> <seed> <empty_pose_index or -1> ; prints "max_abs_err max_abs_x pack_ok".
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../dataset-pipeline_amd/csrc/e3d_math.hpp"

int main(int argc, char** argv) {
  const int s = argc > 1 ? atoi(argv[1]) : 12, np = argc > 2 ? atoi(argv[2]) : 40, seed = argc > 3 ? atoi(argv[3]) : 1;
  const int empty = argc > 4 ? atoi(argv[4]) : -1;
  std::mt19937 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  e3d::ArrowSystem Hs;
  Hs.reset(s, np);
  const int V = Hs.size();
  // H = sum of J J^T over random rows that touch the shared block and ONE pose block each, like the residual Jacobians
  for (int p = 0; p < np; ++p) {
    if (p == empty) continue;                       // an image without observations: all-zero pose block
    for (int k = 0; k < 40; ++k) {
      std::vector<double> J(V, 0.0);
      for (int i = 0; i < s; ++i) J[i] = nd(gen);
      for (int r = 0; r < 6; ++r) J[s + 6 * p + r] = 3.0 * nd(gen);
      const double res = nd(gen);
      for (int i = 0; i < V; ++i) {
        if (J[i] == 0.0) continue;
        for (int j = i; j < V; ++j)
          if (J[j] != 0.0 && !Hs.add(i, j, J[i] * J[j])) { printf("pattern violation\n"); return 2; }
        Hs.b[i] += res * J[i];
      }
    }
  }
  if (Hs.add(s, s + 6, 1.0)) { printf("pose-pose entry accepted\n"); return 2; }
  const double damping = 1.0 + 0.125;
  std::vector<double> x(V), xd(V), H, W;
  std::vector<int> perm;
  Hs.solve(damping, x.data());
  Hs.to_dense(H);
  for (int i = 0; i < V; ++i) H[(size_t)i * V + i] *= damping;
  e3d::ldlt_solve_upper(H.data(), V, Hs.b.data(), xd.data(), W, perm);
  double err = 0, mx = 0;
  for (int i = 0; i < V; ++i) { err = std::max(err, std::fabs(x[i] - xd[i])); mx = std::max(mx, std::fabs(xd[i])); }
  // pack / unpack round trip
  std::vector<double> buf(Hs.packed_size());
  Hs.pack(buf.data());
  e3d::ArrowSystem H2;
  H2.reset(s, np);
  H2.unpack(buf.data());
  const bool pack_ok = H2.A == Hs.A && H2.B == Hs.B && H2.D == Hs.D && H2.b == Hs.b && buf.size() == (size_t)s * s + (size_t)np * (6 * s + 36) + (size_t)V;
  printf("%.3e %.3e %d\n", err, mx, pack_ok ? 1 : 0);
  return 0;
}
