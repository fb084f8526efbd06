// oracle/oracle_shuffle.cc -- TEST INFRASTRUCTURE ONLY.
// Problem::DeterminePointNeighbors (src/opt/problem.cc:706-786) on the oracle's exact k-NN lists, with the reference's own
// random stream: std::shuffle(indices.begin() + 1, indices.end(), std::mt19937(0)) from libstdc++, one generator per call,
// consumed scan by scan, point by point.
#include <algorithm>
#include <cstdint>
#include <random>
#include <vector>

extern "C" {
#include "oracle_kdtree.h"

int oracle_determine_point_neighbors(const float* xyz, size_t n, const uint8_t* scan_indices, int scan_count, int limit_to_same_scan,
                                     int neighbor_count, int candidate_count, uint32_t* out) {
  std::mt19937 generator(0);
  const int k = candidate_count + 1;
  std::vector<int32_t> idx(k);
  std::vector<float> dist(k);
  std::vector<int> indices(k);
  auto run = [&](const std::vector<float>& pts, const std::vector<size_t>* original) -> int {
    const size_t m = pts.size() / 3;
    if (m < (size_t)k) return -1;
    okd_tree* tree = okd_build(pts.data(), m);
    for (size_t i = 0; i < m; ++i) {
      okd_knn(tree, pts.data() + 3 * i, k, idx.data(), dist.data());
      for (int j = 0; j < k; ++j) indices[j] = idx[j];
      std::shuffle(indices.begin() + 1, indices.end(), generator);
      const size_t o = original ? (*original)[i] : i;
      for (int j = 0; j < neighbor_count; ++j) out[o * neighbor_count + j] = (uint32_t)(original ? (*original)[indices[j + 1]] : (size_t)indices[j + 1]);
    }
    okd_free(tree);
    return 0;
  };
  if (limit_to_same_scan) {
    std::vector<std::vector<float>> clouds(scan_count);
    std::vector<std::vector<size_t>> orig(scan_count);
    for (size_t i = 0; i < n; ++i) {
      const int s = scan_indices[i];
      clouds[s].insert(clouds[s].end(), xyz + 3 * i, xyz + 3 * i + 3);
      orig[s].push_back(i);
    }
    for (int s = 0; s < scan_count; ++s) if (run(clouds[s], &orig[s]) < 0) return -1;
    return 0;
  }
  return run(std::vector<float>(xyz, xyz + 3 * n), nullptr);
}
}
