// Host unit test of the PLY readers (dataset-pipeline_amd/csrc/host/io_ply.h): prints the parsed arrays so that the pytest side can
// compare them with what it wrote.  Usage: ply_reader_test cloud|mesh <file.ply>
#include <cstdio>
#include <cstring>

#include "../../dataset-pipeline_amd/csrc/host/io_ply.h"
#include "../../dataset-pipeline_amd/csrc/host/host_types.h"

int main(int argc, char** argv) {
  using namespace e3d_host;
  if (argc < 3) return 2;
  if (!strcmp(argv[1], "cloud")) {
    PointCloud c;
    if (loadPLYFile(argv[2], c, /*want_rgb=*/true) < 0) return 1;
    printf("%zu %d %d %d\n", c.size(), (int)!c.rgb.empty(), (int)!c.normals.empty(), (int)!c.intensity.empty());
    for (size_t i = 0; i < c.size(); ++i) {
      printf("%.9g %.9g %.9g", c.xyz[3 * i], c.xyz[3 * i + 1], c.xyz[3 * i + 2]);
      if (!c.rgb.empty()) printf(" %d %d %d", c.rgb[3 * i], c.rgb[3 * i + 1], c.rgb[3 * i + 2]);
      if (!c.normals.empty()) printf(" %.9g %.9g %.9g", c.normals[3 * i], c.normals[3 * i + 1], c.normals[3 * i + 2]);
      if (!c.intensity.empty()) printf(" %.9g", c.intensity[i]);
      printf("\n");
    }
    return 0;
  }
  if (!strcmp(argv[1], "rotation")) {            // argv[2..10]: linear part, row-major -> Affine3f::rotation()
    if (argc < 11) return 2;
    Affine3f T;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) T(r, c) = (float)atof(argv[2 + 3 * r + c]);
    double R[9];
    T.rotation(R);
    for (int i = 0; i < 9; ++i) printf("%.17g%c", R[i], i == 8 ? '\n' : ' ');
    return 0;
  }
  std::vector<float> xyz; std::vector<uint32_t> tri;
  if (loadPLYMesh(argv[2], xyz, tri) < 0) return 1;
  printf("%zu %zu\n", xyz.size() / 3, tri.size() / 3);
  for (size_t i = 0; i < xyz.size() / 3; ++i) printf("%.9g %.9g %.9g\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  for (size_t i = 0; i < tri.size() / 3; ++i) printf("%u %u %u\n", tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]);
  return 0;
}
