// Mutation fuzzing of the COLMAP text / rigs.json / MeshLab project readers under ASan / UBSan (6 000 damaged files each; clean in round 2).
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -Idataset-pipeline_amd/csrc/host -Iinclude -o /tmp/ft tools/fuzz/text_readers.cc && /tmp/ft   (writes its inputs to /tmp/fuzz/)
#include "io_colmap.h"
#include "io_mlp.h"
#include <cstdlib>
#include <fstream>
using namespace e3d_host;
static std::string mutate(const std::string& f, unsigned* seed) {
  std::string g = f;
  const int nm = 1 + rand_r(seed) % 6;
  static const char* tok[] = {"\"", "[", "]", "{", "}", ",", ":", "-", "1e999", "nan", "<", ">", "/", "=", " ", "\n", "99999999999999999999", "0"};
  for (int m = 0; m < nm && !g.empty(); ++m) {
    const int kind = rand_r(seed) % 5;
    const size_t pos = rand_r(seed) % g.size();
    if (kind == 0) g[pos] = (char)(rand_r(seed) % 256);
    else if (kind == 1) g.erase(pos, 1 + rand_r(seed) % 8);
    else if (kind == 2) g.insert(pos, tok[rand_r(seed) % 18]);
    else if (kind == 3) g.resize(pos);
    else g[pos] = (char)('0' + rand_r(seed) % 10);
  }
  return g;
}
int main() {
  const std::string cams = "# Camera list\n1 PINHOLE 640 480 500 501 320.5 240.5\n2 THIN_PRISM_FISHEYE 6048 4032 3400 3400 3000 2000 0.1 0.01 0 0 0 0 0 0\n3 SIMPLE_RADIAL 64 48 50 32 24 0.1\n";
  const std::string imgs = "# Image list\n1 0.9 0.1 0.2 0.3 1 2 3 1 dslr/a.jpg\n1 2 -1 3 4 5\n2 1 0 0 0 0 0 0 2 cam/b.png\n\n";
  const std::string rigs = "[\n  {\n    \"ref_camera_id\": 1,\n    \"cameras\": [\n      {\"camera_id\": 1, \"image_prefix\": \"cam0\"},\n      {\"camera_id\": 2, \"image_prefix\": \"cam1\"}\n    ]\n  }\n]\n";
  const std::string mlp = "<!DOCTYPE MeshLabDocument>\n<MeshLabProject>\n <MeshGroup>\n  <MLMesh label=\"scan &amp; 1\" filename=\"scan1.ply\">\n   <MLMatrix44>\n1 0 0 0.5 \n0 1 0 0 \n0 0 1 0 \n0 0 0 1 \n</MLMatrix44>\n  </MLMesh>\n  <MLMesh label=\"b\" filename=\"b.ply\"/>\n </MeshGroup>\n <RasterGroup/>\n</MeshLabProject>\n";
  unsigned seed = 4242;
  for (int it = 0; it < 6000; ++it) {
    { std::ofstream o("/tmp/fuzz/c.txt"); o << mutate(cams, &seed); }
    { std::ofstream o("/tmp/fuzz/i.txt"); o << mutate(imgs, &seed); }
    { std::ofstream o("/tmp/fuzz/r.json"); o << mutate(rigs, &seed); }
    { std::ofstream o("/tmp/fuzz/m.mlp"); o << mutate(mlp, &seed); }
    std::map<int, ColmapCamera> C; ReadColmapCameras("/tmp/fuzz/c.txt", &C);
    std::map<int, ColmapImage> I; ReadColmapImages("/tmp/fuzz/i.txt", &I);
    std::vector<ColmapRig> R; ReadColmapRigs("/tmp/fuzz/r.json", &R);
    std::vector<MlpMesh> M; ParseMeshLabProject("/tmp/fuzz/m.mlp", &M);
  }
  printf("done\n");
}
