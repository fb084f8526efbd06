// Mutation fuzzing of loadPLYFile under ASan / UBSan (header digits and bytes, truncation; clean in round 2).
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -Idataset-pipeline_amd/csrc/host -Iinclude -o /tmp/fp tools/fuzz/ply_reader.cc && mkdir -p /tmp/fuzz && /tmp/fp a.ply b.ply
#include "host_types.h"
#include "io_ply.h"
#include <cstdlib>
#include <fstream>
int main(int argc, char** argv) {
  for (int a = 1; a < argc; ++a) {
    std::ifstream s(argv[a], std::ios::binary);
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
    unsigned seed = 777 + a;
    for (int it = 0; it < 3000; ++it) {
      std::vector<uint8_t> g = f;
      const int nm = 1 + rand_r(&seed) % 5;
      for (int m = 0; m < nm; ++m) {
        const int kind = rand_r(&seed) % 4;
        const size_t hdr = 200 < g.size() ? 200 : g.size();
        const size_t pos = (rand_r(&seed) % 3) ? rand_r(&seed) % hdr : rand_r(&seed) % g.size();
        if (kind == 0) g[pos] = (uint8_t)rand_r(&seed);
        else if (kind == 1) g[pos] = (uint8_t)('0' + rand_r(&seed) % 10);
        else if (kind == 2 && g.size() > 30) g.resize(30 + rand_r(&seed) % (g.size() - 30));
        else g[pos] ^= 0x20;
      }
      { std::ofstream o("/tmp/fuzz/t.ply", std::ios::binary); o.write((const char*)g.data(), g.size()); }
      e3d_host::PointCloud c;
      e3d_host::loadPLYFile("/tmp/fuzz/t.ply", c, (it & 1) != 0);
    }
  }
  printf("done\n");
}
