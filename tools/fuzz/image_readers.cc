// Mutation fuzzing of the host image readers (io_image.h, io_jpeg.h) under ASan / UBSan: every input file is damaged 12 000 times (byte
// flips, truncation, injected markers) and decoded as grey and as colour.  Round 2 found one out-of-bounds table index this way.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -Idataset-pipeline_amd/csrc/host \
//       -o /tmp/fuzz_image_readers tools/fuzz/image_readers.cc -lz && /tmp/fuzz_image_readers tests/golden/jpeg_*.jpg some.png
#include "io_image.h"
#include <cstdlib>
int main(int argc, char** argv) {
  // mutate each input file N times in memory, decode grey + colour through the path-based API (writes temp file)
  for (int a = 1; a < argc; ++a) {
    std::ifstream s(argv[a], std::ios::binary);
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
    unsigned seed = 99991 + a * 7;
    for (int it = 0; it < 12000; ++it) {
      std::vector<uint8_t> g = f;
      const int nm = 1 + rand_r(&seed) % 6;
      for (int m = 0; m < nm; ++m) {
        const int kind = rand_r(&seed) % 4;
        const size_t pos = rand_r(&seed) % g.size();
        if (kind == 0) g[pos] = (uint8_t)rand_r(&seed);
        else if (kind == 1) g[pos] ^= (uint8_t)(1u << (rand_r(&seed) % 8));
        else if (kind == 2 && g.size() > 20) g.resize(20 + rand_r(&seed) % (g.size() - 20));
        else if (pos + 2 < g.size()) { g[pos] = 0xFF; g[pos + 1] = (uint8_t)(0xC0 + rand_r(&seed) % 0x30); }
      }
      if (g.empty()) continue;
      int w = 0, h = 0; std::vector<uint8_t> gray, rgb; std::string err;
      if (g.size() >= 2 && g[0] == 0xFF) { e3d_host::load_jpeg(g, &w, &h, &gray, &rgb, &err); e3d_host::load_jpeg(g, &w, &h, &gray, nullptr, &err); }
      else { e3d_host::GrayImage gi; e3d_host::img_detail::load_png(g, &gi, &err, &rgb); }
    }
  }
  printf("done\n");
  return 0;
}
