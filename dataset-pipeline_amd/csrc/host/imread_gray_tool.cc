// e3d_imread_gray <image> <out.pgm> -- decodes an image exactly like the tools do (cv::imread(..., IMREAD_GRAYSCALE) stand-in,
// io_image.h) and writes it as a binary PGM.  Used by the tests to pin the PNG / JPEG decoders; handy for inspecting inputs.
// e3d_imread_gray --color <image> <out>: the cv::imread(path) stand-in (colour); <out> by its extension: .ppm raw, .png, .jpg through
// the cv::imwrite stand-ins (io_image.h: imwrite_color).
#include <cstdio>
#include <cstring>
#include <iostream>

#include "io_image.h"

int main(int argc, char** argv) {
  if (argc == 4 && !strcmp(argv[1], "--color")) {
    std::string err;
    const e3d_host::ColorImage c = e3d_host::imread_color(argv[2], &err);
    if (c.empty()) { std::cerr << err << std::endl; return 1; }
    if (!e3d_host::imwrite_color(argv[3], c, &err)) { std::cerr << err << std::endl; return 1; }
    return 0;
  }
  if (argc != 3) { std::cerr << "Usage: " << argv[0] << " [--color] <image> <out>" << std::endl; return 1; }
  std::string err;
  const e3d_host::GrayImage g = e3d_host::imread_gray(argv[1], &err);
  if (g.empty()) { std::cerr << err << std::endl; return 1; }
  FILE* f = fopen(argv[2], "wb");
  if (!f) { std::cerr << "cannot write " << argv[2] << std::endl; return 1; }
  fprintf(f, "P5\n%d %d\n255\n", g.width, g.height);
  fwrite(g.data.data(), 1, g.data.size(), f);
  fclose(f);
  return 0;
}
