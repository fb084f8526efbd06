// io_image.h -- grey-scale image loading and the pyramids of opt::Image (src/opt/image.cc:40-154) for the host tools.
//
// The reference reads images with cv::imread(path, IMREAD_GRAYSCALE) (image.cc:47) and masks with IMREAD_ANYDEPTH
// (:77).  OpenCV is not available here; this header decodes PNG itself on top of zlib (8/16-bit grey, grey+alpha, RGB,
// RGBA, palette; non-interlaced), binary PGM / PPM, and baseline JPEG (io_jpeg.h: the luminance plane with libjpeg's
// integer IDCT, which is what IMREAD_GRAYSCALE returns).  Colour -> grey for PNG / PPM follows what OpenCV's PNG reader does, libpng's
// png_set_rgb_to_gray(1, 0.299, 0.587): 15-bit fixed-point weights 9797 / 19234 / 3737 (recalled; unpinned).
//
// Pyramid: cv::resize(prev, Size(0.5 * cols, 0.5 * rows), 0.5, 0.5, INTER_AREA) (image.cc:114-119).  For even sizes that
// is OpenCV's integer-scale fast path, the 2x2 box mean (a + b + c + d + 2) >> 2.  For an odd size the destination is
// floor(size / 2) and the scale becomes size / floor(size / 2) (not 2): OpenCV's general area resampling, restated here
// as the exact area-weighted mean in float with round-half-to-even (recalled; unpinned -- the ETH3D image sizes and
// pyramid depths never reach an odd level).
#pragma once

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "io_jpeg.h"
#include "io_jpeg_write.h"

namespace e3d_host {

struct GrayImage {
  int width = 0, height = 0;
  std::vector<uint8_t> data;
  bool empty() const { return data.empty(); }
};

// cv::imread(path) (IMREAD_COLOR): 8-bit, three channels -- stored R, G, B here (OpenCV's B, G, R order only matters when writing)
struct ColorImage {
  int width = 0, height = 0;
  std::vector<uint8_t> rgb;
  bool empty() const { return rgb.empty(); }
};

namespace img_detail {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint8_t rgb_to_gray(int r, int g, int b) {
  if (r == g && g == b) return (uint8_t)r;
  return (uint8_t)((9797 * r + 19234 * g + 3737 * b + 16384) >> 15);
}
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline bool load_png(const std::vector<uint8_t>& f, GrayImage* out, std::string* err, std::vector<uint8_t>* rgb = nullptr) {
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) { *err = "not a PNG file"; return false; }
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, palette;
  while (pos + 12 <= f.size()) {
    const uint32_t len = be32(&f[pos]);
    const char* type = reinterpret_cast<const char*>(&f[pos + 4]);
    if (pos + 12 + len > f.size()) { *err = "truncated PNG"; return false; }
    const uint8_t* d = &f[pos + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) { *err = "PNG with a malformed IHDR chunk"; return false; }
      w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
    } else if (!memcmp(type, "PLTE", 4)) {
      palette.assign(d, d + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), d, d + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  if (w <= 0 || h <= 0) { *err = "PNG without IHDR"; return false; }
  if (w > (1 << 20) || h > (1 << 20)) { *err = "PNG dimensions out of range"; return false; }
  if (interlace) { *err = "interlaced PNG is not supported"; return false; }
  int channels = 0;
  switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; }
  if (!channels || (depth != 8 && depth != 16 && !(ctype == 3 && depth <= 8) && !(ctype == 0 && depth < 8))) { *err = "unsupported PNG format"; return false; }
  const size_t bpp_bits = (size_t)channels * depth;
  const size_t stride = (w * bpp_bits + 7) / 8;
  const size_t bpp = std::max<size_t>(1, bpp_bits / 8);
  // deflate expands at most ~1032 : 1: a raw size the compressed data cannot produce is refused before it is allocated
  if ((stride + 1) * (size_t)h / 1032 > idat.size() + 64) { *err = "PNG image data too short for its dimensions"; return false; }
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { *err = "PNG inflate failed"; return false; }
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  out->width = w; out->height = h; out->data.assign((size_t)w * h, 0);
  if (rgb) rgb->assign((size_t)w * h * 3, 0);
  for (int y = 0; y < h; ++y) {
    const uint8_t* line = &raw[(stride + 1) * (size_t)y];
    const int filter = line[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = line[1 + i];
      switch (filter) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
      cur[i] = (uint8_t)v;
    }
    uint8_t* o = &out->data[(size_t)y * w];
    const size_t step = depth == 16 ? 2 : 1;       // 16-bit samples: the high byte (what IMREAD_GRAYSCALE's 8-bit conversion keeps)
    for (int x = 0; x < w; ++x) {
      if (ctype == 3 || (ctype == 0 && depth < 8)) {
        const int per = 8 / depth, idx = (cur[x / per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1);
        uint8_t* c = rgb ? &(*rgb)[((size_t)y * w + x) * 3] : nullptr;
        if (ctype == 0) { o[x] = (uint8_t)(idx * 255 / ((1 << depth) - 1)); if (c) c[0] = c[1] = c[2] = o[x]; continue; }
        if ((size_t)idx * 3 + 2 >= palette.size()) { *err = "PNG palette index out of range"; return false; }
        o[x] = rgb_to_gray(palette[idx * 3], palette[idx * 3 + 1], palette[idx * 3 + 2]);
        if (c) { c[0] = palette[idx * 3]; c[1] = palette[idx * 3 + 1]; c[2] = palette[idx * 3 + 2]; }
      } else {
        const uint8_t* p = &cur[(size_t)x * channels * step];
        o[x] = (channels >= 3) ? rgb_to_gray(p[0], p[step], p[2 * step]) : p[0];
        if (rgb) {
          uint8_t* c = &(*rgb)[((size_t)y * w + x) * 3];
          if (channels >= 3) { c[0] = p[0]; c[1] = p[step]; c[2] = p[2 * step]; } else c[0] = c[1] = c[2] = p[0];
        }
      }
    }
    prev.swap(cur);
  }
  return true;
}

inline bool load_pnm(const std::vector<uint8_t>& f, GrayImage* out, std::string* err, std::vector<uint8_t>* rgb = nullptr) {
  size_t pos = 2;
  auto token = [&]() {
    while (pos < f.size()) {
      if (f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; }
      else if (isspace(f[pos])) ++pos;
      else break;
    }
    long v = 0;
    while (pos < f.size() && isdigit(f[pos])) v = v * 10 + (f[pos++] - '0');
    return v;
  };
  const bool color = f[1] == '6';
  const long w = token(), h = token(), maxv = token();
  ++pos;
  if (w <= 0 || h <= 0 || maxv != 255 || pos + (size_t)w * h * (color ? 3 : 1) > f.size()) { *err = "unsupported PNM"; return false; }
  out->width = (int)w; out->height = (int)h; out->data.resize((size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; ++i)
    out->data[i] = color ? rgb_to_gray(f[pos + 3 * i], f[pos + 3 * i + 1], f[pos + 3 * i + 2]) : f[pos + i];
  if (rgb) {
    rgb->resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i)
      for (int c = 0; c < 3; ++c) (*rgb)[3 * i + c] = color ? f[pos + 3 * i + c] : f[pos + i];
  }
  return true;
}

}  // namespace img_detail

// cv::imread(path, IMREAD_GRAYSCALE): empty image on failure (the caller reports it)
inline GrayImage imread_gray(const std::string& path, std::string* error = nullptr) {
  GrayImage img;
  std::string err;
  std::ifstream s(path, std::ios::binary);
  if (!s) { if (error) *error = "cannot open " + path; return img; }
  std::vector<uint8_t> f((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
  bool ok = false;
  if (f.size() >= 8 && f[0] == 137 && f[1] == 'P') ok = img_detail::load_png(f, &img, &err);
  else if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) ok = img_detail::load_pnm(f, &img, &err);
  else if (f.size() >= 2 && f[0] == 0xff && f[1] == 0xd8) ok = load_jpeg_gray(f, &img.width, &img.height, &img.data, &err);
  else err = "unknown image format";
  if (!ok) { img = GrayImage(); if (error) *error = path + ": " + err; }
  return img;
}

// cv::imread(path): colour, 8 bits per channel; grey files give three equal channels, 16-bit PNG samples their high byte, alpha is dropped
inline ColorImage imread_color(const std::string& path, std::string* error = nullptr) {
  ColorImage img;
  GrayImage g;
  std::string err;
  std::ifstream s(path, std::ios::binary);
  if (!s) { if (error) *error = "cannot open " + path; return img; }
  std::vector<uint8_t> f((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
  bool ok = false;
  if (f.size() >= 8 && f[0] == 137 && f[1] == 'P') ok = img_detail::load_png(f, &g, &err, &img.rgb);
  else if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) ok = img_detail::load_pnm(f, &g, &err, &img.rgb);
  else if (f.size() >= 2 && f[0] == 0xff && f[1] == 0xd8) ok = load_jpeg(f, &g.width, &g.height, nullptr, &img.rgb, &err);
  else err = "unknown image format";
  if (!ok) { img = ColorImage(); if (error) *error = path + ": " + err; return img; }
  img.width = g.width; img.height = g.height;
  return img;
}

// cv::imwrite(path, image) for the formats the tools write: by extension ".jpg" / ".jpeg" (libjpeg defaults at quality 95,
// io_jpeg_write.h), ".png" (8-bit RGB, filter 0, zlib), ".ppm" (binary P6)
inline bool imwrite_color(const std::string& path, const ColorImage& img, std::string* error = nullptr) {
  std::string ext;
  const size_t dot = path.find_last_of('.');
  if (dot != std::string::npos) for (size_t i = dot + 1; i < path.size(); ++i) ext += (char)tolower(path[i]);
  std::vector<uint8_t> bytes;
  if (img.empty() || img.rgb.size() != (size_t)img.width * img.height * 3) { if (error) *error = "imwrite: empty image"; return false; }
  if (ext == "jpg" || ext == "jpeg" || ext == "jpe") {
    if (img.width > 65535 || img.height > 65535) { if (error) *error = "imwrite: image too large for JPEG"; return false; }
    bytes = encode_jpeg_rgb(img.rgb.data(), img.width, img.height, 95);
  } else if (ext == "png") {
    const size_t stride = (size_t)img.width * 3;
    std::vector<uint8_t> raw((stride + 1) * (size_t)img.height);
    for (int y = 0; y < img.height; ++y) { raw[(stride + 1) * y] = 0; memcpy(&raw[(stride + 1) * y + 1], &img.rgb[stride * y], stride); }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 1) != Z_OK) { if (error) *error = "imwrite: deflate failed"; return false; }
    auto be = [&](uint32_t v) { for (int s2 = 24; s2 >= 0; s2 -= 8) bytes.push_back((uint8_t)(v >> s2)); };
    auto chunk = [&](const char* type, const uint8_t* d, size_t n) {
      be((uint32_t)n);
      const size_t start = bytes.size();
      bytes.insert(bytes.end(), type, type + 4);
      bytes.insert(bytes.end(), d, d + n);
      be((uint32_t)crc32(0L, &bytes[start], (uInt)(n + 4)));
    };
    const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    bytes.insert(bytes.end(), sig, sig + 8);
    uint8_t ihdr[13];
    for (int s2 = 0; s2 < 4; ++s2) { ihdr[s2] = (uint8_t)(img.width >> (24 - 8 * s2)); ihdr[4 + s2] = (uint8_t)(img.height >> (24 - 8 * s2)); }
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), clen);
    chunk("IEND", nullptr, 0);
  } else if (ext == "ppm" || ext == "pnm") {
    char head[64];
    const int n = snprintf(head, sizeof head, "P6\n%d %d\n255\n", img.width, img.height);
    bytes.assign(head, head + n);
    bytes.insert(bytes.end(), img.rgb.begin(), img.rgb.end());
  } else {
    if (error) *error = "imwrite: unsupported file extension ." + ext;
    return false;
  }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { if (error) *error = "cannot write " + path; return false; }
  const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
  fclose(f);
  if (!ok && error) *error = "cannot write " + path;
  return ok;
}

// one INTER_AREA half-size step (see the header comment)
inline GrayImage half_size_area(const GrayImage& a) {
  GrayImage o;
  o.width = (int)(0.5 * a.width); o.height = (int)(0.5 * a.height);
  o.data.resize((size_t)o.width * o.height);
  if (o.width == 0 || o.height == 0) return o;
  if (a.width % 2 == 0 && a.height % 2 == 0) {
    for (int y = 0; y < o.height; ++y) {
      const uint8_t* r0 = &a.data[(size_t)(2 * y) * a.width];
      const uint8_t* r1 = r0 + a.width;
      uint8_t* d = &o.data[(size_t)y * o.width];
      for (int x = 0; x < o.width; ++x) d[x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
    }
    return o;
  }
  const double sx = (double)a.width / o.width, sy = (double)a.height / o.height;
  for (int y = 0; y < o.height; ++y) {
    const double y0 = y * sy, y1 = std::min<double>((y + 1) * sy, a.height);
    for (int x = 0; x < o.width; ++x) {
      const double x0 = x * sx, x1 = std::min<double>((x + 1) * sx, a.width);
      float sum = 0.f;
      for (int yy = (int)y0; yy < (int)std::ceil(y1); ++yy) {
        const float wy = (float)((std::min<double>(yy + 1, y1) - std::max<double>(yy, y0)) / (y1 - y0));
        for (int xx = (int)x0; xx < (int)std::ceil(x1); ++xx) {
          const float wx = (float)((std::min<double>(xx + 1, x1) - std::max<double>(xx, x0)) / (x1 - x0));
          sum += a.data[(size_t)yy * a.width + xx] * (wx * wy);
        }
      }
      const long r = std::lrintf(sum);      // cvRound: half to even
      o.data[(size_t)y * o.width + x] = (uint8_t)std::min<long>(255, std::max<long>(0, r));
    }
  }
  return o;
}

// Image::BuildImagePyramid (image.cc:106-131)
inline std::vector<GrayImage> build_image_pyramid(const GrayImage& level0, int image_scale_count) {
  std::vector<GrayImage> p(image_scale_count);
  p[0] = level0;
  for (int i = 1; i < image_scale_count; ++i) p[i] = half_size_area(p[i - 1]);
  return p;
}

// Image::BuildMaskPyramid (image.cc:133-154): a coarse pixel carries every flag of its four fine pixels
inline std::vector<GrayImage> build_mask_pyramid(const GrayImage& level0, int image_scale_count) {
  std::vector<GrayImage> p(image_scale_count);
  p[0] = level0;
  for (int i = 1; i < image_scale_count; ++i) {
    const GrayImage& a = p[i - 1];
    GrayImage& o = p[i];
    o.height = (int)(0.5 * a.height); o.width = (int)(0.5 * a.width);
    o.data.resize((size_t)o.width * o.height);
    for (int y = 0; y < o.height; ++y)
      for (int x = 0; x < o.width; ++x) {
        const uint8_t* r0 = &a.data[(size_t)(2 * y) * a.width + 2 * x];
        const uint8_t* r1 = r0 + a.width;
        o.data[(size_t)y * o.width + x] = r0[0] | r0[1] | r1[0] | r1[1];
      }
  }
  return p;
}

}  // namespace e3d_host
