// io_jpeg_write.h -- baseline JPEG encoder for cv::imwrite(path, bgr_image) on ".jpg" (GroundTruthCreator's scan renderings,
// src/exe/ground_truth_creator.cc:194-199).
//
// OpenCV hands the image to libjpeg with its defaults: quality 95 (IMWRITE_JPEG_QUALITY), YCbCr with 2x2 chroma subsampling, the
// standard quantisation tables scaled by jpeg_quality_scaling, the standard Huffman tables (no optimisation), baseline sequential,
// one interleaved scan, a JFIF APP0 header.  This header restates that path with libjpeg's integer arithmetic -- RGB -> YCbCr with the
// 16-bit tables of jccolor.c, edge replication up to whole MCUs, the 2x2 box downsampling with its alternating 1 / 2 bias
// (jcsample.c), the "islow" forward DCT (jfdctint.c: the same constants as the inverse in io_jpeg.h) and quantisation with rounding
// half away from zero (jcdctmgr.c) -- so that the coefficients, and therefore every decoder's output, equal libjpeg's (pinned against
// Pillow / libjpeg-turbo in tests/test_cli_host.py: same decoded pixels, same entropy-coded bytes).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace e3d_host {
namespace jpeg_write_detail {

// T.81 Annex K.1 / K.2, natural order
constexpr int kStdLuminanceQuant[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                        69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
                                        81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92,  95,  98,  112, 100, 103, 99};
constexpr int kStdChrominanceQuant[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                          99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// T.81 Annex K.3: code-length counts (1 .. 16 bits) and symbols of the four typical tables
constexpr uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
constexpr uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
constexpr uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
constexpr uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
constexpr uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
constexpr uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
constexpr uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
constexpr int kZigzagOrder[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct EncTable { uint16_t code[256]; uint8_t size[256]; };
inline EncTable make_enc_table(const uint8_t* bits, const uint8_t* vals) {      // T.81 Annex C
  EncTable t;
  memset(&t, 0, sizeof t);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l - 1]; ++i) { t.code[vals[k]] = (uint16_t)code; t.size[vals[k]] = (uint8_t)l; ++code; ++k; }
    code <<= 1;
  }
  return t;
}

struct BitWriter {
  std::vector<uint8_t>* out;
  uint32_t acc = 0; int bits = 0;
  void put(unsigned code, int size) {
    if (!size) return;
    acc = (acc << size) | (code & ((1u << size) - 1u));
    bits += size;
    while (bits >= 8) {
      const uint8_t b = (uint8_t)(acc >> (bits - 8));
      out->push_back(b);
      if (b == 0xFF) out->push_back(0x00);
      bits -= 8;
    }
  }
  void flush() { if (bits) put(0x7F, 8 - bits); }                              // pad the last byte with one-bits
};

// jfdctint.c (CONST_BITS 13, PASS1_BITS 2): data = samples - 128 in, 8 x the DCT coefficients out
inline void fdct_islow(int* d) {
  constexpr int C0298 = 2446, C0390 = 3196, C0541 = 4433, C0765 = 6270, C0899 = 7373, C1175 = 9633, C1501 = 12299, C1847 = 15137, C1961 = 16069,
                C2053 = 16819, C2562 = 20995, C3072 = 25172;
  auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
  for (int pass = 0; pass < 2; ++pass) {
    const int step = pass == 0 ? 1 : 8, stride = pass == 0 ? 8 : 1;
    const int sh_even = pass == 0 ? -2 : 2, sh = pass == 0 ? 11 : 15;           // pass 1 scales up by 2^PASS1_BITS, pass 2 removes it again
    for (int i = 0; i < 8; ++i) {
      int* p = d + i * stride;
      const long t0 = p[0] + p[7 * step], t7 = p[0] - p[7 * step], t1 = p[step] + p[6 * step], t6 = p[step] - p[6 * step];
      const long t2 = p[2 * step] + p[5 * step], t5 = p[2 * step] - p[5 * step], t3 = p[3 * step] + p[4 * step], t4 = p[3 * step] - p[4 * step];
      const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
      if (sh_even < 0) { p[0] = (int)((t10 + t11) << 2); p[4 * step] = (int)((t10 - t11) << 2); }
      else { p[0] = descale(t10 + t11, 2); p[4 * step] = descale(t10 - t11, 2); }
      long z1 = (t12 + t13) * C0541;
      p[2 * step] = descale(z1 + t13 * C0765, sh);
      p[6 * step] = descale(z1 + t12 * (-(long)C1847), sh);
      z1 = t4 + t7;
      long z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
      const long z5 = (z3 + z4) * C1175;
      const long a4 = t4 * C0298, a5 = t5 * C2053, a6 = t6 * C3072, a7 = t7 * C1501;
      z1 *= -(long)C0899; z2 *= -(long)C2562; z3 *= -(long)C1961; z4 *= -(long)C0390;
      z3 += z5; z4 += z5;
      p[7 * step] = descale(a4 + z1 + z3, sh);
      p[5 * step] = descale(a5 + z2 + z4, sh);
      p[3 * step] = descale(a6 + z2 + z3, sh);
      p[step] = descale(a7 + z1 + z4, sh);
    }
  }
}

inline int magnitude_bits(int v) { int n = 0; v = v < 0 ? -v : v; while (v) { ++n; v >>= 1; } return n; }

inline void encode_block(BitWriter& bw, const int* q /* zigzag order */, int* pred, const EncTable& dc, const EncTable& ac) {
  const int diff = q[0] - *pred;
  *pred = q[0];
  int n = magnitude_bits(diff);
  bw.put(dc.code[n], dc.size[n]);
  if (n) bw.put((unsigned)(diff < 0 ? diff - 1 : diff), n);
  int run = 0;
  for (int k = 1; k < 64; ++k) {
    const int v = q[k];
    if (v == 0) { ++run; continue; }
    while (run > 15) { bw.put(ac.code[0xF0], ac.size[0xF0]); run -= 16; }
    n = magnitude_bits(v);
    const int sym = (run << 4) | n;
    bw.put(ac.code[sym], ac.size[sym]);
    bw.put((unsigned)(v < 0 ? v - 1 : v), n);
    run = 0;
  }
  if (run) bw.put(ac.code[0], ac.size[0]);
}

}  // namespace jpeg_write_detail

// rgb: width x height x 3; returns the bytes of a baseline JFIF file (libjpeg defaults at the given quality, 4:2:0)
inline std::vector<uint8_t> encode_jpeg_rgb(const uint8_t* rgb, int W, int H, int quality = 95) {
  using namespace jpeg_write_detail;
  std::vector<uint8_t> out;
  auto put16 = [&](int v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)(v & 255)); };
  auto marker = [&](int m) { out.push_back(0xFF); out.push_back((uint8_t)m); };
  // jpeg_quality_scaling + jpeg_add_quant_table (force_baseline)
  quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  int qt[2][64];
  for (int t = 0; t < 2; ++t)
    for (int i = 0; i < 64; ++i) {
      long v = ((long)(t ? kStdChrominanceQuant[i] : kStdLuminanceQuant[i]) * scale + 50L) / 100L;
      qt[t][i] = (int)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
  marker(0xD8);
  marker(0xE0); put16(16); out.insert(out.end(), {'J', 'F', 'I', 'F', 0, 1, 1, 0}); put16(1); put16(1); out.push_back(0); out.push_back(0);
  for (int t = 0; t < 2; ++t) { marker(0xDB); put16(67); out.push_back((uint8_t)t); for (int k = 0; k < 64; ++k) out.push_back((uint8_t)qt[t][kZigzagOrder[k]]); }
  marker(0xC0); put16(17); out.push_back(8); put16(H); put16(W); out.push_back(3);
  out.insert(out.end(), {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1});
  auto dht = [&](int tc_th, const uint8_t* bits, const uint8_t* vals, int nvals) {
    marker(0xC4); put16(19 + nvals); out.push_back((uint8_t)tc_th);
    out.insert(out.end(), bits, bits + 16); out.insert(out.end(), vals, vals + nvals);
  };
  dht(0x00, kDcLumBits, kDcVals, 12); dht(0x10, kAcLumBits, kAcLumVals, 162);
  dht(0x01, kDcChrBits, kDcVals, 12); dht(0x11, kAcChrBits, kAcChrVals, 162);
  marker(0xDA); put16(12); out.push_back(3); out.insert(out.end(), {1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
  const EncTable dcl = make_enc_table(kDcLumBits, kDcVals), acl = make_enc_table(kAcLumBits, kAcLumVals);
  const EncTable dcc = make_enc_table(kDcChrBits, kDcVals), acc = make_enc_table(kAcChrBits, kAcChrVals);
  // colour conversion (jccolor.c rgb_ycc_convert) into planes padded to whole 16 x 16 MCUs by repeating the last column / row
  const int mx = (W + 15) / 16, my = (H + 15) / 16, PW = mx * 16, PH = my * 16;
  std::vector<uint8_t> Y((size_t)PW * PH), Cb((size_t)PW * PH), Cr((size_t)PW * PH);
  for (int y = 0; y < PH; ++y) {
    const int sy = y < H ? y : H - 1;
    for (int x = 0; x < PW; ++x) {
      const int sx = x < W ? x : W - 1;
      const uint8_t* p = rgb + ((size_t)sy * W + sx) * 3;
      const long r = p[0], g = p[1], b = p[2];
      Y[(size_t)y * PW + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
      Cb[(size_t)y * PW + x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128L << 16) + 32767) >> 16);
      Cr[(size_t)y * PW + x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128L << 16) + 32767) >> 16);
    }
  }
  // h2v2_downsample: bias 1, 2, 1, 2 ... along a row
  const int CW = PW / 2, CHh = PH / 2;
  std::vector<uint8_t> cb((size_t)CW * CHh), cr((size_t)CW * CHh);
  // (the bottom is padded AFTER downsampling, by repeating the last real chroma row -- jcprepct.c expand_bottom_edge on the output --
  // which differs from downsampling repeated image rows when the height is even; the right edge is padded before, jcsample.c)
  const int chroma_rows = (H + 1) / 2;
  for (int y = 0; y < CHh; ++y)
    for (int x = 0; x < CW; ++x) {
      if (y >= chroma_rows) { cb[(size_t)y * CW + x] = cb[(size_t)(chroma_rows - 1) * CW + x]; cr[(size_t)y * CW + x] = cr[(size_t)(chroma_rows - 1) * CW + x]; continue; }
      const int bias = (x & 1) ? 2 : 1;
      const size_t a = (size_t)(2 * y) * PW + 2 * x, b2 = a + PW;
      cb[(size_t)y * CW + x] = (uint8_t)((Cb[a] + Cb[a + 1] + Cb[b2] + Cb[b2 + 1] + bias) >> 2);
      cr[(size_t)y * CW + x] = (uint8_t)((Cr[a] + Cr[a + 1] + Cr[b2] + Cr[b2 + 1] + bias) >> 2);
    }
  BitWriter bw{&out};
  int pred[3] = {0, 0, 0};
  int blk[64];
  // quantised coefficients of one block, zigzag order
  auto quantised = [&](const uint8_t* src, int stride, int table, int* q) {
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x) blk[8 * y + x] = (int)src[(size_t)y * stride + x] - 128;
    fdct_islow(blk);
    for (int k = 0; k < 64; ++k) {                       // jcdctmgr.c: divisor = 8 x the table entry, round half away from zero
      const int zz = kZigzagOrder[k];
      const int qv = qt[table][zz] << 3;
      int t = blk[zz];
      if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
      else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
      q[k] = t;
    }
  };
  // Luminance blocks of an MCU that lie wholly outside the image are libjpeg's dummy blocks (jccoefct.c compress_data): no AC, the DC
  // of the block before them in the MCU -- they only keep the DC prediction chain going, but the bytes should be libjpeg's.
  const int ybw = (W + 7) / 8, ybh = (H + 7) / 8;
  int qy[4][64], qc[64];
  for (int j = 0; j < my; ++j)
    for (int i = 0; i < mx; ++i) {
      for (int by = 0; by < 2; ++by)
        for (int bx = 0; bx < 2; ++bx) {
          int* q = qy[2 * by + bx];
          if (2 * j + by < ybh && 2 * i + bx < ybw) quantised(&Y[(size_t)(16 * j + 8 * by) * PW + 16 * i + 8 * bx], PW, 0, q);
          else { memset(q, 0, sizeof(int) * 64); q[0] = (2 * j + by < ybh) ? qy[2 * by + bx - 1][0] : qy[1][0]; }
        }
      for (int b = 0; b < 4; ++b) encode_block(bw, qy[b], &pred[0], dcl, acl);
      quantised(&cb[(size_t)(8 * j) * CW + 8 * i], CW, 1, qc); encode_block(bw, qc, &pred[1], dcc, acc);
      quantised(&cr[(size_t)(8 * j) * CW + 8 * i], CW, 1, qc); encode_block(bw, qc, &pred[2], dcc, acc);
    }
  bw.flush();
  marker(0xD9);
  return out;
}

}  // namespace e3d_host
