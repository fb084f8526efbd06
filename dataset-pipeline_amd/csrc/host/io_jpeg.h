// io_jpeg.h -- baseline JPEG -> 8-bit grey, for cv::imread(path, cv::IMREAD_GRAYSCALE) on the JPEG images of a dataset
// (src/opt/image.cc:47; the ETH3D DSLR images are JPEGs).
//
// OpenCV's JPEG reader asks libjpeg for JCS_GRAYSCALE output, which for a YCbCr (or grey) file is the decoded luminance
// plane itself -- no colour conversion, chroma is never reconstructed.  This header restates that path: Huffman decoding
// (ITU T.81 F.2), dequantisation, libjpeg's default "islow" inverse DCT (jidctint.c: Loeffler-Ligtenberg-Moschytz, 13-bit
// constants, 2 extra bits after the column pass) and the +128 / clamp of its range-limit table.  The integer arithmetic is
// the specification here, so the result is bit-identical to libjpeg / libjpeg-turbo (pinned against Pillow's libjpeg-turbo
// decoder: tests/golden/jpeg_*.jpg + jpeg_golden.npz, tests/test_cli_host.py).
//
// Supported: SOF0 / SOF1 (baseline / extended sequential, Huffman, 8-bit), 1 or 3 components (YCbCr), any sampling
// factors, restart intervals, sizes that are not multiples of the MCU; SOF2 (progressive, Huffman): DC / AC first and refinement scans of the
// luminance component (T.81 G.1.2; chroma AC scans are skipped, interleaved DC scans are parsed for all components).  Rejected with a message:
// arithmetic coding, 12-bit, CMYK / YCCK, RGB-coded files (Adobe transform 0 or component ids 'R' 'G' 'B').
// The EXIF orientation tag (APP1, TIFF tag 0x0112) is applied like cv::imread does without IMREAD_IGNORE_ORIENTATION
// (OpenCV's ExifTransform: 2 mirror, 3 rotate 180, 4 flip, 5 transpose, 6 rotate 90 cw, 7 transverse, 8 rotate 90 ccw).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace e3d_host {
namespace jpeg_detail {

struct Huff {
  // canonical code tables (T.81 Annex C / F.2.2.3): per code length the smallest code, the largest code and the index of
  // the first symbol of that length
  int mincode[17] = {0}, maxcode[18] = {0}, valptr[17] = {0};
  uint8_t vals[256] = {0};
  bool defined = false;
};

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; };

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int bits = 0;
  bool hit_marker = false;
  void fill() {
    while (bits <= 24) {
      int c = 0;
      if (!hit_marker && p < end) {
        c = *p++;
        if (c == 0xFF) {
          if (p < end && *p == 0x00) ++p;            // stuffed byte
          else { hit_marker = true; --p; c = 0; }    // a marker ends the entropy-coded segment: feed zeros
        }
      }
      acc |= (uint32_t)c << (24 - bits);
      bits += 8;
    }
  }
  int get(int n) {       // n <= 16
    if (n == 0) return 0;
    if (bits < n) fill();
    const int v = (int)(acc >> (32 - n));
    acc <<= n; bits -= n;
    return v;
  }
  void reset() { acc = 0; bits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h, bool* ok) {
  int code = br.get(1);
  for (int l = 1; l <= 16; ++l) {
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    code = (code << 1) | br.get(1);
  }
  *ok = false;
  return 0;
}
inline int extend(int v, int t) { return (t == 0) ? 0 : ((v < (1 << (t - 1))) ? v - (1 << t) + 1 : v); }

constexpr int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// jpeg_idct_islow (jidctint.c): coefficients already dequantised, natural order
inline void idct_islow(const int* in, uint8_t* out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                 F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  long ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* ip = in + c;
    long* wp = ws + c;
    long z2 = ip[16], z3 = ip[48];
    long z1 = (z2 + z3) * F0_541;
    long tmp2 = z1 + z3 * (-F1_847);
    long tmp3 = z1 + z2 * F0_765;
    z2 = ip[0]; z3 = ip[32];
    long tmp0 = (z2 + z3) * (1L << CB);
    long tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ip[56]; tmp1 = ip[40]; tmp2 = ip[24]; tmp3 = ip[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
    wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
    wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
    wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const long* wp = ws + 8 * r;
    long z2 = wp[2], z3 = wp[6];
    long z1 = (z2 + z3) * F0_541;
    long tmp2 = z1 + z3 * (-F1_847);
    long tmp3 = z1 + z2 * F0_765;
    long tmp0 = (wp[0] + wp[4]) * (1L << CB);
    long tmp1 = (wp[0] - wp[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const long o[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
    uint8_t* op = out + (size_t)r * stride;
    for (int k = 0; k < 8; ++k) {
      long v = descale(o[k], CB + P1 + 3) + 128;         // range_limit: centre on 128, clamp to 0..255
      op[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// position of the next marker that ends an entropy-coded segment (not a stuffed 0xFF00, not RSTn), or size
inline size_t next_segment_marker(const std::vector<uint8_t>& f, size_t pos) {
  while (pos + 1 < f.size()) {
    if (f[pos] == 0xFF) {
      const int b = f[pos + 1];
      if (b != 0x00 && b != 0xFF && !(b >= 0xD0 && b <= 0xD7)) return pos;
    }
    ++pos;
  }
  return f.size();
}

// Progressive scans of ONE block (T.81 G.1.2 / figures G.3 - G.7); coefficients in natural order, not dequantised.
struct ProgressiveScan { int ss = 0, se = 0, ah = 0, al = 0; int eobrun = 0; };

inline bool prog_dc(BitReader& br, const Huff& dc, const ProgressiveScan& sc, int* pred, int16_t* coef) {
  if (sc.ah == 0) {
    bool ok = true;
    const int t = decode_symbol(br, dc, &ok);
    if (!ok || t > 15) return false;
    *pred += extend(br.get(t), t);
    coef[0] = (int16_t)(*pred * (1 << sc.al));
  } else if (br.get(1)) {
    coef[0] = (int16_t)(coef[0] | (1 << sc.al));
  }
  return true;
}

inline bool prog_ac_first(BitReader& br, const Huff& ac, ProgressiveScan& sc, int16_t* coef) {
  if (sc.eobrun > 0) { --sc.eobrun; return true; }
  for (int k = sc.ss; k <= sc.se;) {
    bool ok = true;
    const int rs = decode_symbol(br, ac, &ok);
    if (!ok) return false;
    const int r = rs >> 4, s = rs & 15;
    if (s == 0) {
      if (r < 15) {                                   // EOBn: this block and eobrun more end here
        sc.eobrun = (1 << r) - 1;
        if (r) sc.eobrun += br.get(r);
        break;
      }
      k += 16;                                         // ZRL
    } else {
      k += r;
      if (k > 63) return false;
      coef[kZigzag[k]] = (int16_t)(extend(br.get(s), s) * (1 << sc.al));
      ++k;
    }
  }
  return true;
}

// one correction bit for a coefficient that is already non-zero: move it away from zero by 2^al unless that bit is set
inline void prog_refine_bit(BitReader& br, int16_t* c, int p1) {
  if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c - p1);
}

inline bool prog_ac_refine(BitReader& br, const Huff& ac, ProgressiveScan& sc, int16_t* coef) {
  const int p1 = 1 << sc.al;
  int k = sc.ss;
  if (sc.eobrun == 0) {
    while (k <= sc.se) {
      bool ok = true;
      const int rs = decode_symbol(br, ac, &ok);
      if (!ok) return false;
      int r = rs >> 4;
      const int s = rs & 15;
      int value = 0;
      if (s == 0) {
        if (r < 15) {                                 // EOBn: the rest of THIS block is refined below, eobrun counts it
          sc.eobrun = 1 << r;
          if (r) sc.eobrun += br.get(r);
          break;
        }
      } else {
        if (s != 1) return false;
        value = br.get(1) ? p1 : -p1;                 // a coefficient that becomes non-zero in this scan
      }
      // pass r zero-history coefficients (16 for ZRL); every non-zero one on the way takes a correction bit
      while (k <= sc.se) {
        int16_t* c = &coef[kZigzag[k]];
        if (*c != 0) prog_refine_bit(br, c, p1);
        else if (--r < 0) break;
        ++k;
      }
      if (value) {
        if (k > sc.se) return false;
        coef[kZigzag[k]] = (int16_t)value;
      }
      ++k;
    }
  }
  if (sc.eobrun > 0) {
    for (; k <= sc.se; ++k) {
      int16_t* c = &coef[kZigzag[k]];
      if (*c != 0) prog_refine_bit(br, c, p1);
    }
    --sc.eobrun;
  }
  return true;
}

// byte-align and consume the RSTn marker that must follow (restart interval reached)
inline bool take_restart(BitReader& br) {
  br.reset();
  const uint8_t* q = br.p;
  while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
  if (q + 1 >= br.end) return false;
  br.p = q + 2;
  return true;
}

// jdsample.c: chroma plane (real size dw x dh, stride pw) -> w x h.  2:1 ratios use libjpeg's triangle filters when the plane is more
// than two samples wide (do_fancy_upsampling, the default); context rows above / below the image repeat the first / last real row
// (jdmainct.c); other integral ratios replicate.
inline bool upsample_plane(const std::vector<uint8_t>& in, int pw, int dw, int dh, int hr, int vr, int w, int h, std::vector<uint8_t>* out) {
  out->assign((size_t)w * h, 0);
  auto at = [&](int x, int y) { return (int)in[(size_t)(y < 0 ? 0 : (y >= dh ? dh - 1 : y)) * pw + x]; };
  const bool fancy = dw > 2;
  if (hr == 1 && vr == 1) {
    for (int y = 0; y < h; ++y) memcpy(&(*out)[(size_t)y * w], &in[(size_t)y * pw], (size_t)w);
  } else if (hr == 2 && vr == 1 && fancy) {
    std::vector<uint8_t> row((size_t)2 * dw);
    for (int y = 0; y < h; ++y) {
      row[0] = (uint8_t)at(0, y); row[1] = (uint8_t)((at(0, y) * 3 + at(1, y) + 2) >> 2);
      for (int x = 1; x < dw - 1; ++x) {
        const int v = at(x, y) * 3;
        row[2 * x] = (uint8_t)((v + at(x - 1, y) + 1) >> 2); row[2 * x + 1] = (uint8_t)((v + at(x + 1, y) + 2) >> 2);
      }
      row[2 * dw - 2] = (uint8_t)((at(dw - 1, y) * 3 + at(dw - 2, y) + 1) >> 2); row[2 * dw - 1] = (uint8_t)at(dw - 1, y);
      memcpy(&(*out)[(size_t)y * w], row.data(), (size_t)w);
    }
  } else if (hr == 1 && vr == 2 && fancy) {                       // libjpeg-turbo's h1v2_fancy_upsample
    for (int y = 0; y < h; ++y) {
      const int iy = y >> 1, other = (y & 1) ? iy + 1 : iy - 1, bias = (y & 1) ? 2 : 1;
      for (int x = 0; x < w; ++x) (*out)[(size_t)y * w + x] = (uint8_t)((at(x, iy) * 3 + at(x, other) + bias) >> 2);
    }
  } else if (hr == 2 && vr == 2 && fancy) {
    std::vector<uint8_t> row((size_t)2 * dw);
    for (int y = 0; y < h; ++y) {
      const int iy = y >> 1, other = (y & 1) ? iy + 1 : iy - 1;
      auto colsum = [&](int x) { return at(x, iy) * 3 + at(x, other); };
      int last = colsum(0), cur = last, next = colsum(1);
      row[0] = (uint8_t)((cur * 4 + 8) >> 4); row[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
      for (int x = 1; x < dw - 1; ++x) {
        last = cur; cur = next; next = colsum(x + 1);
        row[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4); row[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
      }
      last = cur; cur = next;
      row[2 * dw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); row[2 * dw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
      memcpy(&(*out)[(size_t)y * w], row.data(), (size_t)w);
    }
  } else {
    if (hr < 1 || vr < 1 || hr > 4 || vr > 4) return false;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) (*out)[(size_t)y * w + x] = (uint8_t)at(x / hr < dw ? x / hr : dw - 1, y / vr);
  }
  return true;
}

// jdcolor.c ycc_rgb_convert: SCALEBITS = 16 tables, range-limited
inline void ycc_to_rgb(int y, int cb, int cr, uint8_t* out) {
  const int xb = cb - 128, xr = cr - 128;
  const int r = y + ((91881 * xr + 32768) >> 16);
  const int g = y + ((-22554 * xb + 32768 - 46802 * xr) >> 16);
  const int b = y + ((116130 * xb + 32768) >> 16);
  out[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); out[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g)); out[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
}

}  // namespace jpeg_detail

// grey = the luminance plane, width x height bytes; rgb (optional) = libjpeg's JCS_RGB output, 3 bytes per pixel: chroma planes
// reconstructed, "fancy" (triangle filter) upsampling for 2:1 horizontal, 2:1 vertical and 2x2 subsampling (jdsample.c), YCbCr -> RGB with
// the 16-bit fixed-point tables of jdcolor.c.  false + message on unsupported or damaged files.
inline bool load_jpeg(const std::vector<uint8_t>& f, int* width, int* height, std::vector<uint8_t>* gray, std::vector<uint8_t>* rgb,
                      std::string* err) {
  using namespace jpeg_detail;
  const bool want_rgb = rgb != nullptr;
  if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) { *err = "not a JPEG file"; return false; }
  int qt[4][64] = {{0}};
  bool qt_defined[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Component> comps;
  int W = 0, H = 0, restart_interval = 0;
  int adobe_transform = -1;
  int orientation = 1;
  size_t pos = 2;
  bool have_frame = false, progressive = false;
  // progressive: the luminance coefficients of every block of the MCU-padded grid, natural order, until the last scan is in
  std::vector<int16_t> pcoef[3];
  int pbw[3] = {0, 0, 0}, pbh[3] = {0, 0, 0};
  std::vector<int16_t>& ycoef = pcoef[0];
  int& ybw = pbw[0]; int& ybh = pbh[0];
  auto emit = [&](const std::vector<uint8_t>& plane, int PW) -> bool {
    // EXIF orientation: output pixel (x, y) of the oriented image <- source pixel (sx, sy)
    const bool swap = orientation >= 5;
    const int OW = swap ? H : W, OH = swap ? W : H;
    *width = OW; *height = OH;
    gray->resize((size_t)OW * OH);
    if (orientation == 1) {
      for (int y = 0; y < H; ++y) memcpy(&(*gray)[(size_t)y * W], &plane[(size_t)y * PW], (size_t)W);
      return true;
    }
    for (int y = 0; y < OH; ++y)
      for (int x = 0; x < OW; ++x) {
        int sx, sy;
        switch (orientation) {
          case 2: sx = W - 1 - x; sy = y; break;                   // mirror horizontally
          case 3: sx = W - 1 - x; sy = H - 1 - y; break;           // rotate 180
          case 4: sx = x; sy = H - 1 - y; break;                   // flip vertically
          case 5: sx = y; sy = x; break;                           // transpose
          case 6: sx = y; sy = H - 1 - x; break;                   // rotate 90 clockwise
          case 7: sx = W - 1 - y; sy = H - 1 - x; break;           // transverse
          default: sx = W - 1 - y; sy = x; break;                  // 8: rotate 90 counter-clockwise
        }
        (*gray)[(size_t)y * OW + x] = plane[(size_t)sy * PW + sx];
      }
    return true;
  };
  // colour output from the component planes (plane c: stride pws[c], real size ceil(W h_c / hmax) x ceil(H v_c / vmax))
  auto emit_rgb = [&](const std::vector<uint8_t>* planes, const int* pws) -> bool {
    std::vector<uint8_t> full[3];
    int hmax = 1, vmax = 1;
    for (const Component& c : comps) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
    const int nc = (int)comps.size();
    for (int c = 0; c < nc; ++c) {
      const int ch = nc == 1 ? 1 : comps[c].h, cv = nc == 1 ? 1 : comps[c].v, hm = nc == 1 ? 1 : hmax, vm = nc == 1 ? 1 : vmax;
      if (hm % ch || vm % cv) { *err = "JPEG with fractional chroma subsampling is not supported in colour"; return false; }
      if (!upsample_plane(planes[c], pws[c], (W * ch + hm - 1) / hm, (H * cv + vm - 1) / vm, hm / ch, vm / cv, W, H, &full[c])) {
        *err = "unsupported JPEG subsampling";
        return false;
      }
    }
    std::vector<uint8_t> img((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      if (nc == 1) { img[3 * i] = img[3 * i + 1] = img[3 * i + 2] = full[0][i]; continue; }
      ycc_to_rgb(full[0][i], full[1][i], full[2][i], &img[3 * i]);
    }
    const bool swap = orientation >= 5;
    const int OW = swap ? H : W, OH = swap ? W : H;
    *width = OW; *height = OH;
    rgb->resize((size_t)OW * OH * 3);
    for (int y = 0; y < OH; ++y)
      for (int x = 0; x < OW; ++x) {
        int sx, sy;
        switch (orientation) {
          case 1: sx = x; sy = y; break;
          case 2: sx = W - 1 - x; sy = y; break;
          case 3: sx = W - 1 - x; sy = H - 1 - y; break;
          case 4: sx = x; sy = H - 1 - y; break;
          case 5: sx = y; sy = x; break;
          case 6: sx = y; sy = H - 1 - x; break;
          case 7: sx = W - 1 - y; sy = H - 1 - x; break;
          default: sx = W - 1 - y; sy = x; break;
        }
        memcpy(&(*rgb)[((size_t)y * OW + x) * 3], &img[((size_t)sy * W + sx) * 3], 3);
      }
    return true;
  };
  while (pos + 4 <= f.size()) {
    if (f[pos] != 0xFF) { ++pos; continue; }
    const int marker = f[pos + 1];
    if (marker == 0xFF) { ++pos; continue; }                 // fill bytes
    pos += 2;
    if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) continue;
    if (marker == 0xD9) break;
    if (pos + 2 > f.size()) break;
    const int len = be16(&f[pos]);
    if (len < 2 || pos + (size_t)len > f.size()) { *err = "truncated JPEG segment"; return false; }
    const uint8_t* d = &f[pos + 2];
    const int n = len - 2;
    if (marker == 0xDB) {                                     // DQT
      int i = 0;
      while (i < n) {
        const int pq = d[i] >> 4, tq = d[i] & 15;
        ++i;
        if (tq > 3 || i + (pq ? 128 : 64) > n) { *err = "bad DQT"; return false; }
        for (int k = 0; k < 64; ++k) { qt[tq][kZigzag[k]] = pq ? be16(&d[i + 2 * k]) : d[i + k]; }
        qt_defined[tq] = true;
        i += pq ? 128 : 64;
      }
    } else if (marker == 0xC4) {                              // DHT
      int i = 0;
      while (i + 17 <= n) {
        const int tc = d[i] >> 4, th = d[i] & 15;
        if (tc > 1 || th > 3) { *err = "bad DHT"; return false; }
        Huff& h = tc ? ac[th] : dc[th];
        int counts[17] = {0}, total = 0;
        for (int l = 1; l <= 16; ++l) { counts[l] = d[i + l]; total += counts[l]; }
        i += 17;
        if (total > 256 || i + total > n) { *err = "bad DHT"; return false; }
        memcpy(h.vals, &d[i], (size_t)total);
        i += total;
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
          h.valptr[l] = k; h.mincode[l] = code;
          code += counts[l]; k += counts[l];
          h.maxcode[l] = counts[l] ? code - 1 : -1;
          code <<= 1;
        }
        h.defined = true;
      }
    } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {   // SOF0 / SOF1 / SOF2
      progressive = marker == 0xC2;
      if (n < 6 || d[0] != 8) { *err = "only 8-bit JPEG is supported"; return false; }
      H = be16(&d[1]); W = be16(&d[3]);
      if (W <= 0 || H <= 0 || (size_t)W * (size_t)H > ((size_t)1 << 28)) { *err = "JPEG dimensions out of range"; return false; }
      if (have_frame) { *err = "JPEG with more than one frame header"; return false; }
      const int nc = d[5];
      if ((nc != 1 && nc != 3) || n < 6 + 3 * nc) { *err = nc == 4 ? "CMYK / YCCK JPEG is not supported" : "unsupported JPEG component count"; return false; }
      comps.resize(nc);
      for (int c = 0; c < nc; ++c) {
        comps[c].id = d[6 + 3 * c]; comps[c].h = d[7 + 3 * c] >> 4; comps[c].v = d[7 + 3 * c] & 15; comps[c].tq = d[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) { *err = "bad SOF"; return false; }
      }
      have_frame = true;
    } else if (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
      *err = "unsupported JPEG coding process (lossless, hierarchical or arithmetic)";
      return false;
    } else if (marker == 0xDD) {
      if (n >= 2) restart_interval = be16(d);
    } else if (marker == 0xE1) {                              // APP1: "Exif\0\0" + TIFF header + IFD0
      if (n >= 14 && !memcmp(d, "Exif\0\0", 6)) {
        const uint8_t* t = d + 6;
        const int tn = n - 6;
        const bool le = t[0] == 'I' && t[1] == 'I', be = t[0] == 'M' && t[1] == 'M';
        auto u16 = [&](int o) { return le ? (int)(t[o] | (t[o + 1] << 8)) : (int)((t[o] << 8) | t[o + 1]); };
        auto u32 = [&](int o) { return le ? (long)(t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | ((long)t[o + 3] << 24))
                                          : (long)(((long)t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]); };
        if ((le || be) && u16(2) == 42) {
          const long ifd = u32(4);
          if (ifd >= 8 && ifd + 2 <= tn) {
            const int entries = u16((int)ifd);
            for (int e = 0; e < entries && ifd + 2 + 12 * (e + 1) <= tn; ++e) {
              const int o = (int)ifd + 2 + 12 * e;
              if (u16(o) == 0x0112 && u16(o + 2) == 3 && u32(o + 4) == 1) { const int v = u16(o + 8); if (v >= 1 && v <= 8) orientation = v; }
            }
          }
        }
      }
    } else if (marker == 0xEE) {                              // Adobe
      if (n >= 12 && !memcmp(d, "Adobe", 5)) adobe_transform = d[11];
    } else if (marker == 0xDA) {                              // SOS: baseline -> one scan with all components
      if (!have_frame || W <= 0 || H <= 0) { *err = "JPEG scan before frame header"; return false; }
      if (n < 6) { *err = "bad SOS"; return false; }          // ns + one component + Ss Se AhAl; d[0] is only readable when n >= 1
      if (progressive) {
        const int ns = d[0];
        if (ns < 1 || ns > (int)comps.size() || n < 1 + 2 * ns + 3) { *err = "bad SOS"; return false; }
        std::vector<int> sel(ns);
        for (int sidx = 0; sidx < ns; ++sidx) {
          sel[sidx] = -1;
          for (size_t c = 0; c < comps.size(); ++c)
            if (comps[c].id == d[1 + 2 * sidx]) { comps[c].td = d[2 + 2 * sidx] >> 4; comps[c].ta = d[2 + 2 * sidx] & 15; sel[sidx] = (int)c; }
          if (sel[sidx] < 0 || comps[sel[sidx]].td > 3 || comps[sel[sidx]].ta > 3) { *err = "bad SOS"; return false; }
        }
        ProgressiveScan sc;
        sc.ss = d[1 + 2 * ns]; sc.se = d[2 + 2 * ns]; sc.ah = d[3 + 2 * ns] >> 4; sc.al = d[3 + 2 * ns] & 15;
        if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || (sc.ss == 0 && sc.se != 0) || (sc.ss > 0 && ns != 1)) { *err = "bad progressive scan header"; return false; }
        if (comps.size() == 3 && (adobe_transform == 0 || (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B'))) {
          *err = "RGB-coded JPEG is not supported";
          return false;
        }
        int hmax = 1, vmax = 1;
        for (const Component& c : comps) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
        const bool single_frame = comps.size() == 1;
        const Component& Y = comps[0];
        if (!single_frame && (Y.h != hmax || Y.v != vmax)) { *err = "JPEG with a subsampled first component is not supported"; return false; }
        const int yh = single_frame ? 1 : Y.h, yv = single_frame ? 1 : Y.v;
        const int mcu_w = single_frame ? 8 : 8 * hmax, mcu_h = single_frame ? 8 : 8 * vmax;
        const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
        if (ycoef.empty()) {
          ybw = mcus_x * yh; ybh = mcus_y * yv; ycoef.assign((size_t)ybw * ybh * 64, 0);
          if (want_rgb && comps.size() == 3)
            for (int c = 1; c < 3; ++c) { pbw[c] = mcus_x * comps[c].h; pbh[c] = mcus_y * comps[c].v; pcoef[c].assign((size_t)pbw[c] * pbh[c] * 64, 0); }
        }
        const size_t data = pos + (size_t)len;
        const bool has_y = sel[0] == 0 || (ns > 1 && (sel[1] == 0 || (ns > 2 && sel[2] == 0)));
        if (!has_y && ns == 1 && !want_rgb) { pos = next_segment_marker(f, data); continue; }      // a chroma-only scan: nothing grey needs
        BitReader br{&f[data], f.data() + f.size()};
        for (Component& c : comps) c.pred = 0;
        int restart_count = 0;
        int16_t scratch[64];
        if (ns > 1) {                                                                  // interleaved: DC scans only
          for (int sidx = 0; sidx < ns; ++sidx)
            if (sc.ah == 0 && !dc[comps[sel[sidx]].td].defined) { *err = "JPEG tables missing"; return false; }
          for (int my = 0; my < mcus_y; ++my)
            for (int mx = 0; mx < mcus_x; ++mx) {
              if (restart_interval && restart_count == restart_interval) {
                if (!take_restart(br)) { *err = "missing JPEG restart marker"; return false; }
                for (Component& c : comps) c.pred = 0;
                restart_count = 0;
              }
              ++restart_count;
              for (int sidx = 0; sidx < ns; ++sidx) {
                Component& c = comps[sel[sidx]];
                for (int by = 0; by < c.v; ++by)
                  for (int bx = 0; bx < c.h; ++bx) {
                    const int ci = sel[sidx];
                    int16_t* blk = ci == 0 ? &ycoef[((size_t)(my * yv + by) * ybw + (size_t)(mx * yh + bx)) * 64]
                                           : (want_rgb ? &pcoef[ci][((size_t)(my * c.v + by) * pbw[ci] + (size_t)(mx * c.h + bx)) * 64] : scratch);
                    if (ci != 0 && !want_rgb) scratch[0] = 0;
                    if (!prog_dc(br, dc[c.td], sc, &c.pred, blk)) { *err = "corrupt JPEG data (DC)"; return false; }
                  }
              }
            }
        } else {                                                                       // one component alone
          const int ci = sel[0];
          Component& c = comps[ci];
          if (sc.ss == 0 ? (sc.ah == 0 && !dc[c.td].defined) : !ac[c.ta].defined) { *err = "JPEG tables missing"; return false; }
          // a non-interleaved scan covers the blocks of the component's own size, not the MCU-padded grid
          const int cw = single_frame ? W : (W * c.h + hmax - 1) / hmax, chh = single_frame ? H : (H * c.v + vmax - 1) / vmax;
          const int bw = (cw + 7) / 8, bh = (chh + 7) / 8;
          for (int by = 0; by < bh; ++by)
            for (int bx = 0; bx < bw; ++bx) {
              if (restart_interval && restart_count == restart_interval) {
                if (!take_restart(br)) { *err = "missing JPEG restart marker"; return false; }
                c.pred = 0; sc.eobrun = 0;
                restart_count = 0;
              }
              ++restart_count;
              int16_t* blk = &pcoef[ci][((size_t)by * pbw[ci] + bx) * 64];
              const bool ok = sc.ss == 0 ? prog_dc(br, dc[c.td], sc, &c.pred, blk)
                                         : (sc.ah == 0 ? prog_ac_first(br, ac[c.ta], sc, blk) : prog_ac_refine(br, ac[c.ta], sc, blk));
              if (!ok) { *err = "corrupt JPEG data (progressive scan)"; return false; }
            }
        }
        pos = next_segment_marker(f, (size_t)(br.p - f.data()));
        continue;
      }
      const int ns = d[0];
      if (ns != (int)comps.size() || n < 1 + 2 * ns + 3) { *err = "multi-scan sequential JPEG is not supported"; return false; }
      for (int s = 0; s < ns; ++s) {
        bool found = false;
        for (Component& c : comps)
          if (c.id == d[1 + 2 * s]) { c.td = d[2 + 2 * s] >> 4; c.ta = d[2 + 2 * s] & 15; found = true; }
        if (!found) { *err = "bad SOS"; return false; }
      }
      if (comps.size() == 3 && (adobe_transform == 0 || (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B'))) {
        *err = "RGB-coded JPEG is not supported";
        return false;
      }
      for (const Component& c : comps)
        if (c.td > 3 || c.ta > 3 || !dc[c.td].defined || !ac[c.ta].defined || !qt_defined[c.tq]) { *err = "JPEG tables missing"; return false; }
      int hmax = 1, vmax = 1;
      for (const Component& c : comps) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
      // a single-component scan is non-interleaved: the MCU is one 8x8 block regardless of the sampling factors
      const bool single = comps.size() == 1;
      const int mcu_w = single ? 8 : 8 * hmax, mcu_h = single ? 8 : 8 * vmax;
      const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
      const Component& Y = comps[0];
      const int yh = single ? 1 : Y.h, yv = single ? 1 : Y.v;
      // luminance plane of whole MCUs (the first component is at full resolution when it has the maximal factors)
      if (!single && (Y.h != hmax || Y.v != vmax)) { *err = "JPEG with a subsampled first component is not supported"; return false; }
      const int PW = mcus_x * mcu_w, PH = mcus_y * mcu_h;
      std::vector<uint8_t> plane((size_t)PW * PH);
      std::vector<uint8_t> cplane[2];                        // chroma planes at their own resolution, only for colour output
      int cpw[2] = {0, 0};
      if (want_rgb && comps.size() == 3)
        for (int c = 0; c < 2; ++c) { cpw[c] = mcus_x * comps[c + 1].h * 8; cplane[c].assign((size_t)cpw[c] * mcus_y * comps[c + 1].v * 8, 0); }
      BitReader br{&f[pos + (size_t)len], f.data() + f.size()};
      int restart_count = 0;
      for (Component& c : comps) c.pred = 0;
      int coef[64];
      for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
          if (restart_interval && restart_count == restart_interval) {
            // byte-align, expect RSTn
            br.reset();
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
            if (q + 1 >= br.end) { *err = "missing JPEG restart marker"; return false; }
            br.p = q + 2;
            for (Component& c : comps) c.pred = 0;
            restart_count = 0;
          }
          ++restart_count;
          for (size_t ci = 0; ci < comps.size(); ++ci) {
            Component& c = comps[ci];
            const int bh = single ? 1 : c.h, bv = single ? 1 : c.v;
            for (int by = 0; by < bv; ++by)
              for (int bx = 0; bx < bh; ++bx) {
                bool ok = true;
                memset(coef, 0, sizeof coef);
                const int t = decode_symbol(br, dc[c.td], &ok);
                if (!ok || t > 15) { *err = "corrupt JPEG data (DC)"; return false; }
                c.pred += extend(br.get(t), t);
                coef[0] = c.pred * qt[c.tq][0];
                for (int k = 1; k < 64;) {
                  const int rs = decode_symbol(br, ac[c.ta], &ok);
                  if (!ok) { *err = "corrupt JPEG data (AC)"; return false; }
                  const int r = rs >> 4, s = rs & 15;
                  if (s == 0) {
                    if (r == 15) { k += 16; continue; }
                    break;                                     // EOB
                  }
                  k += r;
                  if (k > 63) { *err = "corrupt JPEG data (run)"; return false; }
                  const int zz = kZigzag[k];
                  coef[zz] = extend(br.get(s), s) * qt[c.tq][zz];
                  ++k;
                }
                if (ci == 0) {
                  uint8_t* o = &plane[(size_t)(my * mcu_h + by * 8) * PW + (size_t)(mx * mcu_w + bx * 8)];
                  idct_islow(coef, o, PW);
                } else if (want_rgb) {
                  uint8_t* o = &cplane[ci - 1][(size_t)((my * c.v + by) * 8) * cpw[ci - 1] + (size_t)((mx * c.h + bx) * 8)];
                  idct_islow(coef, o, cpw[ci - 1]);
                }
              }
          }
          (void)yh; (void)yv;
        }
      if (want_rgb) {
        const std::vector<uint8_t> planes[3] = {std::move(plane), std::move(cplane[0]), std::move(cplane[1])};
        const int pws[3] = {PW, cpw[0], cpw[1]};
        if (!emit_rgb(planes, pws)) return false;
        return gray ? emit(planes[0], PW) : true;
      }
      return emit(plane, PW);
    }
    pos += (size_t)len;
  }
  if (progressive && !ycoef.empty()) {
    std::vector<uint8_t> planes[3];
    int pws[3] = {0, 0, 0};
    int coef[64];
    for (int c = 0; c < (want_rgb ? (int)comps.size() : 1); ++c) {
      if (!qt_defined[comps[c].tq]) { *err = "JPEG tables missing"; return false; }
      pws[c] = pbw[c] * 8;
      planes[c].assign((size_t)pws[c] * pbh[c] * 8, 0);
      for (int by = 0; by < pbh[c]; ++by)
        for (int bx = 0; bx < pbw[c]; ++bx) {
          const int16_t* blk = &pcoef[c][((size_t)by * pbw[c] + bx) * 64];
          for (int k = 0; k < 64; ++k) coef[k] = blk[k] * qt[comps[c].tq][k];
          idct_islow(coef, &planes[c][(size_t)by * 8 * pws[c] + (size_t)bx * 8], pws[c]);
        }
    }
    if (want_rgb && !emit_rgb(planes, pws)) return false;
    return (gray || !want_rgb) ? emit(planes[0], pws[0]) : true;
  }
  *err = "JPEG without image data";
  return false;
}

inline bool load_jpeg_gray(const std::vector<uint8_t>& f, int* width, int* height, std::vector<uint8_t>* gray, std::string* err) {
  return load_jpeg(f, width, height, gray, nullptr, err);
}

}  // namespace e3d_host
