// e3d_reg.hip -- path (B): dense photometric image-registration kernels on gfx950 and their C-ABI
// (include/e3d_hip.h, section "(B) ImageRegistrator").
//
// Reference loops covered: a16 ComputePointIntensityAndJacobians, a17/a18 colour residual
// accumulation, a19 cost, a20/a21 observation creation, a22 neighbour flags, a23 colour update, a24 splat depth,
// a25 trilinear sampling, a27 camera device functions (PINHOLE).  One (image, point scale) per call; the LM driver
// that assembles the per-image blocks is host code above this ABI.
//
// Data layout in HBM: point scale = float4 {x,y,z,0} + u32 neighbour table (n*K) + f32 descriptors (n*K) + i32 counts;
// image = u8 pyramid levels (row-major, tightly packed) + optional u8 masks; observations = SoA {u32 point, f32 x, y,
// scale, u8 flag}; pass 1 writes 48-byte rows {I, Ji[4], Jp[6], pad} (three float4) so that pass 2's K+1 row gathers
// are 16 B/lane loads; `row_of_point` (i32 per point) replaces the reference's unordered_map.
// Reductions are block partials + a fixed-order second stage (f64), no atomics on values.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <chrono>
#include <cmath>
#include <map>
#include <memory>

#include <random>
#include <string>

#include "../../include/e3d_hip.h"
#include "e3d_camera.hpp"
#include "e3d_comm.hpp"
#include "e3d_icp_kernels.hpp"
#include "e3d_math.hpp"

#pragma clang fp contract(off)

namespace e3d {

struct Pose { float R[9]; float t[3]; };

constexpr int kRegMaxLevels = 16;
struct Pyramid {               // passed by value to kernels
  const unsigned char* img[kRegMaxLevels];
  const unsigned char* mask[kRegMaxLevels];
  const unsigned char* cam_mask[kRegMaxLevels];      // Intrinsics::camera_mask (intrinsics.h:104): shared by all images of the camera
  CamLevel cam[kRegMaxLevels];
  int n_levels;
  int min_image_scale;
};

// x86 cvttss2si / cvttsd2si semantics of the reference's `int ix = v + 0.5f;`
__device__ __forceinline__ int f2i(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : INT_MIN; }
__device__ __forceinline__ int d2i(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN; }

__device__ __forceinline__ void rt(const Pose& P, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = dot3e(P.R[0], P.R[1], P.R[2], x, y, z) + P.t[0];
  oy = dot3e(P.R[3], P.R[4], P.R[5], x, y, z) + P.t[1];
  oz = dot3e(P.R[6], P.R[7], P.R[8], x, y, z) + P.t[2];
}

// ---- interpolation (interpolate_bilinear.h:36-74, interpolate_trilinear.h:44-87) ------------------------------------------
// (pixel type T: u8 images, f32 depth maps -- InterpolateTrilinear* are templates in the reference too, interpolate_trilinear.h)
template <class T>
__device__ __forceinline__ float bilinear(const T* img, int w, float x, float y, int ix, int iy) {
  const float fx = x - ix, fxi = 1.f - fx, fy = y - iy, fyi = 1.f - fy;
  const T* r0 = img + (size_t)iy * w;
  const T* r1 = img + (size_t)(iy + 1) * w;
  return fyi * (fxi * r0[ix] + fx * r0[ix + 1]) + fy * (fxi * r1[ix] + fx * r1[ix + 1]);
}
template <class T>
__device__ __forceinline__ void bilinear_d(const T* img, int w, float x, float y, int ix, int iy, float& v,
                                           float& dx, float& dy) {
  const T* r0 = img + (size_t)iy * w;
  const T* r1 = img + (size_t)(iy + 1) * w;
  const float tl = r0[ix], tr = r0[ix + 1], bl = r1[ix], br = r1[ix + 1];
  const float fx = x - ix, fxi = 1.f - fx, fy = y - iy, fyi = 1.f - fy;
  const float top = fxi * tl + fx * tr, bottom = fxi * bl + fx * br;
  v = fyi * top + fy * bottom;
  dx = fy * (br - bl) + fyi * (tr - tl);
  dy = bottom - top;
}
template <class T>
__device__ __forceinline__ float trilinear(const T* i0, int w0, const T* i1, int w1, float x0,
                                           float y0, float z) {
  const float v0 = bilinear(i0, w0, x0, y0, (int)x0, (int)y0);
  const float x1 = 2 * (x0 + 0.5f) - 0.5f, y1 = 2 * (y0 + 0.5f) - 0.5f;
  const float v1 = bilinear(i1, w1, x1, y1, (int)x1, (int)y1);
  return (1 - z) * v0 + z * v1;
}
template <class T>
__device__ __forceinline__ void trilinear_d(const T* i0, int w0, const T* i1, int w1, float x0,
                                            float y0, float z, float& v, float& dx, float& dy, float& dz) {
  float v0, dx0, dy0, v1, dx1, dy1;
  bilinear_d(i0, w0, x0, y0, (int)x0, (int)y0, v0, dx0, dy0);
  const float x1 = 2 * (x0 + 0.5f) - 0.5f, y1 = 2 * (y0 + 0.5f) - 0.5f;
  bilinear_d(i1, w1, x1, y1, (int)x1, (int)y1, v1, dx1, dy1);
  v = (1 - z) * v0 + z * v1;
  dx = (1 - z) * dx0 + z * 2 * dx1;
  dy = (1 - z) * dy0 + z * 2 * dy1;
  dz = v1 - v0;
}
__device__ __forceinline__ float obs_intensity(const Pyramid& Y, float ox, float oy, float os) {
  const int s = f2i(os);
  const int l1 = s - Y.min_image_scale, l0 = l1 + 1;
  return trilinear(Y.img[l0], Y.cam[l0].width, Y.img[l1], Y.cam[l1].width, ox, oy, 1 - (os - (float)s));
}

// ---- robust weighting (robust_weighting.h:61-107) ------------------------------------------------------------------------------
__device__ __forceinline__ float robust_residual(int type, float param, float r) {
  if (type == 1) { const float a = fabsf(r); return (a < param) ? 0.5f * r * r : param * (a - 0.5f * param); }
  if (type == 2) {
    const float a = fabsf(r);
    if (a < param) { const float q = r / param; const float t = 1.f - q * q; return (1 / 6.f) * param * param * (1 - t * t * t); }
    return (1 / 6.f) * param * param;
  }
  return 0.5f * r * r;
}
__device__ __forceinline__ float robust_weight(int type, float param, float r) {
  if (type == 1) { const float a = fabsf(r); return (a < param) ? 1.f : (param / a); }
  if (type == 2) { const float a = fabsf(r); if (a < param) { const float q = r / param; const float t = 1.f - q * q; return t * t; } return 0.f; }
  return 1.f;
}

// ==== a24: OcclusionGeometry::_RenderDepthMapWithSplatsCPU ========================================================================
// depth(x, y) = min z over all points whose splat rectangle covers (x, y): an order-free reduction, so the result is exact
// whatever the schedule.  Splats overlap heavily (a 3 cm splat at 3 m is the 21 x 21 pixel maximum; scan points are
// millimetres apart), so instead of one global atomic per covered pixel the points are binned into 32 x 32 pixel tiles,
// radix-sorted by tile, and each tile is reduced in LDS by one workgroup and written once.
constexpr int kTile = 32;

// the reference's rectangle of one point (occlusion_geometry.cc:423-452), clipped to the image; false if empty
constexpr int kSplatMax = 10;        // max_splat_radius (occlusion_geometry.cc:417)

template <int M>
__device__ __forceinline__ bool splat_rect(const float4 p, const Pose& P, const CamLevel& cam, float point_radius, int& min_x, int& min_y,
                                           int& end_x, int& end_y, unsigned& zb, bool& full_size, int& cxi, int& cyi) {
  full_size = false;
  float X, Y, Z;
  rt(P, p.x, p.y, p.z, X, Y, Z);
  if (!(Z > 0.f)) return false;
  float px, py, d[6];
  const CamTheta th = cam_theta<M>(X / Z, Y / Z);                 // (the fisheye models' atan: once per point)
  cam_normalized_to_image<M>(cam, X / Z, Y / Z, th, px, py);
  cam_image_deriv_by_world<M>(cam, X, Y, Z, th, d);
  float rx = sqrtf(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])) * point_radius;
  float ry = sqrtf(d[3] * d[3] + (d[4] * d[4] + d[5] * d[5])) * point_radius;
  full_size = rx >= 10.f && ry >= 10.f;     // both half-extents clamped: the rectangle is exactly [ix-10, ix+10] x [iy-10, iy+10]
  rx = (10.f < rx) ? 10.f : rx;     // std::min(splat_radius, max_splat_radius)
  ry = (10.f < ry) ? 10.f : ry;
  const int ix = f2i(px + 0.5f), iy = f2i(py + 0.5f);
  cxi = ix; cyi = iy;
  min_x = d2i((double)((float)ix - rx) + 0.5); min_y = d2i((double)((float)iy - ry) + 0.5);
  end_x = d2i((double)((float)ix + rx) + 1.5); end_y = d2i((double)((float)iy + ry) + 1.5);
  min_x = max(min_x, 0); min_y = max(min_y, 0);
  end_x = min(end_x, cam.width); end_y = min(end_y, cam.height);
  zb = __float_as_uint(Z);          // Z > 0: the bit pattern is order preserving
  return min_x < end_x && min_y < end_y;
}

// one (tile, point) pair per tile a rectangle touches (<= 2 x 2: rectangles are at most 22 pixels wide); wave-aggregated append
template <int M>
__global__ __launch_bounds__(kBlock) void k_splat_bin(const float4* __restrict__ pts, size_t n, Pose P, CamLevel cam,
                                                      float point_radius, int tiles_x, uint4* __restrict__ rects,
                                                      unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                      unsigned* __restrict__ counter, unsigned* __restrict__ zbuf) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int min_x = 0, min_y = 0, end_x = 0, end_y = 0, cxi = 0, cyi = 0;
  unsigned zb = 0;
  bool ok = false, full_size = false;
  if (i < n) ok = splat_rect<M>(pts[i], P, cam, point_radius, min_x, min_y, end_x, end_y, zb, full_size, cxi, cyi);
  if (ok && full_size) {
    // Full-size splats (the near, heavily overlapping ones) are not rasterised one by one: their union is the 21 x 21
    // MIN FILTER of the one-pixel-per-point z-buffer, computed separably afterwards (k_min_filter_h / _v).  The buffer
    // carries a 10-pixel apron because a splat centred outside the image can still reach into it.
    const int wp = cam.width + 2 * kSplatMax;
    if (cxi >= -kSplatMax && cyi >= -kSplatMax && cxi < cam.width + kSplatMax && cyi < cam.height + kSplatMax)
      atomicMin(&zbuf[(size_t)(cyi + kSplatMax) * wp + (cxi + kSplatMax)], zb);
    ok = false;
  }
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (ok) {
    tx0 = min_x / kTile; tx1 = (end_x - 1) / kTile; ty0 = min_y / kTile; ty1 = (end_y - 1) / kTile;
    rects[i] = make_uint4((unsigned)min_x | ((unsigned)min_y << 16), (unsigned)end_x | ((unsigned)end_y << 16), zb, 0u);
  }
  const unsigned count = ok ? (unsigned)((tx1 - tx0 + 1) * (ty1 - ty0 + 1)) : 0u;
  const int lane = threadIdx.x & 63;
  unsigned incl = count;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  // one append per workgroup (a single hot address serialises: one atomic per wave costs more than the whole projection)
  __shared__ unsigned wave_total[kBlock / kWave];
  __shared__ unsigned block_base;
  const int wv = threadIdx.x >> 6;
  if (lane == 63) wave_total[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; ++k) tot += wave_total[k];
    block_base = tot ? atomicAdd(counter, tot) : 0u;
  }
  __syncthreads();
  unsigned base = block_base;
  for (int k = 0; k < wv; ++k) base += wave_total[k];
  unsigned o = base + incl - count;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) { keys[o] = (unsigned)(ty * tiles_x + tx); vals[o] = (unsigned)i; ++o; }
}

__global__ __launch_bounds__(kBlock) void k_tile_ranges(const unsigned* __restrict__ keys, size_t n, unsigned* __restrict__ start,
                                                        unsigned* __restrict__ end) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned k = keys[i];
  if (i == 0 || keys[i - 1] != k) start[k] = (unsigned)i;
  if (i + 1 == n || keys[i + 1] != k) end[k] = (unsigned)(i + 1);
}

// one workgroup per tile: z-min of the tile's rectangles in LDS (ds_min_u32), every pixel of the tile written exactly once
__global__ __launch_bounds__(kBlock) void k_splat_tiles(const uint4* __restrict__ rects, const unsigned* __restrict__ vals,
                                                        const unsigned* __restrict__ start, const unsigned* __restrict__ end,
                                                        int tiles_x, int width, int height, unsigned* __restrict__ depth_bits) {
  __shared__ unsigned tile[kTile * kTile];
  const int t = blockIdx.x;
  const int x0 = (t % tiles_x) * kTile, y0 = (t / tiles_x) * kTile;
  for (int i = threadIdx.x; i < kTile * kTile; i += kBlock) tile[i] = 0x7f800000u;
  __syncthreads();
  const unsigned s = start[t], e = end[t];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lx = lane & 31, ly = lane >> 5;
  constexpr unsigned kStep = kBlock / kWave;
  uint4 nxt = (s + wv < e) ? rects[vals[s + wv]] : make_uint4(0, 0, 0, 0);
  for (unsigned j = s + wv; j < e; j += kStep) {
    const uint4 r = nxt;
    if (j + kStep < e) nxt = rects[vals[j + kStep]];       // fetch the next rectangle while this one is applied
    const int mx = max((int)(r.x & 0xffffu), x0) - x0, my = max((int)(r.x >> 16), y0) - y0;
    const int ex = min((int)(r.y & 0xffffu), x0 + kTile) - x0, ey = min((int)(r.y >> 16), y0 + kTile) - y0;
    const int x = mx + lx;
    if (x < ex)
      for (int y = my + ly; y < ey; y += 2) atomicMin(&tile[y * kTile + x], r.z);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kTile * kTile; i += kBlock) {
    const int x = x0 + (i & (kTile - 1)), y = y0 + (i / kTile);
    if (x < width && y < height) depth_bits[(size_t)y * width + x] = tile[i];
  }
}

// separable 21 x 21 min filter of the apron-padded point z-buffer (positive float bits compare like unsigned integers)
__global__ __launch_bounds__(kBlock) void k_min_filter_h(const unsigned* __restrict__ zbuf, int width, int height,
                                                         unsigned* __restrict__ tmp /* (height + 20) x width */) {
  __shared__ unsigned row[kBlock + 2 * kSplatMax];
  const int wp = width + 2 * kSplatMax;
  const int y = blockIdx.y;                                   // padded row
  const int x0 = blockIdx.x * kBlock;
  for (int i = threadIdx.x; i < kBlock + 2 * kSplatMax; i += kBlock) {
    const int xp = x0 + i;                                    // padded column of output column x0 + i - 10
    row[i] = (xp < wp) ? zbuf[(size_t)y * wp + xp] : 0x7f800000u;
  }
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x >= width) return;
  unsigned m = 0x7f800000u;
#pragma unroll
  for (int d = 0; d <= 2 * kSplatMax; ++d) m = min(m, row[threadIdx.x + d]);
  tmp[(size_t)y * width + x] = m;
}
__global__ __launch_bounds__(kBlock) void k_min_filter_v(const unsigned* __restrict__ tmp, int width, int height,
                                                         unsigned* __restrict__ depth_bits) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= width) return;
  unsigned m = depth_bits[(size_t)y * width + x];             // what the rectangle path (smaller splats) produced
#pragma unroll
  for (int d = 0; d <= 2 * kSplatMax; ++d) m = min(m, tmp[(size_t)(y + d) * width + x]);
  depth_bits[(size_t)y * width + x] = m;
}

// The two passes in one kernel (round 4): a block owns a kMfW x kMfH tile of the image, stages the tile of the padded z-buffer with
// its 20-pixel skirt in LDS once, filters it horizontally into a second LDS tile and vertically into the depth map -- the
// intermediate image never goes to memory (two 98 MB images written and read per 24 MP depth map before: 0.22 ms; the vertical
// pass alone read 21 rows per pixel through L2).  A thread produces four neighbouring pixels of a row, then eight of a column:
// 6 + 3.5 LDS reads per pixel instead of 21 + 21.  min is associative: bit-identical to the separable passes.
// MERGE = false: no splat took the rectangle path (the usual case at room scale: every splat is full size), so the depth map holds
// nothing yet -- the filter's result is STORED instead of merged, and the pass that would have filled the map with +inf first
// (k_splat_tiles over all tiles: a 4 B store per pixel, read back here) is not launched.  min(+inf, m) = m: the same bits.
constexpr int kMfW = 64, kMfH = 32, kMfWin = 2 * kSplatMax + 1;
template <bool MERGE>
__global__ __launch_bounds__(kBlock) void k_min_filter_tile(const unsigned* __restrict__ zbuf, int width, int height,
                                                            unsigned* __restrict__ depth_bits) {
  constexpr int IW = kMfW + 2 * kSplatMax, IH = kMfH + 2 * kSplatMax;
  __shared__ unsigned tin[IH][IW];
  __shared__ unsigned th[IH][kMfW];
  const int wp = width + 2 * kSplatMax, hp = height + 2 * kSplatMax;
  const int x0 = blockIdx.x * kMfW, y0 = blockIdx.y * kMfH;           // tile origin (image = padded coordinates of its skirt's corner)
  for (int i = threadIdx.x; i < IH * IW; i += kBlock) {
    const int r = i / IW, c = i - r * IW;
    const int xp = x0 + c, yp = y0 + r;
    tin[r][c] = (xp < wp && yp < hp) ? zbuf[(size_t)yp * wp + xp] : 0x7f800000u;
  }
  __syncthreads();
  // horizontal: IH rows x kMfW columns, four columns per thread
  for (int i = threadIdx.x; i < IH * (kMfW / 4); i += kBlock) {
    const int r = i / (kMfW / 4), c = (i - r * (kMfW / 4)) * 4;
    unsigned mid = 0x7f800000u;                                        // columns c + 3 .. c + 20 are in all four windows
#pragma unroll
    for (int d = 3; d < kMfWin; ++d) mid = min(mid, tin[r][c + d]);
    const unsigned a0 = tin[r][c], a1 = tin[r][c + 1], a2 = tin[r][c + 2];
    const unsigned b0 = tin[r][c + kMfWin], b1 = tin[r][c + kMfWin + 1], b2 = tin[r][c + kMfWin + 2];
    th[r][c] = min(mid, min(a0, min(a1, a2)));
    th[r][c + 1] = min(mid, min(min(a1, a2), b0));
    th[r][c + 2] = min(mid, min(a2, min(b0, b1)));
    th[r][c + 3] = min(mid, min(b0, min(b1, b2)));
  }
  __syncthreads();
  // vertical: one column, eight rows per thread
  const int c = threadIdx.x & (kMfW - 1), r0 = (threadIdx.x / kMfW) * 8;
  const int x = x0 + c;
  unsigned mid = 0x7f800000u;                                          // rows r0 + 7 .. r0 + 20 are in all eight windows
#pragma unroll
  for (int d = 7; d < kMfWin; ++d) mid = min(mid, th[r0 + d][c]);
  unsigned lo[7], hi[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) { lo[d] = th[r0 + d][c]; hi[d] = th[r0 + kMfWin + d][c]; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    unsigned m = mid;
#pragma unroll
    for (int d = k; d < 7; ++d) m = min(m, lo[d]);
#pragma unroll
    for (int d = 0; d < k; ++d) m = min(m, hi[d]);
    const int y = y0 + r0 + k;
    if (x < width && y < height) {
      const size_t o = (size_t)y * width + x;
      if constexpr (MERGE) depth_bits[o] = min(depth_bits[o], m);      // with what the rectangle path (smaller splats) produced
      else depth_bits[o] = m;
    }
  }
}

// ==== f2: occlusion meshes (OcclusionGeometry::RenderDepthMap mesh branch, occlusion_geometry.cc:211-271; the reference
// renders with OpenGL, src/opengl/renderer.cc) ================================================================================
// Software rasteriser with the renderer's conventions: the vertex shader's distortion code per camera model
// (renderer.cc:226-262,470-495,630-653: no cut-off branch for PINHOLE, `* 99` outside the cut-off), projection
// p = f * x'/z + c (SetupProjection :919-974 maps pixel centres to integer coordinates of this code base), depth =
// perspective-correct interpolation of the camera-space z (the fragment shader writes var_depth), nearest fragment wins,
// pixels without geometry are 0 (glClearColor).  Triangles that cross the near plane z = min_depth are clipped the way OpenGL clips
// them: on the vertex shader's OUTPUT (x', y', z) -- linear interpolation along the edge from the vertex inside to the one outside,
// so the cut is shared by the triangles on both sides of an edge -- and before the perspective division; a vertex behind the camera
// goes through the distortion code like any other (its x / z is mirrored), as in the shader.  OpenGL's own rasterisation is implementation-defined at pixel-boundary ties; here: samples at integer pixel
// coordinates, edge functions in f64 (exact for f32 inputs), top-left rule.
template <int M>
__global__ __launch_bounds__(kBlock) void k_mesh_vertices(const float4* __restrict__ v, size_t n, Pose P, CamLevel cam,
                                                          float4* __restrict__ out, float2* __restrict__ out_l) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = v[i];
  float X, Y, Z;
  rt(P, p.x, p.y, p.z, X, Y, Z);
  float lx = X, ly = Y;
  if constexpr (M == kSimpleRadial) {
    // SimpleRadial shader (renderer.cc:402-413): its own r2 expression, 99 outside the cut-off
    float r2 = (X * X + Y * Y) / (Z * Z);
    if (r2 > cam.cutoff2) r2 = 99.0f; else r2 = 1.0f + r2 * cam.q[0];
    lx = r2 * X; ly = r2 * Y;
  } else if constexpr (M == kRadial || M == kPolynomial3) {
    // Radial / Polynomial shaders (renderer.cc:326-337, 288-300)
    const float nx = X / Z, ny = Y / Z;
    float r2 = nx * nx + ny * ny;
    if (r2 <= cam.cutoff2) {
      if constexpr (M == kRadial) r2 = 1.0f + r2 * (cam.q[0] + r2 * cam.q[1]);
      else r2 = 1.0f + r2 * (cam.q[0] + r2 * (cam.q[1] + r2 * cam.q[2]));
    } else {
      r2 = 99.0f;
    }
    lx = r2 * X; ly = r2 * Y;
  } else if constexpr (M == kRadialFisheye || M == kSimpleRadialFisheye) {
    // RadialFisheye / SimpleRadialFisheye shaders (renderer.cc:361-378, 434-452): r2 after the fisheye warp against the OUTER camera's cut-off
    float nx = X / Z, ny = Y / Z;
    float r2 = nx * nx + ny * ny;
    const float r = sqrtf(r2);
    if (r > 1e-6f) {
      const float theta_by_r = e3d_atan2f(r, 1.0f) / r;
      if constexpr (M == kRadialFisheye) { nx = theta_by_r * nx; ny = theta_by_r * ny; r2 = nx * nx + ny * ny; }
      else { r2 = r2 * theta_by_r * theta_by_r; nx = nx * theta_by_r; ny = ny * theta_by_r; }
    }
    if (r2 <= cam.cutoff2) {
      if constexpr (M == kRadialFisheye) r2 = 1.0f + r2 * (cam.q[0] + r2 * cam.q[1]);
      else r2 = 1.0f + r2 * cam.q[0];
    } else {
      r2 = 99.0f;
    }
    lx = Z * r2 * nx; ly = Z * r2 * ny;
  } else if constexpr (M == kFullOpenCV) {
    // FullOpenCV shader (renderer.cc:528-543)
    const float nx = X / Z, ny = Y / Z;
    const float x2 = nx * nx, xy = nx * ny, y2 = ny * ny;
    const float r2 = x2 + y2;
    if (r2 <= cam.cutoff2) {
      const float k1 = cam.q[0], k2 = cam.q[1], p1 = cam.q[2], p2 = cam.q[3], k3 = cam.q[4], k4 = cam.q[5], k5 = cam.q[6], k6 = cam.q[7];
      const float radial = (1.0f + r2 * (k1 + r2 * (k2 + r2 * k3))) / (1.0f + r2 * (k4 + r2 * (k5 + r2 * k6)));
      lx = Z * (radial * nx + 2.0f * p1 * xy + p2 * (r2 + 2.0f * x2));
      ly = Z * (radial * ny + 2.0f * p2 * xy + p1 * (r2 + 2.0f * y2));
    } else {
      lx = X * 99.0f; ly = Y * 99.0f;
    }
  } else if constexpr (M != kPinhole && M != kSimplePinhole) {
    float nx = X / Z, ny = Y / Z;
    float r2 = nx * nx + ny * ny;
    if constexpr (M == kFov) {
      // FisheyeFOV shader (renderer.cc:153-160): r = length(xy) / z; r = atan(r * two_tan_omega_half) / (r * omega); xy *= r.  GLSL
      // leaves 0 / 0 on the optical axis undefined; the camera class's guard (camera_fisheye_fov.h:58-61) is used there
      const float r = sqrtf(X * X + Y * Y) / Z;
      const float f = (r < 1e-6f) ? 1.0f : e3d_atanf(r * cam.q[1]) / (r * cam.q[0]);
      lx = f * X; ly = f * Y;
    } else if constexpr (M == kOpenCVFisheye) {
      // FisheyePolynomial4 shader (renderer.cc:187-205): r2 turns into the radial factor (99 outside the cut-off)
      if (r2 <= cam.cutoff2) {
        const float r = sqrtf(r2);
        if (r > 1e-6f) { const float theta_by_r = e3d_atan2f(r, 1.0f) / r; nx = theta_by_r * nx; ny = theta_by_r * ny; r2 = theta_by_r * theta_by_r * r2; }
        r2 = 1.0f + r2 * (cam.q[0] + r2 * (cam.q[1] + r2 * (cam.q[2] + r2 * cam.q[3])));
      } else {
        r2 = 99.0f;
      }
      lx = Z * r2 * nx; ly = Z * r2 * ny;
    } else if (r2 <= cam.cutoff2) {
      if constexpr (cam_is_fisheye(M)) {           // THIN_PRISM_FISHEYE, FISHEYE_POLYNOMIAL_2_TANGENTIAL_2 (renderer.cc:236-258)
        const float r = sqrtf(r2);
        if (r > 1e-6f) { const float theta_by_r = e3d_atan2f(r, 1.0f) / r; nx = theta_by_r * nx; ny = theta_by_r * ny; }
      }
      const float x2 = nx * nx, xy = nx * ny, y2 = ny * ny;
      r2 = x2 + y2;
      const float k1 = cam.q[0], k2 = cam.q[1], p1 = cam.q[2], p2 = cam.q[3];
      if constexpr (cam_is_poly_tang(M)) {
        const float radial = 1.0f + r2 * (k1 + r2 * k2);
        lx = Z * (radial * nx + 2.0f * p1 * xy + p2 * (r2 + 2.0f * x2));
        ly = Z * (radial * ny + 2.0f * p2 * xy + p1 * (r2 + 2.0f * y2));
      } else {
        const float k3 = cam.q[4], k4 = cam.q[5], sx1 = cam.q[6], sy1 = cam.q[7];
        const float radial = 1.0f + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)));
        lx = Z * (radial * nx + 2.0f * p1 * xy + p2 * (r2 + 2.0f * x2) + sx1 * r2);
        ly = Z * (radial * ny + 2.0f * p2 * xy + p1 * (r2 + 2.0f * y2) + sy1 * r2);
      }
    } else {
      lx = X * 99.0f; ly = Y * 99.0f;
    }
  }
  out[i] = make_float4(cam.fx * (lx / Z) + cam.cx, cam.fy * (ly / Z) + cam.cy, Z, 0.f);
  out_l[i] = make_float2(lx, ly);
}

struct MeshProj { float fx, fy, cx, cy, near; };

// the point where the edge from vertex `in` (z >= near) to vertex `out` (z < near) meets the near plane, projected
__device__ __forceinline__ float4 near_cut(const float4 pin, const float2 lin, const float4 pout, const float2 lout, const MeshProj& m) {
  const float t = (m.near - pin.z) / (pout.z - pin.z);
  const float lx = lin.x + t * (lout.x - lin.x), ly = lin.y + t * (lout.y - lin.y);
  return make_float4(m.fx * (lx / m.near) + m.cx, m.fy * (ly / m.near) + m.cy, m.near, 0.f);
}
// Sub-triangle `sub` of triangle (p0 p1 p2) after near-plane clipping; returns how many sub-triangles there are (0: entirely in
// front of the plane, 1: untouched or one vertex kept, 2: one vertex cut off -- the quad is split along the diagonal from the first
// cut to the second kept vertex)
__device__ __forceinline__ int near_clip(const float4* p, const float2* l, const MeshProj& m, int sub, float4& a, float4& b, float4& c) {
  const bool in0 = p[0].z >= m.near, in1 = p[1].z >= m.near, in2 = p[2].z >= m.near;
  const int n_in = (int)in0 + (int)in1 + (int)in2;
  if (n_in == 3) { a = p[0]; b = p[1]; c = p[2]; return 1; }
  if (n_in == 0) return 0;
  if (n_in == 1) {
    const int i = in0 ? 0 : (in1 ? 1 : 2), j = (i + 1) % 3, k = (i + 2) % 3;
    a = p[i]; b = near_cut(p[i], l[i], p[j], l[j], m); c = near_cut(p[i], l[i], p[k], l[k], m);
    return 1;
  }
  const int i = !in0 ? 0 : (!in1 ? 1 : 2), j = (i + 1) % 3, k = (i + 2) % 3;
  const float4 pj = near_cut(p[j], l[j], p[i], l[i], m);
  if (sub == 0) { a = pj; b = p[j]; c = p[k]; }
  else { a = pj; b = p[k]; c = near_cut(p[k], l[k], p[i], l[i], m); }
  return 2;
}

struct TriSetup { double ax, ay, bx, by, cx, cy; float za, zb, zc; int x0, y0, x1, y1; bool ok; };

// bounding box (inclusive pixel range) and validity of one projected triangle
__device__ __forceinline__ TriSetup tri_setup(const float4 a, const float4 b, const float4 c, int width, int height, float min_depth,
                                              float max_depth) {
  TriSetup t{};
  t.ok = false;
  if (a.z > max_depth && b.z > max_depth && c.z > max_depth) return t;
  if (!(isfinite(a.x) && isfinite(a.y) && isfinite(b.x) && isfinite(b.y) && isfinite(c.x) && isfinite(c.y))) return t;
  const float fx0 = fminf(a.x, fminf(b.x, c.x)), fx1 = fmaxf(a.x, fmaxf(b.x, c.x));
  const float fy0 = fminf(a.y, fminf(b.y, c.y)), fy1 = fmaxf(a.y, fmaxf(b.y, c.y));
  if (fx1 < 0.f || fy1 < 0.f || fx0 > (float)(width - 1) || fy0 > (float)(height - 1)) return t;
  t.x0 = max(0, (int)ceilf(fx0)); t.y0 = max(0, (int)ceilf(fy0));
  t.x1 = min(width - 1, (int)floorf(fx1)); t.y1 = min(height - 1, (int)floorf(fy1));
  if (t.x0 > t.x1 || t.y0 > t.y1) return t;
  t.ax = a.x; t.ay = a.y; t.bx = b.x; t.by = b.y; t.cx = c.x; t.cy = c.y;
  t.za = a.z; t.zb = b.z; t.zc = c.z;
  const double area = (t.bx - t.ax) * (t.cy - t.ay) - (t.by - t.ay) * (t.cx - t.ax);
  if (area == 0.0) return t;
  if (area < 0.0) {     // orient so that interior points have positive edge functions (culling is disabled)
    double d; float f;
    d = t.bx; t.bx = t.cx; t.cx = d; d = t.by; t.by = t.cy; t.cy = d; f = t.zb; t.zb = t.zc; t.zc = f;
  }
  t.ok = true;
  return t;
}
__device__ __forceinline__ bool edge_inside(double ax, double ay, double bx, double by, double px, double py, double& e) {
  const double dx = bx - ax, dy = by - ay;
  e = dx * (py - ay) - dy * (px - ax);
  if (e > 0.0) return true;
  if (e < 0.0) return false;
  return dy < 0.0 || (dy == 0.0 && dx > 0.0);        // top-left rule for samples exactly on an edge
}
// depth of the triangle at an integer pixel, or 0 if the pixel is outside / beyond the depth range
__device__ __forceinline__ float tri_depth(const TriSetup& t, int x, int y, float min_depth, float max_depth) {
  double e0, e1, e2;
  const bool in0 = edge_inside(t.bx, t.by, t.cx, t.cy, x, y, e0);
  const bool in1 = edge_inside(t.cx, t.cy, t.ax, t.ay, x, y, e1);
  const bool in2 = edge_inside(t.ax, t.ay, t.bx, t.by, x, y, e2);
  if (!(in0 && in1 && in2)) return 0.f;
  const double area = e0 + e1 + e2;
  if (!(area > 0.0)) return 0.f;
  const double inv = (e0 / (double)t.za + e1 / (double)t.zb + e2 / (double)t.zc) / area;      // interpolated 1/z
  const float z = (float)(1.0 / inv);
  if (!(z >= min_depth && z <= max_depth)) return 0.f;
  return z;
}

__global__ __launch_bounds__(kBlock) void k_mesh_bin(const float4* __restrict__ pv, const float2* __restrict__ pl,
                                                     const unsigned* __restrict__ tris, size_t n_tris, MeshProj mp,
                                                     int width, int height, float min_depth, float max_depth, int tiles_x,
                                                     unsigned tri_offset, unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                     unsigned* __restrict__ counter, unsigned capacity) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int tx0[2] = {0, 0}, tx1[2] = {-1, -1}, ty0[2] = {0, 0}, ty1[2] = {-1, -1};
  unsigned count = 0;
  if (i < n_tris) {
    const unsigned i0 = tris[3 * i], i1 = tris[3 * i + 1], i2 = tris[3 * i + 2];
    const float4 p[3] = {pv[i0], pv[i1], pv[i2]};
    const float2 l[3] = {pl[i0], pl[i1], pl[i2]};
    int n_sub = 1;
    for (int sub = 0; sub < n_sub; ++sub) {        // sub-triangle 1 exists only where the near plane cuts off one vertex
      float4 a, b, c;
      n_sub = near_clip(p, l, mp, sub, a, b, c);
      if (sub >= n_sub) break;
      const TriSetup t = tri_setup(a, b, c, width, height, min_depth, max_depth);
      if (t.ok) {
        tx0[sub] = t.x0 / kTile; tx1[sub] = t.x1 / kTile; ty0[sub] = t.y0 / kTile; ty1[sub] = t.y1 / kTile;
        count += (unsigned)((tx1[sub] - tx0[sub] + 1) * (ty1[sub] - ty0[sub] + 1));
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned incl = count;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o); if (lane >= o) incl += v; }
  __shared__ unsigned wave_total[kBlock / kWave];
  __shared__ unsigned block_base;
  if (lane == 63) wave_total[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; ++k) tot += wave_total[k];
    block_base = tot ? atomicAdd(counter, tot) : 0u;
  }
  __syncthreads();
  unsigned o = block_base + incl - count;
  for (int k = 0; k < wv; ++k) o += wave_total[k];
  for (int sub = 0; sub < 2; ++sub)
    for (int ty = ty0[sub]; ty <= ty1[sub]; ++ty)
      for (int tx = tx0[sub]; tx <= tx1[sub]; ++tx) {
        if (o < capacity) { keys[o] = (unsigned)(ty * tiles_x + tx); vals[o] = (tri_offset + (unsigned)i) | ((unsigned)sub << 31); }
        ++o;
      }
}

// one workgroup per tile, one LANE per triangle (occlusion meshes are fine: a triangle covers a few pixels)
__global__ __launch_bounds__(kBlock) void k_mesh_tiles(const float4* __restrict__ pv, const float2* __restrict__ pl,
                                                       const unsigned* __restrict__ tris, MeshProj mp,
                                                       const unsigned* __restrict__ vals, const unsigned* __restrict__ start,
                                                       const unsigned* __restrict__ end, int tiles_x, int width, int height,
                                                       float min_depth, float max_depth, unsigned* __restrict__ depth_bits) {
  __shared__ unsigned tile[kTile * kTile];
  const int t = blockIdx.x;
  const int x0 = (t % tiles_x) * kTile, y0 = (t / tiles_x) * kTile;
  for (int i = threadIdx.x; i < kTile * kTile; i += kBlock) tile[i] = 0x7f800000u;
  __syncthreads();
  const unsigned s = start[t], e = end[t];
  for (unsigned j = s + threadIdx.x; j < e; j += kBlock) {
    const unsigned f = vals[j] & 0x7fffffffu;
    const int sub = (int)(vals[j] >> 31);            // second half of a triangle whose near-plane cut is a quad
    const unsigned i0 = tris[3 * (size_t)f], i1 = tris[3 * (size_t)f + 1], i2 = tris[3 * (size_t)f + 2];
    const float4 p[3] = {pv[i0], pv[i1], pv[i2]};
    const float2 l[3] = {pl[i0], pl[i1], pl[i2]};
    float4 a, b, c;
    if (sub >= near_clip(p, l, mp, sub, a, b, c)) continue;
    const TriSetup ts = tri_setup(a, b, c, width, height, min_depth, max_depth);
    if (!ts.ok) continue;
    const int bx0 = max(ts.x0, x0), by0 = max(ts.y0, y0), bx1 = min(ts.x1, x0 + kTile - 1), by1 = min(ts.y1, y0 + kTile - 1);
    for (int y = by0; y <= by1; ++y)
      for (int x = bx0; x <= bx1; ++x) {
        const float z = tri_depth(ts, x, y, min_depth, max_depth);
        if (z > 0.f) atomicMin(&tile[(y - y0) * kTile + (x - x0)], __float_as_uint(z));
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kTile * kTile; i += kBlock) {
    const int x = x0 + (i & (kTile - 1)), y = y0 + (i / kTile);
    if (x < width && y < height) depth_bits[(size_t)y * width + x] = (tile[i] == 0x7f800000u) ? 0u : tile[i];     // no geometry: 0
  }
}

// ---- occlusion boundaries (ComputeEdgeNormalsList / FilterEdgeList :488-645, MaskOutOcclusionBoundaries :284-402) -------------
struct MeshEdge { unsigned v1, v2, f1, f2; int opposite; };       // f2 == 0xFFFFFFFF: boundary edge (one face)

__global__ __launch_bounds__(kBlock) void k_face_normals(const float4* __restrict__ v, const unsigned* __restrict__ tris, size_t n_tris,
                                                         float4* __restrict__ normals, unsigned long long* __restrict__ keys,
                                                         unsigned* __restrict__ vals) {
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_tris) return;
  const unsigned i0 = tris[3 * f], i1 = tris[3 * f + 1], i2 = tris[3 * f + 2];
  const float4 p0 = v[i0], p1 = v[i1], p2 = v[i2];
  const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z, bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
  const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
  const float len = sqrtf(nx * nx + (ny * ny + nz * nz));
  normals[f] = (len > 0.f) ? make_float4(nx / len, ny / len, nz / len, 0.f) : make_float4(nx, ny, nz, 0.f);     // Eigen normalized()
  const unsigned e[3][2] = {{i0, i1}, {i1, i2}, {i2, i0}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {       // AddHalfEdge :466-487: key (smaller, larger) vertex, value (face, swapped)
    const bool swap = e[k][0] > e[k][1];
    const unsigned lo = swap ? e[k][1] : e[k][0], hi = swap ? e[k][0] : e[k][1];
    keys[3 * f + k] = ((unsigned long long)lo << 32) | hi;
    vals[3 * f + k] = ((unsigned)f << 1) | (swap ? 1u : 0u);
  }
}

// one thread per group of equal keys (sorted; stable: faces in increasing index like the reference's insertion order)
__global__ __launch_bounds__(kBlock) void k_filter_edges(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                         size_t n, const float4* __restrict__ v, const float4* __restrict__ normals,
                                                         MeshEdge* __restrict__ edges, unsigned* __restrict__ n_edges) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;                       // not the head of its group
  size_t j = i + 1;
  while (j < n && keys[j] == key) ++j;
  const size_t cnt = j - i;
  MeshEdge E;
  E.v1 = (unsigned)(key >> 32); E.v2 = (unsigned)key;
  E.f1 = vals[i] >> 1; E.f2 = 0xFFFFFFFFu; E.opposite = 0;
  bool keep = true;
  if (cnt >= 2) {
    const float kEpsilon = 1e-4f;
    float factor1 = (vals[i] & 1u) ? -1.f : 1.f, factor2 = (vals[i + 1] & 1u) ? -1.f : 1.f;
    const float4 s0 = v[E.v1], s1 = v[E.v2];
    const float ex = s1.x - s0.x, ey = s1.y - s0.y, ez = s1.z - s0.z;
    const float4 nA = normals[E.f1];
    const float n1x = nA.x * factor1, n1y = nA.y * factor1, n1z = nA.z * factor1;
    E.f2 = vals[i + 1] >> 1;
    const float4 nB = normals[E.f2];
    const float n2x = nB.x * factor2, n2y = nB.y * factor2, n2z = nB.z * factor2;
    E.opposite = (factor1 * factor2 > 0) ? 1 : 0;
    const float l1 = sqrtf(n1x * n1x + (n1y * n1y + n1z * n1z));
    const float bxx = n1x / l1, bxy = n1y / l1, bxz = n1z / l1;                                   // base_x = first_normal.normalized()
    float cyx = bxy * ez - bxz * ey, cyy = bxz * ex - bxx * ez, cyz = bxx * ey - bxy * ex;        // base_x x edge
    const float l2 = sqrtf(cyx * cyx + (cyy * cyy + cyz * cyz));
    cyx /= l2; cyy /= l2; cyz /= l2;
    float n1_2x = 1.f, n1_2y = 0.f;
    float n2_2x = bxx * n2x + (bxy * n2y + bxz * n2z), n2_2y = cyx * n2x + (cyy * n2y + cyz * n2z);
    if (n2_2x < 0 && fabsf(n2_2y) < kEpsilon) keep = false;                                        // coplanar faces: no edge
    else if (cnt > 2) {
      const float cross_n1n2 = n2_2y;
      for (size_t k = i + 2; k < j && keep; ++k) {
        const unsigned f3 = vals[k] >> 1;
        const float factor3 = (vals[k] & 1u) ? -1.f : 1.f;
        const float4 nC = normals[f3];
        const float cx = nC.x * factor3, cy = nC.y * factor3, cz = nC.z * factor3;
        const float n3x = bxx * cx + (bxy * cy + bxz * cz), n3y = cyx * cx + (cyy * cy + cyz * cz);
        const float cross_n1n3 = n1_2x * n3y - n1_2y * n3x;
        const float cross_n2n3 = n2_2x * n3y - n2_2y * n3x;
        const bool sign1 = cross_n1n3 * cross_n1n2 > 0;
        const bool sign2 = cross_n2n3 * cross_n1n2 < 0;
        if (sign1 && !sign2) { n2_2x = n3x; n2_2y = n3y; E.f2 = f3; factor2 = factor3; E.opposite = (factor1 * factor3 != 1) ? 1 : 0; }
        else if (sign2 && !sign1) { n1_2x = n3x; n1_2y = n3y; E.f1 = f3; factor1 = factor3; E.opposite = (factor3 * factor2 != 1) ? 1 : 0; }
        else if (!sign2 && !sign2) keep = false;              // [sic] the reference tests !sign2 twice (:631)
      }
    }
  }
  if (keep) edges[atomicAdd(n_edges, 1u)] = E;
}

// MaskOutOcclusionBoundaries: every silhouette (or boundary) edge that is visible splats -1 over the pixels it is not clearly
// behind.  The reference runs its edge loop under `omp parallel for` while reading the map it writes (:305-334), so its result
// depends on thread timing; here visibility and the per-pixel test read the UNMASKED depth map, which makes the result
// deterministic (a superset of any reference run's mask).
template <int M>
__global__ __launch_bounds__(kBlock) void k_mask_boundaries(const MeshEdge* __restrict__ edges, size_t n_edges, const float4* __restrict__ v,
                                                            const float4* __restrict__ normals, Pose P, float3 image_position,
                                                            CamLevel cam, float splat_radius, const float* __restrict__ depth_in,
                                                            float* __restrict__ depth_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_edges) return;
  const MeshEdge E = edges[i];
  const float4 p1 = v[E.v1], p2 = v[E.v2];
  if (E.f2 != 0xFFFFFFFFu) {
    const float tx = image_position.x - p1.x, ty = image_position.y - p1.y, tz = image_position.z - p1.z;
    const float4 na = normals[E.f1], nb = normals[E.f2];
    const bool face1 = (na.x * tx + (na.y * ty + na.z * tz)) > 0, face2 = (nb.x * tx + (nb.y * ty + nb.z * tz)) > 0;
    if (!((E.opposite && (face1 == face2)) || (face1 != face2 && !E.opposite))) return;
  }
  float a0, a1, a2, b0, b1, b2;
  rt(P, p1.x, p1.y, p1.z, a0, a1, a2);
  if (a2 <= 0) return;
  rt(P, p2.x, p2.y, p2.z, b0, b1, b2);
  if (b2 <= 0) return;
  const float d0 = b0 - a0, d1 = b1 - a1, d2 = b2 - a2;
  const int splat_point_count = 1 + min((int)(sqrtf(d0 * d0 + (d1 * d1 + d2 * d2)) / splat_radius + 0.5f), 150);
  const float kOcclusionDepthThreshold = 0.05f;
  for (int k = 0; k < splat_point_count; ++k) {
    const float factor = k / (splat_point_count - 1.0f);        // 0/0 = NaN for a single point, like the reference
    const float X = a0 + factor * d0, Y = a1 + factor * d1, Z = a2 + factor * d2;
    if (!(Z > 0)) continue;
    float px, py;
    const CamTheta th = cam_theta<M>(X / Z, Y / Z);
    cam_normalized_to_image<M>(cam, X / Z, Y / Z, th, px, py);
    const int ix = f2i(px + 0.5f), iy = f2i(py + 0.5f);
    if (!(px + 0.5f >= 0 && py + 0.5f >= 0 && ix >= 0 && iy >= 0 && ix < cam.width && iy < cam.height)) continue;
    if (!(depth_in[(size_t)iy * cam.width + ix] + kOcclusionDepthThreshold >= Z)) continue;
    float d[6];
    cam_image_deriv_by_world<M>(cam, X, Y, Z, th, d);
    const float rx = sqrtf(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])) * splat_radius;
    const float ry = sqrtf(d[3] * d[3] + (d[4] * d[4] + d[5] * d[5])) * splat_radius;
    const int min_x = max(0, d2i((double)((float)ix - rx) + 0.5)), min_y = max(0, d2i((double)((float)iy - ry) + 0.5));
    const int end_x = min(cam.width, d2i((double)((float)ix + rx) + 1.5)), end_y = min(cam.height, d2i((double)((float)iy + ry) + 1.5));
    for (int y = min_y; y < end_y; ++y)
      for (int x = min_x; x < end_x; ++x) {
        const float old_depth = depth_in[(size_t)y * cam.width + x];
        if (old_depth == 0 || old_depth + kOcclusionDepthThreshold > Z) depth_out[(size_t)y * cam.width + x] = -1.f;
      }
  }
}

// ==== f1: per-point radius range (ComputeMinMaxPointRadius, multi_scale_point_cloud.cc:126-180) ==================================
// CameraBaseImpl::InitializeUndistortionLookup (camera_base_impl.h:252-268): one Undistort per pixel
template <int M>
__global__ __launch_bounds__(kBlock) void k_undistort_lookup(CamLevel c, float2* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)c.width * c.height) return;
  const int x = (int)(i % c.width), y = (int)(i / c.width);
  const float dx = c.fx_inv * x + c.cx_inv, dy = c.fy_inv * y + c.cy_inv;
  float ux, uy;
  if constexpr (M == kPinhole || M == kSimplePinhole) { ux = dx; uy = dy; }
  else if constexpr (M == kFov) { cam_fov_undistort(c, dx, dy, ux, uy); }
  else {
    cam_iterative_undistort<M>(c, dx, dy, dx, dy, ux, uy);
    if constexpr (cam_is_fisheye(M)) {           // FisheyeBase::Undistort (camera_base_impl_fisheye.h:81-92)
      const float r = sqrtf(ux * ux + uy * uy);
      const float factor = (r < kFisheyeEpsilon) ? 1.f : ((r > (float)(M_PI / 2.f)) ? E3D_CAM_INF : e3d_tanf(r) / r);
      ux = factor * ux; uy = factor * uy;
    }
  }
  out[i] = make_float2(ux, uy);
}

// ImageToNormalized(Vector2f) through the lookup table (camera_base_impl.h:188-211); the row index y + 1 is clamped where the
// reference reads one row past the table with weight 0
__device__ __forceinline__ float2 image_to_normalized(const CamLevel& c, const float2* __restrict__ lookup, float px, float py) {
  float cx = px < c.width - 1.001f ? px : c.width - 1.001f;
  float cy = py < c.height - 1.00f ? py : c.height - 1.00f;
  if (!(cx > 0.f)) cx = 0.f;
  if (!(cy > 0.f)) cy = 0.f;
  const int ix = (int)cx, iy = (int)cy;
  const float fx = cx - (float)ix, fy = cy - (float)iy;
  const int iy1 = iy + 1 < c.height ? iy + 1 : c.height - 1;
  const float2 tl = lookup[(size_t)iy * c.width + ix], tr = lookup[(size_t)iy * c.width + ix + 1];
  const float2 bl = lookup[(size_t)iy1 * c.width + ix], br = lookup[(size_t)iy1 * c.width + ix + 1];
  return make_float2((1 - fy) * ((1 - fx) * tl.x + fx * tr.x) + fy * ((1 - fx) * bl.x + fx * br.x),
                     (1 - fy) * ((1 - fx) * tl.y + fx * tr.y) + fy * ((1 - fx) * bl.y + fx * br.y));
}

struct RadiusParams { int image_scale, min_image_scale; float occlusion_threshold, max_valid_intensity; double min_scaling_factor; };

// one image: observations without scale test (visibility_estimator.cc:297-364), then the radius that projects to half a pixel
// at the best image scale; each point is touched by one thread per image and the images are processed one after the other, so
// the running min / max need no atomics
template <int M>
__global__ __launch_bounds__(kBlock) void k_point_radius(const float4* __restrict__ pts, size_t n, Pose P, float4 quat /*w x y z*/,
                                                         CamLevel cam, CamLevel cam_min, const float2* __restrict__ lookup_min,
                                                         const unsigned char* __restrict__ img, const unsigned char* __restrict__ mask,
                                                         const unsigned char* __restrict__ cam_mask,
                                                         const float* __restrict__ occlusion, RadiusParams rp,
                                                         float* __restrict__ min_radius, float* __restrict__ max_radius) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  float X, Y, Z;
  rt(P, p.x, p.y, p.z, X, Y, Z);
  if (!(Z > 0.f)) return;
  float ixf, iyf;
  cam_normalized_to_image<M>(cam, X / Z, Y / Z, ixf, iyf);
  const int ix = f2i(ixf + 0.5f), iy = f2i(iyf + 0.5f);
  if (!(ixf + 0.5f >= 0 && iyf + 0.5f >= 0 && ix >= 0 && iy >= 0 && ix < cam.width && iy < cam.height)) return;
  if (!(occlusion[(size_t)iy * cam.width + ix] + rp.occlusion_threshold >= Z)) return;
  if (mask && mask[(size_t)iy * cam.width + ix] != 0) return;
  if (cam_mask && cam_mask[(size_t)iy * cam.width + ix] != 0) return;                          // visibility_estimator.cc:335-345
  if (img[(size_t)iy * cam.width + ix] > rp.max_valid_intensity) return;
  float returned_scale = rp.image_scale - 1e-6f;
  float ox = ixf, oy = iyf;
  if (returned_scale < 0.f) {
    returned_scale = 0.f;
    ox = 0.5f * (ox + 0.5f) - 0.5f;
    oy = 0.5f * (oy + 0.5f) - 0.5f;
  }
  const int smaller_scale = (int)returned_scale + 1;
  const float up = exp2f((float)(smaller_scale - rp.min_image_scale));     // exact power of two
  const float mx = up * (ox + 0.5f) - 0.5f, my = up * (oy + 0.5f) - 0.5f;
  // image.image_T_global * point as Sophus::SE3f (multi_scale_point_cloud.cc:149)
  const float vx = quat.y, vy = quat.z, vz = quat.w, w = quat.x;
  float ux = vy * p.z - vz * p.y, uy = vz * p.x - vx * p.z, uz = vx * p.y - vy * p.x;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = vy * uz - vz * uy, cy = vz * ux - vx * uz, cz = vx * uy - vy * ux;
  const float G0 = ((p.x + w * ux) + cx) + P.t[0], G1 = ((p.y + w * uy) + cy) + P.t[1], G2 = ((p.z + w * uz) + cz) + P.t[2];
  if (!(G2 > 0.f)) return;
  const float offx = (mx - 0.5f < 0) ? (mx + 0.5f) : (mx - 0.5f);
  float2 nxy;
  if constexpr (M == kFov) {      // FisheyeFOVCamera's own ImageToNormalized: Undistort(ImageToDistorted(p)) (camera_fisheye_fov.h:65-74)
    cam_fov_undistort(cam_min, cam_min.fx_inv * offx + cam_min.cx_inv, cam_min.fy_inv * my + cam_min.cy_inv, nxy.x, nxy.y);
  } else if constexpr (M == kPinhole || M == kSimplePinhole) {   // their own ImageToNormalized: ImageToDistorted (camera_pinhole.h:55-63)
    nxy = make_float2(cam_min.fx_inv * offx + cam_min.cx_inv, cam_min.fy_inv * my + cam_min.cy_inv);
  } else {
    nxy = image_to_normalized(cam_min, lookup_min, offx, my);
  }
  const float d0 = G0 - G2 * nxy.x, d1 = G1 - G2 * nxy.y, d2 = G2 - G2 * 1.f;
  const float point_radius = sqrtf(d0 * d0 + (d1 * d1 + d2 * d2));
  if (point_radius < min_radius[i]) min_radius[i] = point_radius;
  const float mr = (float)((double)point_radius / rp.min_scaling_factor);
  if (mr > max_radius[i]) max_radius[i] = mr;
}

// ==== a20 / a21: observation candidates =============================================================================================
struct ObsParams {
  float point_radius;
  int image_scale, border, current_image_scale, image_scale_count;
  float occlusion_threshold, max_valid_intensity;
  int check;     // 1: occlusion + masks + over-saturation (all points); 0: indexed list
};

// one candidate: true if it stays an observation (its position and scale written to ox / oy / os at k)
template <int M>
__device__ __forceinline__ bool obs_eval_one(const float4 p, const Pose& P, const Pyramid& Y, const float* __restrict__ occlusion,
                                             const ObsParams& q, size_t k, float* __restrict__ ox, float* __restrict__ oy,
                                             float* __restrict__ os) {
  float X, Yc, Z;
  rt(P, p.x, p.y, p.z, X, Yc, Z);
  if (!(Z > 0.f)) return false;
  int lvl = q.image_scale - Y.min_image_scale;
  if (lvl < 0) lvl = 0;
  const CamLevel cam = Y.cam[lvl];
  float ixf, iyf;
  cam_normalized_to_image<M>(cam, X / Z, Yc / Z, ixf, iyf);
  int ix = f2i(ixf + 0.5f), iy = f2i(iyf + 0.5f);
  if (!(ix >= 0 && iy >= 0 && ix < cam.width && iy < cam.height)) return false;
  if (q.check && !(occlusion[(size_t)iy * cam.width + ix] + q.occlusion_threshold >= Z)) return false;
  // CreateObservationIfScaleFits (visibility_estimator.cc:405-532)
  const float prx = X + q.point_radius, pry = Yc + 0.f, prz = Z + 0.f;
  float rxf, ryf;
  cam_normalized_to_image<M>(cam, prx / prz, pry / prz, rxf, ryf);
  const float dx = rxf - ixf, dy = ryf - iyf;
  const float radius_pixels = sqrtf(dx * dx + dy * dy);
  const float observation_scale = q.image_scale + e3d_log2f(2 * radius_pixels);
  const int lo = max(Y.min_image_scale, q.current_image_scale);
  if (!(observation_scale >= lo && f2i(observation_scale) < q.image_scale_count - 1)) return false;
  const int small_scale = f2i(observation_scale) + 1;
  int li = small_scale - Y.min_image_scale;
  if (li < 0) li = 0;
  if (li >= Y.n_levels) return false;
  const CamLevel ic = Y.cam[li];
  const float nx = cam.fx_inv * ixf + cam.cx_inv, ny = cam.fy_inv * iyf + cam.cy_inv;
  const float jx = ic.fx * nx + ic.cx, jy = ic.fy * ny + ic.cy;
  ix = f2i(jx + 0.5f); iy = f2i(jy + 0.5f);
  if (!(jx + 0.5f >= q.border && jy + 0.5f >= q.border && ix >= q.border && iy >= q.border && ix < ic.width - q.border &&
        iy < ic.height - q.border))
    return false;
  if (q.check) {
    const int pl = small_scale - Y.min_image_scale;
    if (Y.mask[pl] && Y.mask[pl][(size_t)iy * ic.width + ix] != 0) return false;
    if (Y.cam_mask[pl] && Y.cam_mask[pl][(size_t)iy * ic.width + ix] != 0) return false;          // visibility_estimator.cc:492-503
    if (Y.img[pl][(size_t)iy * ic.width + ix] > q.max_valid_intensity) return false;
  }
  ox[k] = jx; oy[k] = jy; os[k] = observation_scale;
  return true;
}

// `dropped` (re-projection of a list, where nearly every point stays an observation): number of candidates that did not -- zero
// means the list itself is the result and the compaction can be skipped.  block_counts (an all-points pass, which is always
// compacted): the block's number of kept candidates, i.e. the first stage of the compaction's scan (launch_match_scan's
// k_match_block_counts would read the 4 B per point back for it); block_d2: that scan's second sum, unused here, written as zero.
template <int M>
__global__ __launch_bounds__(kBlock) void k_obs_eval(const float4* __restrict__ pts, const unsigned* __restrict__ indices,
                                                     size_t count, Pose P, Pyramid Y, const float* __restrict__ occlusion,
                                                     ObsParams q, int* __restrict__ valid, float* __restrict__ ox,
                                                     float* __restrict__ oy, float* __restrict__ os,
                                                     unsigned* __restrict__ dropped, unsigned* __restrict__ block_counts,
                                                     double* __restrict__ block_d2) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool kept = false;
  if (k < count) {
    const size_t pi = indices ? indices[k] : k;
    kept = obs_eval_one<M>(pts[pi], P, Y, occlusion, q, k, ox, oy, os);
    valid[k] = kept ? (int)pi : -1;
    if (dropped && !kept) atomicAdd(dropped, 1u);
  }
  if (block_counts) {                                      // (block-uniform)
    __shared__ unsigned sc[kBlock / kWave];
    const unsigned long long b = __ballot(kept);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned c = 0;
      for (int w = 0; w < kBlock / kWave; ++w) c += sc[w];
      block_counts[blockIdx.x] = c; block_d2[blockIdx.x] = 0.0;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_obs_compact(const int* __restrict__ valid, const float* __restrict__ ox,
                                                        const float* __restrict__ oy, const float* __restrict__ os,
                                                        size_t count, const unsigned* __restrict__ block_offsets,
                                                        unsigned* __restrict__ o_idx, float* __restrict__ o_x,
                                                        float* __restrict__ o_y, float* __restrict__ o_s,
                                                        int* __restrict__ row_of_point) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int v = (k < count) ? valid[k] : -1;
  const bool f = v >= 0;
  const unsigned long long b = __ballot(f);
  __shared__ unsigned wbase[kBlock / kWave];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wbase[w] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned base = block_offsets[blockIdx.x];
  for (int j = 0; j < w; ++j) base += wbase[j];
  const size_t o = base + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
  // candidates = all points in point order (row_of_point != nullptr): candidate k IS point k, so the point -> observation row map that
  // finish_observations needs leaves here as one coalesced stream (it cleared the map and scattered the rows into it before)
  if (row_of_point && k < count) row_of_point[k] = f ? (int)o : -1;
  if (!f) return;
  o_idx[o] = (unsigned)v; o_x[o] = ox[k]; o_y[o] = oy[k]; o_s[o] = os[k];
}

// ==== f4: GroundTruthCreator visibility (src/exe/ground_truth_creator.cc:45-86, :152-189) =========================================
// mode 0: counts[i] += 1 for every scan point visible in the image; mode 1: ground-truth depth = min z over the visible
// points that were counted at least `min_count` times; mode 2: scan rendering (:175-187) -- the reference paints a square of
// 2 * radius + 1 pixels around every such point in point order, later points over earlier ones, so a pixel ends up with the colour of
// the LAST point whose square covers it: gt_depth[pixel] = max (point index + 1), 0 = untouched.
// "Visible" = in front of the camera, inside the image at the
// highest available resolution, not behind the occlusion depth (+ threshold), not under an eval-obs mask pixel.
template <int M>
__global__ __launch_bounds__(kBlock) void k_scan_visibility(const float4* __restrict__ pts, size_t n, Pose P, CamLevel cam,
                                                            const float* __restrict__ occlusion, float occlusion_threshold,
                                                            const unsigned char* __restrict__ mask, int excluded_flag, int mode,
                                                            int min_count, int* __restrict__ counts, unsigned* __restrict__ gt_depth,
                                                            int radius) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode != 0 && counts[i] < min_count) return;
  const float4 p = pts[i];
  float X, Yc, Z;
  rt(P, p.x, p.y, p.z, X, Yc, Z);
  if (!(Z > 0.f)) return;
  float ixf, iyf;
  cam_normalized_to_image<M>(cam, X / Z, Yc / Z, ixf, iyf);
  const int ix = f2i(ixf + 0.5f), iy = f2i(iyf + 0.5f);
  if (!(ix >= 0 && iy >= 0 && ix < cam.width && iy < cam.height)) return;
  const size_t px = (size_t)iy * cam.width + ix;
  if (!(occlusion[px] + occlusion_threshold >= Z)) return;
  if (mask && mask[px] == excluded_flag) return;
  if (mode == 0) counts[i] += 1;
  else if (mode == 1) atomicMin(&gt_depth[px], __float_as_uint(Z));       // Z > 0: the bit pattern orders like the value
  else {
    const int x0 = max(0, ix - radius), y0 = max(0, iy - radius), x1 = min(cam.width, ix + radius + 1), y1 = min(cam.height, iy + radius + 1);
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) atomicMax(&gt_depth[(size_t)y * cam.width + x], (unsigned)i + 1u);
  }
}

// ==== a22 ===============================================================================================================================
__global__ __launch_bounds__(kBlock) void k_obs_mark(const unsigned* __restrict__ o_idx, size_t n_obs, int* __restrict__ row_of_point) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_obs) row_of_point[o_idx[i]] = (int)i;
}
// KT > 0: the neighbour count at compile time (round 6) -- the K neighbour indices are requested together and the K row gathers after
// them (two round trips per observation instead of 2 K dependent ones; K = 5 is the reference's default, parameters.h:51).
template <int KT>
__global__ __launch_bounds__(kBlock) void k_obs_flags(const unsigned* __restrict__ o_idx, size_t n_obs,
                                                      const unsigned* __restrict__ nbr, int K,
                                                      const int* __restrict__ row_of_point, unsigned char* __restrict__ flags,
                                                      int* __restrict__ nrow) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_obs) return;
  const size_t p = o_idx[i];
  bool all = true;
  if constexpr (KT > 0) {
    unsigned nb[KT];
    int r[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) nb[k] = nbr[p * KT + k];
#pragma unroll
    for (int k = 0; k < KT; ++k) r[k] = row_of_point[nb[k]];
#pragma unroll
    for (int k = 0; k < KT; ++k) { nrow[i * KT + k] = r[k]; all = all && (r[k] >= 0); }
  } else {
    for (int k = 0; k < K; ++k) {
      const int r = row_of_point[nbr[p * K + k]];
      nrow[i * K + k] = r;            // the dependent gathers are paid once per observation update, not once per pass-2 launch
      all = all && (r >= 0);
    }
  }
  flags[i] = all ? 1 : 0;
}

// ==== a16: pass 1 =========================================================================================================================
// Row of one observation: [intensity, J_intrinsics(I), J_pose(6), 0-padding] as rows4(I) float4.
// NV = number of local unknowns: I + 6, or I + 12 for a non-reference rig image ([intrinsics, rig extrinsics, rig pose]).
__host__ __device__ constexpr int rows4(int NV) { return (NV + 4) / 4; }     // float4 per row: 1 + NV floats, padded

// A non-reference image of a rig frame (intrinsics_and_pose_optimizer.cc:653-670): image_T_rig.rotationMatrix() of its
// camera and the pose of the frame's reference image (Sophus::SE3f: unit quaternion w,x,y,z + translation).
struct RigLink { float R_image_rig[9]; float q_rig_global[4]; float t_rig_global[3]; };

template <int M, bool RIG>
__global__ __launch_bounds__(kBlock) void k_reg_pass1(const float4* __restrict__ pts, float point_radius, Pose P, Pyramid Y,
                                                      const unsigned* __restrict__ o_idx, const float* __restrict__ o_x,
                                                      const float* __restrict__ o_y, const float* __restrict__ o_s,
                                                      size_t n_obs, RigLink L, float4* __restrict__ rows) {
  constexpr int I = cam_param_count(M);
  constexpr int R4 = rows4(I + (RIG ? 12 : 6));
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_obs) return;
  const float4 p = pts[o_idx[i]];
  float T0, T1, T2;
  rt(P, p.x, p.y, p.z, T0, T1, T2);
  const float os = o_s[i], ox = o_x[i], oy = o_y[i];
  const int si = f2i(os);
  const int small_scale = si + 1;
  const int l1 = si - Y.min_image_scale, l0 = l1 + 1;
  float row[4 * R4];
#pragma unroll
  for (int c = 0; c < 4 * R4; ++c) row[c] = 0.f;
  float j0, j1, j2;
  trilinear_d(Y.img[l0], Y.cam[l0].width, Y.img[l1], Y.cam[l1].width, ox, oy, 1 - (os - (float)si), row[0], j0, j1, j2);
  j2 = -1 * j2;
  const float scale_factor = exp2f((float)(Y.min_image_scale - small_scale));     // exact power of two
  const float inv_scale_factor = 1.f / scale_factor;
  j0 *= scale_factor; j1 *= scale_factor;
  const float mx = inv_scale_factor * (ox + 0.5f) - 0.5f, my = inv_scale_factor * (oy + 0.5f) - 0.5f;
  const CamLevel cam = Y.cam[0];                                                   // min_image_scale camera
  const float To0 = T0 + point_radius;
  float offx, offy;
  // the point and its radius-offset twin: one atan each for the seven camera evaluations below (cam_theta)
  const CamTheta th_i = cam_theta<M>(T0 / T2, T1 / T2), th_o = cam_theta<M>(To0 / T2, T1 / T2);
  cam_normalized_to_image<M>(cam, To0 / T2, T1 / T2, th_o, offx, offy);
  const float rdx = offx - mx, rdy = offy - my;
  const float denom = fmaxf(1e-6f, 0.693147180559945f * (rdx * rdx + rdy * rdy));
  {
    float Pi[2 * I], Po[2 * I];
    cam_image_deriv_by_intrinsics<M>(cam, T0, T1, T2, th_i, Pi);
    cam_image_deriv_by_intrinsics<M>(cam, To0, T1, T2, th_o, Po);
#pragma unroll
    for (int c = 0; c < I; ++c) {
      const float scale_row = ((Po[c] - Pi[c]) * rdx + (Po[I + c] - Pi[I + c]) * rdy) / denom;
      row[1 + c] = j0 * Pi[c] + (j1 * Pi[I + c] + j2 * scale_row);
    }
  }
  float W[9], Wo[6], a[3];
  cam_image_deriv_by_world<M>(cam, T0, T1, T2, th_i, W);
  cam_image_deriv_by_world<M>(cam, To0, T1, T2, th_o, Wo);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    W[6 + c] = ((Wo[c] - W[c]) * rdx + (Wo[3 + c] - W[3 + c]) * rdy) / denom;
    a[c] = j0 * W[c] + (j1 * W[3 + c] + j2 * W[6 + c]);
  }
  // [I3 | 0 z -y ; -z 0 x ; y -x 0]
  const float C0[6] = {1, 0, 0, 0, T2, -1 * T1};
  const float C1[6] = {0, 1, 0, -1 * T2, 0, T0};
  const float C2[6] = {0, 0, 1, T1, -1 * T0, 0};
  if constexpr (!RIG) {
#pragma unroll
    for (int c = 0; c < 6; ++c) row[1 + I + c] = a[0] * C0[c] + (a[1] * C1[c] + a[2] * C2[c]);
  } else {
    // extrinsics block: the ordinary pose formula at the transformed point; pose block: through image_T_rig's rotation at
    // rig_T_global * point (intrinsics_and_pose_optimizer.cc:1107-1150)
#pragma unroll
    for (int c = 0; c < 6; ++c) row[1 + I + c] = a[0] * C0[c] + (a[1] * C1[c] + a[2] * C2[c]);
    float ar[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ar[c] = a[0] * L.R_image_rig[c] + (a[1] * L.R_image_rig[3 + c] + a[2] * L.R_image_rig[6 + c]);
    // Sophus SO3 action: p + w * uv + v x uv, uv = 2 (v x p); then + translation
    const float vx = L.q_rig_global[1], vy = L.q_rig_global[2], vz = L.q_rig_global[3], w = L.q_rig_global[0];
    float ux = vy * p.z - vz * p.y, uy = vz * p.x - vx * p.z, uz = vx * p.y - vy * p.x;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const float cx = vy * uz - vz * uy, cy = vz * ux - vx * uz, cz = vx * uy - vy * ux;
    const float G0 = ((p.x + w * ux) + cx) + L.t_rig_global[0];
    const float G1 = ((p.y + w * uy) + cy) + L.t_rig_global[1];
    const float G2 = ((p.z + w * uz) + cz) + L.t_rig_global[2];
    const float D0[6] = {1, 0, 0, 0, G2, -1 * G1};
    const float D1[6] = {0, 1, 0, -1 * G2, 0, G0};
    const float D2[6] = {0, 0, 1, G1, -1 * G0, 0};
#pragma unroll
    for (int c = 0; c < 6; ++c) row[1 + I + 6 + c] = ar[0] * D0[c] + (ar[1] * D1[c] + ar[2] * D2[c]);
  }
#pragma unroll
  for (int r = 0; r < R4; ++r) rows[R4 * i + r] = make_float4(row[4 * r], row[4 * r + 1], row[4 * r + 2], row[4 * r + 3]);
}

// Local system of one (image, point scale): V = I + 6 unknowns [intrinsics(I), pose(6)].  Slot layout of the reduction:
// [upper triangle of H row-major (V(V+1)/2)] [b (V)] [sum_fixed, sum_variable, count_fixed, count_variable].
__host__ __device__ constexpr int reg_h(int V) { return V * (V + 1) / 2; }
__host__ __device__ constexpr int reg_slot(int V) { return reg_h(V) + V + 4; }
__host__ __device__ constexpr int reg_row_start(int V, int r) { return r * V - r * (r - 1) / 2; }   // index of H(r, r)

struct RegWeights { int robust_type; float robust_param; float fixed_weight, var_weight; };

template <int V>
__device__ __forceinline__ void load_row(const float4* __restrict__ rows, size_t r, float* f) {
  constexpr int R4 = rows4(V);
#pragma unroll
  for (int q = 0; q < R4; ++q) {
    const float4 v = rows[R4 * r + q];
    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
  }
}

// ==== depth residuals (off by default: parameters.h:55, "not used in ETH3D pipeline" :165) ==================================================
// ComputePointIntensityAndJacobians, depth part (intrinsics_and_pose_optimizer.cc:1150-1214), images that are not dependent rig images
// (the reference aborts for those, :1199-1207).  Row of one observation: [residual, J_intrinsics(I), J_pose(6), 0-padding] as
// rows4(I + 6) float4.  The projection terms are those of k_reg_pass1 with the depth pyramid's interpolation derivative, scaled by
// -1 / depth^2, in place of the image's; the pose block loses (-1 / z^2) times the z row of d(camera point) / d(pose).
struct DepthPyramid { const float* map[kRegMaxLevels]; };

template <int M>
__global__ __launch_bounds__(kBlock) void k_reg_depth_rows(const float4* __restrict__ pts, float point_radius, Pose P, float4 quat /* w x y z */,
                                                           Pyramid Y, DepthPyramid D, const unsigned* __restrict__ o_idx,
                                                           const float* __restrict__ o_x, const float* __restrict__ o_y,
                                                           const float* __restrict__ o_s, size_t n_obs, float4* __restrict__ rows) {
  constexpr int I = cam_param_count(M);
  constexpr int R4 = rows4(I + 6);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_obs) return;
  const float4 p = pts[o_idx[i]];
  float T0, T1, T2;
  rt(P, p.x, p.y, p.z, T0, T1, T2);
  const float os = o_s[i], ox = o_x[i], oy = o_y[i];
  const int si = f2i(os);
  const int small_scale = si + 1;
  const int l1 = si - Y.min_image_scale, l0 = l1 + 1;
  float row[4 * R4];
#pragma unroll
  for (int c = 0; c < 4 * R4; ++c) row[c] = 0.f;
  float depth, j0, j1, j2;
  trilinear_d(D.map[l0], Y.cam[l0].width, D.map[l1], Y.cam[l1].width, ox, oy, 1 - (os - (float)si), depth, j0, j1, j2);
  j2 = -1 * j2;
  const float scale_factor = exp2f((float)(Y.min_image_scale - small_scale));     // exact power of two
  const float inv_scale_factor = 1.f / scale_factor;
  j0 *= scale_factor; j1 *= scale_factor;
  const float inv_depth = (depth != 0) ? (1.f / depth) : 0.f;
  // pp = image.image_T_global * point (Sophus SE3f): p + w uv + v x uv, uv = 2 (v x p), + translation
  const float vx = quat.y, vy = quat.z, vz = quat.w, w = quat.x;
  float ux = vy * p.z - vz * p.y, uy = vz * p.x - vx * p.z, uz = vx * p.y - vy * p.x;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cz = vx * uy - vy * ux;
  const float ppz = ((p.z + w * uz) + cz) + P.t[2];
  const float point_inv_depth = (ppz != 0.f) ? (1.f / ppz) : 0.f;
  row[0] = inv_depth - point_inv_depth;
  const float j_inv = -1 / (depth * depth);
  j0 = j_inv * j0; j1 = j_inv * j1; j2 = j_inv * j2;
  const float mx = inv_scale_factor * (ox + 0.5f) - 0.5f, my = inv_scale_factor * (oy + 0.5f) - 0.5f;
  const CamLevel cam = Y.cam[0];
  const float To0 = T0 + point_radius;
  float offx, offy;
  // the point and its radius-offset twin: one atan each for the seven camera evaluations below (cam_theta)
  const CamTheta th_i = cam_theta<M>(T0 / T2, T1 / T2), th_o = cam_theta<M>(To0 / T2, T1 / T2);
  cam_normalized_to_image<M>(cam, To0 / T2, T1 / T2, th_o, offx, offy);
  const float rdx = offx - mx, rdy = offy - my;
  const float denom = fmaxf(1e-6f, 0.693147180559945f * (rdx * rdx + rdy * rdy));
  {
    float Pi[2 * I], Po[2 * I];
    cam_image_deriv_by_intrinsics<M>(cam, T0, T1, T2, th_i, Pi);
    cam_image_deriv_by_intrinsics<M>(cam, To0, T1, T2, th_o, Po);
#pragma unroll
    for (int c = 0; c < I; ++c) {
      const float scale_row = ((Po[c] - Pi[c]) * rdx + (Po[I + c] - Pi[I + c]) * rdy) / denom;
      row[1 + c] = j0 * Pi[c] + (j1 * Pi[I + c] + j2 * scale_row);
    }
  }
  float W[9], Wo[6], a[3];
  cam_image_deriv_by_world<M>(cam, T0, T1, T2, th_i, W);
  cam_image_deriv_by_world<M>(cam, To0, T1, T2, th_o, Wo);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    W[6 + c] = ((Wo[c] - W[c]) * rdx + (Wo[3 + c] - W[3 + c]) * rdy) / denom;
    a[c] = j0 * W[c] + (j1 * W[3 + c] + j2 * W[6 + c]);
  }
  const float C0[6] = {1, 0, 0, 0, T2, -1 * T1};
  const float C1[6] = {0, 1, 0, -1 * T2, 0, T0};
  const float C2[6] = {0, 0, 1, T1, -1 * T0, 0};
  const float j_point_inv = -1 / (T2 * T2);
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float v = a[0] * C0[c] + (a[1] * C1[c] + a[2] * C2[c]);
    v -= j_point_inv * C2[c];
    row[1 + I + c] = v;
  }
#pragma unroll
  for (int r = 0; r < R4; ++r) rows[R4 * i + r] = make_float4(row[4 * r], row[4 * r + 1], row[4 * r + 2], row[4 * r + 3]);
}

// AccumulateOnHAndB over the depth rows (:747-757, :1219-1296): weight = robust weight * depth_residuals_weight in f32, entries
// ((weight J_r) J_c) in f32 and summed in f64 like there.  Rows [R0, R1) of the upper triangle per launch (per-thread accumulators,
// the same split as the per-thread colour kernel); WITH_B adds b, the robust residual sum and the count.  Same slot layout as
// k_reg_pass2 (the depth sum / count take the "fixed" places).
template <int V, int R0, int R1, bool WITH_B>
__global__ __launch_bounds__(kBlock) void k_reg_depth_acc(const float4* __restrict__ rows, size_t n_obs, int robust_type, float robust_param,
                                                          float depth_weight, double* __restrict__ partial) {
  constexpr int R4 = rows4(V);
  constexpr int NH = reg_row_start(V, R1) - reg_row_start(V, R0);
  constexpr int NL = NH + (WITH_B ? V + 4 : 0);
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_obs; i += (size_t)gridDim.x * blockDim.x) {
    float f[4 * R4];
    load_row<V>(rows, i, f);
    const float r = f[0];
    if constexpr (WITH_B) {
      acc[NH + V + 2] += 1.0;
      acc[NH + V] += (double)robust_residual(robust_type, robust_param, r);
    }
    float w = robust_weight(robust_type, robust_param, r);
    w *= depth_weight;
    if (w == 0) continue;
    int e = 0;
#pragma unroll
    for (int rr = R0; rr < R1; ++rr) {
      const float wj = w * f[1 + rr];
#pragma unroll
      for (int c = rr; c < V; ++c) { acc[e] += (double)(wj * f[1 + c]); ++e; }
    }
    if constexpr (WITH_B) {
      const float wr = w * r;
#pragma unroll
      for (int c = 0; c < V; ++c) acc[NH + c] += (double)(wr * f[1 + c]);
    }
  }
  __shared__ double s[kBlock / kWave][NL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NL) {
    double v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    const int t = threadIdx.x;
    const int dst = t < NH ? reg_row_start(V, R0) + t : reg_h(V) + (t - NH);
    partial[(size_t)blockIdx.x * reg_slot(V) + dst] = v;
  }
}

// CostCalculator, depth part (cost_calculator.cc:221-245): partial[2 b] = sum of the robust residuals, partial[2 b + 1] = count
__global__ __launch_bounds__(kBlock) void k_reg_depth_cost(const float4* __restrict__ pts, Pose P, float4 quat, Pyramid Y, DepthPyramid D,
                                                           const unsigned* __restrict__ o_idx, const float* __restrict__ o_x,
                                                           const float* __restrict__ o_y, const float* __restrict__ o_s, size_t n_obs,
                                                           int robust_type, float robust_param, double* __restrict__ partial) {
  double sum = 0.0, cnt = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_obs; i += (size_t)gridDim.x * blockDim.x) {
    const float4 p = pts[o_idx[i]];
    const float os = o_s[i];
    const int si = f2i(os);
    const int l1 = si - Y.min_image_scale, l0 = l1 + 1;
    const float depth = trilinear(D.map[l0], Y.cam[l0].width, D.map[l1], Y.cam[l1].width, o_x[i], o_y[i], 1 - (os - (float)si));
    const float inv_depth = (depth != 0) ? (1.f / depth) : 0.f;
    const float vx = quat.y, vy = quat.z, w = quat.x;
    float ux = vy * p.z - quat.w * p.y, uy = quat.w * p.x - vx * p.z, uz = vx * p.y - vy * p.x;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const float cz = vx * uy - vy * ux;
    const float ppz = ((p.z + w * uz) + cz) + P.t[2];
    const float point_inv_depth = (ppz != 0.f) ? (1.f / ppz) : 0.f;
    sum += (double)robust_residual(robust_type, robust_param, inv_depth - point_inv_depth);
    cnt += 1.0;
  }
  __shared__ double s[kBlock / kWave][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  sum = wave_sum(sum); cnt = wave_sum(cnt);
  if (lane == 0) { s[wv][0] = sum; s[wv][1] = cnt; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    partial[2 * (size_t)blockIdx.x + threadIdx.x] = v;
  }
}

// ==== a17 + a18: pass 2 =====================================================================================================================
// One launch accumulates the H rows [R0, R1) (and, if WITH_B, b and the residual sums): 69 f64 accumulators per thread for
// PINHOLE in a single launch; the larger systems (14 ... 24 unknowns) are split into 2 ... 5 launches of <= 75 accumulators
// each so that they stay in registers (the rows are re-read from L2, the arithmetic per launch is proportional to its rows).
template <int K_MAX, int V, int R0, int R1, bool WITH_B>
__global__ __launch_bounds__(kBlock) void k_reg_pass2(const float4* __restrict__ rows, const unsigned* __restrict__ o_idx,
                                                      const unsigned char* __restrict__ flags, size_t n_obs,
                                                      const int* __restrict__ nrow_of_obs, int K,
                                                      const float* __restrict__ fixed_desc, const float* __restrict__ var_desc,
                                                      const int* __restrict__ obs_counts, RegWeights wts,
                                                      double* __restrict__ partial) {
  constexpr int R4 = rows4(V);
  constexpr int NH = reg_row_start(V, R1) - reg_row_start(V, R0);
  constexpr int NL = NH + (WITH_B ? V + 4 : 0);
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_obs; i += (size_t)gridDim.x * blockDim.x) {
    if (!flags[i]) continue;
    const size_t p = o_idx[i];
    float fc[4 * R4];
    load_row<V>(rows, i, fc);
    int nrow[K_MAX];
    float In[K_MAX];
#pragma unroll
    for (int k = 0; k < K_MAX; ++k)
      if (k < K) { nrow[k] = nrow_of_obs[i * K + k]; In[k] = rows[R4 * (size_t)nrow[k]].x; }
    // both residual kinds first (they only need the intensities), then every neighbour row is read ONCE and used for the
    // fixed and the variable residual -- the same f32 products as the reference's two calls of
    // AccumulateHAndBAndResidualForColorObservation; only the order of the f64 additions differs (as it already does
    // between threads)
    float comp[2][K_MAX] = {};       // a skipped kind contributes weight 0 x 0
    float w[2] = {0.f, 0.f};
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      const float sw = kind == 0 ? wts.fixed_weight : wts.var_weight;
      if (!(sw > 0)) continue;
      if (kind == 1 && !(obs_counts[p] >= 2)) continue;
      const float* desc = kind == 0 ? fixed_desc : var_desc;
      float pr = 0.f;
#pragma unroll
      for (int k = 0; k < K_MAX; ++k)
        if (k < K) {
          const float image_descriptor = In[k] - fc[0];
          const float c = image_descriptor - desc[p * K + k];
          comp[kind][k] = c;
          pr += c * c;
        }
      pr = sqrtf(pr);
      if constexpr (WITH_B) {
        acc[NH + V + 2 + kind] += 1.0;
        acc[NH + V + kind] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
      }
      w[kind] = sw * robust_weight(wts.robust_type, wts.robust_param, pr);
    }
    if (w[0] != 0 || w[1] != 0) {
#pragma unroll
      for (int k = 0; k < K_MAX; ++k)
        if (k < K) {
          float fn[4 * R4], J[V];
          double Jd[V];
          load_row<V>(rows, (size_t)nrow[k], fn);
#pragma unroll
          for (int c = 0; c < V; ++c) { J[c] = fn[1 + c] - fc[1 + c]; Jd[c] = (double)J[c]; }
          // AccumulateOnHAndB (intrinsics_and_pose_optimizer.cc:1246-1247): H += ((weight * J^T) * J).cast<double>() for the fixed
          // and the variable residual.  weight * J[r] is formed in f32 like there; the two kinds' row factors are then added
          // (exactly, in f64) and multiplied with J[c] by ONE f64 FMA per entry: the second product is not rounded to f32
          // first, i.e. each term is at least as accurate as the reference's (difference <= 2^-24 relative per term, see
          // DESIGN.md section 10); one instruction per entry instead of six.
          int e = 0;
#pragma unroll
          for (int r = R0; r < R1; ++r) {
            const double wj = (double)(w[0] * J[r]) + (double)(w[1] * J[r]);
#pragma unroll
            for (int c = r; c < V; ++c) { acc[e] = __builtin_fma(wj, Jd[c], acc[e]); ++e; }
          }
          if constexpr (WITH_B) {
            const double wr = (double)(w[0] * comp[0][k]) + (double)(w[1] * comp[1][k]);
#pragma unroll
            for (int c = 0; c < V; ++c) acc[NH + c] = __builtin_fma(wr, Jd[c], acc[NH + c]);
          }
        }
    }
  }
  __shared__ double s[kBlock / kWave][NL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NL) {
    double v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    const int t = threadIdx.x;
    const int dst = t < NH ? reg_row_start(V, R0) + t : reg_h(V) + (t - NH);
    partial[(size_t)blockIdx.x * reg_slot(V) + dst] = v;
  }
}

// Pass 2 on the matrix cores: H = sum over (observation, neighbour) pairs of w J J^T is a rank-4-per-instruction update
// D(16 x 16) += A(16 x 4) B(4 x 16) in f64 (v_mfma_f64_16x16x4_f64).  A wave owns the whole V x V system as one (V <= 16) or three
// (V <= 32: blocks 00, 01, 11) 16 x 16 accumulator tiles -- 4 f64 per lane and tile instead of V (V + 1) / 2 accumulators per
// thread -- so every system size runs in ONE launch and the neighbour rows are gathered once.  Each lane still loads its own
// observation and the K neighbour rows; per neighbour slot the wave's 64 J vectors go through LDS (f32, row stride 36 floats) to
// reach the operand layout (lane l: element l & 15 of pair l >> 4).  A = J_i, B = (w_fixed + w_variable) J_j with the weight sum
// and the product formed in f64 (exact), so each term is at least as accurate as the reference's
// fl32(fl32(w J_i) J_j) per residual kind (difference <= 2^-23 relative).  b = sum (w r) J and the
// residual sums stay per-thread f64 accumulators (V + 4).
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int kP2Stride = 36;     // floats per pair in LDS: 16-byte aligned rows

template <int KT, int V>
__global__ __launch_bounds__(kBlock, 2) void k_reg_pass2_mfma(const float4* __restrict__ rows, const unsigned* __restrict__ o_idx,
                                                              const unsigned char* __restrict__ flags, size_t n_obs,
                                                              const int* __restrict__ nrow_of_obs,
                                                              const float* __restrict__ fixed_desc, const float* __restrict__ var_desc,
                                                              const int* __restrict__ obs_counts, RegWeights wts,
                                                              double* __restrict__ partial) {
  constexpr int R4 = rows4(V);
  constexpr bool kTwo = V > 16;
  constexpr int VP = kTwo ? 32 : 16;
  constexpr int K = KT;
  // the lower-right tile (unknowns 16 .. V-1 against themselves) holds only (V-16)(V-15)/2 useful entries: for V <= 18 these
  // three are cheaper as per-thread f64 FMAs than as a third MFMA per k-batch
  constexpr bool kTile11 = V > 18;
  // V == 18 (12 intrinsics + pose: THIN_PRISM_FISHEYE, the model of the DSLR images) fits ONE tile: the B operand's columns 0, 1
  // carry unknowns 16, 17 instead, so the tile yields H[0..15][2..17]; the six entries this leaves out -- (0,0) (0,1) (1,1) and
  // (16,16) (16,17) (17,17) -- are per-thread FMAs
  constexpr bool kFold = V == 18;
  constexpr int NX = kFold ? 6 : ((kTwo && !kTile11) ? (V - 16) * (V - 15) / 2 : 0);
  static_assert(V <= 32, "local system too large for two 16-wide tiles");
  __shared__ __attribute__((aligned(16))) float s_j[kBlock / kWave][kWave * kP2Stride];
  __shared__ double s_w[kBlock / kWave][kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const Jl = s_j[wv];
  double* const Wl = s_w[wv];
  const int e = lane & 15, pq = lane >> 4;
  // independent accumulator copies per tile (summed at the end): back-to-back MFMAs on ONE accumulator serialise on its latency
  constexpr int NA = 2;
  d4_t acc00[NA], acc01[NA], acc11[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) { acc00[a] = d4_t{0, 0, 0, 0}; acc01[a] = d4_t{0, 0, 0, 0}; acc11[a] = d4_t{0, 0, 0, 0}; }
  double bacc[V + 4];
  double xacc[NX > 0 ? NX : 1];
#pragma unroll
  for (int i = 0; i < V + 4; ++i) bacc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (NX > 0 ? NX : 1); ++i) xacc[i] = 0.0;
  const size_t n_chunks = (n_obs + kWave - 1) / kWave;
  const size_t wave_id = (size_t)blockIdx.x * (kBlock / kWave) + wv, n_waves = (size_t)gridDim.x * (kBlock / kWave);
  for (size_t ch = wave_id; ch < n_chunks; ch += n_waves) {
    const size_t i = ch * kWave + lane;
    const bool on = i < n_obs && flags[i];
    float fc[4 * R4];
    float fnk[KT][4 * R4];           // all neighbour rows are requested up front: their latencies overlap
    float comp[2][KT] = {};
    float w[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4 * R4; ++q) fc[q] = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int q = 0; q < 4 * R4; ++q) fnk[k][q] = 0.f;
    if (on) {
      const size_t p = o_idx[i];
      int nrow[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) nrow[k] = nrow_of_obs[i * K + k];
      load_row<V>(rows, i, fc);
#pragma unroll
      for (int k = 0; k < KT; ++k) load_row<V>(rows, (size_t)nrow[k], fnk[k]);
#pragma unroll
      for (int kind = 0; kind < 2; ++kind) {
        const float sw = kind == 0 ? wts.fixed_weight : wts.var_weight;
        if (!(sw > 0)) continue;
        if (kind == 1 && !(obs_counts[p] >= 2)) continue;
        const float* desc = kind == 0 ? fixed_desc : var_desc;
        float pr = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float image_descriptor = fnk[k][0] - fc[0];
          const float c = image_descriptor - desc[p * K + k];
          comp[kind][k] = c;
          pr += c * c;
        }
        pr = sqrtf(pr);
        bacc[V + 2 + kind] += 1.0;
        bacc[V + kind] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
        w[kind] = sw * robust_weight(wts.robust_type, wts.robust_param, pr);
      }
    }
    const bool act = on && (w[0] != 0 || w[1] != 0);
    const double wsum = act ? (double)w[0] + (double)w[1] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      float J[VP];
#pragma unroll
      for (int c = 0; c < VP; ++c) J[c] = 0.f;
      if (act) {
        const double wr = (double)(w[0] * comp[0][k]) + (double)(w[1] * comp[1][k]);
#pragma unroll
        for (int c = 0; c < V; ++c) {
          J[c] = fnk[k][1 + c] - fc[1 + c];
          bacc[c] = __builtin_fma(wr, (double)J[c], bacc[c]);
        }
        if constexpr (NX > 0) {
          int x = 0;
#pragma unroll
          for (int r = 16; r < V; ++r) {
            const double wj = wsum * (double)J[r];
#pragma unroll
            for (int c = r; c < V; ++c) { xacc[x] = __builtin_fma((double)J[c], wj, xacc[x]); ++x; }
          }
          if constexpr (kFold) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const double wj = wsum * (double)J[r];
#pragma unroll
              for (int c = r; c < 2; ++c) { xacc[x] = __builtin_fma((double)J[c], wj, xacc[x]); ++x; }
            }
          }
        }
      }
      float4* const dst = reinterpret_cast<float4*>(Jl + lane * kP2Stride);
#pragma unroll
      for (int c = 0; c < VP / 4; ++c) dst[c] = make_float4(J[4 * c], J[4 * c + 1], J[4 * c + 2], J[4 * c + 3]);
      Wl[lane] = wsum;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
      for (int q = 0; q < kWave / 4; ++q) {
        const int pair = 4 * q + pq;
        const double a0 = (double)Jl[pair * kP2Stride + e];
        const double ws = Wl[pair];
        const double b0 = kFold ? ws * (double)Jl[pair * kP2Stride + (e < 2 ? 16 + e : e)] : ws * a0;
        acc00[q % NA] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc00[q % NA], 0, 0, 0);
        if constexpr (kTwo && !kFold) {
          const double a1 = (double)Jl[pair * kP2Stride + 16 + e];
          const double b1 = ws * a1;
          acc01[q % NA] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc01[q % NA], 0, 0, 0);
          if constexpr (kTile11) acc11[q % NA] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc11[q % NA], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  // per-wave partial: tile entry (row = (lane >> 4) + 4 r, col = lane & 15) -> upper-triangle slot
  double* const out = partial + wave_id * reg_slot(V);
#pragma unroll
  for (int a = 1; a < NA; ++a) { acc00[0] += acc00[a]; acc01[0] += acc01[a]; acc11[0] += acc11[a]; }      // fixed order
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = pq + 4 * r, col = e;
    if constexpr (kFold) {
      if (col >= 2) { if (row <= col) out[reg_row_start(V, row) + (col - row)] = acc00[0][r]; }
      else out[reg_row_start(V, row) + (16 + col - row)] = acc00[0][r];
    } else if (row <= col && col < V) out[reg_row_start(V, row) + (col - row)] = acc00[0][r];
    if constexpr (kTwo && !kFold) {
      if (row < V && 16 + col < V) out[reg_row_start(V, row) + (16 + col - row)] = acc01[0][r];
      if (kTile11 && row <= col && 16 + col < V) out[reg_row_start(V, 16 + row) + (col - row)] = acc11[0][r];
    }
  }
  if constexpr (NX > 0) {
    int x = 0;
#pragma unroll
    for (int r = 16; r < V; ++r)
#pragma unroll
      for (int c = r; c < V; ++c) {
        const double v = wave_sum(xacc[x]); ++x;
        if (lane == 0) out[reg_row_start(V, r) + (c - r)] = v;
      }
    if constexpr (kFold) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = r; c < 2; ++c) {
          const double v = wave_sum(xacc[x]); ++x;
          if (lane == 0) out[reg_row_start(V, r) + (c - r)] = v;
        }
    }
  }
#pragma unroll
  for (int c = 0; c < V + 4; ++c) {
    const double v = wave_sum(bacc[c]);
    if (lane == 0) out[reg_h(V) + c] = v;
  }
}

// Chunk ranges per XCD: workgroup b runs on XCD b % 8 (observed placement, not a contract -- a different placement is slower, not
// wrong), each XCD has its own 4 MB L2, and the rows a chunk gathers are those of nearby chunks (the neighbours of a point are close in
// the observation order, or one scan line away).  Handing every XCD one contiguous eighth of the chunks keeps a row in ONE L2 instead
// of fetching it into several: chunk_first / chunk_last / chunk_step of a wave.
struct ChunkRange { size_t first, last, step; };
__device__ __forceinline__ ChunkRange xcd_chunk_range(size_t n_chunks, int waves_per_block, int wave_in_block) {
  constexpr unsigned kXcds = 8;
  if (gridDim.x % kXcds != 0 || n_chunks < 64 * kXcds)
    return ChunkRange{(size_t)blockIdx.x * waves_per_block + wave_in_block, n_chunks, (size_t)gridDim.x * waves_per_block};
  const size_t xcd = blockIdx.x % kXcds, local = blockIdx.x / kXcds, per = gridDim.x / kXcds;
  const size_t c0 = n_chunks * xcd / kXcds, c1 = n_chunks * (xcd + 1) / kXcds;
  return ChunkRange{c0 + local * waves_per_block + wave_in_block, c1, per * waves_per_block};
}

// The same update on the f32 matrix instruction (v_mfma_f32_16x16x4_f32: 32 cycles per instruction and SIMD against the ~97 measured for
// the f64 form).  A = J_i and B = fl32((w_fixed + w_variable) J_j) are f32 -- the reference forms weight * J in
// f32 as well (intrinsics_and_pose_optimizer.cc:1246) -- and the instruction is an f32 fma chain over its 4 pairs.  The chain is kept
// short: after the 16 instructions of one neighbour slot (64 pairs per accumulator entry, two independent accumulators of 32) the f32
// tile is added into an f64 master tile and cleared, so H is a sum of f64 additions of 32-pair f32 partial sums instead of an f64 sum of
// single f32 products.  Deviation from the f64 kernel: ~1e-9 of the entry scale (tests/test_gpu_reg.py::test_pass2_variants_agree);
// b, the residual sums and the entries outside the tile stay per-thread f64 as in k_reg_pass2_mfma.  Pair order inside a slot:
// pair(q, pq) = 4 pq + (q & 3) + 16 (q >> 2), which puts the two 16-lane groups of one LDS pass 16 banks apart (row stride 36 floats).
typedef float f4_t __attribute__((ext_vector_type(4)));

template <int KT, int V>
__global__ __launch_bounds__(kBlock, 2) void k_reg_pass2_mfma32(const float4* __restrict__ rows, const unsigned* __restrict__ o_idx,
                                                                const unsigned char* __restrict__ flags, size_t n_obs,
                                                                const int* __restrict__ nrow_of_obs,
                                                                const float* __restrict__ fixed_desc, const float* __restrict__ var_desc,
                                                                const int* __restrict__ obs_counts, RegWeights wts,
                                                                double* __restrict__ partial) {
  constexpr int R4 = rows4(V);
  constexpr bool kTwo = V > 16;
  constexpr int VP = kTwo ? 32 : 16;
  constexpr int K = KT;
  constexpr bool kTile11 = V > 18;
  constexpr bool kFold = V == 18;
  constexpr int NX = kFold ? 6 : ((kTwo && !kTile11) ? (V - 16) * (V - 15) / 2 : 0);
  static_assert(V <= 32 && VP < kP2Stride, "local system too large for two 16-wide tiles");
  __shared__ __attribute__((aligned(16))) float s_j[kBlock / kWave][kWave * kP2Stride];      // slot VP of a row: the pair's weight
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const Jl = s_j[wv];
  const int e = lane & 15, pq = lane >> 4;
  constexpr int NA = 2;
  f4_t t00[NA], t01[NA], t11[NA];
  d4_t m00 = d4_t{0, 0, 0, 0}, m01 = d4_t{0, 0, 0, 0}, m11 = d4_t{0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < NA; ++a) { t00[a] = f4_t{0, 0, 0, 0}; t01[a] = f4_t{0, 0, 0, 0}; t11[a] = f4_t{0, 0, 0, 0}; }
  double bacc[V + 4];
  double xacc[NX > 0 ? NX : 1];
#pragma unroll
  for (int i = 0; i < V + 4; ++i) bacc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (NX > 0 ? NX : 1); ++i) xacc[i] = 0.0;
  const size_t n_chunks = (n_obs + kWave - 1) / kWave;
  const size_t wave_id = (size_t)blockIdx.x * (kBlock / kWave) + wv;
  const ChunkRange cr = xcd_chunk_range(n_chunks, kBlock / kWave, wv);
  for (size_t ch = cr.first; ch < cr.last; ch += cr.step) {
    const size_t i = ch * kWave + lane;
    const bool on = i < n_obs && flags[i];
    float fc[4 * R4];
    float fnk[KT][4 * R4];
    float comp[2][KT] = {};
    float w[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4 * R4; ++q) fc[q] = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int q = 0; q < 4 * R4; ++q) fnk[k][q] = 0.f;
    if (on) {
      const size_t p = o_idx[i];
      int nrow[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) nrow[k] = nrow_of_obs[i * K + k];
      load_row<V>(rows, i, fc);
#pragma unroll
      for (int k = 0; k < KT; ++k) load_row<V>(rows, (size_t)nrow[k], fnk[k]);
#pragma unroll
      for (int kind = 0; kind < 2; ++kind) {
        const float sw = kind == 0 ? wts.fixed_weight : wts.var_weight;
        if (!(sw > 0)) continue;
        if (kind == 1 && !(obs_counts[p] >= 2)) continue;
        const float* desc = kind == 0 ? fixed_desc : var_desc;
        float pr = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float image_descriptor = fnk[k][0] - fc[0];
          const float c = image_descriptor - desc[p * K + k];
          comp[kind][k] = c;
          pr += c * c;
        }
        pr = sqrtf(pr);
        bacc[V + 2 + kind] += 1.0;
        bacc[V + kind] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
        w[kind] = sw * robust_weight(wts.robust_type, wts.robust_param, pr);
      }
    }
    const bool act = on && (w[0] != 0 || w[1] != 0);
    const double wsum = act ? (double)w[0] + (double)w[1] : 0.0;
    const float wsum32 = act ? w[0] + w[1] : 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      float J[VP];
#pragma unroll
      for (int c = 0; c < VP; ++c) J[c] = 0.f;
      if (act) {
        const double wr = (double)(w[0] * comp[0][k]) + (double)(w[1] * comp[1][k]);
#pragma unroll
        for (int c = 0; c < V; ++c) {
          J[c] = fnk[k][1 + c] - fc[1 + c];
          bacc[c] = __builtin_fma(wr, (double)J[c], bacc[c]);
        }
        if constexpr (NX > 0) {
          int x = 0;
#pragma unroll
          for (int r = 16; r < V; ++r) {
            const double wj = wsum * (double)J[r];
#pragma unroll
            for (int c = r; c < V; ++c) { xacc[x] = __builtin_fma((double)J[c], wj, xacc[x]); ++x; }
          }
          if constexpr (kFold) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const double wj = wsum * (double)J[r];
#pragma unroll
              for (int c = r; c < 2; ++c) { xacc[x] = __builtin_fma((double)J[c], wj, xacc[x]); ++x; }
            }
          }
        }
      }
      float4* const dst = reinterpret_cast<float4*>(Jl + lane * kP2Stride);
#pragma unroll
      for (int c = 0; c < VP / 4; ++c) dst[c] = make_float4(J[4 * c], J[4 * c + 1], J[4 * c + 2], J[4 * c + 3]);
      Jl[lane * kP2Stride + VP] = wsum32;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < kWave / 4; ++q) {
        const float* const row = Jl + (4 * pq + (q & 3) + 16 * (q >> 2)) * kP2Stride;
        const float a0 = row[e];
        const float ws = row[VP];
        float bj = a0;
        if constexpr (kFold) bj = row[e < 2 ? 16 + e : e];
        const float b0 = ws * bj;
        t00[q % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, t00[q % NA], 0, 0, 0);
        if constexpr (kTwo && !kFold) {
          const float a1 = row[16 + e];
          const float b1 = ws * a1;
          t01[q % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, t01[q % NA], 0, 0, 0);
          if constexpr (kTile11) t11[q % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, t11[q % NA], 0, 0, 0);
        }
      }
      // end of the f32 chains: into the f64 master tiles (fixed order)
#pragma unroll
      for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          m00[r] += (double)t00[a][r];
          if constexpr (kTwo && !kFold) { m01[r] += (double)t01[a][r]; if constexpr (kTile11) m11[r] += (double)t11[a][r]; }
        }
        t00[a] = f4_t{0, 0, 0, 0}; t01[a] = f4_t{0, 0, 0, 0}; t11[a] = f4_t{0, 0, 0, 0};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  // per-wave partial: f32 tile entry (row = 4 (lane >> 4) + r, col = lane & 15) -> upper-triangle slot
  double* const out = partial + wave_id * reg_slot(V);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * pq + r, col = e;
    if constexpr (kFold) {
      if (col >= 2) { if (row <= col) out[reg_row_start(V, row) + (col - row)] = m00[r]; }
      else out[reg_row_start(V, row) + (16 + col - row)] = m00[r];
    } else if (row <= col && col < V) out[reg_row_start(V, row) + (col - row)] = m00[r];
    if constexpr (kTwo && !kFold) {
      if (row < V && 16 + col < V) out[reg_row_start(V, row) + (16 + col - row)] = m01[r];
      if (kTile11 && row <= col && 16 + col < V) out[reg_row_start(V, 16 + row) + (col - row)] = m11[r];
    }
  }
  if constexpr (NX > 0) {
    int x = 0;
#pragma unroll
    for (int r = 16; r < V; ++r)
#pragma unroll
      for (int c = r; c < V; ++c) {
        const double v = wave_sum(xacc[x]); ++x;
        if (lane == 0) out[reg_row_start(V, r) + (c - r)] = v;
      }
    if constexpr (kFold) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = r; c < 2; ++c) {
          const double v = wave_sum(xacc[x]); ++x;
          if (lane == 0) out[reg_row_start(V, r) + (c - r)] = v;
        }
    }
  }
#pragma unroll
  for (int c = 0; c < V + 4; ++c) {
    const double v = wave_sum(bacc[c]);
    if (lane == 0) out[reg_h(V) + c] = v;
  }
}

// Single-tile systems (V <= 16, and V == 18 through the folded tile) with the per-pair arithmetic in packed f32: a row pair is
// (n - c) for ten float2 at once (element 0 = the intensity difference), the B operand w (n - c) likewise, and b = sum (w r) (n - c) and
// the six entries outside the folded tile are f32 fma chains over kChainChunks chunks (5 pairs each) that end in per-lane f64 sums.
// LDS row of a pair: the 20 floats of (n - c), then the 20 floats of the B operand with its elements 1, 2 replaced by the ones of
// unknowns 16, 17 when folded, so that lane e reads A and B with one ds_read2_b32 (offsets 1 + e and 21 + e); row stride 44 floats
// (conflict-free b128 writes and, with the pair order of k_reg_pass2_mfma32, conflict-free reads).
typedef float f2_t __attribute__((ext_vector_type(2)));
constexpr int kT32Stride = 44;
constexpr int kChainChunks = 4;

// 32 per-lane values -> lane l (and l + 32) receives the wave's sum of value l & 31: a butterfly that halves the number of live values
// with every exchange (31 + 1 exchanges instead of 32 x 6)
__device__ __forceinline__ float wave_reduce_scatter32(float (&v)[32], int lane) {
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int o = 1 << s;
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < (16 >> s); ++i) {
      const float a = v[2 * i], b = v[2 * i + 1];
      const float send = up ? a : b, keep = up ? b : a;
      v[i] = keep + __shfl_xor(send, o, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 32, 64);
}

template <int KT, int V>
__global__ __launch_bounds__(kBlock, 2) void k_reg_pass2_tile32(const float4* __restrict__ rows, const unsigned* __restrict__ o_idx,
                                                                const unsigned char* __restrict__ flags, size_t n_obs,
                                                                const int* __restrict__ nrow_of_obs,
                                                                const float* __restrict__ fixed_desc, const float* __restrict__ var_desc,
                                                                const int* __restrict__ obs_counts, RegWeights wts,
                                                                double* __restrict__ partial) {
  static_assert(V <= 16 || V == 18, "single 16 x 16 tile");
  constexpr int R4 = rows4(V);
  constexpr int K = KT;
  constexpr bool kFold = V == 18;
  constexpr int NX = kFold ? 6 : 0;
  __shared__ __attribute__((aligned(16))) float s_j[kBlock / kWave][kWave * kT32Stride];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const Jl = s_j[wv];
  const int e = lane & 15, pq = lane >> 4;
  constexpr int NA = 2;
  f4_t t00[NA];
  d4_t m00 = d4_t{0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < NA; ++a) t00[a] = f4_t{0, 0, 0, 0};
  f2_t pb[10];                       // f32 chains of b, paired like the row elements (pb[0].x and pb[9].y are not unknowns)
  float px[NX > 0 ? NX : 1];
  double racc[4] = {0.0, 0.0, 0.0, 0.0};       // residual sums and counts of the two kinds
  double bm = 0.0;                              // lane l: f64 sum of chain value l & 31 (0 .. 19: b by row element, 20 .. 25: px)
#pragma unroll
  for (int i = 0; i < 10; ++i) pb[i] = f2_t{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (NX > 0 ? NX : 1); ++i) px[i] = 0.f;
  auto end_chains = [&]() {
    float v[32];
#pragma unroll
    for (int j = 0; j < 10; ++j) { v[2 * j] = pb[j].x; v[2 * j + 1] = pb[j].y; pb[j] = f2_t{0.f, 0.f}; }
#pragma unroll
    for (int x = 0; x < 12; ++x) v[20 + x] = 0.f;
#pragma unroll
    for (int x = 0; x < NX; ++x) { v[20 + x] = px[x]; px[x] = 0.f; }
    bm += (double)wave_reduce_scatter32(v, lane);
  };
  const size_t n_chunks = (n_obs + kWave - 1) / kWave;
  const size_t wave_id = (size_t)blockIdx.x * (kBlock / kWave) + wv;
  int chain = 0;
  const ChunkRange cr = xcd_chunk_range(n_chunks, kBlock / kWave, wv);
  for (size_t ch = cr.first; ch < cr.last; ch += cr.step) {
    const size_t i = ch * kWave + lane;
    const bool on = i < n_obs && flags[i];
    float4 rc[5], rn[KT][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      rc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < KT; ++k) rn[k][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float comp[2][KT] = {};
    float w[2] = {0.f, 0.f};
    if (on) {
      const size_t p = o_idx[i];
      int nrow[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) nrow[k] = nrow_of_obs[i * K + k];
#pragma unroll
      for (int q = 0; q < R4; ++q) rc[q] = rows[R4 * i + q];
#pragma unroll
      for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int q = 0; q < R4; ++q) rn[k][q] = rows[R4 * (size_t)nrow[k] + q];
#pragma unroll
      for (int kind = 0; kind < 2; ++kind) {
        const float sw = kind == 0 ? wts.fixed_weight : wts.var_weight;
        if (!(sw > 0)) continue;
        if (kind == 1 && !(obs_counts[p] >= 2)) continue;
        const float* desc = kind == 0 ? fixed_desc : var_desc;
        float pr = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float image_descriptor = rn[k][0].x - rc[0].x;
          const float c = image_descriptor - desc[p * K + k];
          comp[kind][k] = c;
          pr += c * c;
        }
        pr = sqrtf(pr);
        racc[2 + kind] += 1.0;
        racc[kind] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
        w[kind] = sw * robust_weight(wts.robust_type, wts.robust_param, pr);
      }
    }
    const float ws = w[0] + w[1];
    const f2_t ws2 = f2_t{ws, ws};
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float wr = w[0] * comp[0][k] + w[1] * comp[1][k];
      const f2_t wr2 = f2_t{wr, wr};
      f2_t D[10], B[10];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        D[2 * q] = f2_t{rn[k][q].x, rn[k][q].y} - f2_t{rc[q].x, rc[q].y};
        D[2 * q + 1] = f2_t{rn[k][q].z, rn[k][q].w} - f2_t{rc[q].z, rc[q].w};
      }
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        pb[j] = __builtin_elementwise_fma(wr2, D[j], pb[j]);
        B[j] = ws2 * D[j];
      }
      if constexpr (kFold) {
        // unknown c = row element 1 + c: unknowns 0, 1 = D[0].y, D[1].x; 16, 17 = D[8].y, D[9].x
        px[0] = fmaf(B[8].y, D[8].y, px[0]); px[1] = fmaf(B[8].y, D[9].x, px[1]); px[2] = fmaf(B[9].x, D[9].x, px[2]);
        px[3] = fmaf(B[0].y, D[0].y, px[3]); px[4] = fmaf(B[0].y, D[1].x, px[4]); px[5] = fmaf(B[1].x, D[1].x, px[5]);
        B[0].y = B[8].y; B[1].x = B[9].x;          // columns 0, 1 of the tile carry unknowns 16, 17
      }
      float4* const dst = reinterpret_cast<float4*>(Jl + lane * kT32Stride);
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        dst[q] = make_float4(D[2 * q].x, D[2 * q].y, D[2 * q + 1].x, D[2 * q + 1].y);
        dst[5 + q] = make_float4(B[2 * q].x, B[2 * q].y, B[2 * q + 1].x, B[2 * q + 1].y);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // all 16 operand pairs first (one LDS latency for the slot instead of one per instruction), then the instructions back to back
      float opa[kWave / 4], opb[kWave / 4];
#pragma unroll
      for (int q = 0; q < kWave / 4; ++q) {
        const float* const row = Jl + (4 * pq + (q & 3) + 16 * (q >> 2)) * kT32Stride + 1 + e;
        opa[q] = row[0]; opb[q] = row[20];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < kWave / 4; ++q) t00[q % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(opa[q], opb[q], t00[q % NA], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) m00[r] += (double)t00[a][r];
        t00[a] = f4_t{0, 0, 0, 0};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (++chain == kChainChunks) { chain = 0; end_chains(); }
  }
  end_chains();
  // per-wave partial: f32 tile entry (row = 4 (lane >> 4) + r, col = lane & 15) -> upper-triangle slot
  double* const out = partial + wave_id * reg_slot(V);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * pq + r, col = e;
    if constexpr (kFold) {
      if (col >= 2) { if (row <= col) out[reg_row_start(V, row) + (col - row)] = m00[r]; }
      else out[reg_row_start(V, row) + (16 + col - row)] = m00[r];
    } else if (row <= col && col < V) out[reg_row_start(V, row) + (col - row)] = m00[r];
  }
  if (lane >= 1 && lane <= V) out[reg_h(V) + (lane - 1)] = bm;
  if constexpr (kFold) {
    if (lane >= 20 && lane < 26) {
      const int x = lane - 20;
      const int xr = x < 2 ? 16 : (x == 2 ? 17 : (x < 5 ? 0 : 1)), xc = x == 0 ? 16 : (x < 3 ? 17 : (x == 3 ? 0 : 1));
      out[reg_row_start(V, xr) + (xc - xr)] = bm;
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double v = wave_sum(racc[c]);
    if (lane == 0) out[reg_h(V) + V + c] = v;
  }
}

// InitCutoff (camera_base_impl.h:410-463) of the non-fisheye model M: one WAVE per border test point.  The 10 x 10 seeds of
// UndistortFromInside (:278-328) are independent Gauss-Newton runs -- two per lane -- whose results go to LDS; lane 0 then
// replays the reference's order-dependent best / second-best bookkeeping over them in seed order, so the outcome is the
// sequential one bit for bit.  min_candidate / max_candidate are order-free max / min reductions (non-negative floats: their
// bit patterns are ordered).
template <int M>
__global__ __launch_bounds__(kBlock) void k_cam_cutoff(CamLevel c, unsigned* __restrict__ out /* [max r2 bits, min second r2 bits] */) {
  __shared__ float s_rx[kBlock / kWave][100], s_ry[kBlock / kWave][100];
  __shared__ unsigned char s_ok[kBlock / kWave][100];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * (kBlock / kWave) + w;
  const int total = 2 * c.width + 2 * c.height;
  if (t >= total) return;                                   // whole wave
  float px, py;
  if (t < 2 * c.width) { px = (float)(t >> 1); py = (t & 1) ? (float)(c.height - 1) : 0.f; }
  else { const int u = t - 2 * c.width; py = (float)(u >> 1); px = (u & 1) ? (float)(c.width - 1) : 0.f; }
  const float dx = c.fx_inv * px + c.cx_inv, dy = c.fy_inv * py + c.cy_inv;
  for (int sd = lane; sd < 100; sd += kWave) {
    const int yi = sd / 10, xi = sd - 10 * yi;
    const float iy = dy + 1.5f * (yi - 0.5f * 10) / (0.5f * 10);
    const float ix = dx + 1.5f * (xi - 0.5f * 10) / (0.5f * 10);
    float rx = 0.f, ry = 0.f;
    const bool ok = cam_iterative_undistort<M>(c, dx, dy, ix, iy, rx, ry);
    s_rx[w][sd] = rx; s_ry[w][sd] = ry; s_ok[w][sd] = ok ? 1 : 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane != 0) return;
  bool converged = false, second_available = false;
  float best_radius = E3D_CAM_INF, second_best_radius = E3D_CAM_INF;
  float bx = 0.f, by = 0.f, sbx = 0.f, sby = 0.f;
  for (int sd = 0; sd < 100; ++sd) {
    if (!s_ok[w][sd]) continue;
    const float rx = s_rx[w][sd], ry = s_ry[w][sd];
    const float radius = sqrtf(rx * rx + ry * ry);
    if (radius < 0.99f * best_radius) {
      second_best_radius = best_radius;
      sbx = bx; sby = by;
      second_available = converged;
      best_radius = radius; bx = rx; by = ry;
      converged = true;
    } else if (radius > 1 / 0.99f * best_radius && radius < 0.99f * second_best_radius) {
      second_best_radius = radius;
      sbx = rx; sby = ry;
      second_available = true;
    }
  }
  if (converged) {
    atomicMax(&out[0], __float_as_uint(bx * bx + by * by));
    if (second_available) atomicMin(&out[1], __float_as_uint(sbx * sbx + sby * sby));
  }
}

// one wave per slot: lane l sums blocks l, l+64, ... in order, then a fixed-shape butterfly -> deterministic, no serial chain
__global__ __launch_bounds__(kWave) void k_reg_reduce(const double* __restrict__ partial, int nblocks, int slot, double* __restrict__ out) {
  const int t = blockIdx.x;
  double v = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += kWave) v += partial[(size_t)b * slot + t];
  v = wave_sum(v);
  if (threadIdx.x == 0) out[t] = v;
}

// ==== a19: cost ================================================================================================================================
// per observation (kept in Obs::inten while the observations stay)
__global__ __launch_bounds__(kBlock) void k_reg_intensity(Pyramid Y, const float* __restrict__ o_x, const float* __restrict__ o_y,
                                                          const float* __restrict__ o_s, size_t n_obs, float* __restrict__ inten) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_obs) return;
  inten[i] = obs_intensity(Y, o_x[i], o_y[i], o_s[i]);
}


// (inten = intensity per observation; nrow = observation row of each neighbour point, k_obs_flags -- an observation with its flag
// set has all K of them: the values the reference reads from its per-point intensity array, cost_calculator.cc:150-200, without
// clearing and scattering that array for every image)
// KK > 0: the neighbour count as a compile-time constant (the default 5): all row indices, then all gathers and descriptor loads of
// an observation are requested together instead of one dependent pair after the other
template <int KK>
__global__ __launch_bounds__(kBlock) void k_reg_cost(const float* __restrict__ inten, const unsigned* __restrict__ o_idx,
                                                     const unsigned char* __restrict__ flags, size_t n_obs,
                                                     const int* __restrict__ nrow, int K_rt, const float* __restrict__ fixed_desc,
                                                     const float* __restrict__ var_desc, const int* __restrict__ obs_counts,
                                                     RegWeights wts, double* __restrict__ partial) {
  constexpr int KM = KK > 0 ? KK : 1;
  const int K = KK > 0 ? KK : K_rt;
  double acc[4] = {0, 0, 0, 0};
  const bool use_f = wts.fixed_weight > 0, use_v = wts.var_weight > 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_obs; i += (size_t)gridDim.x * blockDim.x) {
    if (!flags[i]) continue;
    const size_t p = o_idx[i];
    const float Ic = inten[i];
    if (KK > 0) {
      int r[KM];
      float nb[KM], fd[KM], vd[KM];
#pragma unroll
      for (int k = 0; k < KM; ++k) r[k] = nrow[i * KM + k];
      const bool var_ok = use_v && obs_counts[p] >= 2;
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        nb[k] = inten[r[k]];
        fd[k] = use_f ? fixed_desc[p * KM + k] : 0.f;
        vd[k] = var_ok ? var_desc[p * KM + k] : 0.f;
      }
      if (use_f) {
        float pr = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) { const float c = (nb[k] - Ic) - fd[k]; pr += c * c; }
        pr = sqrtf(pr);
        acc[0] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
        acc[2] += 1.0;
      }
      if (var_ok) {
        float pr = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) { const float c = (nb[k] - Ic) - vd[k]; pr += c * c; }
        pr = sqrtf(pr);
        acc[1] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
        acc[3] += 1.0;
      }
      continue;
    }
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      const float sw = kind == 0 ? wts.fixed_weight : wts.var_weight;
      if (!(sw > 0)) continue;
      if (kind == 1 && !(obs_counts[p] >= 2)) continue;
      const float* desc = kind == 0 ? fixed_desc : var_desc;
      float pr = 0.f;
      for (int k = 0; k < K; ++k) {
        const float c = (inten[nrow[i * K + k]] - Ic) - desc[p * K + k];
        pr += c * c;
      }
      pr = sqrtf(pr);
      acc[kind] += (double)robust_residual(wts.robust_type, wts.robust_param, pr);
      acc[2 + kind] += 1.0;
    }
  }
  __shared__ double s[kBlock / kWave][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    partial[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
  }
}

// ==== a23: colour update ========================================================================================================================
// Within one image every point is observed at most once, so the per-image accumulation needs no atomics; images are
// processed one after the other (same f32 summation order as a sequential loop over images).
template <int KT>
__global__ __launch_bounds__(kBlock) void k_color_accumulate(const float* __restrict__ inten, const unsigned* __restrict__ o_idx,
                                                             const unsigned char* __restrict__ flags, size_t n_obs,
                                                             const int* __restrict__ nrow, int K, float* __restrict__ desc,
                                                             int* __restrict__ counts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_obs || !flags[i]) return;
  const size_t p = o_idx[i];
  const float Ic = inten[i];
  counts[p] += 1;
  if constexpr (KT > 0) {            // K at compile time: row indices, neighbour intensities and descriptors requested together
    int r[KT]; float In[KT], dk[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) { r[k] = nrow[i * KT + k]; dk[k] = desc[p * KT + k]; }
#pragma unroll
    for (int k = 0; k < KT; ++k) In[k] = inten[r[k]];
#pragma unroll
    for (int k = 0; k < KT; ++k) desc[p * KT + k] = dk[k] + (In[k] - Ic);
  } else {
    for (int k = 0; k < K; ++k) desc[p * K + k] += inten[nrow[i * K + k]] - Ic;
  }
}
__global__ __launch_bounds__(kBlock) void k_color_finish(size_t n, int K, float* __restrict__ desc, const int* __restrict__ counts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = counts[i];
  if (c > 1)
    for (int k = 0; k < K; ++k) desc[i * K + k] /= c;
}

__global__ __launch_bounds__(kBlock) void k_fill_f32(float* p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// the same with 16-byte stores (p from hipMalloc: 256-byte aligned); the last n % 4 entries one by one
__global__ __launch_bounds__(kBlock) void k_fill_f32x4(float* __restrict__ p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n4 = n / 4;
  if (i < n4) reinterpret_cast<float4*>(p)[i] = make_float4(v, v, v, v);
  if (i < (n & 3)) p[4 * n4 + i] = v;
}
// nearest of two mesh depth maps; 0 = no geometry
__global__ __launch_bounds__(kBlock) void k_depth_merge(float* __restrict__ a, const float* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i], y = b[i];
  a[i] = (x == 0.f) ? y : ((y == 0.f) ? x : fminf(x, y));
}
__global__ __launch_bounds__(kBlock) void k_fill_i32(int* p, size_t n, int v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_xyz_to_float4(const float* __restrict__ xyz, size_t n, float4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
}
// 30-bit Morton code of a point in a 1024^3 lattice over the bounding box (NaN / out of range: clamped) -- only an ORDER for the
// splat points: the depth map is a minimum over them, whatever their order
__device__ __forceinline__ unsigned morton_spread10(unsigned v) {
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ __launch_bounds__(kBlock) void k_morton_keys(const float* __restrict__ xyz, size_t n, float ox, float oy, float oz, float inv,
                                                        unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float fx = (xyz[3 * i] - ox) * inv, fy = (xyz[3 * i + 1] - oy) * inv, fz = (xyz[3 * i + 2] - oz) * inv;
  const unsigned qx = (unsigned)fminf(fmaxf(fx, 0.f), 1023.f), qy = (unsigned)fminf(fmaxf(fy, 0.f), 1023.f), qz = (unsigned)fminf(fmaxf(fz, 0.f), 1023.f);
  keys[i] = morton_spread10(qx) | (morton_spread10(qy) << 1) | (morton_spread10(qz) << 2);
  vals[i] = (unsigned)i;
}
__global__ __launch_bounds__(kBlock) void k_gather_xyz_to_float4(const float* __restrict__ xyz, const unsigned* __restrict__ order, size_t n,
                                                                 float4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = order[i];
  out[i] = make_float4(xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2], 0.f);
}

// =====================================================================================================================================================
// host side
// =====================================================================================================================================================
struct PointScale {
  size_t n = 0;
  float radius = 0;
  DevBuf<float4> pts;
  DevBuf<unsigned> nbr;
  DevBuf<float> fixed_desc, var_desc;
  DevBuf<int> obs_counts, row_of_point;
  bool has_fixed = false;
};
struct Intrin {
  int type = 0, min_image_scale = 0, n_params = 4;
  int width = 0, height = 0;
  float params[12] = {0};
  std::vector<CamLevel> levels;
  std::shared_ptr<std::vector<DevBuf<unsigned char>>> cam_mask;     // per level (empty buffer = none); survives parameter updates
};
struct Obs {
  bool active = false;            // false: the reference would hold no vector for this (image, scale); buffers are kept for reuse
  size_t n = 0;
  DevBuf<unsigned> idx;
  DevBuf<float> x, y, s;
  DevBuf<unsigned char> flags;
  DevBuf<int> nrow;               // K per observation: observation row of each neighbour point (-1: not observed), for pass 2
  DevBuf<float4> rows;
  bool rows_valid = false;
  DevBuf<float4> drows;           // depth residual rows (k_reg_depth_rows), recomputed per use
  // the interpolated image intensity of every observation (k_reg_intensity): a function of (x, y, s) and the image alone, so the
  // colour update and the cost that follows it at the same observations sample the pyramid once
  DevBuf<float> inten;
  bool inten_valid = false;
  int inten_w = 0, inten_h = 0, inten_min_scale = -1, inten_levels = 0;      // (the pyramid geometry they were sampled with)
  // flags / nrow describe the list `flags_src` (device pointer of an indexed re-projection's candidates) of flags_count entries:
  // a trial state of Apply that keeps every point of the visibility list visible reuses them (finish_observations skipped)
  const void* flags_src = nullptr;
  size_t flags_count = 0;
};
struct ImageDev {
  int intrinsics_id = -1;
  std::vector<DevBuf<unsigned char>> pix, mask;
  std::vector<bool> has_mask;
  Pose pose{};                    // so3().matrix() + translation of image_T_global, as the kernels read it
  SE3f pose_q;                    // image_T_global (Sophus::SE3f state)
  DevBuf<float> depth;            // last rendered occlusion depth (as float bits)
  int depth_scale = -1;
  std::vector<DevBuf<float>> depth_maps;   // Problem::depth_maps_ (fixed depth maps, one per pyramid level); empty = none
  std::map<int, Obs> obs;         // per point scale
  std::map<int, DevBuf<unsigned>> vis;   // per point scale: visibility list of the running Apply (grow-only scratch)
  // ObservationsCache::image_id_to_visibility_lists_ (observations_cache.h): per point scale, the observed point indices
  std::map<int, std::pair<DevBuf<unsigned>, size_t>> observed;
  bool has_observed = false;
  // rig membership (opt::RigImages): camera_index 0 = the frame's reference image, whose pose is the rig pose
  int rig_id = -1, camera_index = 0, ref_image_id = -1;
  bool dependent() const { return rig_id >= 0 && camera_index > 0; }
};
// one occlusion mesh (OcclusionGeometry::AddMesh): vertices in the global frame, triangles, and -- if requested -- the
// silhouette-candidate edges with the normals of their two outermost faces
struct MeshDev {
  size_t n_vertices = 0, n_triangles = 0, n_edges = 0;
  DevBuf<float4> vertices, normals;
  DevBuf<unsigned> triangles;
  DevBuf<MeshEdge> edges;
};
struct RigState { std::vector<SE3f> image_T_rig; };                 // opt::Rig (rig.h:41-76)
struct RigFrame { int rig_id; std::vector<int> image_ids; };        // opt::RigImages

static void set_pose(ImageDev& im, const SE3f& T) {
  im.pose_q = T;
  quat_to_matrix<float>(T.q.w, T.q.x, T.q.y, T.q.z, im.pose.R);
  for (int i = 0; i < 3; ++i) im.pose.t[i] = T.t[i];
}

}  // namespace e3d

using namespace e3d;

struct e3d_reg {
  int device = 0;
  hipStream_t stream = nullptr;
  e3d_reg_params prm{};
  std::map<int, PointScale> scales;
  std::map<int, Intrin> intr;
  std::map<int, ImageDev> images;
  std::map<int, RigState> rigs;
  std::vector<RigFrame> frames;
  DevBuf<float4> splat;
  size_t n_splat = 0;
  std::vector<std::unique_ptr<MeshDev>> meshes;
  DevBuf<float4> mesh_projected;          // per mesh vertex: pixel position and camera-space z
  DevBuf<float2> mesh_shaded;             // ... and the vertex shader output (x', y') the near-plane clipping interpolates
  DevBuf<float> depth_unmasked, ztmp_f;
  float min_occlusion_depth = 0.05f, max_occlusion_depth = 100.f;     // opt::Parameters defaults (parameters.h:60-61)
  bool mask_occlusion_boundaries = true;
  bool cache_observations = false;        // Optimizer::cache_observations_ (optimizer.h)
  DevBuf<float4> scan_pts;                // f4: evaluation scan points + their observation counts
  DevBuf<int> scan_counts;
  DevBuf<unsigned char> eval_mask;
  DevBuf<unsigned> gt_depth;
  size_t n_scan = 0;
  // scratch
  DevBuf<int> valid;
  DevBuf<float> tx, ty, ts;
  DevBuf<unsigned> cand, block_counts, block_offsets;
  PinBuf<unsigned char> mailbox;     // read_back
  DevBuf<double> block_d2, chunk_d2, d_total_d2, partial, red, red_all, acc_all;
  DevBuf<unsigned long long> chunk_sum, d_total;
  DevBuf<float> dummy_d2;
  DevBuf<unsigned> cut;
  // multi-GPU: image id mod world == rank -> owned
  int rank = 0, world = 1;
  e3d_allreduce_fn allreduce = nullptr;
  e3d_allreduce_device_fn allreduce_dev = nullptr;
  void* ar_user = nullptr;
  e3d_comm* comm = nullptr;                     // native RCCL collectives (e3d_reg_set_comm); not owned
  // HIP-event times of the two kernels of e3d_reg_accumulate, summed since the last e3d_reg_kernel_times(reset)
  std::unique_ptr<EventTimer> t_pass1, t_pass2;
  double pass1_ms = 0, pass2_ms = 0, pass_observations = 0, pass_calls = 0;
  DevBuf<double> comm_stage;
  // e3d_reg_profile: wall-clock time per phase of RunOnCurrentScale (the stream is synchronised at every phase boundary while
  // the profile is on, so a profiled run is slower than a plain one; the split is what it is for)
  bool profile_on = false;
  std::map<std::string, double> profile_ms;
  // ... and, while the profile is on, HIP-event time, launches and work units (points / observations / pixels: what the bench
  // prices a group's algorithmic bytes with) of every kernel group of an iteration (KT below); read lazily, no synchronisation
  struct KGroup { double ms = 0, units = 0; long long calls = 0; };
  std::map<std::string, KGroup> kgroups;
  struct KPending { hipEvent_t a, b; KGroup* g; };
  std::vector<KPending> kpending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kfree;
  void kflush() {
    for (KPending& p : kpending) {
      float t = 0.f;
      if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) p.g->ms += (double)t;
      kfree.emplace_back(p.a, p.b);
    }
    kpending.clear();
  }
  bool owns(int image_id) const { return world <= 1 || ((image_id % world) + world) % world == rank; }
  // splat depth: per-point rectangles, (tile, point) pairs (double-buffered for the sort), tile ranges
  DevBuf<uint4> rects;
  DevBuf<unsigned> sp_keys[2], sp_vals[2], sp_counter, tile_start, tile_end, zbuf, ztmp;
  DevBuf<char> sort_temp;
  ~e3d_reg() {
    kflush();
    for (auto& e : kfree) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace e3d {

static void rsync(e3d_reg* h) { E3D_HIP(hipStreamSynchronize(h->stream)); }
// A few words back from the device, through the handle's pinned mailbox, and the stream synchronised: hipMemcpyAsync into pageable
// memory (a stack variable) goes through the runtime's staging path and costs several times the copy into pinned memory -- an
// iteration of RunOnCurrentScale reads ~120 such results (pair counts, kept / dropped candidates, the sums of every pass).
static void read_back(e3d_reg* h, void* dst, const void* src_dev, size_t bytes) {
  static const bool pinned = [] { const char* e = getenv("E3D_REG_PINNED"); return !(e && e[0] == '0'); }();     // (0: as rounds 1 - 5, for A / B timing)
  if (bytes == 0 || !pinned) { copy_out(dst, src_dev, bytes, h->stream); rsync(h); return; }
  if (bytes > h->mailbox.cap) h->mailbox.reserve(std::max(bytes, (size_t)65536));
  E3D_HIP(hipMemcpyAsync(h->mailbox.p, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
  E3D_HIP(hipStreamSynchronize(h->stream));
  memcpy(dst, h->mailbox.p, bytes);
}

// stop-watch of one kernel group (HIP events on the handle's stream; only while e3d_reg_profile is on)
struct KT {
  e3d_reg* h; hipEvent_t a = nullptr, b = nullptr; e3d_reg::KGroup* g = nullptr;
  KT(e3d_reg* hh, const char* name, double units) : h(hh) {
    if (!h->profile_on) return;
    if (h->kpending.size() >= 8192) h->kflush();
    if (h->kfree.empty()) { hipEvent_t x, y; E3D_HIP(hipEventCreate(&x)); E3D_HIP(hipEventCreate(&y)); h->kfree.emplace_back(x, y); }
    a = h->kfree.back().first; b = h->kfree.back().second; h->kfree.pop_back();
    g = &h->kgroups[name];
    g->calls += 1; g->units += units;
    (void)hipEventRecord(a, h->stream);
  }
  ~KT() {
    if (!g) return;
    (void)hipEventRecord(b, h->stream);
    h->kpending.push_back({a, b, g});
  }
};

struct Phase {
  e3d_reg* h; const char* name; std::chrono::steady_clock::time_point t0;
  Phase(e3d_reg* hh, const char* n) : h(hh), name(n) {
    if (h->profile_on) { (void)hipStreamSynchronize(h->stream); t0 = std::chrono::steady_clock::now(); }
  }
  ~Phase() {
    if (h->profile_on) {
      (void)hipStreamSynchronize(h->stream);
      h->profile_ms[name] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
};
static unsigned nblk(size_t n) { return (unsigned)div_up(n ? n : 1, kBlock); }

// Sums over the ranks.  With the library's own communicator (e3d_reg_set_comm) both run as RCCL all-reduces on the handle's
// stream -- the small f64 blocks staged through HBM, the descriptors in place -- otherwise through the callbacks.
static void allreduce_host(e3d_reg* h, double* buf, size_t n) {
  if (!n) return;
  if (h->comm) {
    h->comm_stage.reserve(n);
    copy_in(h->comm_stage.p, buf, sizeof(double) * n, h->stream);
    comm_allreduce_f64(h->comm, h->comm_stage.p, n, h->stream);
    copy_out(buf, h->comm_stage.p, sizeof(double) * n, h->stream);
    rsync(h);
    return;
  }
  if (h->world <= 1) return;
  if (!h->allreduce || h->allreduce(buf, n, h->ar_user) != 0) throw Error(E3D_ERR_INVALID, "all-reduce callback failed");
}
static void allreduce_device(e3d_reg* h, void* dev, size_t n, int dtype) {
  if (!n) return;
  if (h->comm) {
    if (dtype == 0) comm_allreduce_f32(h->comm, static_cast<float*>(dev), n, h->stream);
    else comm_allreduce_i32(h->comm, static_cast<int*>(dev), n, h->stream);
    return;
  }
  if (h->world <= 1) return;
  rsync(h);
  if (!h->allreduce_dev || h->allreduce_dev(dev, n, dtype, h->ar_user) != 0) throw Error(E3D_ERR_INVALID, "device all-reduce callback failed");
}

// run `stmt` with the camera model as the compile-time constant M
#define E3D_CAM_SWITCH(model, stmt)                                                        \
  switch (model) {                                                                         \
    case kPinhole: { constexpr int M = kPinhole; stmt; } break;                            \
    case kOpenCV: { constexpr int M = kOpenCV; stmt; } break;                              \
    case kThinPrismFisheye: { constexpr int M = kThinPrismFisheye; stmt; } break;          \
    case kOpenCVFisheye: { constexpr int M = kOpenCVFisheye; stmt; } break;                \
    case kFov: { constexpr int M = kFov; stmt; } break;                                    \
    case kSimplePinhole: { constexpr int M = kSimplePinhole; stmt; } break;                \
    case kSimpleRadial: { constexpr int M = kSimpleRadial; stmt; } break;                  \
    case kRadial: { constexpr int M = kRadial; stmt; } break;                              \
    case kPolynomial3: { constexpr int M = kPolynomial3; stmt; } break;                    \
    case kFisheyePolyTang: { constexpr int M = kFisheyePolyTang; stmt; } break;            \
    case kFullOpenCV: { constexpr int M = kFullOpenCV; stmt; } break;                      \
    case kRadialFisheye: { constexpr int M = kRadialFisheye; stmt; } break;                \
    case kSimpleRadialFisheye: { constexpr int M = kSimpleRadialFisheye; stmt; } break;    \
    default: throw Error(E3D_ERR_INVALID, "unknown camera model");                         \
  }

// RadialBase::InitCutoff of the Polynomial4Camera inside OPENCV_FISHEYE (camera_base_impl_radial.h:59-170): the farthest image
// corner, ten start radii, a 1-D Gauss-Newton each -- a few hundred scalar operations, done on the host in the reference's f32 /
// f64 mix (the start radius is `float + float * double / float`).
static float radial_init_cutoff(const CamLevel& c) {
  const float* q = c.q;
  const int model = c.model;
  // DistortionFactor(r2) / DistortedDerivativeByNormalized(r2) of the RadialBase child, in its own expression order
  // (camera_polynomial_4.h:55-61,100-110; camera_radial.h:60-65,103-107; camera_polynomial.h:58-64,103-107)
  auto factor = [&](float r2) {
    if (cam_is_radial2(model)) return 1.0f + r2 * (q[0] + r2 * q[1]);
    if (model == kPolynomial3) return 1.0f + r2 * (q[0] + r2 * (q[1] + r2 * q[2]));
    return 1.0f + r2 * (q[0] + r2 * (q[1] + r2 * (q[2] + r2 * q[3])));
  };
  auto dfactor = [&](float r2) {
    if (cam_is_radial2(model)) return 1.f + r2 * (3.f * q[0] + r2 * 5.f * q[1]);
    if (model == kPolynomial3) return 1.0f + r2 * (3.0f * q[0] + r2 * (5.0f * q[1] + r2 * 7.0f * q[2]));
    return 1.0f + r2 * (3.0f * q[0] + r2 * (5.0f * q[1] + r2 * (7.0f * q[2] + r2 * (9.0f * q[3]))));
  };
  float test_r = 0.f;
  for (int k = 0; k < 4; ++k) {
    const float px = (k & 2) ? (float)c.width : 0.f, py = (k & 1) ? (float)c.height : 0.f;
    const float x = c.fx_inv * px + c.cx_inv, y = c.fy_inv * py + c.cy_inv;
    const float r = sqrtf(x * x + y * y);
    if (k == 0 || r > test_r) test_r = r;
  }
  bool converged = false, second_available = false;
  float best = INFINITY, second = INFINITY;
  for (int i = 0; i < 10; ++i) {
    const float init_radius = (float)((double)test_r + (double)1.5f * ((double)i - 0.5 * 10) / (double)(0.5f * 10));
    bool tc = false;
    float ur = init_radius, ur2 = init_radius * init_radius;
    for (int it = 0; it < 100; ++it) {
      const float r_candidate = ur * factor(ur2);
      const float delta_r = r_candidate - test_r;
      if (delta_r * delta_r < 1e-10f) { tc = true; break; }
      const float step = delta_r / dfactor(ur2);
      ur -= step;
      ur2 = ur * ur;
    }
    if (tc) {
      if (ur < 0.99f * best) { second = best; second_available = converged; best = ur; converged = true; }
      else if (ur > 1 / 0.99f * best && ur < 0.99f * second) { second = ur; second_available = true; }
    }
  }
  if (converged && best > 0) {
    if (second_available && second > 0) { const float a = best * best * 1.01f, b = second * second; return (b < a) ? b : a; }
    return best * best * 1.01f;
  }
  return INFINITY;
}

// One camera of the pyramid = one constructor call of the reference's camera class: pixel mapping (camera_base.cc:81-86) and,
// for the distorted models, InitCutoff -- run on the device (k_cam_cutoff), 2(W+H) border points in parallel.
static CamLevel make_level(e3d_reg* h, int model, int w, int h_px, const float* p) {
  CamLevel c{};
  c.model = model; c.width = w; c.height = h_px;
  if (cam_unique_focal(model)) {                    // [f cx cy ...]: the class is constructed with fx = fy = f (camera_radial.cc:43-46)
    c.fx = p[0]; c.fy = p[0]; c.cx = p[1]; c.cy = p[2];
    for (int i = 0; i < cam_distortion_count(model); ++i) c.q[i] = p[3 + i];
  } else {
    c.fx = p[0]; c.fy = p[1]; c.cx = p[2]; c.cy = p[3];
    for (int i = 0; i < cam_distortion_count(model); ++i) c.q[i] = p[4 + i];
  }
  c.fx_inv = (float)(1.0 / (double)c.fx); c.fy_inv = (float)(1.0 / (double)c.fy);
  c.cx_inv = (float)(-1.0 * (double)c.cx / (double)c.fx); c.cy_inv = (float)(-1.0 * (double)c.cy / (double)c.fy);
  c.cutoff2 = INFINITY; c.inner_cutoff2 = INFINITY;
  if (model == kPinhole || model == kSimplePinhole) return c;   // no InitCutoff (camera_pinhole.cc:35-43, camera_simple_pinhole.cc:36-41)
  if (model == kOpenCVFisheye || model == kRadialFisheye) { c.inner_cutoff2 = radial_init_cutoff(c); return c; }   // the RadialBase child inside
  if (model == kSimpleRadialFisheye) {              // the SimpleRadialCamera inside (camera_simple_radial.cc:51-57)
    if (c.q[0] < 0) c.inner_cutoff2 = -1.f / (3 * c.q[0]);
    return c;
  }
  if (model == kRadial || model == kPolynomial3) { c.cutoff2 = radial_init_cutoff(c); return c; }   // RadialBase::InitCutoff on the camera itself
  if (model == kSimpleRadial) {                     // camera_simple_radial.cc:51-57: where d(distorted r)/dr = 0
    if (c.q[0] < 0) c.cutoff2 = -1.f / (3 * c.q[0]);
    return c;
  }
  if (model == kFov) {                              // camera_fisheye_fov.cc:37-51: derived constants, no InitCutoff
    c.q[1] = 2.0f * e3d_tanf(0.5f * c.q[0]);
    c.q[2] = (float)(M_PI / (double)(2 * c.q[0]));
    return c;
  }
  h->cut.reserve(2);
  const unsigned init[2] = {0u, 0x7f800000u};       // min_candidate = 0, max_candidate = +inf
  copy_in(h->cut.p, init, sizeof init, h->stream);
  const int total = 2 * w + 2 * h_px;
  const unsigned cut_blocks = (unsigned)div_up((size_t)total, kBlock / kWave);      // one wave per border test point
  if (cam_is_poly_tang(model)) hipLaunchKernelGGL(k_cam_cutoff<kOpenCV>, dim3(cut_blocks), dim3(kBlock), 0, h->stream, c, h->cut.p);   // (the inner PolynomialTangentialCamera of kFisheyePolyTang)
  else if (model == kFullOpenCV) hipLaunchKernelGGL(k_cam_cutoff<kFullOpenCV>, dim3(cut_blocks), dim3(kBlock), 0, h->stream, c, h->cut.p);
  else hipLaunchKernelGGL(k_cam_cutoff<kThinPrismFisheye>, dim3(cut_blocks), dim3(kBlock), 0, h->stream, c, h->cut.p);   // inner ThinPrismCamera
  unsigned out[2];
  copy_out(out, h->cut.p, sizeof out, h->stream);
  rsync(h);
  float mn, mx;
  std::memcpy(&mn, &out[0], 4); std::memcpy(&mx, &out[1], 4);
  const float a = mn * 1.01f;
  const float cutoff = (mx < a) ? mx : a;           // std::min(min_candidate * kIncreaseFactor, max_candidate)
  if (model == kOpenCV || model == kFullOpenCV) c.cutoff2 = cutoff; else c.inner_cutoff2 = cutoff;
  return c;
}

// Intrinsics::BuildModelPyramid (intrinsics.cc:46-51): level l = ScaledBy(0.5) of level l-1 (camera_base_impl.h:70-89)
static void build_model_pyramid(e3d_reg* h, Intrin& in, int n_levels) {
  in.levels.clear();
  in.levels.push_back(make_level(h, in.type, in.width, in.height, in.params));
  for (int l = 1; l < n_levels; ++l) {
    const CamLevel& p = in.levels.back();
    const float f = 0.5f;
    float q[12];
    for (int i = 0; i < in.n_params; ++i) q[i] = in.params[i];
    if (cam_unique_focal(in.type)) { q[0] = p.fx * f; q[1] = f * (p.cx + 0.5f) - 0.5f; q[2] = f * (p.cy + 0.5f) - 0.5f; }
    else { q[0] = p.fx * f; q[1] = p.fy * f; q[2] = f * (p.cx + 0.5f) - 0.5f; q[3] = f * (p.cy + 0.5f) - 0.5f; }
    in.levels.push_back(make_level(h, in.type, (int)(f * p.width + 0.5f), (int)(f * p.height + 0.5f), q));
  }
}

static int image_model(e3d_reg* h, const ImageDev& im) { return h->intr.at(im.intrinsics_id).type; }
static int local_unknowns(e3d_reg* h, const ImageDev& im) { return h->intr.at(im.intrinsics_id).n_params + (im.dependent() ? 12 : 6); }

// image_T_global of the non-reference images of every rig frame = image_T_rig[camera] * reference image_T_global
// (CreateDeltaState, intrinsics_and_pose_optimizer.cc:539-548)
static void compose_rig_poses(e3d_reg* h) {
  for (const RigFrame& f : h->frames) {
    const RigState& rig = h->rigs.at(f.rig_id);
    const SE3f ref = h->images.at(f.image_ids[0]).pose_q;
    for (size_t c = 1; c < f.image_ids.size(); ++c) {
      ImageDev& im = h->images.at(f.image_ids[c]);
      set_pose(im, se3_mul(rig.image_T_rig[c], ref));
      for (auto& o : im.obs) o.second.rows_valid = false;
    }
  }
}
static RigLink rig_link(e3d_reg* h, const ImageDev& im) {
  RigLink L{};
  if (!im.dependent()) return L;
  const SE3f& ir = h->rigs.at(im.rig_id).image_T_rig[im.camera_index];
  quat_to_matrix<float>(ir.q.w, ir.q.x, ir.q.y, ir.q.z, L.R_image_rig);
  const SE3f& rg = h->images.at(im.ref_image_id).pose_q;
  L.q_rig_global[0] = rg.q.w; L.q_rig_global[1] = rg.q.x; L.q_rig_global[2] = rg.q.y; L.q_rig_global[3] = rg.q.z;
  for (int i = 0; i < 3; ++i) L.t_rig_global[i] = rg.t[i];
  return L;
}

static Pyramid make_pyramid(e3d_reg* h, const ImageDev& im) {
  const Intrin& in = h->intr.at(im.intrinsics_id);
  if (im.pix.size() != in.levels.size()) throw Error(E3D_ERR_INVALID, "the pyramid of this image is not resident on this rank");
  Pyramid Y{};
  Y.n_levels = (int)in.levels.size();
  Y.min_image_scale = in.min_image_scale;
  for (int l = 0; l < Y.n_levels; ++l) {
    Y.img[l] = im.pix[l].p;
    Y.mask[l] = im.has_mask[l] ? im.mask[l].p : nullptr;
    Y.cam_mask[l] = (in.cam_mask && l < (int)in.cam_mask->size() && (*in.cam_mask)[l].p) ? (*in.cam_mask)[l].p : nullptr;
    Y.cam[l] = in.levels[l];
  }
  return Y;
}

static PointScale& get_scale(e3d_reg* h, int s) {
  auto it = h->scales.find(s);
  if (it == h->scales.end()) throw Error(E3D_ERR_INDEX, fmt("point scale %d not set", s));
  return it->second;
}
static ImageDev& get_image(e3d_reg* h, int id) {
  auto it = h->images.find(id);
  if (it == h->images.end()) throw Error(E3D_ERR_INDEX, fmt("image %d not set", id));
  return it->second;
}
static bool has_obs(const ImageDev& im, int s) {
  auto it = im.obs.find(s);
  return it != im.obs.end() && it->second.active;
}
static Obs& get_obs(ImageDev& im, int s) {
  auto it = im.obs.find(s);
  if (it == im.obs.end() || !it->second.active) throw Error(E3D_ERR_INVALID, fmt("no observations for point scale %d (call e3d_reg_observe first)", s));
  return it->second;
}

static void check_params(const e3d_reg_params* p) {
  if (!p) throw Error(E3D_ERR_INVALID, "null params");
  if (p->point_neighbor_count < 1 || p->point_neighbor_count > 8) throw Error(E3D_ERR_INVALID, "point_neighbor_count must be in [1, 8]");
  if (p->robust_weighting_type < 0 || p->robust_weighting_type > 2) throw Error(E3D_ERR_INVALID, "robust_weighting_type must be 0, 1 or 2");
}

// flags + row_of_point for the current observation list
// (map_written: the compaction of an all-points pass has written S.row_of_point for this list already)
static void finish_observations(e3d_reg* h, PointScale& S, Obs& O, bool map_written = false) {
  hipStream_t s = h->stream;
  KT kt(h, "obs.neighbour_flags", (double)O.n);
  O.flags_src = nullptr; O.flags_count = 0;
  if (!map_written) hipLaunchKernelGGL(k_fill_i32, dim3(nblk(S.n)), dim3(kBlock), 0, s, S.row_of_point.p, S.n, -1);
  O.flags.reserve(O.n);
  O.nrow.reserve(O.n * (size_t)h->prm.point_neighbor_count);
  if (O.n) {
    if (!map_written) hipLaunchKernelGGL(k_obs_mark, dim3(nblk(O.n)), dim3(kBlock), 0, s, O.idx.p, O.n, S.row_of_point.p);
    if (h->prm.point_neighbor_count == 5)
      hipLaunchKernelGGL(k_obs_flags<5>, dim3(nblk(O.n)), dim3(kBlock), 0, s, O.idx.p, O.n, S.nbr.p, 5, S.row_of_point.p, O.flags.p, O.nrow.p);
    else
      hipLaunchKernelGGL(k_obs_flags<0>, dim3(nblk(O.n)), dim3(kBlock), 0, s, O.idx.p, O.n, S.nbr.p, h->prm.point_neighbor_count,
                         S.row_of_point.p, O.flags.p, O.nrow.p);
  }
  O.rows_valid = false;
}

// pass 1 of the observation list (row_of_point is scratch of finish_observations; pass 2 reads the per-observation O.nrow)
static void prepare_rows(e3d_reg* h, ImageDev& im, PointScale& S, Obs& O) {
  hipStream_t s = h->stream;
  const int model = image_model(h, im);
  O.rows.reserve((size_t)rows4(local_unknowns(h, im)) * O.n);
  if (O.n) {
    const RigLink L = rig_link(h, im);
    if (im.dependent()) {
      E3D_CAM_SWITCH(model, hipLaunchKernelGGL((k_reg_pass1<M, true>), dim3(nblk(O.n)), dim3(kBlock), 0, s, S.pts.p, S.radius, im.pose,
                                               make_pyramid(h, im), O.idx.p, O.x.p, O.y.p, O.s.p, O.n, L, O.rows.p));
    } else {
      E3D_CAM_SWITCH(model, hipLaunchKernelGGL((k_reg_pass1<M, false>), dim3(nblk(O.n)), dim3(kBlock), 0, s, S.pts.p, S.radius, im.pose,
                                               make_pyramid(h, im), O.idx.p, O.x.p, O.y.p, O.s.p, O.n, L, O.rows.p));
    }
  }
  O.rows_valid = true;
}

static void obs_intensities(e3d_reg* h, ImageDev& im, Obs& O) {
  hipStream_t s = h->stream;
  const Intrin& in = h->intr.at(im.intrinsics_id);
  const bool same_pyramid = O.inten_w == in.levels[0].width && O.inten_h == in.levels[0].height && O.inten_min_scale == in.min_image_scale &&
                            O.inten_levels == (int)in.levels.size();
  if (O.n && !(O.inten_valid && same_pyramid)) {
    KT kt(h, "intensity.sample", (double)O.n);
    O.inten.reserve(O.n);
    hipLaunchKernelGGL(k_reg_intensity, dim3(nblk(O.n)), dim3(kBlock), 0, s, make_pyramid(h, im), O.x.p, O.y.p, O.s.p, O.n, O.inten.p);
    O.inten_valid = true;
    O.inten_w = in.levels[0].width; O.inten_h = in.levels[0].height; O.inten_min_scale = in.min_image_scale; O.inten_levels = (int)in.levels.size();
  }
}

}  // namespace e3d

#define R_TRY try {
// entry points that take a handle run on the handle's device, whatever the calling thread's current device is
#define R_TRYH try { if (h) E3D_HIP(hipSetDevice(h->device));
#define R_CATCH()                                                                            \
  } catch (const e3d::Error& e) { e3d::set_last_error(e.what()); return e.code; }            \
  catch (const std::exception& e) { e3d::set_last_error(e.what()); return E3D_ERR_INVALID; }

extern "C" {

e3d_reg_t* e3d_reg_create(const e3d_reg_params* params) {
  try {
    check_params(params);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(E3D_ERR_NO_DEVICE, "no HIP device visible (libe3dhip needs an MI355X / gfx950 GPU)");
    std::unique_ptr<e3d_reg> h(new e3d_reg());
    E3D_HIP(hipGetDevice(&h->device));
    E3D_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->prm = *params;
    return h.release();
  } catch (const std::exception& e) {
    e3d::set_last_error(e.what());
    return nullptr;
  }
}
void e3d_reg_destroy(e3d_reg_t* reg) { delete reg; }

int e3d_reg_set_params(e3d_reg_t* h, const e3d_reg_params* params) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  check_params(params);
  if (params->point_neighbor_count != h->prm.point_neighbor_count && !h->scales.empty())
    throw Error(E3D_ERR_INVALID, "point_neighbor_count cannot change after point scales were set");
  h->prm = *params;
  return 0;
  R_CATCH()
}

int e3d_reg_set_point_scale(e3d_reg_t* h, int point_scale, const float* xyz, size_t n, float point_radius,
                            const uint32_t* neighbor_indices, const float* fixed_descriptors) {
  R_TRYH
  if (!h || (!xyz && n) || (!neighbor_indices && n)) throw Error(E3D_ERR_INVALID, "e3d_reg_set_point_scale: null argument");
  hipStream_t s = h->stream;
  const int K = h->prm.point_neighbor_count;
  PointScale& S = h->scales[point_scale];
  S.n = n; S.radius = point_radius;
  S.pts.reserve(n); S.nbr.reserve(n * K); S.fixed_desc.reserve(n * K); S.var_desc.reserve(n * K);
  S.obs_counts.reserve(n); S.row_of_point.reserve(n);
  DevBuf<float> tmp; tmp.reserve(3 * n);
  copy_in(tmp.p, xyz, sizeof(float) * 3 * n, s);
  hipLaunchKernelGGL(k_xyz_to_float4, dim3(nblk(n)), dim3(kBlock), 0, s, tmp.p, n, S.pts.p);
  copy_in(S.nbr.p, neighbor_indices, sizeof(unsigned) * n * K, s);
  S.has_fixed = fixed_descriptors != nullptr;
  if (fixed_descriptors) copy_in(S.fixed_desc.p, fixed_descriptors, sizeof(float) * n * K, s);
  else E3D_HIP(hipMemsetAsync(S.fixed_desc.p, 0, sizeof(float) * n * K, s));
  E3D_HIP(hipMemsetAsync(S.var_desc.p, 0, sizeof(float) * n * K, s));
  hipLaunchKernelGGL(k_fill_i32, dim3(nblk(n)), dim3(kBlock), 0, s, S.obs_counts.p, n, fixed_descriptors ? 99999 : 0);   // problem.cc:568-570
  rsync(h);
  for (auto& kv : h->images) kv.second.obs.erase(point_scale);
  return 0;
  R_CATCH()
}

int e3d_reg_set_variable_descriptors(e3d_reg_t* h, int point_scale, const float* descriptors, const int32_t* counts) {
  R_TRYH
  if (!h || !descriptors || !counts) throw Error(E3D_ERR_INVALID, "null argument");
  PointScale& S = get_scale(h, point_scale);
  copy_in(S.var_desc.p, descriptors, sizeof(float) * S.n * h->prm.point_neighbor_count, h->stream);
  copy_in(S.obs_counts.p, counts, sizeof(int) * S.n, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}
int e3d_reg_get_variable_descriptors(e3d_reg_t* h, int point_scale, float* descriptors, int32_t* counts) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  PointScale& S = get_scale(h, point_scale);
  if (descriptors) copy_out(descriptors, S.var_desc.p, sizeof(float) * S.n * h->prm.point_neighbor_count, h->stream);
  if (counts) copy_out(counts, S.obs_counts.p, sizeof(int) * S.n, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}

int e3d_reg_set_intrinsics(e3d_reg_t* h, int intrinsics_id, int camera_type, int width, int height, const float* parameters,
                           int n_parameters, int min_image_scale, int n_levels) {
  R_TRYH
  if (!h || !parameters) throw Error(E3D_ERR_INVALID, "null argument");
  if (camera_type < 0 || camera_type >= kNumCameraModels)
    throw Error(E3D_ERR_INVALID, "unknown camera model (E3D_CAMERA_* of e3d_hip.h)");
  if (n_parameters != cam_param_count(camera_type)) throw Error(E3D_ERR_INVALID, fmt("camera model %d takes %d parameters, got %d", camera_type, cam_param_count(camera_type), n_parameters));
  if (n_levels < 1 || n_levels > kRegMaxLevels || width < 2 || height < 2 || min_image_scale < 0) throw Error(E3D_ERR_INVALID, "bad pyramid description");
  Intrin in;
  in.type = camera_type; in.min_image_scale = min_image_scale; in.n_params = n_parameters;
  in.width = width; in.height = height;
  for (int i = 0; i < n_parameters; ++i) in.params[i] = parameters[i];
  build_model_pyramid(h, in, n_levels);
  h->intr[intrinsics_id] = in;                          // (a camera mask belongs to the previous description of this id and goes with it)
  return 0;
  R_CATCH()
}

/* Intrinsics::camera_mask (src/opt/intrinsics.h:104, loaded by Image::LoadImageData, image.cc:62-72): one u8 mask per pyramid level
 * of the camera, shared by all its images; an observation is dropped where the image mask OR the camera mask is non-zero
 * (visibility_estimator.cc:335-345,482-503).  level_masks[l] = width_l x height_l bytes (host or device) or NULL; level_masks == NULL
 * removes the mask.  Call after e3d_reg_set_intrinsics. */
int e3d_reg_set_camera_mask(e3d_reg_t* h, int intrinsics_id, const uint8_t* const* level_masks) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  auto it = h->intr.find(intrinsics_id);
  if (it == h->intr.end()) throw Error(E3D_ERR_INDEX, "no such intrinsics");
  Intrin& in = it->second;
  if (!level_masks) { in.cam_mask.reset(); }
  else {
    auto masks = std::make_shared<std::vector<DevBuf<unsigned char>>>(in.levels.size());
    for (size_t l = 0; l < in.levels.size(); ++l) {
      if (!level_masks[l]) continue;
      const size_t bytes = (size_t)in.levels[l].width * (size_t)in.levels[l].height;
      (*masks)[l].reserve(bytes);
      copy_in((*masks)[l].p, level_masks[l], bytes, h->stream);
    }
    rsync(h);
    in.cam_mask = masks;
  }
  for (auto& kv : h->images)
    if (kv.second.intrinsics_id == intrinsics_id)
      for (auto& o : kv.second.obs) o.second.rows_valid = false;
  return 0;
  R_CATCH()
}

int e3d_reg_get_intrinsics_level(e3d_reg_t* h, int intrinsics_id, int level, int* width, int* height, float* parameters,
                                 float* cutoff2) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  auto it = h->intr.find(intrinsics_id);
  if (it == h->intr.end() || level < 0 || level >= (int)it->second.levels.size()) throw Error(E3D_ERR_INDEX, "no such intrinsics level");
  const CamLevel& c = it->second.levels[level];
  if (width) *width = c.width;
  if (height) *height = c.height;
  if (parameters) {
    if (cam_unique_focal(c.model)) {
      parameters[0] = c.fx; parameters[1] = c.cx; parameters[2] = c.cy;
      for (int i = 3; i < it->second.n_params; ++i) parameters[i] = c.q[i - 3];
    } else {
      parameters[0] = c.fx; parameters[1] = c.fy; parameters[2] = c.cx; parameters[3] = c.cy;
      for (int i = 4; i < it->second.n_params; ++i) parameters[i] = c.q[i - 4];
    }
  }
  if (cutoff2) *cutoff2 = cam_is_fisheye(c.model) ? c.inner_cutoff2 : c.cutoff2;
  return 0;
  R_CATCH()
}

int e3d_reg_set_image(e3d_reg_t* h, int image_id, int intrinsics_id, const uint8_t* const* level_pixels,
                      const uint8_t* const* level_masks) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null argument");
  auto it = h->intr.find(intrinsics_id);
  if (it == h->intr.end()) throw Error(E3D_ERR_INDEX, "intrinsics not set");
  const Intrin& in = it->second;
  if (!h->owns(image_id)) {                     // another rank's image: only its id, intrinsics and pose are kept here
    ImageDev& im = h->images[image_id];
    im.intrinsics_id = intrinsics_id;
    im.pix.clear(); im.mask.clear(); im.has_mask.clear(); im.obs.clear();
    im.depth_scale = -1;
    return 0;
  }
  if (!level_pixels) throw Error(E3D_ERR_INVALID, "null argument");
  ImageDev& im = h->images[image_id];
  im.intrinsics_id = intrinsics_id;
  const int L = (int)in.levels.size();
  im.pix.resize(L); im.mask.resize(L); im.has_mask.assign(L, false);
  for (int l = 0; l < L; ++l) {
    const size_t bytes = (size_t)in.levels[l].width * in.levels[l].height;
    if (!level_pixels[l]) throw Error(E3D_ERR_INVALID, "missing pyramid level");
    im.pix[l].reserve(bytes);
    copy_in(im.pix[l].p, level_pixels[l], bytes, h->stream);
    if (level_masks && level_masks[l]) {
      im.mask[l].reserve(bytes);
      copy_in(im.mask[l].p, level_masks[l], bytes, h->stream);
      im.has_mask[l] = true;
    }
  }
  rsync(h);
  im.obs.clear();
  im.depth_scale = -1;
  return 0;
  R_CATCH()
}

int e3d_reg_set_image_pose(e3d_reg_t* h, int image_id, const float q[4], const float t[3]) {
  R_TRYH
  if (!h || !q || !t) throw Error(E3D_ERR_INVALID, "null argument");
  ImageDev& im = get_image(h, image_id);
  SE3f T;
  T.q.w = q[0]; T.q.x = q[1]; T.q.y = q[2]; T.q.z = q[3];
  for (int i = 0; i < 3; ++i) T.t[i] = t[i];
  set_pose(im, T);
  for (auto& kv : im.obs) kv.second.rows_valid = false;
  if (!h->frames.empty()) compose_rig_poses(h);
  return 0;
  R_CATCH()
}

/* opt::Rig: image_T_rig of every camera of a rig (camera 0 = reference, normally identity) */
int e3d_reg_set_rig(e3d_reg_t* h, int rig_id, int n_cameras, const float* q, const float* t) {
  R_TRYH
  if (!h || !q || !t || n_cameras < 1) throw Error(E3D_ERR_INVALID, "bad rig");
  RigState r;
  r.image_T_rig.resize(n_cameras);
  for (int c = 0; c < n_cameras; ++c) {
    SE3f T;
    T.q.w = q[4 * c]; T.q.x = q[4 * c + 1]; T.q.y = q[4 * c + 2]; T.q.z = q[4 * c + 3];
    for (int i = 0; i < 3; ++i) T.t[i] = t[3 * c + i];
    r.image_T_rig[c] = T;
  }
  h->rigs[rig_id] = r;
  if (!h->frames.empty()) compose_rig_poses(h);
  return 0;
  R_CATCH()
}
int e3d_reg_get_rig(e3d_reg_t* h, int rig_id, int camera_index, float q[4], float t[3]) {
  R_TRYH
  if (!h || !q || !t) throw Error(E3D_ERR_INVALID, "null argument");
  auto it = h->rigs.find(rig_id);
  if (it == h->rigs.end() || camera_index < 0 || camera_index >= (int)it->second.image_T_rig.size()) throw Error(E3D_ERR_INDEX, "no such rig camera");
  const SE3f& T = it->second.image_T_rig[camera_index];
  q[0] = T.q.w; q[1] = T.q.x; q[2] = T.q.y; q[3] = T.q.z;
  for (int i = 0; i < 3; ++i) t[i] = T.t[i];
  return 0;
  R_CATCH()
}
/* opt::RigImages: one frame of a rig = one image per camera, image_ids[0] is the reference image.  The poses of the other
 * images become image_T_rig[camera] * image_T_global(reference). */
int e3d_reg_add_rig_images(e3d_reg_t* h, int rig_id, const int* image_ids, int n_cameras) {
  R_TRYH
  if (!h || !image_ids) throw Error(E3D_ERR_INVALID, "null argument");
  auto it = h->rigs.find(rig_id);
  if (it == h->rigs.end() || (int)it->second.image_T_rig.size() != n_cameras) throw Error(E3D_ERR_INVALID, "rig not set or camera count mismatch");
  RigFrame f; f.rig_id = rig_id;
  for (int c = 0; c < n_cameras; ++c) {
    ImageDev& im = get_image(h, image_ids[c]);
    if (im.rig_id >= 0) throw Error(E3D_ERR_INVALID, fmt("image %d already belongs to a rig frame", image_ids[c]));
    f.image_ids.push_back(image_ids[c]);
  }
  for (int c = 0; c < n_cameras; ++c) {
    ImageDev& im = h->images.at(image_ids[c]);
    im.rig_id = rig_id; im.camera_index = c; im.ref_image_id = image_ids[0];
  }
  h->frames.push_back(f);
  compose_rig_poses(h);
  return 0;
  R_CATCH()
}

int e3d_reg_get_image_pose(e3d_reg_t* h, int image_id, float q[4], float t[3]) {
  R_TRYH
  if (!h || !q || !t) throw Error(E3D_ERR_INVALID, "null argument");
  const ImageDev& im = get_image(h, image_id);
  q[0] = im.pose_q.q.w; q[1] = im.pose_q.q.x; q[2] = im.pose_q.q.y; q[3] = im.pose_q.q.z;
  for (int i = 0; i < 3; ++i) t[i] = im.pose_q.t[i];
  return 0;
  R_CATCH()
}

}  // extern "C"

namespace e3d {
// OcclusionGeometry::RenderDepthMap, mesh branch (:211-271): rasterise all meshes, then mask the occlusion boundaries
static void render_depth_meshes(e3d_reg* h, ImageDev& im, const Intrin& in, const CamLevel& cam) {
  hipStream_t s = h->stream;
  const size_t px = (size_t)cam.width * cam.height;
  const int tiles_x = (int)div_up(cam.width, kTile), tiles_y = (int)div_up(cam.height, kTile);
  const size_t n_tiles = (size_t)tiles_x * tiles_y;
  size_t total_tris = 0, max_vertices = 0;
  for (auto& m : h->meshes) { total_tris += m->n_triangles; max_vertices = std::max(max_vertices, m->n_vertices); }
  if (total_tris >= ((size_t)1 << 31)) throw Error(E3D_ERR_INVALID, "more than 2^31 occlusion triangles");
  h->sp_counter.reserve(1); h->tile_start.reserve(n_tiles); h->tile_end.reserve(n_tiles);
  // depth of all meshes = min over meshes; each mesh is binned, sorted and reduced on its own into im.depth
  DevBuf<float>& D = (h->mask_occlusion_boundaries ? h->depth_unmasked : im.depth);
  D.reserve(px);
  bool first = true;
  for (auto& m : h->meshes) {
    h->mesh_projected.reserve(m->n_vertices); h->mesh_shaded.reserve(m->n_vertices);
    E3D_CAM_SWITCH(in.type, hipLaunchKernelGGL(k_mesh_vertices<M>, dim3(nblk(m->n_vertices)), dim3(kBlock), 0, s, m->vertices.p,
                                               m->n_vertices, im.pose, cam, h->mesh_projected.p, h->mesh_shaded.p));
    const MeshProj mp{cam.fx, cam.fy, cam.cx, cam.cy, h->min_occlusion_depth};
    // capacity: most triangles touch one tile; retried with the exact count if the first guess was too small
    size_t capacity = m->n_triangles + m->n_triangles / 2 + 1024;
    unsigned n_pairs = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      for (int k = 0; k < 2; ++k) { h->sp_keys[k].reserve(capacity); h->sp_vals[k].reserve(capacity); }
      E3D_HIP(hipMemsetAsync(h->sp_counter.p, 0, sizeof(unsigned), s));
      hipLaunchKernelGGL(k_mesh_bin, dim3(nblk(m->n_triangles)), dim3(kBlock), 0, s, h->mesh_projected.p, h->mesh_shaded.p, m->triangles.p,
                         m->n_triangles, mp, cam.width, cam.height, h->min_occlusion_depth, h->max_occlusion_depth, tiles_x, 0u, h->sp_keys[0].p,
                         h->sp_vals[0].p, h->sp_counter.p, (unsigned)std::min<size_t>(capacity, 0xFFFFFFFFu));
      read_back(h, &n_pairs, h->sp_counter.p, sizeof n_pairs);
      if ((size_t)n_pairs <= capacity) break;
      capacity = n_pairs;
    }
    E3D_HIP(hipMemsetAsync(h->tile_start.p, 0, sizeof(unsigned) * n_tiles, s));
    E3D_HIP(hipMemsetAsync(h->tile_end.p, 0, sizeof(unsigned) * n_tiles, s));
    if (n_pairs) {
      int bits = 1;
      while (((size_t)1 << bits) < n_tiles) ++bits;
      sort_pairs_u32_u32(h->sp_keys[0].p, h->sp_keys[1].p, h->sp_vals[0].p, h->sp_vals[1].p, n_pairs, bits, h->sort_temp, s);
      hipLaunchKernelGGL(k_tile_ranges, dim3(nblk(n_pairs)), dim3(kBlock), 0, s, h->sp_keys[1].p, (size_t)n_pairs, h->tile_start.p, h->tile_end.p);
    }
    DevBuf<float>& target = first ? D : h->ztmp_f;
    target.reserve(px);
    hipLaunchKernelGGL(k_mesh_tiles, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, h->mesh_projected.p, h->mesh_shaded.p, m->triangles.p, mp,
                       h->sp_vals[1].p,
                       h->tile_start.p, h->tile_end.p, tiles_x, cam.width, cam.height, h->min_occlusion_depth, h->max_occlusion_depth,
                       reinterpret_cast<unsigned*>(target.p));
    if (!first) hipLaunchKernelGGL(k_depth_merge, dim3(nblk(px)), dim3(kBlock), 0, s, D.p, h->ztmp_f.p, px);
    first = false;
  }
  if (h->mask_occlusion_boundaries) {
    E3D_HIP(hipMemcpyAsync(im.depth.p, D.p, sizeof(float) * px, hipMemcpyDeviceToDevice, s));
    // image position = global_T_image.translation() = -R^T t
    const float* R = im.pose.R; const float* t = im.pose.t;
    float3 pos;
    pos.x = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]); pos.y = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]); pos.z = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
    for (auto& m : h->meshes)
      if (m->n_edges)
        E3D_CAM_SWITCH(in.type, hipLaunchKernelGGL(k_mask_boundaries<M>, dim3(nblk(m->n_edges)), dim3(kBlock), 0, s, m->edges.p, m->n_edges,
                                                   m->vertices.p, m->normals.p, im.pose, pos, cam, h->prm.splat_radius, D.p, im.depth.p));
  }
}
}  // namespace e3d

extern "C" {

/* OcclusionGeometry::AddMesh / AddSplats (occlusion_geometry.cc:64-182): one more triangle mesh (vertices already in the
 * global frame).  compute_edges != 0 also extracts the edges used for masking occlusion boundaries (meshes: yes, splat
 * geometry: no). */
int e3d_reg_add_occlusion_mesh(e3d_reg_t* h, const float* vertices, size_t n_vertices, const uint32_t* triangles, size_t n_triangles,
                               int compute_edges) {
  R_TRYH
  if (!h || !vertices || !triangles || !n_vertices || !n_triangles) throw Error(E3D_ERR_INVALID, "empty mesh");
  if (n_triangles >= ((size_t)1 << 31) || n_vertices >= ((size_t)1 << 32)) throw Error(E3D_ERR_INVALID, "mesh too large");
  hipStream_t s = h->stream;
  std::unique_ptr<MeshDev> m(new MeshDev());
  m->n_vertices = n_vertices; m->n_triangles = n_triangles;
  DevBuf<float> tmp; tmp.reserve(3 * n_vertices);
  copy_in(tmp.p, vertices, sizeof(float) * 3 * n_vertices, s);
  m->vertices.reserve(n_vertices);
  hipLaunchKernelGGL(k_xyz_to_float4, dim3(nblk(n_vertices)), dim3(kBlock), 0, s, tmp.p, n_vertices, m->vertices.p);
  m->triangles.reserve(3 * n_triangles);
  copy_in(m->triangles.p, triangles, sizeof(unsigned) * 3 * n_triangles, s);
  rsync(h);
  for (size_t i = 0; i < 3 * n_triangles; ++i) if (triangles[i] >= n_vertices) throw Error(E3D_ERR_INDEX, "triangle refers to a missing vertex");
  if (compute_edges) {
    const size_t ne = 3 * n_triangles;
    DevBuf<unsigned long long> ka, kb;
    DevBuf<unsigned> va, vb, cnt;
    ka.reserve(ne); kb.reserve(ne); va.reserve(ne); vb.reserve(ne); cnt.reserve(1);
    m->normals.reserve(n_triangles);
    hipLaunchKernelGGL(k_face_normals, dim3(nblk(n_triangles)), dim3(kBlock), 0, s, m->vertices.p, m->triangles.p, n_triangles, m->normals.p,
                       ka.p, va.p);
    sort_pairs_u64_u32(ka.p, kb.p, va.p, vb.p, ne, 64, h->sort_temp, s);
    m->edges.reserve(ne);
    E3D_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_filter_edges, dim3(nblk(ne)), dim3(kBlock), 0, s, kb.p, vb.p, ne, m->vertices.p, m->normals.p, m->edges.p, cnt.p);
    unsigned n_edges = 0;
    copy_out(&n_edges, cnt.p, sizeof n_edges, s);
    rsync(h);
    m->n_edges = n_edges;
  }
  h->meshes.push_back(std::move(m));
  for (auto& kv : h->images) kv.second.depth_scale = -1;
  return (int)h->meshes.size();
  R_CATCH()
}
int e3d_reg_clear_occlusion_meshes(e3d_reg_t* h) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  h->meshes.clear();
  for (auto& kv : h->images) kv.second.depth_scale = -1;
  return 0;
  R_CATCH()
}
/* min_occlusion_depth / max_occlusion_depth (near / far plane of the mesh renderer) and mask_occlusion_boundaries of
 * OcclusionGeometry::RenderDepthMap (occlusion_geometry.h:80-86); defaults 0.05, 100, true */
int e3d_reg_set_occlusion_options(e3d_reg_t* h, float min_depth, float max_depth, int mask_occlusion_boundaries) {
  R_TRYH
  if (!h || !(min_depth > 0) || !(max_depth > min_depth)) throw Error(E3D_ERR_INVALID, "bad occlusion depth range");
  h->min_occlusion_depth = min_depth; h->max_occlusion_depth = max_depth; h->mask_occlusion_boundaries = mask_occlusion_boundaries != 0;
  return 0;
  R_CATCH()
}
int64_t e3d_reg_occlusion_edge_count(e3d_reg_t* h, int mesh_index) {
  R_TRYH
  if (!h || mesh_index < 0 || mesh_index >= (int)h->meshes.size()) throw Error(E3D_ERR_INDEX, "no such mesh");
  return (int64_t)h->meshes[mesh_index]->n_edges;
  R_CATCH()
}

int e3d_reg_set_splat_points(e3d_reg_t* h, const float* xyz, size_t n) {
  R_TRYH
  if (!h || (!xyz && n)) throw Error(E3D_ERR_INVALID, "null argument");
  DevBuf<float> tmp; tmp.reserve(3 * n);
  copy_in(tmp.p, xyz, sizeof(float) * 3 * n, h->stream);
  h->splat.reserve(n);
  // The splat points are only ever reduced to a depth map (a minimum: order free), so the library keeps them in Morton order of
  // their positions: neighbouring threads of k_splat_bin then hit neighbouring pixels, and the one atomicMin per point lands in
  // cache lines other lanes of the wave touch too (measured: 0.41 -> see DESIGN 10.3 ms per 10 M points of a 24 MP image with the
  // points in random order before).  E3D_REG_SPLAT_ORDER=0 keeps the caller's order.
  static const bool reorder = [] { const char* e = getenv("E3D_REG_SPLAT_ORDER"); return !(e && e[0] == '0'); }();
  if (reorder && n > 1 && n < ((size_t)1 << 32)) {
    hipStream_t s = h->stream;
    DevBuf<float> bbp, bbo;
    bbp.reserve(6 * (size_t)kMaxBboxBlocks); bbo.reserve(6);
    launch_bbox_aos(tmp.p, n, bbp.p, bbo.p, s);
    float bb[6];
    copy_out(bb, bbo.p, sizeof bb, s);
    rsync(h);
    float ext = 0.f;
    for (int k = 0; k < 3; ++k) ext = std::max(ext, bb[3 + k] - bb[k]);
    const float inv = (ext > 0.f && std::isfinite(ext)) ? 1023.f / ext : 0.f;
    for (int k = 0; k < 2; ++k) { h->sp_keys[k].reserve(n); h->sp_vals[k].reserve(n); }
    hipLaunchKernelGGL(k_morton_keys, dim3(nblk(n)), dim3(kBlock), 0, s, tmp.p, n, std::isfinite(bb[0]) ? bb[0] : 0.f, std::isfinite(bb[1]) ? bb[1] : 0.f,
                       std::isfinite(bb[2]) ? bb[2] : 0.f, inv, h->sp_keys[0].p, h->sp_vals[0].p);
    sort_pairs_u32_u32(h->sp_keys[0].p, h->sp_keys[1].p, h->sp_vals[0].p, h->sp_vals[1].p, n, 30, h->sort_temp, s);
    hipLaunchKernelGGL(k_gather_xyz_to_float4, dim3(nblk(n)), dim3(kBlock), 0, s, tmp.p, h->sp_vals[1].p, n, h->splat.p);
  } else {
    hipLaunchKernelGGL(k_xyz_to_float4, dim3(nblk(n)), dim3(kBlock), 0, h->stream, tmp.p, n, h->splat.p);
  }
  rsync(h);
  h->n_splat = n;
  return 0;
  R_CATCH()
}

int e3d_reg_render_depth(e3d_reg_t* h, int image_id, int image_scale, float* depth_out) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  ImageDev& im = get_image(h, image_id);
  if (!h->owns(image_id)) throw Error(E3D_ERR_INVALID, fmt("image %d belongs to rank %d", image_id, e3d_reg_image_owner(h, image_id)));
  const Intrin& in = h->intr.at(im.intrinsics_id);
  const int lvl = std::max(0, image_scale - in.min_image_scale);
  if (lvl >= (int)in.levels.size()) throw Error(E3D_ERR_INDEX, "image scale beyond the pyramid");
  const CamLevel& cam = in.levels[lvl];
  const size_t px = (size_t)cam.width * cam.height;
  im.depth.reserve(px);
  if (h->n_splat == 0 && !h->meshes.empty()) {
    double tris = 0;
    for (auto& m : h->meshes) tris += (double)m->n_triangles;
    KT kt(h, "depth.mesh_raster", tris);
    render_depth_meshes(h, im, in, cam);
  } else if (h->n_splat == 0) {
    // no occlusion geometry: everything is visible (occlusion_geometry.cc:272-281)
    hipLaunchKernelGGL(k_fill_f32, dim3(nblk(px)), dim3(kBlock), 0, h->stream, im.depth.p, px, INFINITY);
  } else {
    hipStream_t s = h->stream;
    const size_t n = h->n_splat;
    const int tiles_x = (int)div_up(cam.width, kTile), tiles_y = (int)div_up(cam.height, kTile);
    const size_t n_tiles = (size_t)tiles_x * tiles_y;
    if (cam.width > 65535 || cam.height > 65535) throw Error(E3D_ERR_INVALID, "image too large");
    h->rects.reserve(n);
    for (int k = 0; k < 2; ++k) { h->sp_keys[k].reserve(4 * n); h->sp_vals[k].reserve(4 * n); }
    h->sp_counter.reserve(1); h->tile_start.reserve(n_tiles); h->tile_end.reserve(n_tiles);
    E3D_HIP(hipMemsetAsync(h->sp_counter.p, 0, sizeof(unsigned), s));
    E3D_HIP(hipMemsetAsync(h->tile_start.p, 0, sizeof(unsigned) * n_tiles, s));
    E3D_HIP(hipMemsetAsync(h->tile_end.p, 0, sizeof(unsigned) * n_tiles, s));
    unsigned n_pairs = 0;
    static const bool separable = [] { const char* e = getenv("E3D_REG_MIN_FILTER"); return e && !strcmp(e, "separable"); }();
    const size_t zpx = (size_t)(cam.width + 2 * kSplatMax) * (cam.height + 2 * kSplatMax);
    h->zbuf.reserve(zpx);
    {
      KT kt(h, "depth.zbuffer_clear", (double)zpx);
      hipLaunchKernelGGL(k_fill_f32x4, dim3(nblk(zpx / 4 + 4)), dim3(kBlock), 0, s, reinterpret_cast<float*>(h->zbuf.p), zpx, INFINITY);
    }
    if (n) {
      {
        KT kt(h, "depth.splat_bin", (double)n);
        E3D_CAM_SWITCH(in.type, hipLaunchKernelGGL(k_splat_bin<M>, dim3(nblk(n)), dim3(kBlock), 0, s, h->splat.p, n, im.pose, cam,
                                                   h->prm.splat_radius, tiles_x, h->rects.p, h->sp_keys[0].p, h->sp_vals[0].p,
                                                   h->sp_counter.p, h->zbuf.p));
      }
      read_back(h, &n_pairs, h->sp_counter.p, sizeof n_pairs);
    }
    {
      KT kt(h, "depth.small_splat_tiles", (double)n_pairs);
      if (n_pairs) {
        int bits = 1;
        while (((size_t)1 << bits) < n_tiles) ++bits;
        sort_pairs_u32_u32(h->sp_keys[0].p, h->sp_keys[1].p, h->sp_vals[0].p, h->sp_vals[1].p, n_pairs, bits, h->sort_temp, s);
        hipLaunchKernelGGL(k_tile_ranges, dim3(nblk(n_pairs)), dim3(kBlock), 0, s, h->sp_keys[1].p, (size_t)n_pairs, h->tile_start.p,
                           h->tile_end.p);
      }
      if (n_pairs || separable)
      hipLaunchKernelGGL(k_splat_tiles, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, h->rects.p, h->sp_vals[1].p, h->tile_start.p,
                         h->tile_end.p, tiles_x, cam.width, cam.height, reinterpret_cast<unsigned*>(im.depth.p));
    }
    {
      KT kt(h, "depth.min_filter", (double)px);
      if (separable) {
        h->ztmp.reserve((size_t)cam.width * (cam.height + 2 * kSplatMax));
        hipLaunchKernelGGL(k_min_filter_h, dim3((unsigned)div_up(cam.width, kBlock), (unsigned)(cam.height + 2 * kSplatMax)), dim3(kBlock), 0, s,
                           h->zbuf.p, cam.width, cam.height, h->ztmp.p);
        hipLaunchKernelGGL(k_min_filter_v, dim3((unsigned)div_up(cam.width, kBlock), (unsigned)cam.height), dim3(kBlock), 0, s, h->ztmp.p,
                           cam.width, cam.height, reinterpret_cast<unsigned*>(im.depth.p));
      } else {
        const dim3 mf_grid((unsigned)div_up(cam.width, kMfW), (unsigned)div_up(cam.height, kMfH));
        if (n_pairs) hipLaunchKernelGGL(k_min_filter_tile<true>, mf_grid, dim3(kBlock), 0, s, h->zbuf.p, cam.width, cam.height, reinterpret_cast<unsigned*>(im.depth.p));
        else hipLaunchKernelGGL(k_min_filter_tile<false>, mf_grid, dim3(kBlock), 0, s, h->zbuf.p, cam.width, cam.height, reinterpret_cast<unsigned*>(im.depth.p));
      }
    }
  }
  if (depth_out) { copy_out(depth_out, im.depth.p, sizeof(float) * px, h->stream); rsync(h); }     // (else: every reader runs on the stream)
  im.depth_scale = image_scale;
  return 0;
  R_CATCH()
}

int64_t e3d_reg_observe(e3d_reg_t* h, int image_id, int point_scale, int image_scale, int border_size, const uint32_t* indices,
                        size_t n_indices) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  hipStream_t s = h->stream;
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  const bool all = (indices == nullptr);
  if (all && im.depth_scale != image_scale) throw Error(E3D_ERR_INVALID, "render the occlusion depth map of this image and scale first (e3d_reg_render_depth)");
  const size_t count = all ? S.n : n_indices;
  Obs& O = im.obs[point_scale];
  O.active = true;
  h->valid.reserve(count); h->tx.reserve(count); h->ty.reserve(count); h->ts.reserve(count);
  const unsigned* d_idx = nullptr;
  if (!all) { h->cand.reserve(count); copy_in(h->cand.p, indices, sizeof(unsigned) * count, s); d_idx = h->cand.p; }
  ObsParams q{};
  q.point_radius = S.radius; q.image_scale = image_scale; q.border = border_size;
  q.current_image_scale = h->prm.current_image_scale; q.image_scale_count = h->prm.image_scale_count;
  q.occlusion_threshold = h->prm.occlusion_depth_threshold; q.max_valid_intensity = h->prm.maximum_valid_intensity;
  q.check = all ? 1 : 0;
  O.n = 0;
  O.inten_valid = false;
  bool map_written = false;      // S.row_of_point written by the compaction (all-points pass)
  // the flags of the list the candidates came from still describe the result if every candidate stays an observation
  const bool flags_kept = !all && O.flags_src == (const void*)indices && O.flags_count == count;
  if (count) {
    unsigned* d_dropped = nullptr;
    if (!all) {
      h->d_total.reserve(2);
      d_dropped = reinterpret_cast<unsigned*>(h->d_total.p + 1);
      E3D_HIP(hipMemsetAsync(d_dropped, 0, sizeof(unsigned), s));
    }
    const size_t nb = div_up(count, kBlock);
    h->block_counts.reserve(nb); h->block_offsets.reserve(nb); h->block_d2.reserve(nb);
    {
      KT kt(h, all ? "obs.eval_all_points" : "obs.eval_listed_points", (double)count);
      E3D_CAM_SWITCH(image_model(h, im), hipLaunchKernelGGL(k_obs_eval<M>, dim3(nblk(count)), dim3(kBlock), 0, s, S.pts.p, d_idx, count,
                                                            im.pose, make_pyramid(h, im), all ? im.depth.p : nullptr, q, h->valid.p,
                                                            h->tx.p, h->ty.p, h->ts.p, d_dropped, all ? h->block_counts.p : nullptr,
                                                            all ? h->block_d2.p : nullptr));
    }
    unsigned dropped = 1;
    if (!all) {
      read_back(h, &dropped, d_dropped, sizeof dropped);
    }
    if (dropped == 0) {
      // every listed point is still an observation: the list and the freshly written positions ARE the compacted result
      O.n = count;
      std::swap(O.idx, h->cand); std::swap(O.x, h->tx); std::swap(O.y, h->ty); std::swap(O.s, h->ts);
    } else {
      h->chunk_sum.reserve(div_up(nb, 256) + 1); h->chunk_d2.reserve(div_up(nb, 256) + 1);
      h->d_total.reserve(2); h->d_total_d2.reserve(1);
      {
        KT kt(h, "obs.scan", (double)count);
        // (an all-points pass: k_obs_eval has written the per-block counts itself)
        launch_match_scan(all ? nullptr : h->valid.p, nullptr, count, h->block_counts.p, h->block_offsets.p, h->block_d2.p, h->chunk_sum.p,
                          h->chunk_d2.p, h->d_total.p, h->d_total_d2.p, s);
      }
      unsigned long long total = 0;
      read_back(h, &total, h->d_total.p, sizeof total);
      O.n = (size_t)total;
      O.idx.reserve(O.n); O.x.reserve(O.n); O.y.reserve(O.n); O.s.reserve(O.n);
      if (O.n) {
        KT kt(h, "obs.compact", (double)count);
        hipLaunchKernelGGL(k_obs_compact, dim3(nblk(count)), dim3(kBlock), 0, s, h->valid.p, h->tx.p, h->ty.p, h->ts.p, count,
                           h->block_offsets.p, O.idx.p, O.x.p, O.y.p, O.s.p, all ? S.row_of_point.p : nullptr);
        map_written = all;
      }
    }
  }
  if (flags_kept && O.n == count) {
    // same points in the same order as the list the flags were made for: flags and neighbour rows are unchanged
  } else {
    finish_observations(h, S, O, map_written);
  }
  return (int64_t)O.n;        // (the stream is not synchronised here: every reader of O runs on it)
  R_CATCH()
}

int e3d_reg_get_observations(e3d_reg_t* h, int image_id, int point_scale, uint32_t* idx, float* x, float* y, float* scale,
                             uint8_t* flags) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  Obs& O = get_obs(get_image(h, image_id), point_scale);
  if (idx) copy_out(idx, O.idx.p, sizeof(unsigned) * O.n, h->stream);
  if (x) copy_out(x, O.x.p, sizeof(float) * O.n, h->stream);
  if (y) copy_out(y, O.y.p, sizeof(float) * O.n, h->stream);
  if (scale) copy_out(scale, O.s.p, sizeof(float) * O.n, h->stream);
  if (flags) copy_out(flags, O.flags.p, O.n, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}

int e3d_reg_set_observations(e3d_reg_t* h, int image_id, int point_scale, size_t n, const uint32_t* idx, const float* x,
                             const float* y, const float* scale) {
  R_TRYH
  if (!h || (n && (!idx || !x || !y || !scale))) throw Error(E3D_ERR_INVALID, "null argument");
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = im.obs[point_scale];
  O.active = true;
  O.n = n;
  O.idx.reserve(n); O.x.reserve(n); O.y.reserve(n); O.s.reserve(n);
  copy_in(O.idx.p, idx, sizeof(unsigned) * n, h->stream); copy_in(O.x.p, x, sizeof(float) * n, h->stream);
  copy_in(O.y.p, y, sizeof(float) * n, h->stream); copy_in(O.s.p, scale, sizeof(float) * n, h->stream);
  O.inten_valid = false;
  finish_observations(h, S, O);
  rsync(h);
  return 0;
  R_CATCH()
}

int e3d_reg_pass1(e3d_reg_t* h, int image_id, int point_scale, float* intensities, float* j_intrinsics, float* j_pose) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = get_obs(im, point_scale);
  prepare_rows(h, im, S, O);
  const int I = cam_param_count(image_model(h, im));
  const int off_pose = 1 + I + (im.dependent() ? 6 : 0);
  const size_t stride = 4 * (size_t)rows4(local_unknowns(h, im));
  std::vector<float> rows(stride * O.n);
  copy_out(rows.data(), O.rows.p, sizeof(float) * stride * O.n, h->stream);
  rsync(h);
  for (size_t i = 0; i < O.n; ++i) {
    const float* r = rows.data() + stride * i;
    if (intensities) intensities[i] = r[0];
    if (j_intrinsics) for (int c = 0; c < I; ++c) j_intrinsics[(size_t)I * i + c] = r[1 + c];
    if (j_pose) for (int c = 0; c < 6; ++c) j_pose[6 * i + c] = r[off_pose + c];
  }
  return 0;
  R_CATCH()
}

}  // extern "C"
namespace e3d {
// The accumulation of one (image, point scale): pass 1, pass 2 and the reduction of the partials enqueued; the reg_slot(V) numbers
// {upper triangle of H by rows, b, fixed sum, variable sum, fixed count, variable count} land at d_out (device).  e3d_reg_accumulate
// reads them at once; the optimizer's driver enqueues every image of an Apply and reads all of them with one copy.
static int accumulate_enqueue(e3d_reg* h, int image_id, int point_scale, double* d_out, bool use_own_red) {
  hipStream_t s = h->stream;
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = get_obs(im, point_scale);
  if (!h->t_pass1) { h->t_pass1.reset(new EventTimer()); h->t_pass2.reset(new EventTimer()); }
  h->t_pass1->start(s);
  { KT kt(h, "accumulate.pass1", (double)O.n); prepare_rows(h, im, S, O); }
  h->t_pass1->stop(s);
  std::unique_ptr<KT> kt2(new KT(h, "accumulate.pass2", (double)O.n));
  // one resident round of workgroups (two of these 256-thread groups fit a CU): 512 on MI355X; measured 0.684 / 0.694 / 0.704 / 0.711 ms
  // for 512 / 1024 / 2048 / 4096 at the configs[3] shape
  static const int max_blocks = [] {
    if (const char* e = getenv("E3D_REG_PASS2_BLOCKS")) return std::max(8, atoi(e));
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1024;
    return 2 * cus;
  }();
  const int nb = (int)std::min<size_t>(std::max<size_t>(div_up(O.n, kBlock * 4), 1), (size_t)max_blocks);
  const int V = local_unknowns(h, im), slot = reg_slot(V);
  h->partial.reserve((size_t)nb * slot); h->red.reserve(slot);
  const RegWeights w{h->prm.robust_weighting_type, h->prm.robust_weighting_parameter, h->prm.fixed_residuals_weight,
                     h->prm.variable_residuals_weight};
  static const bool valu_pass2 = [] { const char* e = getenv("E3D_REG_PASS2"); return e && !strcmp(e, "valu"); }();
  int n_partials = nb;
  // matrix-core kernel: systems that would need several per-thread-accumulator launches (V > 10), default neighbour count
  static const bool mfma10 = getenv("E3D_REG_PASS2_MFMA10") != nullptr;     // experiment: the single-launch system on the matrix cores too
  h->t_pass2->start(s);
  if (!valu_pass2 && (V > 10 || mfma10) && h->prm.point_neighbor_count == 5) {
    n_partials = nb * (kBlock / kWave);       // one partial per wave
    h->partial.reserve((size_t)n_partials * slot);
    // Default: the f64 matrix instruction on exact products -- every term at least as accurate as the reference's
    // fl32(fl32(w J_i) J_j), all sums in f64 as in intrinsics_and_pose_optimizer.cc:1246-1247.  E3D_REG_PASS2 = tile32 / mfma32: the
    // narrower f32-chain kernels (opt-in: faster, ~1e-9 of the entry scale away).
    static const std::string p2 = [] { const char* e = getenv("E3D_REG_PASS2"); return std::string(e ? e : ""); }();
    static const bool tile32 = p2 == "tile32", mfma32g = p2 == "mfma32";
    static const bool mfma64 = !(tile32 || mfma32g);
#define E3D_PASS2M(V_)                                                                                                         \
  if (mfma64)                                                                                                                  \
    hipLaunchKernelGGL((k_reg_pass2_mfma<5, V_>), dim3(nb), dim3(kBlock), 0, s, O.rows.p, O.idx.p, O.flags.p, O.n, O.nrow.p,   \
                       S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p);                                          \
  else if constexpr ((V_) <= 16 || (V_) == 18) {                                                                               \
    if (mfma32g)                                                                                                               \
      hipLaunchKernelGGL((k_reg_pass2_mfma32<5, V_>), dim3(nb), dim3(kBlock), 0, s, O.rows.p, O.idx.p, O.flags.p, O.n, O.nrow.p, \
                         S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p);                                        \
    else                                                                                                                       \
      hipLaunchKernelGGL((k_reg_pass2_tile32<5, V_>), dim3(nb), dim3(kBlock), 0, s, O.rows.p, O.idx.p, O.flags.p, O.n, O.nrow.p, \
                         S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p);                                        \
  } else                                                                                                                       \
    hipLaunchKernelGGL((k_reg_pass2_mfma32<5, V_>), dim3(nb), dim3(kBlock), 0, s, O.rows.p, O.idx.p, O.flags.p, O.n, O.nrow.p, \
                       S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p)
    switch (V) {
      case 10: E3D_PASS2M(10); break;
      case 11: E3D_PASS2M(11); break;
      case 13: E3D_PASS2M(13); break;
      case 15: E3D_PASS2M(15); break;
      case 19: E3D_PASS2M(19); break;
      case 14: E3D_PASS2M(14); break;
      case 16: E3D_PASS2M(16); break;
      case 17: E3D_PASS2M(17); break;
      case 18: E3D_PASS2M(18); break;
      case 20: E3D_PASS2M(20); break;
      case 24: E3D_PASS2M(24); break;
      default: throw Error(E3D_ERR_INVALID, "unsupported local system size");
    }
#undef E3D_PASS2M
  } else {
#define E3D_PASS2(V_, R0_, R1_, B_)                                                                                              \
  hipLaunchKernelGGL((k_reg_pass2<8, V_, R0_, R1_, B_>), dim3(nb), dim3(kBlock), 0, s, O.rows.p, O.idx.p, O.flags.p, O.n, O.nrow.p, \
                     h->prm.point_neighbor_count, S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p)
  switch (V) {     // E3D_REG_PASS2=valu: per-thread accumulators; row ranges chosen so that every launch keeps <= 75 of them
    case 9: E3D_PASS2(9, 0, 9, true); break;
    case 10: E3D_PASS2(10, 0, 10, true); break;
    case 13: E3D_PASS2(13, 0, 4, true); E3D_PASS2(13, 4, 13, false); break;
    case 15: E3D_PASS2(15, 0, 3, true); E3D_PASS2(15, 3, 8, false); E3D_PASS2(15, 8, 15, false); break;
    case 19: E3D_PASS2(19, 0, 2, true); E3D_PASS2(19, 2, 6, false); E3D_PASS2(19, 6, 11, false); E3D_PASS2(19, 11, 19, false); break;
    case 11: E3D_PASS2(11, 0, 5, true); E3D_PASS2(11, 5, 11, false); break;
    case 14: E3D_PASS2(14, 0, 4, true); E3D_PASS2(14, 4, 14, false); break;
    case 16: E3D_PASS2(16, 0, 3, true); E3D_PASS2(16, 3, 8, false); E3D_PASS2(16, 8, 16, false); break;
    case 17: E3D_PASS2(17, 0, 3, true); E3D_PASS2(17, 3, 8, false); E3D_PASS2(17, 8, 17, false); break;
    case 18: E3D_PASS2(18, 0, 3, true); E3D_PASS2(18, 3, 7, false); E3D_PASS2(18, 7, 18, false); break;
    case 20: E3D_PASS2(20, 0, 2, true); E3D_PASS2(20, 2, 6, false); E3D_PASS2(20, 6, 11, false); E3D_PASS2(20, 11, 20, false); break;
    case 24: E3D_PASS2(24, 0, 2, true); E3D_PASS2(24, 2, 5, false); E3D_PASS2(24, 5, 9, false); E3D_PASS2(24, 9, 14, false);
             E3D_PASS2(24, 14, 24, false); break;
    default: throw Error(E3D_ERR_INVALID, "unsupported local system size");
  }
#undef E3D_PASS2
  }
  h->t_pass2->stop(s);
  kt2.reset();
  hipLaunchKernelGGL(k_reg_reduce, dim3(slot), dim3(kWave), 0, s, h->partial.p, n_partials, slot, use_own_red ? h->red.p : d_out);
  h->pass_observations += (double)O.n; h->pass_calls += 1;
  return V;
}

// the numbers of accumulate_enqueue as the dense V x V upper triangle, b, sums and counts
static void accumulate_unpack(int V, const double* r, double* H, double* b, double sums[2], int64_t counts[2]) {
  const int NH = reg_h(V);
  std::fill(H, H + V * V, 0.0);
  int e = 0;
  for (int i = 0; i < V; ++i) for (int j = i; j < V; ++j) H[i * V + j] = r[e++];
  for (int i = 0; i < V; ++i) b[i] = r[NH + i];
  sums[0] = r[NH + V]; sums[1] = r[NH + V + 1];
  counts[0] = (int64_t)r[NH + V + 2]; counts[1] = (int64_t)r[NH + V + 3];
}
}  // namespace e3d
extern "C" {

int e3d_reg_accumulate(e3d_reg_t* h, int image_id, int point_scale, double* H, double* b, double sums[2], int64_t counts[2]) {
  R_TRYH
  if (!h || !H || !b || !sums || !counts) throw Error(E3D_ERR_INVALID, "null argument");
  const int V = accumulate_enqueue(h, image_id, point_scale, nullptr, true);
  const int slot = reg_slot(V);
  std::vector<double> r(slot);
  read_back(h, r.data(), h->red.p, sizeof(double) * slot);
  h->pass1_ms += h->t_pass1->ms(); h->pass2_ms += h->t_pass2->ms();      // (the pass timers of the call just made: the stream is idle)
  accumulate_unpack(V, r.data(), H, b, sums, counts);
  return 0;
  R_CATCH()
}

int e3d_reg_kernel_times(e3d_reg_t* h, double out[4], int reset) {
  R_TRYH
  if (!h || !out) throw Error(E3D_ERR_INVALID, "null argument");
  out[0] = h->pass1_ms; out[1] = h->pass2_ms; out[2] = h->pass_observations; out[3] = h->pass_calls;
  if (reset) { h->pass1_ms = h->pass2_ms = h->pass_observations = h->pass_calls = 0; }
  return 0;
  R_CATCH()
}

int e3d_reg_profile(e3d_reg_t* h, int enable, char* out, size_t capacity) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  if (out && capacity) {
    std::string t;
    for (const auto& kv : h->profile_ms) t += kv.first + "=" + fmt("%.4f", kv.second) + ";";
    h->kflush();
    for (const auto& kv : h->kgroups) t += "k:" + kv.first + "=" + fmt("%.4f,%lld,%.0f", kv.second.ms, kv.second.calls, kv.second.units) + ";";
    std::snprintf(out, capacity, "%s", t.c_str());
  }
  if (enable != (h->profile_on ? 1 : 0)) { h->profile_ms.clear(); h->kflush(); h->kgroups.clear(); }
  h->profile_on = enable != 0;
  return 0;
  R_CATCH()
}

}  // extern "C"
namespace e3d {
// CostCalculator::ComputeCost of one (image, point scale): kernels enqueued, the four results {fixed sum, variable sum, fixed count,
// variable count} land in d_out[0..3] (device); the callers of the driver read many of them with one copy
static void cost_enqueue(e3d_reg* h, int image_id, int point_scale, double* d_out) {
  hipStream_t s = h->stream;
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = get_obs(im, point_scale);
  obs_intensities(h, im, O);
  // enough blocks for full occupancy (the loop is a chain of dependent gathers): 8 waves per SIMD on 256 CUs = 8192 blocks
  const int nb = (int)std::min<size_t>(std::max<size_t>(div_up(O.n, kBlock * 4), 1), 8192);
  h->partial.reserve((size_t)nb * 4);
  const RegWeights w{h->prm.robust_weighting_type, h->prm.robust_weighting_parameter, h->prm.fixed_residuals_weight,
                     h->prm.variable_residuals_weight};
  KT kt(h, "cost", (double)O.n);
  if (h->prm.point_neighbor_count == 5)
    hipLaunchKernelGGL(k_reg_cost<5>, dim3(nb), dim3(kBlock), 0, s, O.inten.p, O.idx.p, O.flags.p, O.n, O.nrow.p,
                       h->prm.point_neighbor_count, S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p);
  else
    hipLaunchKernelGGL(k_reg_cost<0>, dim3(nb), dim3(kBlock), 0, s, O.inten.p, O.idx.p, O.flags.p, O.n, O.nrow.p,
                       h->prm.point_neighbor_count, S.fixed_desc.p, S.var_desc.p, S.obs_counts.p, w, h->partial.p);
  hipLaunchKernelGGL(k_reg_reduce, dim3(4), dim3(kWave), 0, s, h->partial.p, nb, 4, d_out);
}
}  // namespace e3d
extern "C" {

int e3d_reg_cost(e3d_reg_t* h, int image_id, int point_scale, double sums[2], int64_t counts[2]) {
  R_TRYH
  if (!h || !sums || !counts) throw Error(E3D_ERR_INVALID, "null argument");
  h->red.reserve(4);
  cost_enqueue(h, image_id, point_scale, h->red.p);
  double r[4];
  read_back(h, r, h->red.p, sizeof r);
  sums[0] = r[0]; sums[1] = r[1]; counts[0] = (int64_t)r[2]; counts[1] = (int64_t)r[3];
  return 0;
  R_CATCH()
}

// ---- depth residuals -------------------------------------------------------------------------------------------------------------
/* Problem::SetFixedDepthMaps for one image (problem.cc:593-595): one f32 depth map per pyramid level of the image's camera
 * (level_depths[l]: width_l x height_l floats, host or device memory); level_depths == NULL removes them. */
int e3d_reg_set_depth_maps(e3d_reg_t* h, int image_id, const float* const* level_depths) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  ImageDev& im = get_image(h, image_id);
  const Intrin& in = h->intr.at(im.intrinsics_id);
  im.depth_maps.clear();
  if (!level_depths) return 0;
  im.depth_maps.resize(in.levels.size());
  for (size_t l = 0; l < in.levels.size(); ++l) {
    if (!level_depths[l]) throw Error(E3D_ERR_INVALID, "e3d_reg_set_depth_maps: every pyramid level needs a depth map");
    const size_t px = (size_t)in.levels[l].width * (size_t)in.levels[l].height;
    im.depth_maps[l].reserve(px);
    copy_in(im.depth_maps[l].p, level_depths[l], sizeof(float) * px, h->stream);
  }
  rsync(h);
  return 0;
  R_CATCH()
}

namespace e3d {
static DepthPyramid make_depth_pyramid(e3d_reg* h, const ImageDev& im, int image_id) {
  const Intrin& in = h->intr.at(im.intrinsics_id);
  if (im.depth_maps.size() != in.levels.size())
    throw Error(E3D_ERR_INVALID, fmt("depth residuals are enabled but image %d has no depth maps (e3d_reg_set_depth_maps)", image_id));
  if (im.dependent())     // intrinsics_and_pose_optimizer.cc:1199-1207: LOG(FATAL) << "Not implemented yet"
    throw Error(E3D_ERR_INVALID, "depth residuals for the non-reference images of a rig are not implemented (nor are they in the reference)");
  DepthPyramid D{};
  for (size_t l = 0; l < in.levels.size(); ++l) D.map[l] = im.depth_maps[l].p;
  return D;
}
}  // namespace e3d

/* The depth residuals of one (image, point scale): normal equations of the V = I + 6 local unknowns [intrinsics, pose] (row-major V x V,
 * upper triangle), b, the sum of the robust residuals and their count (intrinsics_and_pose_optimizer.cc:747-757, 1150-1214, 1219-1296). */
int e3d_reg_depth_accumulate(e3d_reg_t* h, int image_id, int point_scale, double* H, double* b, double* sum, int64_t* count) {
  R_TRYH
  if (!h || !H || !b || !sum || !count) throw Error(E3D_ERR_INVALID, "null argument");
  hipStream_t s = h->stream;
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = get_obs(im, point_scale);
  const DepthPyramid D = make_depth_pyramid(h, im, image_id);
  const int model = image_model(h, im);
  const int V = h->intr.at(im.intrinsics_id).n_params + 6, NH = reg_h(V), slot = reg_slot(V);
  O.drows.reserve((size_t)rows4(V) * std::max<size_t>(O.n, 1));
  const float4 quat = make_float4(im.pose_q.q.w, im.pose_q.q.x, im.pose_q.q.y, im.pose_q.q.z);
  if (O.n)
    E3D_CAM_SWITCH(model, hipLaunchKernelGGL(k_reg_depth_rows<M>, dim3(nblk(O.n)), dim3(kBlock), 0, s, S.pts.p, S.radius, im.pose, quat,
                                             make_pyramid(h, im), D, O.idx.p, O.x.p, O.y.p, O.s.p, O.n, O.drows.p));
  const int nb = (int)std::min<size_t>(std::max<size_t>(div_up(O.n, kBlock * 4), 1), 1024);
  h->partial.reserve((size_t)nb * slot); h->red.reserve(slot);
  const int rt = h->prm.depth_robust_weighting_type;
  const float rp = h->prm.depth_robust_weighting_parameter, dw = h->prm.depth_residuals_weight;
#define E3D_DEPTH(V_, R0_, R1_, B_) \
  hipLaunchKernelGGL((k_reg_depth_acc<V_, R0_, R1_, B_>), dim3(nb), dim3(kBlock), 0, s, O.drows.p, O.n, rt, rp, dw, h->partial.p)
  switch (V) {     // row ranges: <= 75 accumulators per launch, as for the per-thread colour kernel
    case 9: E3D_DEPTH(9, 0, 9, true); break;
    case 10: E3D_DEPTH(10, 0, 10, true); break;
    case 11: E3D_DEPTH(11, 0, 5, true); E3D_DEPTH(11, 5, 11, false); break;
    case 13: E3D_DEPTH(13, 0, 4, true); E3D_DEPTH(13, 4, 13, false); break;
    case 14: E3D_DEPTH(14, 0, 4, true); E3D_DEPTH(14, 4, 14, false); break;
    case 18: E3D_DEPTH(18, 0, 3, true); E3D_DEPTH(18, 3, 7, false); E3D_DEPTH(18, 7, 18, false); break;
    default: throw Error(E3D_ERR_INVALID, "unsupported local system size");
  }
#undef E3D_DEPTH
  hipLaunchKernelGGL(k_reg_reduce, dim3(slot), dim3(kWave), 0, s, h->partial.p, nb, slot, h->red.p);
  std::vector<double> r(slot);
  read_back(h, r.data(), h->red.p, sizeof(double) * slot);
  std::fill(H, H + V * V, 0.0);
  int e = 0;
  for (int i = 0; i < V; ++i) for (int j = i; j < V; ++j) H[i * V + j] = r[e++];
  for (int i = 0; i < V; ++i) b[i] = r[NH + i];
  *sum = r[NH + V]; *count = (int64_t)r[NH + V + 2];
  return 0;
  R_CATCH()
}

/* CostCalculator, depth part (cost_calculator.cc:221-245): sum of the robust depth residuals of the stored observations, and their count */
int e3d_reg_depth_cost(e3d_reg_t* h, int image_id, int point_scale, double* sum, int64_t* count) {
  R_TRYH
  if (!h || !sum || !count) throw Error(E3D_ERR_INVALID, "null argument");
  hipStream_t s = h->stream;
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = get_obs(im, point_scale);
  const DepthPyramid D = make_depth_pyramid(h, im, image_id);
  const int nb = (int)std::min<size_t>(std::max<size_t>(div_up(O.n, kBlock * 4), 1), 1024);
  h->partial.reserve((size_t)nb * 2); h->red.reserve(2);
  const float4 quat = make_float4(im.pose_q.q.w, im.pose_q.q.x, im.pose_q.q.y, im.pose_q.q.z);
  hipLaunchKernelGGL(k_reg_depth_cost, dim3(nb), dim3(kBlock), 0, s, S.pts.p, im.pose, quat, make_pyramid(h, im), D, O.idx.p, O.x.p, O.y.p,
                     O.s.p, O.n, h->prm.depth_robust_weighting_type, h->prm.depth_robust_weighting_parameter, h->partial.p);
  hipLaunchKernelGGL(k_reg_reduce, dim3(2), dim3(kWave), 0, s, h->partial.p, nb, 2, h->red.p);
  double r[2];
  read_back(h, r, h->red.p, sizeof r);
  *sum = r[0]; *count = (int64_t)r[1];
  return 0;
  R_CATCH()
}

int e3d_reg_color_begin(e3d_reg_t* h, int point_scale) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  PointScale& S = get_scale(h, point_scale);
  {
    KT kt(h, "color.clear", (double)S.n);
    E3D_HIP(hipMemsetAsync(S.var_desc.p, 0, sizeof(float) * S.n * h->prm.point_neighbor_count, h->stream));
    E3D_HIP(hipMemsetAsync(S.obs_counts.p, 0, sizeof(int) * S.n, h->stream));
  }
  rsync(h);
  return 0;
  R_CATCH()
}
}  // extern "C"
namespace e3d {
static void color_accumulate_enqueue(e3d_reg* h, int image_id, int point_scale) {
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  Obs& O = get_obs(im, point_scale);
  obs_intensities(h, im, O);
  KT kt(h, "color.accumulate", (double)O.n);
  if (O.n && h->prm.point_neighbor_count == 5)
    hipLaunchKernelGGL(k_color_accumulate<5>, dim3(nblk(O.n)), dim3(kBlock), 0, h->stream, O.inten.p, O.idx.p, O.flags.p, O.n,
                       O.nrow.p, 5, S.var_desc.p, S.obs_counts.p);
  else if (O.n)
    hipLaunchKernelGGL(k_color_accumulate<0>, dim3(nblk(O.n)), dim3(kBlock), 0, h->stream, O.inten.p, O.idx.p, O.flags.p, O.n,
                       O.nrow.p, h->prm.point_neighbor_count, S.var_desc.p, S.obs_counts.p);
}
}  // namespace e3d
extern "C" {
int e3d_reg_color_accumulate(e3d_reg_t* h, int image_id, int point_scale) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  color_accumulate_enqueue(h, image_id, point_scale);
  rsync(h);
  return 0;
  R_CATCH()
}
int e3d_reg_color_finish(e3d_reg_t* h, int point_scale) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  PointScale& S = get_scale(h, point_scale);
  {
    KT kt(h, "color.finish", (double)S.n);
    hipLaunchKernelGGL(k_color_finish, dim3(nblk(S.n)), dim3(kBlock), 0, h->stream, S.n, h->prm.point_neighbor_count, S.var_desc.p,
                       S.obs_counts.p);
  }
  rsync(h);
  return 0;
  R_CATCH()
}


// =====================================================================================================================================
// Optimizer driver (host): the alternation of opt::Optimizer::RunOnCurrentScale (src/opt/optimizer.cc:49-182) and
// IntrinsicsAndPoseOptimizer::Apply (src/opt/intrinsics_and_pose_optimizer.cc:48-259) over the kernel-level operators above.
// Images are visited in ascending image id (the reference iterates an unordered_map, whose order is unspecified).
// Non-rig images, colour residuals.
// =====================================================================================================================================
}  // extern "C"

namespace e3d {

struct RegState {
  std::map<int, Intrin> intr;
  std::map<int, SE3f> poses;
  std::map<int, RigState> rigs;
};

static RegState get_state(e3d_reg* h) {
  RegState st;
  st.intr = h->intr;
  st.rigs = h->rigs;
  for (auto& kv : h->images) st.poses[kv.first] = kv.second.pose_q;
  return st;
}
static void set_state(e3d_reg* h, const RegState& st) {
  h->intr = st.intr;
  h->rigs = st.rigs;
  for (auto& kv : h->images) {
    set_pose(kv.second, st.poses.at(kv.first));
    for (auto& o : kv.second.obs) o.second.rows_valid = false;
  }
}

static int best_available_scale(const e3d_reg* h, const Intrin& in) {
  // Intrinsics::best_available_image_scale(max(min_occlusion_check_image_scale (0), current_image_scale))
  const int want = std::max(0, (int)h->prm.current_image_scale);
  return std::min<int>(in.min_image_scale + (int)in.levels.size() - 1, std::max<int>(in.min_image_scale, want));
}

// Problem::ComputeCost (problem.cc:602-631): [fixed colour, variable colour, depth]
static double compute_cost_value(const e3d_reg* h, const double sums[3], const int64_t counts[3]) {
  const bool use_f = h->prm.fixed_residuals_weight > 0, use_v = h->prm.variable_residuals_weight > 0, use_d = h->prm.depth_residuals_weight > 0;
  double r = 0;
  if (use_f && counts[0] > 0) r += h->prm.fixed_residuals_weight * sums[0] / (double)counts[0];
  if (use_v && counts[1] > 0) r += h->prm.variable_residuals_weight * sums[1] / (double)counts[1];
  if (use_d && counts[2] > 0) r += h->prm.depth_residuals_weight * sums[2] / (double)counts[2];
  if ((!use_f && !use_v && !use_d) || (counts[0] == 0 && counts[1] == 0 && counts[2] == 0)) r = std::numeric_limits<float>::infinity();
  return r;
}

// VisibilityEstimator::CreateObservationsForAllImages + DetermineIfAllNeighborsAreObserved; with cache_observations the
// ObservationsCache::GetObservations path (observations_cache.cc:52-68): the cached point indices of each image are
// re-projected with the current state, without occlusion / mask / saturation tests (visibility_estimator.cc:140-168).
static void update_observations(e3d_reg* h, int border) {
  constexpr size_t kManyObservationsCount = 100;
  for (auto& kv : h->images) {
    if (!h->owns(kv.first)) continue;
    ImageDev& im = kv.second;
    const int scale = best_available_scale(h, h->intr.at(im.intrinsics_id));
    const bool cached = h->cache_observations;
    if (cached && !im.has_observed)
      throw Error(E3D_ERR_INVALID, fmt("no observed point indices for image %d (e3d_reg_determine_observed_indices / e3d_reg_set_observed_indices)", kv.first));
    {
      Phase ph(h, "observations.occlusion_depth_map");
      if (!cached && e3d_reg_render_depth(h, kv.first, scale, nullptr) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
    }
    Phase ph(h, "observations.visibility");
    for (auto& o : im.obs) { o.second.active = false; o.second.n = 0; }      // keep the device buffers
    bool had_many = false;
    for (auto it = h->scales.rbegin(); it != h->scales.rend(); ++it) {
      int64_t n;
      if (cached) {
        auto ci = im.observed.find(it->first);
        const size_t nv = (ci == im.observed.end()) ? 0 : ci->second.second;
        static const unsigned dummy = 0;
        n = e3d_reg_observe(h, kv.first, it->first, scale, border, nv ? ci->second.first.p : &dummy, nv);
      } else {
        n = e3d_reg_observe(h, kv.first, it->first, scale, border, nullptr, 0);
      }
      if (n < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
      if ((size_t)n > kManyObservationsCount) had_many = true;
      else if (n == 0 && had_many) break;
    }
  }
}

static void color_update(e3d_reg* h) {
  for (auto& sc : h->scales) {
    if (e3d_reg_color_begin(h, sc.first) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
    for (auto& kv : h->images)
      if (h->owns(kv.first) && has_obs(kv.second, sc.first)) color_accumulate_enqueue(h, kv.first, sc.first);     // (one stream: no sync between images)
    // the exchange step of (B): descriptor sums and observation counts over all images = over all ranks
    allreduce_device(h, sc.second.var_desc.p, sc.second.n * (size_t)h->prm.point_neighbor_count, 0);
    allreduce_device(h, sc.second.obs_counts.p, sc.second.n, 1);
    if (e3d_reg_color_finish(h, sc.first) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
  }
}

static void reduce_sums(e3d_reg* h, double sums[3], int64_t counts[3]) {
  if (h->world <= 1) return;
  double buf[6] = {sums[0], sums[1], sums[2], (double)counts[0], (double)counts[1], (double)counts[2]};      // counts < 2^53: exact in f64
  allreduce_host(h, buf, 6);
  for (int i = 0; i < 3; ++i) { sums[i] = buf[i]; counts[i] = (int64_t)buf[3 + i]; }
}

// depth residuals are in use: every image needs its depth maps, and none may be a dependent rig image (as in the reference)
static bool depth_in_use(const e3d_reg* h) { return h->prm.depth_residuals_weight > 0; }

// CostCalculator::ComputeCost over the stored observations
static double total_cost(e3d_reg* h) {
  double sums[3] = {0, 0, 0};
  int64_t counts[3] = {0, 0, 0};
  // the colour costs of all (image, scale) pairs are enqueued back to back and read with one copy; summed in the loop's order
  std::vector<std::pair<int, int>> jobs;
  for (auto& kv : h->images)
    for (auto& sc : h->scales)
      if (h->owns(kv.first) && has_obs(kv.second, sc.first)) jobs.emplace_back(kv.first, sc.first);
  h->red_all.reserve(4 * std::max<size_t>(jobs.size(), 1));
  for (size_t j = 0; j < jobs.size(); ++j) cost_enqueue(h, jobs[j].first, jobs[j].second, h->red_all.p + 4 * j);
  std::vector<double> r(4 * jobs.size());
  read_back(h, r.data(), h->red_all.p, sizeof(double) * r.size());
  for (size_t j = 0; j < jobs.size(); ++j) {
    sums[0] += r[4 * j]; sums[1] += r[4 * j + 1]; counts[0] += (int64_t)r[4 * j + 2]; counts[1] += (int64_t)r[4 * j + 3];
    if (depth_in_use(h)) {
      double sd; int64_t cd;
      if (e3d_reg_depth_cost(h, jobs[j].first, jobs[j].second, &sd, &cd) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
      sums[2] += sd; counts[2] += cd;
    }
  }
  reduce_sums(h, sums, counts);
  if (counts[0] == 0 && counts[1] == 0 && counts[2] == 0) return std::numeric_limits<double>::infinity();
  return compute_cost_value(h, sums, counts);
}

// The normal equations of IntrinsicsAndPoseOptimizer::Apply at the current state over the current observations
// (intrinsics_and_pose_optimizer.cc:55-185): variable indices, H and b in arrow form, the residual sums and counts behind
// "Initial residual", and the visibility lists (device copies of the observed point indices) the LM tries re-project.
struct NormalSystem {
  std::map<int, int> intr_index, image_index, rig_index;
  int V = 0;
  ArrowSystem Hb;
  double sums[3] = {0, 0, 0};
  int64_t counts[3] = {0, 0, 0};
  std::map<int, std::map<int, std::pair<DevBuf<unsigned>*, size_t>>> vis;
};

static void accumulate_system(e3d_reg* h, NormalSystem& N) {
  hipStream_t s = h->stream;
  // CountAndIndexVariables: [intrinsics blocks][6 per image]
  std::map<int, int>&intr_index = N.intr_index, &image_index = N.image_index, &rig_index = N.rig_index;
  intr_index.clear(); image_index.clear(); rig_index.clear(); N.vis.clear();
  int V = 0;
  for (auto& kv : h->intr) { intr_index[kv.first] = V; V += kv.second.n_params; }
  for (auto& kv : h->rigs) { rig_index[kv.first] = V; V += 6 * ((int)kv.second.image_T_rig.size() - 1); }   // reference camera excluded (:455-460)
  for (auto& kv : h->images) {                                                                             // dependent rig images share the
    if (kv.second.dependent()) continue;                                                                   // reference image's pose (:461-472)
    image_index[kv.first] = V; V += 6;
  }
  N.V = V;
  // H and b in arrow form (e3d_math.hpp): shared block = intrinsics + rig extrinsics, then one 6 x 6 block per pose
  const int n_shared = V - 6 * (int)image_index.size();
  ArrowSystem& Hb = N.Hb;
  Hb.reset(n_shared, (int)image_index.size());
  double* sums = N.sums;
  int64_t* counts = N.counts;
  for (int i = 0; i < 3; ++i) { sums[i] = 0; counts[i] = 0; }
  // visibility lists = observed point indices of the current observations (device copies)
  auto& vis = N.vis;
  // Every (image, scale) of the Apply is enqueued back to back and its numbers read with ONE copy (one host round trip per image
  // before: the device idled between images); then the blocks are added in the loop's order -- the same sums in the same order.
  struct Job { int image_id, scale, Vl; size_t off; };
  std::vector<Job> jobs;
  size_t total = 0;
  for (auto& kv : h->images) {
    if (!h->owns(kv.first)) continue;
    ImageDev& im = kv.second;
    for (auto& sc : h->scales) {
      if (!has_obs(im, sc.first)) continue;
      const int Vl = local_unknowns(h, im);
      jobs.push_back({kv.first, sc.first, Vl, total});
      total += (size_t)reg_slot(Vl);
    }
  }
  h->acc_all.reserve(std::max<size_t>(total, 1));
  {
    Phase ph(h, "apply.accumulate");
    for (const Job& j : jobs) {
      ImageDev& im = h->images.at(j.image_id);
      Obs& O = im.obs.at(j.scale);
      DevBuf<unsigned>* buf = &im.vis[j.scale];
      buf->reserve(O.n);
      if (O.n) E3D_HIP(hipMemcpyAsync(buf->p, O.idx.p, sizeof(unsigned) * O.n, hipMemcpyDeviceToDevice, s));
      vis[j.image_id][j.scale] = {buf, O.n};
      O.flags_src = O.n ? (const void*)buf->p : nullptr; O.flags_count = O.n;     // flags / nrow belong to exactly this list
      if (accumulate_enqueue(h, j.image_id, j.scale, h->acc_all.p + j.off, false) != j.Vl) throw Error(E3D_ERR_INVALID, "accumulate: local system size");
    }
  }
  std::vector<double> all(total);
  {
    Phase ph(h, "apply.accumulate");
    read_back(h, all.data(), h->acc_all.p, sizeof(double) * total);
  }
  for (const Job& j : jobs) {
    ImageDev& im = h->images.at(j.image_id);
    const int I = h->intr.at(im.intrinsics_id).n_params;
    const bool dep = im.dependent();
    const int ii = intr_index.at(im.intrinsics_id);
    const int pi = image_index.at(dep ? im.ref_image_id : j.image_id);
    const int ri = dep ? rig_index.at(im.rig_id) + 6 * (im.camera_index - 1) : -1;
    const int Vl = j.Vl;
    std::vector<double> Hl((size_t)Vl * Vl), bl(Vl);
    double s2[2]; int64_t c2[2];
    accumulate_unpack(Vl, all.data() + j.off, Hl.data(), bl.data(), s2, c2);
    sums[0] += s2[0]; sums[1] += s2[1]; counts[0] += c2[0]; counts[1] += c2[1];
    // scatter the local [intrinsics(I), (rig extrinsics(6),) pose(6)] block (AccumulateOnHAndB's block updates)
    auto gidx = [&](int l) { return l < I ? ii + l : ((dep && l < I + 6) ? ri + (l - I) : pi + (l - (Vl - 6))); };
    for (int r = 0; r < Vl; ++r) {
      for (int c = r; c < Vl; ++c)
        if (!Hb.add(gidx(r), gidx(c), Hl[(size_t)r * Vl + c])) throw Error(E3D_ERR_INVALID, "normal equations: entry outside the arrow pattern");
      Hb.b[gidx(r)] += bl[r];
    }
    if (depth_in_use(h)) {
      // depth residuals of every observation (intrinsics_and_pose_optimizer.cc:747-757); [intrinsics(I), pose(6)] block
      std::vector<double> Hd((size_t)(I + 6) * (I + 6)), bd(I + 6);
      double sd; int64_t cd;
      if (e3d_reg_depth_accumulate(h, j.image_id, j.scale, Hd.data(), bd.data(), &sd, &cd) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
      sums[2] += sd; counts[2] += cd;
      auto didx = [&](int l) { return l < I ? ii + l : pi + (l - I); };
      for (int r = 0; r < I + 6; ++r) {
        for (int c = r; c < I + 6; ++c)
          if (!Hb.add(didx(r), didx(c), Hd[(size_t)r * (I + 6) + c])) throw Error(E3D_ERR_INVALID, "normal equations: entry outside the arrow pattern");
        Hb.b[didx(r)] += bd[r];
      }
    }
  }
  E3D_HIP(hipStreamSynchronize(s));
  if (h->world > 1) {                       // one exchange per Apply: [non-zero blocks of H, b, sums, counts] -- block-sparse:
    const size_t np = Hb.packed_size();     // s^2 + (6 s + 36 + 6) per pose instead of V^2 (512 images: 0.5 MB instead of 76 MB)
    std::vector<double> buf(np + 6);
    Hb.pack(buf.data());
    double* tail = buf.data() + np;
    for (int i = 0; i < 3; ++i) { tail[i] = sums[i]; tail[3 + i] = (double)counts[i]; }
    allreduce_host(h, buf.data(), buf.size());
    Hb.unpack(buf.data());
    for (int i = 0; i < 3; ++i) { sums[i] = tail[i]; counts[i] = (int64_t)tail[3 + i]; }
  }
}

// IntrinsicsAndPoseOptimizer::Apply.  `ready`: the normal equations of exactly this state and these observations, accumulated
// already (e3d_reg_run_on_current_scale takes an iteration's cost from them, see there); else they are accumulated here.
static void apply_update(e3d_reg* h, bool print, bool* applied_update, float* lambda, float* max_change, NormalSystem* ready = nullptr) {
  NormalSystem own;
  if (!ready) accumulate_system(h, own);
  NormalSystem& N = ready ? *ready : own;
  std::map<int, int>&intr_index = N.intr_index, &image_index = N.image_index, &rig_index = N.rig_index;
  const int V = N.V;
  ArrowSystem& Hb = N.Hb;
  auto& vis = N.vis;
  const double* sums = N.sums;
  const int64_t* counts = N.counts;
  const double initial_residual = compute_cost_value(h, sums, counts);
  if (print)
    printf("    Initial residual: %g (#fixed residuals: %lld, #variable residuals: %lld)\n", initial_residual, (long long)counts[0], (long long)counts[1]);

  const RegState old_state = get_state(h);
  *applied_update = false;
  // Small systems: the reference's dense pivoted LDLT, operation for operation.  Large ones (V > 384, i.e. more than ~60 images):
  // Schur complement on the shared block -- same solution to f64 rounding, O(images) instead of O(V^3) host work.
  // E3D_REG_SOLVER=dense|arrow forces one of them (tests).
  static const int forced_solver = [] { const char* e = getenv("E3D_REG_SOLVER"); return !e ? 0 : (!strcmp(e, "dense") ? 1 : (!strcmp(e, "arrow") ? 2 : 0)); }();
  const bool dense_solve = forced_solver == 1 || (forced_solver == 0 && V <= 384);
  std::vector<double> H, Hlm, x(V), W;
  if (dense_solve) Hb.to_dense(H);
  const std::vector<double>& b = Hb.b;
  std::vector<int> perm;
  constexpr int kNumLMTries = 10;
  for (int lm = 0; lm < kNumLMTries; ++lm) {
    {
      Phase ph(h, "apply.host_solve");
      if (dense_solve) {
        Hlm = H;
        for (int i = 0; i < V; ++i) Hlm[(size_t)i * V + i] *= (1 + (*lambda));       // multiplicative damping (:206)
        ldlt_solve_upper(Hlm.data(), V, b.data(), x.data(), W, perm);
      } else {
        Hb.solve((double)(1 + (*lambda)), x.data());
      }
    }
    // CreateDeltaState(-x)
    std::unique_ptr<Phase> ph_state(new Phase(h, "apply.trial_state_and_camera_pyramids"));
    RegState trial = old_state;
    for (auto& kv : trial.intr) {
      Intrin& in = kv.second;
      const int base = intr_index.at(kv.first);
      for (int i = 0; i < in.n_params; ++i) in.params[i] += -1 * x[base + i];      // float += double (intrinsics.cc:71-73)
      build_model_pyramid(h, in, (int)in.levels.size());
    }
    for (auto& kv : trial.rigs)                                                                    // Rig::Update (rig.cc:9-23)
      for (size_t c = 1; c < kv.second.image_T_rig.size(); ++c)
        kv.second.image_T_rig[c] = se3_apply_update(&x[rig_index.at(kv.first) + 6 * ((int)c - 1)], old_state.rigs.at(kv.first).image_T_rig[c]);
    for (auto& kv : trial.poses)
      if (image_index.count(kv.first)) kv.second = se3_apply_update(&x[image_index.at(kv.first)], old_state.poses.at(kv.first));   // exp(-x) * T
    // ComputeResidualForState with the visibility lists fixed
    set_state(h, trial);
    compose_rig_poses(h);
    ph_state.reset();
    double ts[3] = {0, 0, 0}; int64_t tc[3] = {0, 0, 0};
    constexpr size_t kManyObservationsCount = 100;
    std::vector<std::pair<int, int>> trial_jobs;
    const size_t trial_cap = h->images.size() * h->scales.size();        // (red_all must not move while results are pending)
    h->red_all.reserve(4 * std::max<size_t>(trial_cap, 1));
    for (auto& kv : h->images) {
      if (!h->owns(kv.first)) continue;
      const int scale = best_available_scale(h, h->intr.at(kv.second.intrinsics_id));
      bool had_many = false;
      std::vector<int> done;
      std::unique_ptr<Phase> ph_obs(new Phase(h, "apply.trial_reprojection"));
      for (auto it = h->scales.rbegin(); it != h->scales.rend(); ++it) {
        auto vi = vis[kv.first].find(it->first);
        const size_t nv = (vi == vis[kv.first].end()) ? 0 : vi->second.second;
        const unsigned* ptr = nv ? vi->second.first->p : nullptr;
        static const unsigned dummy = 0;
        const int64_t n = e3d_reg_observe(h, kv.first, it->first, scale, 1, ptr ? ptr : &dummy, nv);
        if (n < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
        done.push_back(it->first);
        if ((size_t)n > kManyObservationsCount) had_many = true;
        else if (n == 0 && had_many) break;
      }
      ph_obs.reset();
      Phase ph_cost(h, "apply.trial_cost");
      for (auto& sc : h->scales) {
        // scales skipped by the early-out have empty observation vectors in the reference
        if (std::find(done.begin(), done.end(), sc.first) == done.end()) continue;
        trial_jobs.emplace_back(kv.first, sc.first);
        h->red_all.reserve(4 * std::max<size_t>(trial_cap, trial_jobs.size()));
        cost_enqueue(h, kv.first, sc.first, h->red_all.p + 4 * (trial_jobs.size() - 1));
      }
    }
    {
      // every trial cost was enqueued behind its image's re-projection: one copy, summed in the loop's order
      Phase ph_cost(h, "apply.trial_cost");
      std::vector<double> r(4 * trial_jobs.size());
      read_back(h, r.data(), h->red_all.p, sizeof(double) * r.size());
      for (size_t j = 0; j < trial_jobs.size(); ++j) {
        ts[0] += r[4 * j]; ts[1] += r[4 * j + 1]; tc[0] += (int64_t)r[4 * j + 2]; tc[1] += (int64_t)r[4 * j + 3];
        if (depth_in_use(h)) {
          double sd; int64_t cd;
          if (e3d_reg_depth_cost(h, trial_jobs[j].first, trial_jobs[j].second, &sd, &cd) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
          ts[2] += sd; tc[2] += cd;
        }
      }
    }
    reduce_sums(h, ts, tc);
    const double new_residual = compute_cost_value(h, ts, tc);
    if (new_residual < initial_residual || lm == kNumLMTries - 1) {      // kAlwaysApplyLastUpdate
      if (print) printf("    LM update accepted, new residual: %g\n", new_residual);
      double mx = -std::numeric_limits<double>::infinity();
      for (int i = 0; i < V; ++i) mx = std::max(mx, x[i]);               // x.maxCoeff(): signed max [QUIRK]
      *max_change = (float)mx;
      *lambda = 0.5f * (*lambda);
      *applied_update = true;
      break;                                                             // state stays at `trial`
    } else {
      *lambda = 2.f * (*lambda);
      if (print) printf("    [%d of %d] LM update rejected (bad residual: %g), lambda increased to %g\n", lm + 1, kNumLMTries, new_residual, (double)*lambda);
      set_state(h, old_state);
    }
  }
}

}  // namespace e3d

extern "C" {

int e3d_reg_set_shard(e3d_reg_t* h, int rank, int world_size, e3d_allreduce_fn allreduce, e3d_allreduce_device_fn allreduce_device,
                      void* user) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  if (world_size < 1 || rank < 0 || rank >= world_size) throw Error(E3D_ERR_INVALID, "bad rank / world size");
  if (world_size > 1 && (!allreduce || !allreduce_device)) throw Error(E3D_ERR_INVALID, "world_size > 1 needs both all-reduce callbacks");
  if (!h->images.empty()) throw Error(E3D_ERR_INVALID, "e3d_reg_set_shard must be called before images are set");
  h->rank = rank; h->world = world_size; h->allreduce = allreduce; h->allreduce_dev = allreduce_device; h->ar_user = user;
  return 0;
  R_CATCH()
}
int e3d_reg_set_comm(e3d_reg_t* h, e3d_comm_t* comm) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  if (!h->images.empty()) throw Error(E3D_ERR_INVALID, "e3d_reg_set_comm must be called before images are set");
  if (comm && comm->device != h->device) throw Error(E3D_ERR_INVALID, "e3d_reg_set_comm: communicator and handle live on different devices");
  h->comm = comm;
  h->rank = comm ? comm->rank : 0; h->world = comm ? comm->world : 1;
  h->allreduce = nullptr; h->allreduce_dev = nullptr; h->ar_user = nullptr;
  return 0;
  R_CATCH()
}
int e3d_reg_image_owner(e3d_reg_t* h, int image_id) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  return h->world <= 1 ? 0 : ((image_id % h->world) + h->world) % h->world;
  R_CATCH()
}

/* CreateMultiScalePointCloud, first part (multi_scale_point_cloud.cc:236-262): the radius range of every point over all
 * images of this rank -- observations at the best available scale of max(min_occlusion_check_image_scale (0),
 * current_image_scale) with occlusion, mask and saturation tests but no scale test, then the point radius that projects to
 * half a pixel.  min_radius starts at +inf, max_radius at -inf (points seen by no image keep those values). */
int e3d_reg_point_radius_minmax(e3d_reg_t* h, const float* xyz, size_t n, float* min_radius, float* max_radius) {
  R_TRYH
  if (!h || (n && (!xyz || !min_radius || !max_radius))) throw Error(E3D_ERR_INVALID, "null argument");
  hipStream_t s = h->stream;
  DevBuf<float> tmp, d_min, d_max;
  DevBuf<float4> pts;
  tmp.reserve(3 * n); pts.reserve(n); d_min.reserve(n); d_max.reserve(n);
  copy_in(tmp.p, xyz, sizeof(float) * 3 * n, s);
  hipLaunchKernelGGL(k_xyz_to_float4, dim3(nblk(n)), dim3(kBlock), 0, s, tmp.p, n, pts.p);
  hipLaunchKernelGGL(k_fill_f32, dim3(nblk(n)), dim3(kBlock), 0, s, d_min.p, n, INFINITY);
  hipLaunchKernelGGL(k_fill_f32, dim3(nblk(n)), dim3(kBlock), 0, s, d_max.p, n, -INFINITY);
  std::map<int, std::shared_ptr<DevBuf<float2>>> lookups;       // per intrinsics block: model(0)'s undistortion table
  for (auto& kv : h->images) {
    if (!h->owns(kv.first)) continue;
    ImageDev& im = kv.second;
    const Intrin& in = h->intr.at(im.intrinsics_id);
    const int scale = best_available_scale(h, in);
    if (e3d_reg_render_depth(h, kv.first, scale, nullptr) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
    const int lvl = std::max(0, scale - in.min_image_scale);
    const CamLevel& cam = in.levels[lvl];
    const CamLevel& cam_min = in.levels[0];
    auto& lk = lookups[im.intrinsics_id];
    if (!lk) {
      lk = std::make_shared<DevBuf<float2>>();
      const size_t px = (size_t)cam_min.width * cam_min.height;
      lk->reserve(px);
      E3D_CAM_SWITCH(in.type, hipLaunchKernelGGL(k_undistort_lookup<M>, dim3(nblk(px)), dim3(kBlock), 0, s, cam_min, lk->p));
    }
    RadiusParams rp{};
    rp.image_scale = scale; rp.min_image_scale = in.min_image_scale;
    rp.occlusion_threshold = h->prm.occlusion_depth_threshold; rp.max_valid_intensity = h->prm.maximum_valid_intensity;
    rp.min_scaling_factor = std::pow(2.0, -1.0 * (h->prm.image_scale_count - 1));
    const float4 quat = make_float4(im.pose_q.q.w, im.pose_q.q.x, im.pose_q.q.y, im.pose_q.q.z);
    if (n)
      E3D_CAM_SWITCH(in.type, hipLaunchKernelGGL(k_point_radius<M>, dim3(nblk(n)), dim3(kBlock), 0, s, pts.p, n, im.pose, quat, cam, cam_min,
                                                 lk->p, im.pix[lvl].p, im.has_mask[lvl] ? im.mask[lvl].p : nullptr,
                                                 (in.cam_mask && lvl < (int)in.cam_mask->size()) ? (*in.cam_mask)[lvl].p : nullptr, im.depth.p, rp,
                                                 d_min.p, d_max.p));
  }
  copy_out(min_radius, d_min.p, sizeof(float) * n, s);
  copy_out(max_radius, d_max.p, sizeof(float) * n, s);
  rsync(h);
  E3D_HIP(hipGetLastError());
  return 0;
  R_CATCH()
}

/* Problem::DeterminePointNeighbors (src/opt/problem.cc:706-786): the (candidates + 1) nearest neighbours of every point --
 * within its own scan if limit_to_same_scan -- from the GPU k-NN of path (A'), then libstdc++'s
 * std::shuffle(indices + 1, end, std::mt19937(0)) with ONE generator consumed in the reference's point order (scan by scan,
 * point by point), keeping the first neighbor_count. */
int e3d_determine_point_neighbors(const float* xyz, size_t n, const uint8_t* scan_indices, int scan_count, int limit_to_same_scan,
                                  int neighbor_count, int candidate_count, uint32_t* neighbor_indices) {
  R_TRY
  if ((!xyz && n) || !neighbor_indices || neighbor_count < 1 || candidate_count < neighbor_count) throw Error(E3D_ERR_INVALID, "bad argument");
  if (limit_to_same_scan && (!scan_indices || scan_count < 1)) throw Error(E3D_ERR_INVALID, "scan indices required");
  const int k = candidate_count + 1;
  std::mt19937 generator(/*seed*/ 0);
  std::vector<int> indices(k);
  const float vp[3] = {0.f, 0.f, 0.f};
  auto run = [&](const std::vector<float>& pts, const std::vector<size_t>* original, bool check_self) {
    const size_t m = pts.size() / 3;
    if (m < (size_t)k) throw Error(E3D_ERR_INVALID, fmt("a cloud of %zu points cannot provide %d neighbour candidates", m, candidate_count));
    std::vector<float> nrm(3 * m), curv(m);
    std::vector<int32_t> knn(m * (size_t)k);
    if (e3d_normals_knn(pts.data(), m, k, vp, nrm.data(), curv.data(), knn.data()) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
    for (size_t i = 0; i < m; ++i) {
      for (int j = 0; j < k; ++j) indices[j] = knn[i * k + j];
      if (check_self && indices[0] != (int)i) throw Error(E3D_ERR_INVALID, fmt("point %zu is not its own nearest neighbour (duplicate points)", i));
      std::shuffle(indices.begin() + 1, indices.end(), generator);
      const size_t o = original ? (*original)[i] : i;
      for (int j = 0; j < neighbor_count; ++j)
        neighbor_indices[o * neighbor_count + j] = (uint32_t)(original ? (*original)[indices[j + 1]] : (size_t)indices[j + 1]);
    }
  };
  if (limit_to_same_scan) {
    std::vector<std::vector<float>> clouds(scan_count);
    std::vector<std::vector<size_t>> orig(scan_count);
    for (size_t i = 0; i < n; ++i) {
      const int sidx = scan_indices[i];
      if (sidx >= scan_count) throw Error(E3D_ERR_INDEX, "scan index out of range");
      clouds[sidx].insert(clouds[sidx].end(), xyz + 3 * i, xyz + 3 * i + 3);
      orig[sidx].push_back(i);
    }
    for (int sidx = 0; sidx < scan_count; ++sidx)
      if (clouds[sidx].size() / 3 < (size_t)k) throw Error(E3D_ERR_INVALID, fmt("scan %d has fewer than %d points at this scale", sidx, k));
    for (int sidx = 0; sidx < scan_count; ++sidx) run(clouds[sidx], &orig[sidx], false);
  } else {
    run(std::vector<float>(xyz, xyz + 3 * n), nullptr, true);
  }
  return 0;
  R_CATCH()
}

// ---- f4: GroundTruthCreator -------------------------------------------------------------------------------------------------
int e3d_reg_set_scan_points(e3d_reg_t* h, const float* xyz, size_t n) {
  R_TRYH
  if (!h || (n && !xyz)) throw Error(E3D_ERR_INVALID, "null argument");
  if (n >= (size_t)1 << 31) throw Error(E3D_ERR_INVALID, "more than 2^31-1 scan points");
  h->n_scan = n;
  h->scan_pts.reserve(n); h->scan_counts.reserve(n);
  if (n) {
    DevBuf<float> raw;
    raw.reserve(3 * n);
    copy_in(raw.p, xyz, sizeof(float) * 3 * n, h->stream);
    hipLaunchKernelGGL(k_xyz_to_float4, dim3(nblk(n)), dim3(kBlock), 0, h->stream, raw.p, n, h->scan_pts.p);
    E3D_HIP(hipMemsetAsync(h->scan_counts.p, 0, sizeof(int) * n, h->stream));
    rsync(h);
  }
  return 0;
  R_CATCH()
}
static void scan_visibility(e3d_reg* h, int image_id, const uint8_t* mask, int excluded_flag, int mode, int min_count, int radius = 0) {
  ImageDev& im = get_image(h, image_id);
  const Intrin& in = h->intr.at(im.intrinsics_id);
  // RenderDepthMap(intrinsics, image, intrinsics.min_image_scale, ...) + intrinsics.model(0): the highest resolution
  if (e3d_reg_render_depth(h, image_id, in.min_image_scale, nullptr) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
  const CamLevel& cam = in.levels[0];
  const size_t px = (size_t)cam.width * cam.height;
  if (mask) { h->eval_mask.reserve(px); copy_in(h->eval_mask.p, mask, px, h->stream); }
  if (h->n_scan)
    E3D_CAM_SWITCH(in.type, hipLaunchKernelGGL(k_scan_visibility<M>, dim3(nblk(h->n_scan)), dim3(kBlock), 0, h->stream, h->scan_pts.p,
                                               h->n_scan, im.pose, cam, im.depth.p, h->prm.occlusion_depth_threshold,
                                               mask ? h->eval_mask.p : nullptr, excluded_flag, mode, min_count, h->scan_counts.p,
                                               h->gt_depth.p, radius));
}
int e3d_reg_count_scan_observations(e3d_reg_t* h, int image_id, const uint8_t* mask, int excluded_flag) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  scan_visibility(h, image_id, mask, excluded_flag, 0, 0);
  rsync(h);
  return 0;
  R_CATCH()
}
int e3d_reg_get_scan_observation_counts(e3d_reg_t* h, int32_t* counts) {
  R_TRYH
  if (!h || (!counts && h->n_scan)) throw Error(E3D_ERR_INVALID, "null argument");
  if (h->n_scan) copy_out(counts, h->scan_counts.p, sizeof(int) * h->n_scan, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}
int e3d_reg_set_scan_observation_counts(e3d_reg_t* h, const int32_t* counts) {
  R_TRYH
  if (!h || (!counts && h->n_scan)) throw Error(E3D_ERR_INVALID, "null argument");
  if (h->n_scan) copy_in(h->scan_counts.p, counts, sizeof(int) * h->n_scan, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}
int e3d_reg_ground_truth_depth(e3d_reg_t* h, int image_id, const uint8_t* mask, int excluded_flag, int min_count, float* gt_depth,
                               float* occlusion_depth) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  ImageDev& im = get_image(h, image_id);
  const CamLevel& cam = h->intr.at(im.intrinsics_id).levels[0];
  const size_t px = (size_t)cam.width * cam.height;
  h->gt_depth.reserve(px);
  hipLaunchKernelGGL(k_fill_f32, dim3(nblk(px)), dim3(kBlock), 0, h->stream, reinterpret_cast<float*>(h->gt_depth.p), px, INFINITY);
  scan_visibility(h, image_id, mask, excluded_flag, 1, min_count);
  if (gt_depth) copy_out(gt_depth, h->gt_depth.p, sizeof(float) * px, h->stream);
  if (occlusion_depth) copy_out(occlusion_depth, im.depth.p, sizeof(float) * px, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}

/* CreateGroundTruthForImage, scan rendering part (ground_truth_creator.cc:149,175-187): which scan point ends up on top of every pixel
 * when the visible points counted at least min_count times are painted as squares in point order.  winner[pixel] = point index + 1, 0 where
 * the image keeps its own colour. */
int e3d_reg_scan_rendering(e3d_reg_t* h, int image_id, const uint8_t* mask, int excluded_flag, int min_count, int point_radius,
                           uint32_t* winner) {
  R_TRYH
  if (!h || !winner) throw Error(E3D_ERR_INVALID, "null argument");
  if (point_radius < 0 || point_radius > 64) throw Error(E3D_ERR_INVALID, "scan_point_radius out of range [0, 64]");
  if (h->n_scan >= 0xFFFFFFFFull) throw Error(E3D_ERR_INVALID, "too many scan points for 32-bit indices");
  ImageDev& im = get_image(h, image_id);
  const CamLevel& cam = h->intr.at(im.intrinsics_id).levels[0];
  const size_t px = (size_t)cam.width * cam.height;
  h->gt_depth.reserve(px);
  E3D_HIP(hipMemsetAsync(h->gt_depth.p, 0, sizeof(unsigned) * px, h->stream));
  scan_visibility(h, image_id, mask, excluded_flag, 2, min_count, point_radius);
  copy_out(winner, h->gt_depth.p, sizeof(unsigned) * px, h->stream);
  rsync(h);
  return 0;
  R_CATCH()
}

int e3d_reg_update_observations(e3d_reg_t* h, int border_size) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  update_observations(h, border_size);
  return 0;
  R_CATCH()
}
// Optimizer::set_cache_observations (optimizer.h)
int e3d_reg_set_cache_observations(e3d_reg_t* h, int enabled) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  h->cache_observations = enabled != 0;
  return 0;
  R_CATCH()
}
// ObservationsCache::DetermineAndSaveObservedPointIndices (observations_cache.cc:104-158) without the files: the full
// visibility test at image scale 0, whose observed point indices become the cached lists.
int e3d_reg_determine_observed_indices(e3d_reg_t* h) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  const int old_scale = h->prm.current_image_scale;
  const bool old_cache = h->cache_observations;
  h->prm.current_image_scale = 0;
  h->cache_observations = false;
  try { update_observations(h, 1); } catch (...) { h->prm.current_image_scale = old_scale; h->cache_observations = old_cache; throw; }
  h->prm.current_image_scale = old_scale;
  h->cache_observations = old_cache;
  for (auto& kv : h->images) {
    if (!h->owns(kv.first)) continue;
    ImageDev& im = kv.second;
    for (auto& sc : h->scales) {
      auto& slot = im.observed[sc.first];
      slot.second = 0;
      if (!has_obs(im, sc.first)) continue;
      Obs& O = im.obs.at(sc.first);
      slot.first.reserve(O.n);
      if (O.n) E3D_HIP(hipMemcpyAsync(slot.first.p, O.idx.p, sizeof(unsigned) * O.n, hipMemcpyDeviceToDevice, h->stream));
      slot.second = O.n;
    }
    im.has_observed = true;
  }
  rsync(h);
  return 0;
  R_CATCH()
}
int64_t e3d_reg_get_observed_indices(e3d_reg_t* h, int image_id, int point_scale, uint64_t* indices) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  ImageDev& im = get_image(h, image_id);
  get_scale(h, point_scale);
  if (!im.has_observed) throw Error(E3D_ERR_INVALID, fmt("image %d has no observed point indices", image_id));
  auto ci = im.observed.find(point_scale);
  const size_t n = (ci == im.observed.end()) ? 0 : ci->second.second;
  if (indices && n) {
    std::vector<unsigned> tmp(n);
    copy_out(tmp.data(), ci->second.first.p, sizeof(unsigned) * n, h->stream);
    rsync(h);
    for (size_t i = 0; i < n; ++i) indices[i] = tmp[i];
  }
  return (int64_t)n;
  R_CATCH()
}
int e3d_reg_set_observed_indices(e3d_reg_t* h, int image_id, int point_scale, const uint64_t* indices, size_t count) {
  R_TRYH
  if (!h || (count && !indices)) throw Error(E3D_ERR_INVALID, "null argument");
  ImageDev& im = get_image(h, image_id);
  PointScale& S = get_scale(h, point_scale);
  std::vector<unsigned> tmp(count);
  for (size_t i = 0; i < count; ++i) {
    if (indices[i] >= S.n) throw Error(E3D_ERR_INDEX, fmt("observed point index %llu out of range (scale %d has %zu points)", (unsigned long long)indices[i], point_scale, S.n));
    tmp[i] = (unsigned)indices[i];
  }
  auto& slot = im.observed[point_scale];
  slot.first.reserve(count);
  if (count) copy_in(slot.first.p, tmp.data(), sizeof(unsigned) * count, h->stream);
  rsync(h);
  slot.second = count;
  im.has_observed = true;
  return 0;
  R_CATCH()
}
int e3d_reg_color_update(e3d_reg_t* h) {
  R_TRYH
  if (!h) throw Error(E3D_ERR_INVALID, "null handle");
  color_update(h);
  return 0;
  R_CATCH()
}
int e3d_reg_compute_cost(e3d_reg_t* h, double* cost) {
  R_TRYH
  if (!h || !cost) throw Error(E3D_ERR_INVALID, "null argument");
  *cost = total_cost(h);
  return 0;
  R_CATCH()
}
int e3d_reg_apply(e3d_reg_t* h, int print_progress, int* applied_update, float* lambda, float* max_change) {
  R_TRYH
  if (!h || !applied_update || !lambda || !max_change) throw Error(E3D_ERR_INVALID, "null argument");
  bool applied = false;
  apply_update(h, print_progress != 0, &applied, lambda, max_change);
  *applied_update = applied ? 1 : 0;
  return 0;
  R_CATCH()
}

// bool Optimizer::RunOnCurrentScale(...)  (src/opt/optimizer.cc:49-182); the observation source follows
// e3d_reg_set_cache_observations (the files of the cache are the host side's business)
int e3d_reg_run_on_current_scale(e3d_reg_t* h, int max_num_iterations, float max_change_convergence_threshold,
                                 int iterations_without_new_optimum_threshold, int print_progress, double* optimum_cost,
                                 int* iterations_done) {
  R_TRYH
  if (!h || !optimum_cost) throw Error(E3D_ERR_INVALID, "null argument");
  const bool print = print_progress != 0;
  // never use the highest image scale (optimizer.cc:60-61)
  h->prm.current_image_scale = std::min<int>(h->prm.current_image_scale, h->prm.image_scale_count - 1 - 1);
  if (print) printf("--- Optimizing at scaling factor %g ---\n", std::pow(2.0, -1.0 * h->prm.current_image_scale));
  if (h->cache_observations) {       // ObservationsCache constructor (observations_cache.cc:39-50), "path does not exist" branch
    bool missing = false;
    for (auto& kv : h->images) missing = missing || (h->owns(kv.first) && !kv.second.has_observed);
    if (missing && e3d_reg_determine_observed_indices(h) < 0) throw Error(E3D_ERR_INVALID, e3d_last_error());
  }
  bool converged = false;
  float lambda = 64.0f;
  int without = 0;
  *optimum_cost = std::numeric_limits<double>::infinity();
  RegState optimum = get_state(h);
  int it = 0;
  NormalSystem pending;                 // the next Apply's normal equations, when an iteration's cost came out of them
  bool have_pending = false;
  static const bool fuse_cost = [] { const char* e = getenv("E3D_REG_FUSE_COST"); return !e || atoi(e) != 0; }();
  for (; it < max_num_iterations; ++it) {
    if (print) printf("Iteration %d\n", it + 1);
    bool applied = true;
    float max_change = std::numeric_limits<float>::infinity();
    if (it > 0) {
      if (print) printf("  Intrinsics and poses update ...\n");
      applied = false;
      max_change = 0;
      Phase ph(h, "apply (all of it)");
      apply_update(h, print, &applied, &lambda, &max_change, have_pending ? &pending : nullptr);
      have_pending = false;
    }
    if (print) printf("  Observations update ...\n");
    { Phase ph(h, "observations (all of it)"); update_observations(h, /*kBorderSize*/ 1); }
    if (h->prm.variable_residuals_weight > 0) {
      if (print) printf("  Color update ...\n");
      Phase ph(h, "color_update");
      color_update(h);
    }
    if (print) printf("  Determining cost ...\n");
    double current_cost;
    // CostCalculator::ComputeCost here and the residual sums of the NEXT iteration's Apply ("Initial residual",
    // intrinsics_and_pose_optimizer.cc:176-185) are the same sums over the same observations at the same state -- the reference
    // evaluates every residual twice.  When the loop is certain to go on whatever the cost turns out to be (the exit test below
    // with the larger of the two values `without` can take), the next Apply's accumulation runs here, its sums give this
    // iteration's cost, and the Apply finds its normal equations ready: one intensity + cost pass per image and iteration less.
    // Otherwise (last iteration, convergence by max_change, optimum counter at its limit): the cost pass alone.
    const bool goes_on = fuse_cost && it + 1 < max_num_iterations && applied && !(max_change < max_change_convergence_threshold) &&
                         without + 1 < iterations_without_new_optimum_threshold;
    if (goes_on) {
      Phase ph(h, "total_cost");
      accumulate_system(h, pending);
      have_pending = true;
      const bool none = pending.counts[0] == 0 && pending.counts[1] == 0 && pending.counts[2] == 0;      // cost_calculator.cc:84-90
      current_cost = none ? std::numeric_limits<double>::infinity() : compute_cost_value(h, pending.sums, pending.counts);
    } else {
      Phase ph(h, "total_cost");
      current_cost = total_cost(h);
    }
    if (print) printf("  Cost (considering occlusions) is: %g\n", current_cost);
    if (current_cost < *optimum_cost) {
      *optimum_cost = current_cost;
      without = 0;
      optimum = get_state(h);
    } else {
      ++without;
    }
    if (!applied || max_change < max_change_convergence_threshold || without >= iterations_without_new_optimum_threshold) {
      if (print) printf("Assuming convergence (applied_update: %d, max_change: %g, iterations_without_new_optimum: %d)\n", (int)applied, (double)max_change, without);
      converged = true;
      ++it;
      break;
    } else if (print) {
      printf("max_change in this iteration: %g\n", (double)max_change);
    }
  }
  set_state(h, optimum);
  if (iterations_done) *iterations_done = it;
  if (print) fflush(stdout);
  return converged ? 1 : 0;
  R_CATCH()
}
}  // extern "C"
