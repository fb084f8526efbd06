// e3d_kernels.hpp -- device-side building blocks shared by the ICP and normal-estimation
// kernels (gfx950, wave64).  All parity-critical f32 arithmetic lives here; the translation
// unit is compiled with -ffp-contract=off and the operation orders below are the ones
// documented in DESIGN.md ("f32 operation orders").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace e3d {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;

struct HashEntry {
  unsigned long long key;
  unsigned start, end;
};

// Row-major 3x4 affine passed by value (lands in SGPRs).
struct Affine { float m[12]; };

// Static per-cloud grid description used by build and query kernels.
struct GridDesc {
  float origin[3];
  float inv_cell;
  unsigned mask;         // hash table size - 1 (power of two)
};

// Maps a global-frame query into the target cloud's local frame: ql = Linv * (q - t).
struct InvMap { float Linv[9]; float t[3]; };

// ---- f32 arithmetic with pinned operation order -----------------------------------------------
// pcl::transformPointCloudWithNormals (PCL 1.10 Transformer::se3/so3, recalled):
//   p' = x*c0 + (y*c1 + (z*c2 + c3)),  n' = x*c0 + (y*c1 + z*c2)
__device__ __forceinline__ float3 pcl_se3(const Affine& T, float x, float y, float z) {
  float3 r;
  r.x = x * T.m[0] + (y * T.m[1] + (z * T.m[2] + T.m[3]));
  r.y = x * T.m[4] + (y * T.m[5] + (z * T.m[6] + T.m[7]));
  r.z = x * T.m[8] + (y * T.m[9] + (z * T.m[10] + T.m[11]));
  return r;
}
__device__ __forceinline__ float3 pcl_so3(const Affine& T, float x, float y, float z) {
  float3 r;
  r.x = x * T.m[0] + (y * T.m[1] + z * T.m[2]);
  r.y = x * T.m[4] + (y * T.m[5] + z * T.m[6]);
  r.z = x * T.m[8] + (y * T.m[9] + z * T.m[10]);
  return r;
}
// Eigen 3-term inner product: e0 + (e1 + e2)
__device__ __forceinline__ float dot3e(float a0, float a1, float a2, float b0, float b1, float b2) {
  const float e0 = a0 * b0, e1 = a1 * b1, e2 = a2 * b2;
  return e0 + (e1 + e2);
}
// FLANN L2_Simple: ((dx*dx) + dy*dy) + dz*dz
__device__ __forceinline__ float sqdist_l2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float acc = dx * dx;
  acc = acc + dy * dy;
  acc = acc + dz * dz;
  return acc;
}

// ---- grid cells --------------------------------------------------------------------------------
__device__ __forceinline__ int cell_coord(float v, float origin, float inv_cell) {
  return (int)floorf((v - origin) * inv_cell);
}
// 21 bits per axis; callers guarantee 0 <= c < 2^21 for stored points.
__device__ __forceinline__ unsigned long long cell_key(int cx, int cy, int cz) {
  return ((unsigned long long)(unsigned)cz << 42) | ((unsigned long long)(unsigned)cy << 21) |
         (unsigned long long)(unsigned)cx;
}
__device__ __forceinline__ unsigned hash_key(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned)k;
}

// ---- wave / block reductions ---------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace e3d
