// e3d_icp_kernels.hip -- HIP kernels of the point-to-plane ICP path (gfx950 / CDNA4, wave64).
//
// Data layout in HBM (per cloud, all arrays in *grid-cell order* of the cloud's static local grid):
//   L4[n]  float4 {lx, ly, lz, bits(original index)}   local frame, written once
//   LN[n]  float4 {nx, ny, nz, 0}                      local normals, written once
//   G4[n]  float4 {gx, gy, gz, bits(original index)}   global frame, rewritten every outer iteration
//   table  HashEntry{key, start, end}                  cell -> [start,end) into the sorted arrays
// Correspondences of all directed pairs of a rank, concatenated (three float4 planes, 48 B each):
//   A = {sp.x, sp.y, sp.z, sn.x}  B = {sn.y, sn.z, tp.x, tp.y}  C = {tp.z, tn.x, tn.y, tn.z}
// so that every LM pass is a pure coalesced 16 B/lane stream.
//
// Reference loops covered: a3 (transform+bbox), a5 (1-NN within radius),
// a6 (count / distance sum), a7 (accumulate), a8 (cost).
#include <cstdlib>

#include "e3d_icp_kernels.hpp"

#pragma clang fp contract(off)

namespace e3d {

// =================================================================================================
// a3: pcl::transformPointCloudWithNormals + AlignedBox::extend   (icp_point_to_plane.cc:189-205)
// =================================================================================================
// AoS variant (fixed clouds at AddPointCloud, and the stand-alone e3d_transform_cloud entry point).
__global__ __launch_bounds__(kBlock) void k_transform_aos(const float* __restrict__ xyz,
                                                          const float* __restrict__ nrm, size_t n, Affine T,
                                                          float* __restrict__ oxyz, float* __restrict__ onrm,
                                                          float* __restrict__ bbox_partial) {
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const float3 p = pcl_se3(T, x, y, z);
    oxyz[3 * i] = p.x; oxyz[3 * i + 1] = p.y; oxyz[3 * i + 2] = p.z;
    if (nrm) {
      const float3 q = pcl_so3(T, nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
      onrm[3 * i] = q.x; onrm[3 * i + 1] = q.y; onrm[3 * i + 2] = q.z;
    }
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float s[kBlock / kWave][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
  if (lane == 0) { for (int d = 0; d < 3; ++d) { s[w][d] = mn[d]; s[w][3 + d] = mx[d]; } }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v = (threadIdx.x < 3) ? fminf(v, s[k][threadIdx.x]) : fmaxf(v, s[k][threadIdx.x]);
    bbox_partial[6 * blockIdx.x + threadIdx.x] = v;
  }
}

// Sorted float4 variant used every outer iteration: L4 -> G4 (32 B/point of HBM traffic).
__global__ __launch_bounds__(kBlock) void k_transform_bbox(const float4* __restrict__ L4, size_t n, Affine T,
                                                           float4* __restrict__ G4,
                                                           float* __restrict__ bbox_partial) {
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 l = ld_stream(L4 + i);
    const float3 p = pcl_se3(T, l.x, l.y, l.z);
    st_stream(G4 + i, make_float4(p.x, p.y, p.z, l.w));
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float s[kBlock / kWave][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
  if (lane == 0) { for (int d = 0; d < 3; ++d) { s[w][d] = mn[d]; s[w][3 + d] = mx[d]; } }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v = (threadIdx.x < 3) ? fminf(v, s[k][threadIdx.x]) : fmaxf(v, s[k][threadIdx.x]);
    bbox_partial[6 * blockIdx.x + threadIdx.x] = v;
  }
}

// single block: reduce the per-block bbox partials to 6 floats
__global__ void k_bbox_final(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
  __shared__ float s[256][6];
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], partial[6 * b + d]); mx[d] = fmaxf(mx[d], partial[6 * b + 3 + d]); }
  }
  for (int d = 0; d < 3; ++d) { s[threadIdx.x][d] = mn[d]; s[threadIdx.x][3 + d] = mx[d]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int k = 1; k < (int)blockDim.x; ++k) v = (threadIdx.x < 3) ? fminf(v, s[k][threadIdx.x]) : fmaxf(v, s[k][threadIdx.x]);
    out[threadIdx.x] = v;
  }
}

// bbox only (local frame, for the grid origin)
__global__ __launch_bounds__(kBlock) void k_bbox_aos(const float* __restrict__ xyz, size_t n,
                                                     float* __restrict__ bbox_partial) {
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
    mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
  }
  __shared__ float s[kBlock / kWave][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
  if (lane == 0) { for (int d = 0; d < 3; ++d) { s[w][d] = mn[d]; s[w][3 + d] = mx[d]; } }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / kWave; ++k) v = (threadIdx.x < 3) ? fminf(v, s[k][threadIdx.x]) : fmaxf(v, s[k][threadIdx.x]);
    bbox_partial[6 * blockIdx.x + threadIdx.x] = v;
  }
}

// =================================================================================================
// Static grid build (replaces the per-pair FLANN kd-tree build of icp_point_to_plane.cc:46-51)
// =================================================================================================
__global__ __launch_bounds__(kBlock) void k_cell_keys(const float* __restrict__ xyz, size_t n, GridDesc g,
                                                      unsigned long long* __restrict__ keys,
                                                      unsigned* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cx = cell_coord(xyz[3 * i], g.origin[0], g.inv_cell);
  const int cy = cell_coord(xyz[3 * i + 1], g.origin[1], g.inv_cell);
  const int cz = cell_coord(xyz[3 * i + 2], g.origin[2], g.inv_cell);
  keys[i] = cell_key(cx, cy, cz);
  vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(kBlock) void k_permute(const float* __restrict__ xyz, const float* __restrict__ nrm,
                                                    const unsigned* __restrict__ order, size_t n,
                                                    float4* __restrict__ L4, float4* __restrict__ LN) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned i = order[j];
  L4[j] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
  if (nrm) LN[j] = make_float4(nrm[3 * (size_t)i], nrm[3 * (size_t)i + 1], nrm[3 * (size_t)i + 2], 0.f);
}

__global__ __launch_bounds__(kBlock) void k_count_cells(const unsigned long long* __restrict__ keys, size_t n,
                                                        unsigned* __restrict__ counter) {
  unsigned c = 0;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
    c += (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __shared__ unsigned sc[kBlock / kWave];
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(counter, sc[0] + sc[1] + sc[2] + sc[3]);
}

__device__ __forceinline__ unsigned table_find_or_insert(HashEntry* table, unsigned mask, unsigned long long key) {
  unsigned h = hash_key(key) & mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&table[h].key, kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) return h;
    h = (h + 1) & mask;
  }
}

__global__ __launch_bounds__(kBlock) void k_build_table(const unsigned long long* __restrict__ keys, size_t n,
                                                        HashEntry* __restrict__ table, unsigned mask) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long k = keys[j];
  const bool is_start = (j == 0) || (keys[j - 1] != k);
  const bool is_end = (j + 1 == n) || (keys[j + 1] != k);
  if (is_start || is_end) {
    const unsigned h = table_find_or_insert(table, mask, k);
    if (is_start) table[h].start = (unsigned)j;
    if (is_end) table[h].end = (unsigned)(j + 1);
  }
}

// ---- half cells: every grid cell is split into 2 x 2 x 2 sub-cells and its points are stored in sub-cell order (z, y, x).  A second
// dense directory over the sub-cells (8 entries per grid cell, same construction as the cell directory) lets the bounded search
// (k_nn_bounded_half) visit only the sub-cell rows its ball touches: a 2 - 3 mm ball in 10 mm cells of 62 points touches 2 - 3
// cells but only ~7 half cells of 8 points.  The half-cell coordinate is floor((v - origin) * (2 inv_cell)); scaling by 2 is
// exact in f32, so its upper bits ARE the cell coordinate (floor(2 t) >> 1 == floor(t)), and it is monotone in v like the cell
// coordinate -- the completeness argument of the cells carries over with a cell of half the size.
__device__ __forceinline__ unsigned half_code(float x, float y, float z, const GridDesc& g) {
  const float inv_h = 2.f * g.inv_cell;
  const unsigned fx = (unsigned)cell_coord(x, g.origin[0], inv_h) & 1u;
  const unsigned fy = (unsigned)cell_coord(y, g.origin[1], inv_h) & 1u;
  const unsigned fz = (unsigned)cell_coord(z, g.origin[2], inv_h) & 1u;
  return (fz * 2u + fy) * 2u + fx;
}
// first sort pass of the build: 3-bit sub-cell code per point (the stable sort by cell key that follows keeps this order
// inside every cell)
__global__ __launch_bounds__(kBlock) void k_half_keys(const float* __restrict__ xyz, size_t n, GridDesc g, unsigned* __restrict__ keys,
                                                      unsigned* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = half_code(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], g);
  vals[i] = (unsigned)i;
}
// cell keys of the points in a given order (second, stable sort pass)
__global__ __launch_bounds__(kBlock) void k_cell_keys_ordered(const float* __restrict__ xyz, const unsigned* __restrict__ order, size_t n,
                                                              GridDesc g, unsigned long long* __restrict__ keys) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t i = order[j];
  keys[j] = cell_key(cell_coord(xyz[3 * i], g.origin[0], g.inv_cell), cell_coord(xyz[3 * i + 1], g.origin[1], g.inv_cell),
                     cell_coord(xyz[3 * i + 2], g.origin[2], g.inv_cell));
}
// The half-cell directory: 8 bytes per grid cell, byte k = number of the cell's points with sub-cell code <= k (an inclusive
// prefix: sub-cell k is the run [S[cell] + byte(k - 1), S[cell] + byte(k))).  Cells with more than 255 points hold 0xFF in every
// byte and are scanned whole.  Two steps: the last point of every (cell, code) run writes its end offset; the first point of every
// cell then turns the 8 bytes into a running maximum (empty sub-cells wrote nothing) and marks the overflow.
__global__ __launch_bounds__(kBlock) void k_half_ends(const unsigned long long* __restrict__ keys, const float4* __restrict__ L4, size_t n,
                                                      GridDesc g, QueryRange qr, const unsigned* __restrict__ S,
                                                      unsigned char* __restrict__ H8) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long k = keys[j];
  const float4 p = L4[j];
  const unsigned code = half_code(p.x, p.y, p.z, g);
  if (j + 1 < n && keys[j + 1] == k) {
    const float4 pn = L4[j + 1];
    if (half_code(pn.x, pn.y, pn.z, g) == code) return;       // only the last point of a run writes
  }
  const int cx = (int)(k & 0x1FFFFFull), cy = (int)((k >> 21) & 0x1FFFFFull), cz = (int)((k >> 42) & 0x1FFFFFull);
  const size_t lin = ((size_t)(unsigned)(cz - qr.lo[2]) * qr.D[1] + (unsigned)(cy - qr.lo[1])) * qr.D[0] + (unsigned)(cx - qr.lo[0]);
  const unsigned end_rel = (unsigned)(j + 1) - S[lin];
  H8[8 * lin + code] = (unsigned char)min(end_rel, 255u);
}
__global__ __launch_bounds__(kBlock) void k_half_fix(const unsigned long long* __restrict__ keys, size_t n, QueryRange qr,
                                                     const unsigned* __restrict__ S, unsigned long long* __restrict__ H8) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long k = keys[j];
  if (j > 0 && keys[j - 1] == k) return;                      // the first point of every cell
  const int cx = (int)(k & 0x1FFFFFull), cy = (int)((k >> 21) & 0x1FFFFFull), cz = (int)((k >> 42) & 0x1FFFFFull);
  const size_t lin = ((size_t)(unsigned)(cz - qr.lo[2]) * qr.D[1] + (unsigned)(cy - qr.lo[1])) * qr.D[0] + (unsigned)(cx - qr.lo[0]);
  if (S[lin + 1] - S[lin] > 255u) { H8[lin] = ~0ull; return; }
  unsigned long long w = H8[lin], out = 0ull;
  unsigned run = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    run = max(run, (unsigned)((w >> (8 * b)) & 0xFFull));
    out |= (unsigned long long)run << (8 * b);
  }
  H8[lin] = out;
}

// dense cell-start directory: ends[lin(cell)] = end index of the cell's run in the sorted arrays (written by the
// last point of each run, 0 elsewhere); an exclusive MAX scan turns it into starts, with
// starts[lin + 1] = max(starts[lin], ends[lin]) = end of the cell's run (or its start if the cell is empty).
__global__ __launch_bounds__(kBlock) void k_dense_counts(const unsigned long long* __restrict__ keys, size_t n,
                                                         QueryRange qr, unsigned* __restrict__ ends) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long k = keys[j];
  if (j + 1 < n && keys[j + 1] == k) return;          // only the last point of a run writes
  const int cx = (int)(k & 0x1FFFFFull), cy = (int)((k >> 21) & 0x1FFFFFull), cz = (int)((k >> 42) & 0x1FFFFFull);
  const size_t lin = ((size_t)(unsigned)(cz - qr.lo[2]) * qr.D[1] + (unsigned)(cy - qr.lo[1])) * qr.D[0] +
                     (unsigned)(cx - qr.lo[0]);
  ends[lin] = (unsigned)(j + 1);
}

// =================================================================================================
// a5: FindCorrespondencesFast (icp_point_to_plane.cc:42-105): exact nearest target point with
//     f32 squared distance < r2, lowest original target index on ties.
// One thread per source point (in the source cloud's cell order, so that a wave's 64 queries share
// their 27-cell neighbourhoods and the candidate reads are L1/L2 broadcast hits).
// =================================================================================================
__global__ __launch_bounds__(kBlock) void k_nn_query(const float4* __restrict__ Gsrc, size_t n_src,
                                                     const float4* __restrict__ Gtgt,
                                                     const HashEntry* __restrict__ table, GridDesc g, InvMap im,
                                                     float r2, int* __restrict__ match_pos,
                                                     float* __restrict__ match_d2) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_src) return;
  const float4 q = Gsrc[j];
  // query in the target's local frame (approximate; only selects candidate cells)
  const float dx = q.x - im.t[0], dy = q.y - im.t[1], dz = q.z - im.t[2];
  const float lx = im.Linv[0] * dx + im.Linv[1] * dy + im.Linv[2] * dz;
  const float ly = im.Linv[3] * dx + im.Linv[4] * dy + im.Linv[5] * dz;
  const float lz = im.Linv[6] * dx + im.Linv[7] * dy + im.Linv[8] * dz;
  const int cx = cell_coord(lx, g.origin[0], g.inv_cell);
  const int cy = cell_coord(ly, g.origin[1], g.inv_cell);
  const int cz = cell_coord(lz, g.origin[2], g.inv_cell);

  float best_d2 = r2;        // strict: a candidate at exactly r2 can never win (best_oi == 0)
  unsigned best_oi = 0u;
  int best_pos = -1;
  constexpr int kMaxC = (1 << 21) - 1;
  // reject early when the whole neighbourhood is outside the key range (NaN coordinates land here too)
  if (!(cx >= -1 && cy >= -1 && cz >= -1 && cx <= kMaxC + 1 && cy <= kMaxC + 1 && cz <= kMaxC + 1)) {
    match_pos[j] = -1; match_d2[j] = 0.f;
    return;
  }
  for (int oz = -1; oz <= 1; ++oz) {
    const int z = cz + oz;
    if (z < 0 || z > kMaxC) continue;
    for (int oy = -1; oy <= 1; ++oy) {
      const int y = cy + oy;
      if (y < 0 || y > kMaxC) continue;
      for (int ox = -1; ox <= 1; ++ox) {
        const int x = cx + ox;
        if (x < 0 || x > kMaxC) continue;
        const unsigned long long key = cell_key(x, y, z);
        unsigned h = hash_key(key) & g.mask;
        unsigned s = 0, e = 0;
        for (;;) {
          const HashEntry en = table[h];
          if (en.key == key) { s = en.start; e = en.end; break; }
          if (en.key == kEmptyKey) break;
          h = (h + 1) & g.mask;
        }
        for (unsigned m = s; m < e; ++m) {
          const float4 c = Gtgt[m];
          const float d2 = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
          const unsigned oi = __float_as_uint(c.w);
          if (d2 < best_d2 || (d2 == best_d2 && oi < best_oi)) { best_d2 = d2; best_oi = oi; best_pos = (int)m; }
        }
      }
    }
  }
  match_pos[j] = best_pos;
  match_d2[j] = best_d2;
}

// -------------------------------------------------------------------------------------------------
// Dense-data path of a5: queries are radix-sorted by the TARGET grid cell they fall into, then one
// wave handles 64 consecutive sorted queries.  For every distinct cell in its chunk the wave stages the
// cell's 27-neighbourhood (contiguous runs of the target's cell-ordered G4 array, found with 27 parallel
// hash lookups) in LDS with coalesced loads, and the lanes -- arranged as (queries of that cell) x
// (candidate slices) -- scan the bucket with broadcast LDS reads.  Same arithmetic, same (d2, index)
// ordering and strict radius test as k_nn_query, so both kernels return identical results.
// -------------------------------------------------------------------------------------------------
// block_dist (optional): distance, in cells, from the mapped query to the nearest face of the 3 x 3 x 3 cell block around its
// cell (1 .. 1.5): every target point outside the block is at least that far away in the target's local frame.
__device__ __forceinline__ unsigned long long query_cell_key(const float4 q, const InvMap& im, const GridDesc& g,
                                                             const QueryRange& qr, int& cx, int& cy, int& cz,
                                                             float* block_dist = nullptr) {
  const float dx = q.x - im.t[0], dy = q.y - im.t[1], dz = q.z - im.t[2];
  const float lx = im.Linv[0] * dx + im.Linv[1] * dy + im.Linv[2] * dz;
  const float ly = im.Linv[3] * dx + im.Linv[4] * dy + im.Linv[5] * dz;
  const float lz = im.Linv[6] * dx + im.Linv[7] * dy + im.Linv[8] * dz;
  cx = cell_coord(lx, g.origin[0], g.inv_cell);
  cy = cell_coord(ly, g.origin[1], g.inv_cell);
  cz = cell_coord(lz, g.origin[2], g.inv_cell);
  if (block_dist) {
    const float ux = (lx - g.origin[0]) * g.inv_cell - (float)cx, uy = (ly - g.origin[1]) * g.inv_cell - (float)cy,
                uz = (lz - g.origin[2]) * g.inv_cell - (float)cz;
    const float m = fminf(fminf(fminf(ux, 1.0f - ux), fminf(uy, 1.0f - uy)), fminf(uz, 1.0f - uz));
    *block_dist = 1.0f + fminf(fmaxf(m, 0.0f), 0.5f);
  }
  const unsigned kx = (unsigned)(cx - qr.lo[0]), ky = (unsigned)(cy - qr.lo[1]), kz = (unsigned)(cz - qr.lo[2]);
  if (kx >= qr.D[0] || ky >= qr.D[1] || kz >= qr.D[2]) return kEmptyKey;   // no target cell within reach
  return ((unsigned long long)kz * qr.D[1] + ky) * qr.D[0] + kx;
}

__global__ __launch_bounds__(kBlock) void k_query_keys(const float4* __restrict__ Gsrc, size_t n, GridDesc g, InvMap im,
                                                       QueryRange qr, unsigned long long* __restrict__ keys,
                                                       unsigned* __restrict__ vals) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int cx, cy, cz;
  keys[j] = query_cell_key(Gsrc[j], im, g, qr, cx, cy, cz);
  vals[j] = (unsigned)j;
}

// 32-bit variant of the sort keys (grids with fewer than 2^32 - 1 cells in the query range): 8 instead of 12
// bytes per (key, index) pair through every radix pass.
__global__ __launch_bounds__(kBlock) void k_query_keys32(const float4* __restrict__ Gsrc, size_t n, GridDesc g, InvMap im,
                                                         QueryRange qr, unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ vals) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int cx, cy, cz;
  const unsigned long long k = query_cell_key(Gsrc[j], im, g, qr, cx, cy, cz);
  keys[j] = (k == kEmptyKey) ? 0xFFFFFFFFu : (unsigned)k;
  vals[j] = (unsigned)j;
}

// the same keys for a LIST of queries (the ones the certificates did not settle): vals = their source positions
__global__ __launch_bounds__(kBlock) void k_query_keys_list(const float4* __restrict__ Gsrc, const unsigned* __restrict__ list, size_t n,
                                                            GridDesc g, InvMap im, QueryRange qr, unsigned long long* __restrict__ keys,
                                                            unsigned* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned jf = list[i], j = jf & kListIndexMask;            // (a far-list entry may carry kListNoPartner: it travels with the value)
  int cx, cy, cz;
  keys[i] = query_cell_key(Gsrc[j], im, g, qr, cx, cy, cz);
  vals[i] = jf;
}
__global__ __launch_bounds__(kBlock) void k_query_keys32_list(const float4* __restrict__ Gsrc, const unsigned* __restrict__ list, size_t n,
                                                              GridDesc g, InvMap im, QueryRange qr, unsigned* __restrict__ keys,
                                                              unsigned* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned jf = list[i], j = jf & kListIndexMask;
  int cx, cy, cz;
  const unsigned long long k = query_cell_key(Gsrc[j], im, g, qr, cx, cy, cz);
  keys[i] = (k == kEmptyKey) ? 0xFFFFFFFFu : (unsigned)k;
  vals[i] = jf;
}

// accumulated motion bound of query q (MotionBound, e3d_icp_kernels.hpp): a * rho + b, rho = |q - cs| widened (up) or narrowed (lo)
// by the rounding of the global coordinates and of this evaluation
__device__ __forceinline__ float motion_rho(const float4 q, const MotionBound& m) {
  const float dx = q.x - m.cs[0], dy = q.y - m.cs[1], dz = q.z - m.cs[2];
  return sqrtf(dx * dx + (dy * dy + dz * dz));
}
__device__ __forceinline__ float motion_up(const float4 q, const MotionBound& m) {
  return (m.a * (motion_rho(q, m) * 1.000002f + m.rho_err) + m.b) * 1.000001f;
}
__device__ __forceinline__ float motion_lo(const float4 q, const MotionBound& m) {
  return (m.a * fmaxf(motion_rho(q, m) * 0.999998f - m.rho_err, 0.f) + m.b) * 0.999999f;
}

// -------------------------------------------------------------------------------------------------
// Occupancy of the 27-cell blocks (round 5).  While two scans are still centimetres apart most queries have no target point anywhere
// in the 27 cells around them; k_nn_rows keyed, sorted and visited them all the same, read the nine directory rows of their block
// and left with "no partner" (9 of the 9.4 ms of such an outer iteration at 2 x 50 M points).  One bit per cell of the dense
// directory's range says whether ANY of the 27 cells around it holds a point (k_occ_cells: the occupied cells, 64 cells per wave and
// ballot; k_occ_dilate: the OR over the 3 x 3 x 3 neighbourhood, rows padded to whole 32-bit words): 1e9 cells are 125 MB, built
// once per grid.  k_query_keys_prune tests it with ONE load per query.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_occ_cells(const unsigned* __restrict__ S, QueryRange qr, unsigned stride_w, unsigned* __restrict__ occ) {
  // one wave per 64 consecutive (padded) cells of a row
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_row = (size_t)stride_w * 32u;
  const size_t row = t / per_row;
  if (row >= (size_t)qr.D[1] * qr.D[2]) return;                 // (per_row is a multiple of 64 or of 32: see the launch)
  const unsigned x = (unsigned)(t % per_row);
  bool on = false;
  if (x < qr.D[0]) { const size_t lin = row * qr.D[0] + x; on = S[lin + 1] > S[lin]; }
  const unsigned long long b = __ballot(on);
  const int lane = threadIdx.x & 63;
  const size_t w0 = row * stride_w + (x >> 5);
  if (lane == 0) occ[w0] = (unsigned)b;
  if (lane == 32) occ[w0] = (unsigned)(b >> 32);
}
__global__ __launch_bounds__(kBlock) void k_occ_dilate(const unsigned* __restrict__ in, unsigned D1, unsigned D2, unsigned stride_w, unsigned* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = t / stride_w;
  if (row >= (size_t)D1 * D2) return;
  const unsigned xw = (unsigned)(t % stride_w);
  const int y = (int)(row % D1), z = (int)(row / D1);
  unsigned acc = 0u;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy, zz = z + dz;
      if (yy < 0 || zz < 0 || yy >= (int)D1 || zz >= (int)D2) continue;
      const unsigned* r = in + ((size_t)zz * D1 + (size_t)yy) * stride_w;
      const unsigned w = r[xw], lo = xw > 0u ? r[xw - 1] : 0u, hi = xw + 1u < stride_w ? r[xw + 1] : 0u;
      acc |= w | (w << 1) | (w >> 1) | (lo >> 31) | (hi << 31);
    }
  out[t] = acc;
}

// Keys of the queries k_nn_rows has to look at, COMPACTED: a query whose block is empty (its cell's bit in `occ` is clear, or its
// cell lies outside the directory's range) is settled here as k_nn_rows settles a query without candidates -- no partner, d2 = r2,
// the certificate of the block's faces -- and only the others are keyed for the sort.  The RESULTS (partner, distance) are those of
// k_nn_rows; the certificate STATE need not be: k_nn_rows scores a query against every candidate of its segment's span, so its
// bound can be min(sqrt(b2), faces) with a finite b2 where this kernel writes the faces' bound -- either is a valid lower bound.
// list == nullptr: all n queries.  A block takes kPruneBlock queries, kPrunePerThread per thread, and reserves its stretch of the
// output with ONE atomic (a first version with an atomic per wave on the one counter took 15 ms for 1.5 M of them).  The order of
// the kept pairs depends on the order in which blocks reach the counter; the radix sort orders them by key and k_nn_rows treats
// every query on its own, so results do not.  count[0] = number of pairs written.
constexpr int kPrunePerThread = 8, kPruneBlock = kBlock * kPrunePerThread;
static_assert((unsigned)kPruneBlock == kQueryKeysBlock, "queries per block of the key kernels");
// prune == false (a pair whose far list has stopped being mostly empty blocks, sort_query_keys_pruned): every listed query is keyed,
// none is settled -- what k_query_keys32_list does.  key_or / key_mask: the batch's key kernel puts the pair's index above the cell
// key (one sort for the far lists of a whole batch of pairs); count2: that pair's own counter.
template <typename KeyT>
__device__ __forceinline__ void query_keys_prune_body(const unsigned bx, const float4* __restrict__ Gsrc, const unsigned* __restrict__ list, size_t n,
                                                      const unsigned* __restrict__ occ, unsigned stride_w, const GridDesc& g, const InvMap& im,
                                                      const QueryRange& qr, float r2, const CertParams& cert, KeyT* __restrict__ keys,
                                                      unsigned* __restrict__ vals, unsigned* __restrict__ count, unsigned* __restrict__ count2,
                                                      const KeyT key_mask, const KeyT key_or, const bool prune,
                                                      int* __restrict__ match, int* __restrict__ match2,
                                                      float* __restrict__ match_d2, float* __restrict__ lbe, int from_state) {
  __shared__ unsigned s_cnt[kBlock / kWave][kPrunePerThread];
  __shared__ unsigned s_base;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t i0 = (size_t)bx * kPruneBlock + (size_t)w * kWave + (size_t)lane;      // step u: + u * kBlock
  KeyT kk[kPrunePerThread];
  unsigned vv[kPrunePerThread];
  unsigned long long keep_mask[kPrunePerThread];       // wave-uniform ballots
  unsigned jf[kPrunePerThread];
  float4 qq[kPrunePerThread];
#pragma unroll
  for (int u = 0; u < kPrunePerThread; ++u) {
    const size_t i = i0 + (size_t)u * kBlock;
    // from_state (all queries of a pair whose certificates were not tested this time, see certify_now in e3d_icp.hip): the flag
    // k_nn_certify would have set comes from the state itself, and so does the r2 it would have stored
    jf[u] = (i < n) ? (list ? list[i] : ((unsigned)i | ((from_state && match[i] < 0) ? kListNoPartner : 0u))) : 0u;
  }
#pragma unroll
  for (int u = 0; u < kPrunePerThread; ++u) qq[u] = Gsrc[jf[u] & kListIndexMask];
  unsigned wordv[kPrunePerThread];
  int bitv[kPrunePerThread];
  float bdist[kPrunePerThread];
#pragma unroll
  for (int u = 0; u < kPrunePerThread; ++u) {
    int cx = 0, cy = 0, cz = 0;
    bdist[u] = 2.0f;
    const unsigned long long key = query_cell_key(qq[u], im, g, qr, cx, cy, cz, &bdist[u]);
    kk[u] = ((KeyT)key & key_mask) | key_or;
    wordv[u] = 0u; bitv[u] = -1;
    if (prune && key != kEmptyKey) {
      const int kx = cx - qr.lo[0], ky = cy - qr.lo[1], kz = cz - qr.lo[2];
      wordv[u] = occ[((size_t)kz * qr.D[1] + (size_t)ky) * stride_w + (size_t)(kx >> 5)];
      bitv[u] = kx & 31;
    } else {
      bdist[u] = 2.0f;                                   // outside the directory range: two empty cells all around (k_nn_rows)
    }
  }
#pragma unroll
  for (int u = 0; u < kPrunePerThread; ++u) {
    const size_t i = i0 + (size_t)u * kBlock;
    const bool valid = i < n;
    const bool keep = valid && (!prune || (bitv[u] >= 0 && ((wordv[u] >> bitv[u]) & 1u)));
    const unsigned j = jf[u] & kListIndexMask;
    vv[u] = jf[u];
    if (valid && from_state && (jf[u] & kListNoPartner)) match_d2[j] = r2;
    if (valid && !keep) {
      if (!(jf[u] & kListNoPartner)) {                   // (see k_nn_rows: a query that had no partner holds these values already)
        match[j] = -1; match_d2[j] = r2;
        if (match2) match2[j] = -1;
      }
      const float kInf = __uint_as_float(0x7f800000u);
      const float lb_out = bdist[u] * cert.cell_scale - cert.cell_sub;
      lbe[j] = fmaxf(fminf(sqrtf(kInf), lb_out), 0.0f) * 0.999999f + motion_lo(qq[u], cert.lo);
    }
    keep_mask[u] = __ballot(keep);
    if (lane == 0) s_cnt[w][u] = (unsigned)__popcll(keep_mask[u]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0u;
    for (int ww = 0; ww < kBlock / kWave; ++ww)
      for (int u = 0; u < kPrunePerThread; ++u) tot += s_cnt[ww][u];
    s_base = tot ? atomicAdd(count, tot) : 0u;
    if (count2 && tot) atomicAdd(count2, tot);
  }
  __syncthreads();
  unsigned off = s_base;
  for (int ww = 0; ww < w; ++ww)
#pragma unroll
    for (int u = 0; u < kPrunePerThread; ++u) off += s_cnt[ww][u];
#pragma unroll
  for (int u = 0; u < kPrunePerThread; ++u) {
    if ((keep_mask[u] >> lane) & 1ull) {
      const unsigned slot = off + (unsigned)__popcll(keep_mask[u] & ((1ull << lane) - 1ull));
      keys[slot] = kk[u];
      vals[slot] = vv[u];
    }
    off += (unsigned)__popcll(keep_mask[u]);
  }
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_query_keys_prune(const float4* __restrict__ Gsrc, const unsigned* __restrict__ list, size_t n,
                                                             const unsigned* __restrict__ occ, unsigned stride_w, GridDesc g, InvMap im,
                                                             QueryRange qr, float r2, CertParams cert, KeyT* __restrict__ keys,
                                                             unsigned* __restrict__ vals, unsigned* __restrict__ count,
                                                             int* __restrict__ match, int* __restrict__ match2,
                                                             float* __restrict__ match_d2, float* __restrict__ lbe, int from_state) {
  query_keys_prune_body<KeyT>(blockIdx.x, Gsrc, list, n, occ, stride_w, g, im, qr, r2, cert, keys, vals, count, nullptr, (KeyT)~(KeyT)0, (KeyT)0, true,
                              match, match2, match_d2, lbe, from_state);
}


__device__ __forceinline__ int rdlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned rdlane_u(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }

__global__ __launch_bounds__(kBlock, 6) void k_nn_cells(const float4* __restrict__ Gsrc, const unsigned* __restrict__ order,
                                                     size_t n, const float4* __restrict__ Gtgt,
                                                     const HashEntry* __restrict__ table,
                                                     const unsigned* __restrict__ dense_start, GridDesc g, InvMap im,
                                                     QueryRange qr, float r2, int* __restrict__ match_pos,
                                                     float* __restrict__ match_d2) {
  __shared__ float4 s_c[kBlock / kWave][kNNCap];
  __shared__ int s_p[kBlock / kWave][kNNCap];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t pos = ((size_t)blockIdx.x * (kBlock / kWave) + w) * kWave + lane;
  const bool valid = pos < n;
  const unsigned j = valid ? (order[pos] & kListIndexMask) : 0u;
  const float4 q = valid ? Gsrc[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  int cx = 0, cy = 0, cz = 0;
  const unsigned long long key = valid ? query_cell_key(q, im, g, qr, cx, cy, cz) : kEmptyKey;

  float best_d2 = r2;      // strict radius: with best_oi == 0 a candidate at exactly r2 can never win
  unsigned best_oi = 0u;
  int best_pos = -1;
  constexpr int kMaxC = (1 << 21) - 1;

  unsigned long long remaining = __ballot(key != kEmptyKey);
  while (remaining) {
    const int f = __ffsll((long long)remaining) - 1;                      // first lane of the next cell group
    const unsigned klo = rdlane_u((unsigned)key, f), khi = rdlane_u((unsigned)(key >> 32), f);
    const unsigned long long cur = ((unsigned long long)khi << 32) | klo;
    const unsigned long long group = __ballot(key == cur);                // contiguous lanes (sorted input)
    const int a = __popcll(group);
    const int ccx = rdlane_i(cx, f), ccy = rdlane_i(cy, f), ccz = rdlane_i(cz, f);

    // 27 parallel cell lookups: dense cell-start directory (one coalesced round trip, runs of x-neighbours
    // are adjacent words) or, for grids too large for it, the hash table
    unsigned st = 0u, cnt = 0u;
    if (lane < 27) {
      const int x = ccx + (lane % 3) - 1, y = ccy + ((lane / 3) % 3) - 1, z = ccz + (lane / 9) - 1;
      if (dense_start) {
        const unsigned kx = (unsigned)(x - qr.lo[0]), ky = (unsigned)(y - qr.lo[1]), kz = (unsigned)(z - qr.lo[2]);
        if (kx < qr.D[0] && ky < qr.D[1] && kz < qr.D[2]) {
          const size_t lin = ((size_t)kz * qr.D[1] + ky) * qr.D[0] + kx;
          st = dense_start[lin];
          cnt = dense_start[lin + 1] - st;
        }
      } else if (x >= 0 && y >= 0 && z >= 0 && x <= kMaxC && y <= kMaxC && z <= kMaxC) {
        const unsigned long long ck = cell_key(x, y, z);
        unsigned h = hash_key(ck) & g.mask;
        for (;;) {
          const HashEntry en = table[h];
          if (en.key == ck) { st = en.start; cnt = en.end - en.start; break; }
          if (en.key == kEmptyKey) break;
          h = (h + 1) & g.mask;
        }
      }
    }
    unsigned inc = cnt;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const unsigned t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    const unsigned off = inc - cnt;
    const unsigned total = rdlane_u(inc, 26);

    // lanes = (query slot) x (candidate slice)
    int lg = 0;
    while ((1 << lg) < a) ++lg;
    const int A2 = 1 << lg, slices = kWave >> lg;
    const int qi = lane & (A2 - 1), sl = lane >> lg;
    const bool act = qi < a;
    const int owner = (f + qi) & 63;
    const float qx = __shfl(q.x, owner, 64), qy = __shfl(q.y, owner, 64), qz = __shfl(q.z, owner, 64);
    float lb_d2 = r2;
    unsigned lb_oi = 0u;
    int lb_pos = -1;

    for (unsigned base = 0; base < total; base += kNNCap) {
      const unsigned nb = min((unsigned)kNNCap, total - base);
      // stage the bucket: every lane first resolves the source positions of its kNNCap/64 slots (flat
      // candidate index -> (cell run, offset) via the 27 prefix sums), then all loads are issued
      // back-to-back, then the LDS stores -- one memory round trip per batch instead of one per run.
      static_assert(kNNCap == 4 * kWave, "staging below is written for 4 slots per lane");
      const unsigned t0 = base + (unsigned)lane, t1 = t0 + kWave, t2 = t1 + kWave, t3 = t2 + kWave;
      unsigned m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu, m3 = 0xFFFFFFFFu;
      for (int r = 0; r < 27; ++r) {
        const unsigned c_r = rdlane_u(cnt, r);
        if (c_r == 0u) continue;
        const unsigned o_r = rdlane_u(off, r), s_r = rdlane_u(st, r);
        if (t0 - o_r < c_r) m0 = s_r + (t0 - o_r);     // unsigned wrap makes this "o_r <= t < o_r + c_r"
        if (t1 - o_r < c_r) m1 = s_r + (t1 - o_r);
        if (t2 - o_r < c_r) m2 = s_r + (t2 - o_r);
        if (t3 - o_r < c_r) m3 = s_r + (t3 - o_r);
      }
      float4 c0, c1, c2, c3;
      if (m0 != 0xFFFFFFFFu) c0 = Gtgt[m0];
      if (m1 != 0xFFFFFFFFu) c1 = Gtgt[m1];
      if (m2 != 0xFFFFFFFFu) c2 = Gtgt[m2];
      if (m3 != 0xFFFFFFFFu) c3 = Gtgt[m3];
      if (m0 != 0xFFFFFFFFu) { s_c[w][lane] = c0; s_p[w][lane] = (int)m0; }
      if (m1 != 0xFFFFFFFFu) { s_c[w][kWave + lane] = c1; s_p[w][kWave + lane] = (int)m1; }
      if (m2 != 0xFFFFFFFFu) { s_c[w][2 * kWave + lane] = c2; s_p[w][2 * kWave + lane] = (int)m2; }
      if (m3 != 0xFFFFFFFFu) { s_c[w][3 * kWave + lane] = c3; s_p[w][3 * kWave + lane] = (int)m3; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int lb_t = -1;
      if (act) {
#pragma unroll 4
        for (unsigned t = sl; t < nb; t += slices) {
          const float4 c = s_c[w][t];
          const float d2 = sqdist_l2(qx, qy, qz, c.x, c.y, c.z);
          const unsigned oi = __float_as_uint(c.w);
          if (d2 < lb_d2 || (d2 == lb_d2 && oi < lb_oi)) { lb_d2 = d2; lb_oi = oi; lb_t = (int)t; }
        }
        if (lb_t >= 0) lb_pos = s_p[w][lb_t];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // min over the candidate slices of each query slot
    for (int stx = A2; stx < kWave; stx <<= 1) {
      const float od2 = __shfl_xor(lb_d2, stx, 64);
      const unsigned ooi = (unsigned)__shfl_xor((int)lb_oi, stx, 64);
      const int opos = __shfl_xor(lb_pos, stx, 64);
      if (od2 < lb_d2 || (od2 == lb_d2 && ooi < lb_oi)) { lb_d2 = od2; lb_oi = ooi; lb_pos = opos; }
    }
    // hand the result to the lane that owns the query
    const int srcl = (lane - f) & 63;
    const float rd2 = __shfl(lb_d2, srcl, 64);
    const unsigned roi = (unsigned)__shfl((int)lb_oi, srcl, 64);
    const int rpos = __shfl(lb_pos, srcl, 64);
    if ((group >> lane) & 1ull) {
      if (rd2 < best_d2 || (rd2 == best_d2 && roi < best_oi)) { best_d2 = rd2; best_oi = roi; best_pos = rpos; }
    }
    remaining &= ~group;
  }
  if (valid) { match_pos[pos] = best_pos; match_d2[pos] = best_d2; }
}

// -------------------------------------------------------------------------------------------------
// Dense-directory path of a5 (the default on MI355X: 4 B per grid cell is cheap in 288 GB of HBM).
// Queries are sorted by target cell in (z, y, x) order -- the same order the target's points are stored
// in -- so the candidates of a run of cells [xa, xb] inside one neighbour ROW (y+dy, z+dz) are ONE contiguous
// run [S[lin(xa)], S[lin(xb)+1]) of the target array.  A wave handles a whole ROW SEGMENT of its 64 sorted
// queries at once (consecutive lanes in the same row, x-extent <= kRowSpan cells): 18 directory words
// (one round trip), the 9 union runs staged in LDS with all loads in flight together (second round trip),
// then lanes arranged as (query slot) x (candidate slice) scan the staged superset.  Candidates outside a
// query's own 27 cells are farther than the radius, so they can never win: results are identical to the
// other two kernels.
//
// The scan is VALU-issue bound (rocprofv3 PMC: SQ_ACTIVE_INST_VALU ~ 75 % of the kernel), so the inner loop is
// written for instruction count: candidates are staged as SoA PAIRS {x0,x1,y0,y1} {z0,z1} so that the three
// subtractions, three squares and two sums run as packed f32 ops (v_pk_add_f32 / v_pk_mul_f32, each still an
// individually rounded IEEE op in the (dx*dx + dy*dy) + dz*dz order); the bucket is padded with +inf sentinels
// so the loop is uniform and branch free with four pairs prefetched; and the (d2, index) tie rule is applied
// lazily: the fast loop tracks only (d2, slot) with a strict '<' and records whether ANY equality was seen, in
// which case (rare: lattices, duplicates) the batch is re-scanned with the full comparator.
// -------------------------------------------------------------------------------------------------
typedef float f2_t __attribute__((ext_vector_type(2)));

struct RowLds {
  float4 xy[kRowCap / 2 + 2 * kWave];   // {x0, x1, y0, y1} per candidate pair (+ sentinel padding)
  float2 z[kRowCap / 2 + 2 * kWave];    // {z0, z1}
  unsigned oi[kRowCap];             // original target index (tie rule, result)
};

// Fast scan: per QUAD of candidates (two staged pairs) only the minimum distance is compared with the running
// best -- 2 v_min + 2 v_cmp + 2 v_cndmask per four candidates instead of a compare/select chain per candidate.
// The winning quad is re-evaluated exactly afterwards (row_scan_resolve); `tie` records whether a quad minimum
// ever EQUALLED the running best (cross-quad tie), in which case the caller re-scans the batch with the full
// comparator.  trips2 = number of quads per lane; quad i of a lane covers pairs sl + 2*i*slices and
// sl + (2*i+1)*slices (sentinel padded).
__device__ __forceinline__ f2_t row_pair_d2(const float4 A, const float2 Zv, const f2_t QX, const f2_t QY, const f2_t QZ) {
  const f2_t cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {Zv.x, Zv.y};
  const f2_t dx = QX - cx, dy = QY - cy, dz = QZ - cz;
  f2_t d = dx * dx;
  d = d + dy * dy;
  d = d + dz * dz;
  return d;
}

// b2 follows the second smallest value at quad granularity (a displaced best, or the minimum of a quad that did not win);
// the other members of the winning quad are folded in by row_scan_resolve, so that afterwards b2 is the exact second
// smallest squared distance among all candidates seen (the certificate bound of k_nn_certify).
__device__ __forceinline__ void row_scan_fast(const RowLds& L, int sl, int slices, int trips2, float qx, float qy,
                                              float qz, float& bd, int& bq, bool& tie, float& b2) {
  const f2_t QX = {qx, qx}, QY = {qy, qy}, QZ = {qz, qz};
  int p = sl;
  int i = 0;
#define E3D_QUAD_STEP(A0, Z0, A1, Z1, I)                                      \
  {                                                                          \
    const f2_t d0 = row_pair_d2(A0, Z0, QX, QY, QZ);                          \
    const f2_t d1 = row_pair_d2(A1, Z1, QX, QY, QZ);                          \
    const float mq = fminf(fminf(d0.x, d0.y), fminf(d1.x, d1.y));            \
    b2 = fminf(b2, fmaxf(bd, mq));                                           \
    const bool lt = mq < bd, le = mq <= bd;                                  \
    tie = tie || (le && !lt);                                                \
    bd = lt ? mq : bd;                                                       \
    bq = lt ? (I) : bq;                                                      \
  }
  for (; i + 2 <= trips2; i += 2) {
    const float4 a0 = L.xy[p], a1 = L.xy[p + slices], a2 = L.xy[p + 2 * slices], a3 = L.xy[p + 3 * slices];
    const float2 z0 = L.z[p], z1 = L.z[p + slices], z2 = L.z[p + 2 * slices], z3 = L.z[p + 3 * slices];
    E3D_QUAD_STEP(a0, z0, a1, z1, i)
    E3D_QUAD_STEP(a2, z2, a3, z3, i + 1)
    p += 4 * slices;
  }
  for (; i < trips2; ++i) {
    const float4 a0 = L.xy[p], a1 = L.xy[p + slices];
    const float2 z0 = L.z[p], z1 = L.z[p + slices];
    E3D_QUAD_STEP(a0, z0, a1, z1, i)
    p += 2 * slices;
  }
#undef E3D_QUAD_STEP
}

// exact (d2, original index) comparator over the four candidates of quad `bq` of this lane
__device__ __forceinline__ void row_scan_resolve(const RowLds& L, int sl, int slices, int bq, unsigned base, unsigned nb,
                                                 float qx, float qy, float qz, float& bd, unsigned& boi, unsigned& bt, float& b2) {
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int p = sl + (2 * bq + h2) * slices;
    const float4 A = L.xy[p];
    const float2 Zv = L.z[p];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const unsigned t = 2u * (unsigned)p + (unsigned)hh;
      if (t >= nb) continue;
      const float d2 = sqdist_l2(qx, qy, qz, hh ? A.y : A.x, hh ? A.w : A.z, hh ? Zv.y : Zv.x);
      const unsigned oi = L.oi[t];
      if (d2 < bd || (d2 == bd && oi < boi)) { b2 = fminf(b2, bd); bd = d2; boi = oi; bt = base + t; }
      else b2 = fminf(b2, d2);
    }
  }
}

// exact re-scan with the full (d2, original index) comparator
__device__ __forceinline__ void row_scan_exact(const RowLds& L, int sl, int slices, int trips, unsigned base, unsigned nb,
                                               float qx, float qy, float qz, float& bd, unsigned& boi, unsigned& bt, float& b2) {
  int p = sl;
  for (int i = 0; i < trips; ++i, p += slices) {
    const float4 A = L.xy[p];
    const float2 Zv = L.z[p];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const unsigned t = 2u * (unsigned)p + (unsigned)hh;
      if (t >= nb) continue;
      const float d2 = sqdist_l2(qx, qy, qz, hh ? A.y : A.x, hh ? A.w : A.z, hh ? Zv.y : Zv.x);
      const unsigned oi = L.oi[t];
      if (d2 < bd || (d2 == bd && oi < boi)) { b2 = fminf(b2, bd); bd = d2; boi = oi; bt = base + t; }
      else b2 = fminf(b2, d2);
    }
  }
}

// Results are written at the query's SOURCE position order[pos] (the queries may be a sorted sub-list of the source cloud):
// match_pos / match_d2 as before, and lbe = (distance every target point other than the partner exceeds) + cert.cum_lo, the
// state k_nn_certify tests in the following outer iterations.
__device__ __forceinline__ void nn_rows_body(const unsigned bx, const float4* __restrict__ Gsrc, const unsigned* __restrict__ order,
                                             size_t n, const float4* __restrict__ Gtgt,
                                             const unsigned* __restrict__ S, const GridDesc& g, const InvMap& im,
                                             const QueryRange& qr, float r2, int row_span, const CertParams& cert,
                                             int* __restrict__ match_pos, float* __restrict__ match_d2,
                                             float* __restrict__ lbe, int* __restrict__ match2) {
  __shared__ RowLds lds[kBlock / kWave];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  RowLds& L = lds[w];
  const size_t pos = ((size_t)bx * (kBlock / kWave) + w) * kWave + lane;
  const bool valid = pos < n;
  const unsigned jf = valid ? order[pos] : 0u, j = jf & kListIndexMask;
  const float4 q = valid ? Gsrc[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  int cx = 0, cy = 0, cz = 0;
  float block_dist = 2.0f;
  const unsigned long long key = valid ? query_cell_key(q, im, g, qr, cx, cy, cz, &block_dist) : kEmptyKey;
  if (key == kEmptyKey) block_dist = 2.0f;     // outside the directory range (occupied cells +- 2): two empty cells all around
  const int kx = cx - qr.lo[0], ky = cy - qr.lo[1], kz = cz - qr.lo[2];   // in [0, D) for valid keys

  float best_d2 = r2;      // strict radius (see k_nn_query)
  unsigned best_oi = 0u;
  int best_pos = -1;
  const float kInf = __uint_as_float(0x7f800000u);
  float best_b2 = kInf;    // second smallest squared distance among the candidates seen

  unsigned long long remaining = __ballot(key != kEmptyKey);
  while (remaining) {
    const int f = __ffsll((long long)remaining) - 1;
    const int fky = rdlane_i(ky, f), fkz = rdlane_i(kz, f), fkx = rdlane_i(kx, f);
    // segment: lanes of the same row whose cells lie within kRowSpan cells of the first (contiguous: sorted input)
    const unsigned long long seg =
        __ballot(key != kEmptyKey && ky == fky && kz == fkz && (unsigned)(kx - fkx) <= (unsigned)row_span) & remaining;
    const int a = __popcll(seg);
    const int last = 63 - __clzll((long long)seg);
    const int lkx = rdlane_i(kx, last);
    const int xa = max(fkx - 1, 0), xb = min(lkx + 1, (int)qr.D[0] - 1);

    // directory words of the 9 neighbour rows (lanes 0..8)
    unsigned dst = 0u, dcnt = 0u;
    if (lane < 9) {
      const int y = fky + (lane % 3) - 1, z = fkz + (lane / 3) - 1;
      if (y >= 0 && z >= 0 && y < (int)qr.D[1] && z < (int)qr.D[2]) {
        const size_t row = ((size_t)z * qr.D[1] + (size_t)y) * qr.D[0];
        dst = S[row + xa];
        dcnt = S[row + xb + 1] - dst;
      }
    }
    unsigned rs[9], rl[9];
    unsigned total = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { rs[r] = rdlane_u(dst, r); rl[r] = rdlane_u(dcnt, r); total += rl[r]; }

    // lanes = (query slot) x (candidate slice)
    int lg = 0;
    while ((1 << lg) < a) ++lg;
    const int A2 = 1 << lg, slices = kWave >> lg;
    const int qi = lane & (A2 - 1), sl = lane >> lg;
    const int owner = (f + qi) & 63;
    const float qx = __shfl(q.x, owner, 64), qy = __shfl(q.y, owner, 64), qz = __shfl(q.z, owner, 64);
    float lb_d2 = r2;
    unsigned lb_oi = 0u;
    unsigned lb_t = 0xFFFFFFFFu;      // flat index into the concatenated 9 runs
    float lb_b2 = kInf;

    for (unsigned base = 0; base < total; base += kRowCap) {
      const unsigned nb = min((unsigned)kRowCap, total - base);
      const int np = (int)((nb + 1u) >> 1);
      const int trips2 = (np + 2 * slices - 1) / (2 * slices);   // quads (2 pairs) per lane
      const int trips = 2 * trips2;
      // ---- stage [base, base + nb) of the concatenated runs.  Each lane handles candidate PAIRS (2u, 2u + 1): the SoA-pair
      // layout is then one 16-byte and two 8-byte LDS stores per pair, and the flat index -> target position map
      // (position = flat + off_r for the run r that contains it, off_r wave-uniform) costs two instructions per run.
      constexpr int kPairsPerLane = kRowCap / 2 / kWave;
      unsigned pm[kPairsPerLane][2];
#pragma unroll
      for (int k = 0; k < kPairsPerLane; ++k) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned fl = base + 2u * (unsigned)(k * kWave + lane) + (unsigned)hh;
          unsigned off = rs[0];
          unsigned P = rl[0];
#pragma unroll
          for (int r = 1; r < 9; ++r) {
            off = (fl >= P) ? rs[r] - P : off;          // an empty run r is overridden by the next one (same P)
            P += rl[r];
          }
          pm[k][hh] = (fl < total) ? fl + off : 0xFFFFFFFFu;
        }
      }
      float4 cv[kPairsPerLane][2];
#pragma unroll
      for (int k = 0; k < kPairsPerLane; ++k)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          cv[k][hh] = (pm[k][hh] != 0xFFFFFFFFu) ? Gtgt[pm[k][hh]] : make_float4(kInf, 0.f, 0.f, 0.f);
      const unsigned padded_pairs = (unsigned)(trips * slices);            // pairs incl. sentinels (<= kRowCap / 2 + 2 * 64)
      uint2* const oi2 = reinterpret_cast<uint2*>(L.oi);
#pragma unroll
      for (int k = 0; k < kPairsPerLane + 2; ++k) {
        const unsigned u = (unsigned)(k * kWave + lane);
        if (u < padded_pairs) {
          const float4 c0 = (k < kPairsPerLane) ? cv[k < kPairsPerLane ? k : 0][0] : make_float4(kInf, 0.f, 0.f, 0.f);
          const float4 c1 = (k < kPairsPerLane) ? cv[k < kPairsPerLane ? k : 0][1] : make_float4(kInf, 0.f, 0.f, 0.f);
          L.xy[u] = make_float4(c0.x, c1.x, c0.y, c1.y);    // x = +inf for slots past nb: d2 = +inf never wins and never ties
          L.z[u] = make_float2(c0.z, c1.z);
          if (2u * u < nb) oi2[u] = make_uint2(__float_as_uint(c0.w), __float_as_uint(c1.w));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      const float in_d2 = lb_d2, in_b2 = lb_b2;
      const unsigned in_oi = lb_oi, in_t = lb_t;
      bool tie = false;
      int bq = -1;
      float fd = lb_d2;
      row_scan_fast(L, sl, slices, trips2, qx, qy, qz, fd, bq, tie, lb_b2);
      if (__ballot(tie)) {                    // rare: cross-quad exact f32 distance tie -> full comparator for this batch
        lb_d2 = in_d2; lb_oi = in_oi; lb_t = in_t; lb_b2 = in_b2;
        row_scan_exact(L, sl, slices, trips, base, nb, qx, qy, qz, lb_d2, lb_oi, lb_t, lb_b2);
      } else if (bq >= 0) {                   // winner quad: exact (d2, index) order among its four candidates
        row_scan_resolve(L, sl, slices, bq, base, nb, qx, qy, qz, lb_d2, lb_oi, lb_t, lb_b2);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // flat index of the slice winner -> position in the target array
    int lb_pos = -1;
    {
      unsigned p = 0;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (lb_t - p < rl[r]) lb_pos = (int)(rs[r] + (lb_t - p));
        p += rl[r];
      }
    }
    // min over the candidate slices of each query slot
    for (int stx = A2; stx < kWave; stx <<= 1) {
      const float od2 = __shfl_xor(lb_d2, stx, 64);
      const unsigned ooi = (unsigned)__shfl_xor((int)lb_oi, stx, 64);
      const int opos = __shfl_xor(lb_pos, stx, 64);
      const float ob2 = __shfl_xor(lb_b2, stx, 64);
      lb_b2 = fminf(fminf(lb_b2, ob2), fmaxf(lb_d2, od2));      // second smallest of the two slices together
      if (od2 < lb_d2 || (od2 == lb_d2 && ooi < lb_oi)) { lb_d2 = od2; lb_oi = ooi; lb_pos = opos; }
    }
    // hand the result to the lane that owns the query
    const int srcl = (lane - f) & 63;
    const float rd2 = __shfl(lb_d2, srcl, 64);
    const unsigned roi = (unsigned)__shfl((int)lb_oi, srcl, 64);
    const int rpos = __shfl(lb_pos, srcl, 64);
    const float rb2 = __shfl(lb_b2, srcl, 64);
    if ((seg >> lane) & 1ull) {
      best_b2 = fminf(fminf(best_b2, rb2), fmaxf(best_d2, rd2));
      if (rd2 < best_d2 || (rd2 == best_d2 && roi < best_oi)) { best_d2 = rd2; best_oi = roi; best_pos = rpos; }
    }
    remaining &= ~seg;
  }
  if (valid) {
    // A query that had no partner and still has none (kListNoPartner, set by k_nn_certify -- most queries of the first outer
    // iterations of a poorly aligned pair) holds match = match2 = -1 already, and k_nn_certify has stored r2 as its distance:
    // only the certificate changes.  Three of the four scattered 4-byte stores of this kernel, a third of its time there.
    if (!((jf & kListNoPartner) && best_pos < 0)) {
      match_pos[j] = best_pos; match_d2[j] = best_d2;
      if (match2) match2[j] = -1;               // this kernel remembers the partner only
    }
    // every candidate in the 27 cells was evaluated: the others are >= sqrt(best_b2) away (all of them, if there is no partner:
    // best_d2 stayed r2, so best_b2 is the smallest distance seen); points outside the block are >= block_dist cells away in
    // the local frame (2 cells if the query's cell lies outside the directory range, i.e. occupied cells +- 2)
    const float lb_out = block_dist * cert.cell_scale - cert.cell_sub;
    lbe[j] = fmaxf(fminf(sqrtf(best_b2), lb_out), 0.0f) * 0.999999f + motion_lo(q, cert.lo);
  }
}

__global__ __launch_bounds__(kBlock, 6) void k_nn_rows(const float4* __restrict__ Gsrc, const unsigned* __restrict__ order,
                                                       size_t n, const float4* __restrict__ Gtgt,
                                                       const unsigned* __restrict__ S, GridDesc g, InvMap im,
                                                       QueryRange qr, float r2, int row_span, CertParams cert,
                                                       int* __restrict__ match_pos, float* __restrict__ match_d2,
                                                       float* __restrict__ lbe, int* __restrict__ match2) {
  nn_rows_body(blockIdx.x, Gsrc, order, n, Gtgt, S, g, im, qr, r2, row_span, cert, match_pos, match_d2, lbe, match2);
}

// -------------------------------------------------------------------------------------------------
// MFMA-filtered variant of k_nn_rows (same segments, same staging runs, same results).
//
// The scan of a segment is a dense (queries x candidates) squared-distance matrix.  Its entries are first computed
// APPROXIMATELY on the matrix cores -- |q|^2 + |c|^2 - 2 q.c as one K = 16 f16 MFMA per 32 x 32 tile, with coordinates
// taken relative to a per-segment origin, scaled by a power of two and split into f16 hi + lo parts so that the result
// is within a PROVEN bound of the exact f32 value (host: mfma_filter_params) -- and only candidates whose approximate
// distance is within that bound of the running approximate minimum (or of the radius) are evaluated EXACTLY, with the
// very arithmetic (sqdist_l2) and the (d2, original index) order of the other kernels.  Every candidate that could be
// the exact winner passes the filter, so the output is bit-identical; the VALU work per pair drops from ~6
// instructions to ~0.02 (a min-tree over the accumulators) plus the per-candidate operand packing.
// The query's own cell row is scanned first: most queries meet their nearest neighbour there, after which the
// filter rarely fires.
//
// Operand layout (v_mfma_f32_32x32x16_f16: A = candidates as rows, B = queries as columns; lane l supplies row / column
// l & 31 and the eight k values 8 * (l >> 5) .. + 7):
//   k:   0     1     2     3    4     5     6     7   |  8     9     10    11   12    13    14    15
//   A:   chx   chy   chz   nch  clx   cly   clz   ncl |  chx   chy   chz   1    clx   cly   clz   1
//   B:  -2qhx -2qhy -2qhz  1   -2qhx -2qhy -2qhz  1   | -2qlx -2qly -2qlz  nqh -2qlx -2qly -2qlz  nql
// (h / l = f16 hi / lo part of the scaled relative coordinate, n.. = hi / lo part of |hi + lo|^2.)
// -------------------------------------------------------------------------------------------------
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

constexpr int kMfCap = 256;                     // candidates staged per wave and batch
constexpr float kMfSentinel = 60000.0f;         // |.|^2 part of padding rows / columns: never passes the filter

struct MfLds {
  uint4 a[kMfCap / 32][2][32];                  // operand A parts: [tile][k half][row]
  float4 c[kMfCap];                             // exact coordinates + original index (for the exact evaluation)
};

__device__ __forceinline__ unsigned pk_rtz(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}
__device__ __forceinline__ float h_lo(unsigned w) { return (float)__builtin_bit_cast(fp16x2_t, w)[0]; }
__device__ __forceinline__ float h_hi(unsigned w) { return (float)__builtin_bit_cast(fp16x2_t, w)[1]; }

// hi / lo split of a scaled relative position: words (hx,hy) (hz,0) (lx,ly) (lz,0) and |hi + lo|^2
struct MfSplit { unsigned hxy, hz0, lxy, lz0; float n; };
__device__ __forceinline__ MfSplit mf_split(float x, float y, float z) {
  MfSplit r;
  r.hxy = pk_rtz(x, y); r.hz0 = pk_rtz(z, 0.f);
  const float hx = h_lo(r.hxy), hy = h_hi(r.hxy), hz = h_lo(r.hz0);
  const float rx = x - hx, ry = y - hy, rz = z - hz;                // exact
  r.lxy = pk_rtz(rx, ry); r.lz0 = pk_rtz(rz, 0.f);
  const float cx = hx + h_lo(r.lxy), cy = hy + h_hi(r.lxy), cz = hz + h_lo(r.lz0);   // exact sums
  r.n = cx * cx + (cy * cy + cz * cz);
  return r;
}
// (hi, lo) f16 parts of a non-negative norm as the high halves of two words whose low halves are given
__device__ __forceinline__ void mf_norm_parts(float n, unsigned lowA, unsigned lowB, unsigned& wA, unsigned& wB) {
  const unsigned nh = pk_rtz(0.f, n);                               // high half = rtz(n)
  const float rest = n - h_hi(nh);
  const unsigned nl = pk_rtz(0.f, rest);
  wA = (lowA & 0xFFFFu) | (nh & 0xFFFF0000u);
  wB = (lowB & 0xFFFFu) | (nl & 0xFFFF0000u);
}

// MfParams (e3d_icp_kernels.hpp): S = power-of-two scale of the relative coordinates, r2s = r^2 S^2, eta2 = 2 eta (eta bounds
// |mfma value - scaled exact f32 d2| apart from the coordinate representation), delta4 = 4 Delta, delta4sq = 4 Delta^2
// (Delta bounds the error of a scaled point-to-point distance caused by the hi/lo representation).
__device__ __forceinline__ float mf_threshold(float am, const MfParams& P) {
  return am + P.eta2 + P.delta4 * sqrtf(fmaxf(am, 0.f)) + P.delta4sq;
}

__device__ __forceinline__ float mf_min16(const f16_t& v) {
  const float a = fminf(fminf(v[0], v[1]), v[2]), b = fminf(fminf(v[3], v[4]), v[5]);
  const float c = fminf(fminf(v[6], v[7]), v[8]), d = fminf(fminf(v[9], v[10]), v[11]);
  const float e = fminf(fminf(v[12], v[13]), v[14]);
  return fminf(fminf(fminf(a, b), fminf(c, d)), fminf(e, v[15]));
}

// one query column group: running approximate minimum + filter threshold, exact best so far
struct MfBest {
  float am, thr;
  float qx, qy, qz;
  float bd; unsigned boi, bt;
};

__device__ __forceinline__ void mf_process(const f16_t& acc, MfBest& B, const MfLds& L, int tile, int g, unsigned base, unsigned nb,
                                           const MfParams& P) {
  const float tm = mf_min16(acc);
  if (!__ballot(tm <= B.thr)) return;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    if (acc[reg] <= B.thr) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * g;
      const unsigned t = (unsigned)(tile * 32 + row);
      const float4 cc = L.c[t];
      const float d2 = sqdist_l2(B.qx, B.qy, B.qz, cc.x, cc.y, cc.z);
      const unsigned oi = __float_as_uint(cc.w);
      if (t < nb && (d2 < B.bd || (d2 == B.bd && oi < B.boi))) { B.bd = d2; B.boi = oi; B.bt = base + t; }
      if (acc[reg] < B.am) { B.am = acc[reg]; B.thr = mf_threshold(B.am, P); }
    }
  }
}

__global__ __launch_bounds__(kBlock, 3) void k_nn_mfma(const float4* __restrict__ Gsrc, const unsigned* __restrict__ order,
                                                       size_t n, const float4* __restrict__ Gtgt,
                                                       const unsigned* __restrict__ S, GridDesc g, InvMap im,
                                                       QueryRange qr, float r2, int row_span, MfParams P,
                                                       int* __restrict__ match_pos, float* __restrict__ match_d2) {
  __shared__ MfLds lds[kBlock / kWave];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  MfLds& L = lds[w];
  const int col = lane & 31, gh = lane >> 5;
  const size_t pos = ((size_t)blockIdx.x * (kBlock / kWave) + w) * kWave + lane;
  const bool valid = pos < n;
  const unsigned j = valid ? (order[pos] & kListIndexMask) : 0u;
  const float4 q = valid ? Gsrc[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  int cx = 0, cy = 0, cz = 0;
  const unsigned long long key = valid ? query_cell_key(q, im, g, qr, cx, cy, cz) : kEmptyKey;
  const int kx = cx - qr.lo[0], ky = cy - qr.lo[1], kz = cz - qr.lo[2];

  float best_d2 = r2;
  unsigned best_oi = 0u;
  int best_pos = -1;
  const float kInf = __uint_as_float(0x7f800000u);
  const float thr0 = mf_threshold(P.r2s, P);
  const f16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned one_hi = 0x3C000000u;                              // f16 1.0 in the high half

  unsigned long long remaining = __ballot(key != kEmptyKey);
  while (remaining) {
    const int f = __ffsll((long long)remaining) - 1;
    const int fky = rdlane_i(ky, f), fkz = rdlane_i(kz, f), fkx = rdlane_i(kx, f);
    const unsigned long long seg =
        __ballot(key != kEmptyKey && ky == fky && kz == fkz && (unsigned)(kx - fkx) <= (unsigned)row_span) & remaining;
    const int a = __popcll(seg);
    const int last = 63 - __clzll((long long)seg);
    const int lkx = rdlane_i(kx, last);
    const int xa = max(fkx - 1, 0), xb = min(lkx + 1, (int)qr.D[0] - 1);

    // directory words of the 9 neighbour rows, the queries' own row first (lanes 0..8)
    unsigned dst = 0u, dcnt = 0u;
    if (lane < 9) {
      const int rid = (int)((0x862075314ull >> (4 * lane)) & 15ull);       // 4, 1, 3, 5, 7, 0, 2, 6, 8
      const int y = fky + (rid % 3) - 1, z = fkz + (rid / 3) - 1;
      if (y >= 0 && z >= 0 && y < (int)qr.D[1] && z < (int)qr.D[2]) {
        const size_t row = ((size_t)z * qr.D[1] + (size_t)y) * qr.D[0];
        dst = S[row + xa];
        dcnt = S[row + xb + 1] - dst;
      }
    }
    unsigned rs[9], rl[9];
    unsigned total = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { rs[r] = rdlane_u(dst, r); rl[r] = rdlane_u(dcnt, r); total += rl[r]; }

    // segment origin: midpoint of its first and last query (wave-uniform)
    const float ox = 0.5f * (__shfl(q.x, f, 64) + __shfl(q.x, last, 64));
    const float oy = 0.5f * (__shfl(q.y, f, 64) + __shfl(q.y, last, 64));
    const float oz = 0.5f * (__shfl(q.z, f, 64) + __shfl(q.z, last, 64));

    // B operands: column group G holds query slots 32 G + col
    h8_t Bop[2];
    MfBest Q[2];
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      const int slot = 32 * G + col;
      const int owner = (f + slot) & 63;
      const float qx = __shfl(q.x, owner, 64), qy = __shfl(q.y, owner, 64), qz = __shfl(q.z, owner, 64);
      const bool qv = slot < a;
      Q[G].qx = qx; Q[G].qy = qy; Q[G].qz = qz;
      Q[G].am = P.r2s; Q[G].thr = qv ? thr0 : -kInf;
      Q[G].bd = r2; Q[G].boi = 0u; Q[G].bt = 0xFFFFFFFFu;
      const MfSplit sp = mf_split(qv ? (qx - ox) * P.S : 0.f, qv ? (qy - oy) * P.S : 0.f, qv ? (qz - oz) * P.S : 0.f);
      uint4 wv;
      if (gh == 0) {
        const unsigned w0 = pk_rtz(-2.f * h_lo(sp.hxy), -2.f * h_hi(sp.hxy));
        const unsigned w1 = (pk_rtz(-2.f * h_lo(sp.hz0), 0.f) & 0xFFFFu) | one_hi;
        wv = make_uint4(w0, w1, w0, w1);
      } else {
        const unsigned w0 = pk_rtz(-2.f * h_lo(sp.lxy), -2.f * h_hi(sp.lxy));
        const unsigned w1l = pk_rtz(-2.f * h_lo(sp.lz0), 0.f);
        unsigned w1, w3;
        mf_norm_parts(qv ? sp.n : kMfSentinel, w1l, w1l, w1, w3);
        wv = make_uint4(w0, w1, w0, w3);
      }
      Bop[G] = __builtin_bit_cast(h8_t, wv);
    }
    const bool two = a > 32;

    for (unsigned base = 0; base < total; base += kMfCap) {
      const unsigned nb = min((unsigned)kMfCap, total - base);
      const int ntiles = (int)((nb + 31u) >> 5);
      // ---- stage [base, base + nb): resolve indices, issue the loads, pack the operands ----
      unsigned m[kMfCap / kWave];
#pragma unroll
      for (int k = 0; k < kMfCap / kWave; ++k) m[k] = 0xFFFFFFFFu;
      {
        unsigned p = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
#pragma unroll
          for (int k = 0; k < kMfCap / kWave; ++k) {
            const unsigned t = base + (unsigned)(k * kWave + lane) - p;
            if (t < rl[r]) m[k] = rs[r] + t;
          }
          p += rl[r];
        }
      }
      float4 cv[kMfCap / kWave];
#pragma unroll
      for (int k = 0; k < kMfCap / kWave; ++k) cv[k] = (m[k] != 0xFFFFFFFFu) ? Gtgt[m[k]] : make_float4(kInf, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < kMfCap / kWave; ++k) {
        const unsigned t = (unsigned)(k * kWave + lane);
        if (t < (unsigned)(ntiles * 32)) {
          const bool cvd = m[k] != 0xFFFFFFFFu;
          const float4 c = cv[k];
          const MfSplit sp = mf_split(cvd ? (c.x - ox) * P.S : 0.f, cvd ? (c.y - oy) * P.S : 0.f, cvd ? (c.z - oz) * P.S : 0.f);
          unsigned w1, w3;
          mf_norm_parts(cvd ? sp.n : kMfSentinel, sp.hz0, sp.lz0, w1, w3);
          L.a[t >> 5][0][t & 31] = make_uint4(sp.hxy, w1, sp.lxy, w3);
          L.a[t >> 5][1][t & 31] = make_uint4(sp.hxy, (sp.hz0 & 0xFFFFu) | one_hi, sp.lxy, (sp.lz0 & 0xFFFFu) | one_hi);
          L.c[t] = c;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      for (int tile = 0; tile < ntiles; ++tile) {
        const h8_t Aop = __builtin_bit_cast(h8_t, L.a[tile][gh][col]);
        const f16_t acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop, Bop[0], zero16, 0, 0, 0);
        if (two) {
          const f16_t acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop, Bop[1], zero16, 0, 0, 0);
          mf_process(acc0, Q[0], L, tile, gh, base, nb, P);
          mf_process(acc1, Q[1], L, tile, gh, base, nb, P);
        } else {
          mf_process(acc0, Q[0], L, tile, gh, base, nb, P);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // per column group: flat index -> target position, combine the two row halves, hand to the owning lane
    float rd2 = r2; unsigned roi = 0u; int rpos = -1;
    const int slot_of_lane = (lane - f) & 63;
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      int lb_pos = -1;
      {
        unsigned p = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          if (Q[G].bt - p < rl[r]) lb_pos = (int)(rs[r] + (Q[G].bt - p));
          p += rl[r];
        }
      }
      float lb_d2 = Q[G].bd; unsigned lb_oi = Q[G].boi;
      {
        const float od2 = __shfl_xor(lb_d2, 32, 64);
        const unsigned ooi = (unsigned)__shfl_xor((int)lb_oi, 32, 64);
        const int opos = __shfl_xor(lb_pos, 32, 64);
        if (od2 < lb_d2 || (od2 == lb_d2 && ooi < lb_oi)) { lb_d2 = od2; lb_oi = ooi; lb_pos = opos; }
      }
      const float sd2 = __shfl(lb_d2, slot_of_lane & 31, 64);
      const unsigned soi = (unsigned)__shfl((int)lb_oi, slot_of_lane & 31, 64);
      const int spos = __shfl(lb_pos, slot_of_lane & 31, 64);
      if ((slot_of_lane >> 5) == G) { rd2 = sd2; roi = soi; rpos = spos; }
    }
    if ((seg >> lane) & 1ull) {
      if (rd2 < best_d2 || (rd2 == best_d2 && roi < best_oi)) { best_d2 = rd2; best_oi = roi; best_pos = rpos; }
    }
    remaining &= ~seg;
  }
  if (valid) { match_pos[pos] = best_pos; match_d2[pos] = best_d2; }
}

// -------------------------------------------------------------------------------------------------
// Certificates: is last outer iteration's partner provably still the unique nearest neighbour within the radius?
//
// State of query j of a directed pair (source order): match[j] = partner position m in the target arrays (or -1) and
// lbe[j] = LB + cum(s), where at the query's last search (outer iteration s) every target point other than m was at (true
// Euclidean) distance >= LB of the query in the global frame, and cum(.) is the host's running bound of how far any point of
// either cloud has moved since -- round 5: of how far THIS query has moved relative to the target, a rho + b (MotionBound; cum_up adds
// the f32 rounding of the transforms).  Triangle inequality: now every other point is at distance >= thr = lbe - cum_up.  If the partner's new f32 squared distance v
// is < thr^2 (with room for the f32 evaluation error of the others' distances) and < r2, the exact search would return (m, v):
// nothing else can be nearer or tie.  For m = -1 "no partner" stands as long as thr^2 >= r2.  All other queries are appended
// to the todo list (in source order inside a block of 2048 queries; one atomic per block) and searched by k_nn_rows, which
// renews their state.
// -------------------------------------------------------------------------------------------------
constexpr int kCertPerWave = 512;          // consecutive queries per wave (8 steps of 64): one atomic per list and block of 2048 queries
// The state may hold a second candidate match2[j] (the runner-up of the last search, lbe then bounds everything but these two).
// Lists: todo_near = queries whose old partner is within sqrt(near2) (searched by k_nn_bounded: only the cells the ball of that
// distance touches), todo_far = all others (no partner, or a far one: sorted by target cell and searched by k_nn_rows).
// counts[0] / counts[1] = list lengths.
// (bx: the block's index among the blocks of ITS pair -- blockIdx.x for the one-pair kernel, the offset inside the pair's block range
// for the kernel that walks a batch of pairs, k_nn_certify_multi)
template <int kCertUnroll>
__device__ __forceinline__ void nn_certify_body(const unsigned bx, const float4* __restrict__ Gsrc, size_t n, const float4* __restrict__ Gtgt,
                                                const MotionBound cum_up, float r2, float near2, int none_near, int* __restrict__ match,
                                                int* __restrict__ match2, const float* __restrict__ lbe,
                                                float* __restrict__ match_d2, unsigned* __restrict__ todo_near,
                                                unsigned* __restrict__ todo_far, unsigned* __restrict__ counts,
                                                unsigned* __restrict__ upd_counts = nullptr, double* __restrict__ upd_d2 = nullptr,
                                                unsigned* __restrict__ upd_groups = nullptr, unsigned char* __restrict__ upd_done = nullptr) {
  __shared__ unsigned s_list[2][kBlock / kWave][kCertPerWave];
  __shared__ unsigned s_cnt[2][kBlock / kWave];
  __shared__ unsigned s_base[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t j0 = ((size_t)bx * (kBlock / kWave) + (size_t)w) * kCertPerWave;
  unsigned cn = 0, cf = 0;                                     // wave-uniform
  // kCertUnroll steps at a time: the state words, the queries and the partner gathers of all of them are requested before any
  // is evaluated (the kernel is a chain match[j] -> Gtgt[m] per query; kCertUnroll chains in flight per lane)
  for (int step0 = 0; step0 < kCertPerWave / kWave; step0 += kCertUnroll) {
    size_t jj[kCertUnroll];
    bool vv[kCertUnroll];
    int mm[kCertUnroll], mm2[kCertUnroll];
    float ll[kCertUnroll];
    float4 qq[kCertUnroll], c1[kCertUnroll], c2v[kCertUnroll];
#pragma unroll
    for (int u = 0; u < kCertUnroll; ++u) {
      jj[u] = j0 + (size_t)(step0 + u) * kWave + (size_t)lane;
      vv[u] = jj[u] < n;
      const size_t js = vv[u] ? jj[u] : 0;
      mm[u] = ld_stream(match + js); mm2[u] = ld_stream(match2 + js); ll[u] = ld_stream(lbe + js); qq[u] = ld_stream(Gsrc + js);
    }
#pragma unroll
    for (int u = 0; u < kCertUnroll; ++u) {
      c1[u] = Gtgt[mm[u] >= 0 ? mm[u] : 0];
      c2v[u] = Gtgt[(mm[u] >= 0 && mm2[u] >= 0) ? mm2[u] : 0];
    }
    // (round 6, upd_done != nullptr) A 256-query block of the row update whose queries are ALL settled here, none of them by its
    // runner-up, has nothing to rewrite, and its count, distance sum and active groups are known in this wave: four 64-query
    // steps = one such block, the same wave_sum over the same lanes and the same sequential sum over the four groups as
    // corr_update_body -- the same bits.  The update then skips the block on one flag instead of reading 12 B per query.
    unsigned ub_c = 0, ub_g = 0, ub_gm = 0;
    double ub_t = 0.0;
    bool ub_bad = false;
    static_assert(kCertUnroll == 4, "four steps of 64 queries = one block of the row update");
#pragma unroll
    for (int u = 0; u < kCertUnroll; ++u) {
    const size_t j = jj[u];
    const bool valid = vv[u];
    bool ok = false, near = false;
    bool swapped = false;
    float v_ok = 0.f;
    if (valid) {
      const float thr = (ll[u] - motion_up(qq[u], cum_up)) * 0.999999f;
      const float lim = thr * thr * 0.999999f;
      const int m = mm[u];
      if (m >= 0) {
        const int m2 = mm2[u];
        const float4 q = qq[u];
        const float4 c = c1[u];
        float v = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
        if (m2 >= 0) {
          // two remembered candidates: the nearer one (exact (d2, original index) order) is the partner if both beat the bound
          // of everything else; the other one stays remembered
          const float4 c2 = c2v[u];
          const float v2 = sqdist_l2(q.x, q.y, q.z, c2.x, c2.y, c2.z);
          if (v2 < v || (v2 == v && __float_as_uint(c2.w) < __float_as_uint(c.w))) {
            v = v2;
            if (thr > 0.f && v < lim && v < r2) { match[j] = m2; match2[j] = m; swapped = true; }
          }
        }
        ok = (thr > 0.f) && (v < lim) && (v < r2);
        near = v < near2;
        if (ok) { st_stream(match_d2 + j, v); v_ok = v; }
      } else {
        ok = (thr > 0.f) && (lim >= r2);
        near = none_near != 0;                          // k_nn_bounded searches these beyond the radius
        st_stream(match_d2 + j, r2);                    // settled or not: what a search that finds nothing would write (k_nn_rows then skips it)
      }
    }
    const unsigned long long fn = __ballot(valid && !ok && near), ff = __ballot(valid && !ok && !near);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (valid && !ok) {
      if (near) s_list[0][w][cn + (unsigned)__popcll(fn & below)] = (unsigned)j;
      else s_list[1][w][cf + (unsigned)__popcll(ff & below)] = (unsigned)j | (mm[u] < 0 ? kListNoPartner : 0u);
    }
    cn += (unsigned)__popcll(fn); cf += (unsigned)__popcll(ff);
    if (upd_done) {
      const bool f = valid && mm[u] >= 0;
      const unsigned gc = (unsigned)__popcll(__ballot(f));
      const double gd = wave_sum((f && ok) ? (double)v_ok : 0.0);
      ub_c += gc; ub_t += gd;
      if (gc) { ++ub_g; ub_gm |= 1u << u; }
      if (__ballot(valid && (!ok || swapped))) ub_bad = true;
    }
    }
    if (upd_done) {
      const size_t blk = j0 / kBlock + (size_t)(step0 / kCertUnroll);
      if (blk < (n + kBlock - 1) / kBlock && lane == 0) {
        if (!ub_bad) { upd_counts[blk] = ub_c; upd_d2[blk] = ub_t; upd_groups[blk] = ub_g | (ub_gm << 8); }
        upd_done[blk] = ub_bad ? 0 : 1;
      }
    }
  }
  if (lane == 0) { s_cnt[0][w] = cn; s_cnt[1][w] = cf; }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned tot = 0;
    for (int k = 0; k < kBlock / kWave; ++k) tot += s_cnt[threadIdx.x][k];
    s_base[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x], tot) : 0u;
  }
  __syncthreads();
  unsigned bn = s_base[0], bf = s_base[1];
  for (int k = 0; k < w; ++k) { bn += s_cnt[0][k]; bf += s_cnt[1][k]; }
  for (unsigned i = (unsigned)lane; i < cn; i += kWave) todo_near[bn + i] = s_list[0][w][i];
  for (unsigned i = (unsigned)lane; i < cf; i += kWave) todo_far[bf + i] = s_list[1][w][i];
}

template <int kCertUnroll>
__global__ __launch_bounds__(kBlock) void k_nn_certify(const float4* __restrict__ Gsrc, size_t n, const float4* __restrict__ Gtgt,
                                                       MotionBound cum_up, float r2, float near2, int none_near, int* __restrict__ match,
                                                       int* __restrict__ match2, const float* __restrict__ lbe,
                                                       float* __restrict__ match_d2, unsigned* __restrict__ todo_near,
                                                       unsigned* __restrict__ todo_far, unsigned* __restrict__ counts) {
  nn_certify_body<kCertUnroll>(blockIdx.x, Gsrc, n, Gtgt, cum_up, r2, near2, none_near, match, match2, lbe, match_d2, todo_near, todo_far, counts);
}

// ---- a BATCH of directed pairs per launch (round 5) ---------------------------------------------------------------------------
// An all-pairs job runs the certificate search of 240 directed pairs per outer iteration: with one launch per pair and kernel that is
// ~2000 launches of 0.04 - 0.1 ms each, a cost that does not shrink when the job is spread over more GPUs (a rank's slices get
// shorter, its launches do not get fewer).  The kernels below walk a table of pairs instead: a block finds its pair from the
// exclusive ends of the pairs' block ranges (at most kNnBatchPairs words, block-uniform: scalar loads and compares) and runs the
// one-pair body on that pair's pointers and constants -- the same code, so the same bits.
__device__ __forceinline__ int nn_find_range(const unsigned* __restrict__ ends, int n, unsigned b) {
  int lo = 0, hi = n - 1;                                   // first index with b < ends[index]
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b < ends[mid]) hi = mid; else lo = mid + 1;
  }
  return __builtin_amdgcn_readfirstlane(lo);
}

__global__ __launch_bounds__(kBlock) void k_nn_certify_multi(const NnBatchDev* __restrict__ B, float r2) {
  const int p = nn_find_range(B->cert_end, B->n_pairs, blockIdx.x);
  const unsigned bx = blockIdx.x - (p ? B->cert_end[p - 1] : 0u);
  const NnPairDev& P = B->pair[p];
  nn_certify_body<4>(bx, P.Gsrc, (size_t)P.n, P.Gtgt, P.cum_up, r2, P.near2, P.none_near, P.match, P.match2, P.lbe, P.match_d2,
                     P.todo_near, P.todo_far, P.counts, P.upd_counts, P.upd_d2, P.upd_groups, P.upd_done);
}

// -------------------------------------------------------------------------------------------------
// Bounded search of the listed queries (one thread per query, no sort): the old partner at f32 squared distance d1 is still a
// candidate, so the nearest neighbour lies within d1; all target points whose squared distance is <= cover2 = (sqrt(d1) +
// margin)^2 (at most r2) lie in the cells the box [l - rho, l + rho] touches (l = query in the target's local frame, rho =
// sqrt(cover2) / sigma_min (1 + 1e-5) + slack; same completeness argument as for the 27 cells of the radius).  Those cells are
// scanned row by row -- the cells [xa, xb] of one (y, z) row are ONE run of the dense cell-start directory -- with the exact
// (d2, original index) order.  Afterwards every point other than the winner is farther than min(second smallest d2 seen,
// cover2): the next certificate.
// -------------------------------------------------------------------------------------------------
template <int LPQ>
__global__ __launch_bounds__(kBlock) void k_nn_bounded(const float4* __restrict__ Gsrc, const unsigned* __restrict__ list,
                                                       unsigned n_list, const float4* __restrict__ Gtgt, const unsigned* __restrict__ S,
                                                       GridDesc g, InvMap im, QueryRange qr, float r2, BoundParams bp,
                                                       int* __restrict__ match, int* __restrict__ match2,
                                                       float* __restrict__ match_d2, float* __restrict__ lbe) {
  // LPQ lanes per query (1 or 4): the lanes of a quad share the box and split every cell-row run -- lane k takes candidates s0 + k,
  // s0 + k + 4, ... (one 64-byte read per quad and step instead of four serial 16-byte reads per lane) -- then lane 0 merges the
  // partial top lists.  Which candidate wins does not depend on the split: distinct distances order themselves, and any exact
  // equality among the leaders (within a lane or at the merge) raises `tie`, which sends lane 0 through the exact (d2, original
  // index) scan.  Short lists are latency bound and gain from the quads (1.85 -> 1.49 ms per iteration at 2 x 50 M once the
  // poses settle); lists of millions of queries are bound by the candidate traffic and run one lane per query.
  const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
  const int sub = threadIdx.x % LPQ;
  if (i >= n_list) return;
  const unsigned j = list[i] & kListIndexMask;
  const float4 q = Gsrc[j];
  const int m = match[j];
  float d1 = r2;                                     // no old partner: the whole radius (the 27 cells)
  if (m >= 0) {
    const float4 pc = Gtgt[m];
    d1 = sqdist_l2(q.x, q.y, q.z, pc.x, pc.y, pc.z);
    const int m2 = match2[j];
    if (m2 >= 0) { const float4 pc2 = Gtgt[m2]; d1 = fminf(d1, sqdist_l2(q.x, q.y, q.z, pc2.x, pc2.y, pc2.z)); }
  }
  // A query WITHOUT a partner searches radius r + np_extra: it costs little where the target is absent (a row of empty cells
  // is two directory words) and yields "nothing within r + np_extra", a certificate that outlives the next pose updates -- the
  // 27-cell block of the row kernel only certifies the distance to its faces, 1 - 1.5 cells, and most of those queries came
  // back in every outer iteration.
  const float dr = sqrtf(fminf(d1, r2)) + ((m < 0) ? bp.np_extra : bp.margin);
  const float cover2 = (d1 < r2) ? fminf(dr * dr * 1.000001f, r2) : ((m < 0) ? dr * dr * 1.000001f : r2);      // NaN distances search the whole radius too
  const float rho = sqrtf(cover2) * bp.rho_scale + bp.rho_pad;
  const float dx = q.x - im.t[0], dy = q.y - im.t[1], dz = q.z - im.t[2];
  const float lx = im.Linv[0] * dx + im.Linv[1] * dy + im.Linv[2] * dz;
  const float ly = im.Linv[3] * dx + im.Linv[4] * dy + im.Linv[5] * dz;
  const float lz = im.Linv[6] * dx + im.Linv[7] * dy + im.Linv[8] * dz;
  const int x0 = max(cell_coord(lx - rho, g.origin[0], g.inv_cell) - qr.lo[0], 0), x1 = min(cell_coord(lx + rho, g.origin[0], g.inv_cell) - qr.lo[0], (int)qr.D[0] - 1);
  const int y0 = max(cell_coord(ly - rho, g.origin[1], g.inv_cell) - qr.lo[1], 0), y1 = min(cell_coord(ly + rho, g.origin[1], g.inv_cell) - qr.lo[1], (int)qr.D[1] - 1);
  const int z0 = max(cell_coord(lz - rho, g.origin[2], g.inv_cell) - qr.lo[2], 0), z1 = min(cell_coord(lz + rho, g.origin[2], g.inv_cell) - qr.lo[2], (int)qr.D[2] - 1);
  const float kInf = __uint_as_float(0x7f800000u);
  // What the scanned box really covers: every target point outside it is at least `fd` cells away from the query in the target's
  // local frame (faces at the edge of the dense grid do not count: no point lies beyond them), i.e. at global distance >=
  // fd * cell_scale - cell_sub.  Usually more than sqrt(cover2) -- and for a query WITHOUT a partner the only way to a
  // certificate that outlives the next pose update ("nothing within more than the radius").
  float cover_box2 = 0.f;
  if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
    const float ux = (lx - g.origin[0]) * g.inv_cell - (float)qr.lo[0], uy = (ly - g.origin[1]) * g.inv_cell - (float)qr.lo[1],
                uz = (lz - g.origin[2]) * g.inv_cell - (float)qr.lo[2];
    float fd = kInf;
    if (x0 > 0) fd = fminf(fd, ux - (float)x0);
    if (x1 < (int)qr.D[0] - 1) fd = fminf(fd, (float)(x1 + 1) - ux);
    if (y0 > 0) fd = fminf(fd, uy - (float)y0);
    if (y1 < (int)qr.D[1] - 1) fd = fminf(fd, (float)(y1 + 1) - uy);
    if (z0 > 0) fd = fminf(fd, uz - (float)z0);
    if (z1 < (int)qr.D[2] - 1) fd = fminf(fd, (float)(z1 + 1) - uz);
    fd = fminf(fd, 4.0f);                                   // (a box that spans the whole grid: keep the bound finite)
    const float cb = fd * bp.cell_scale - bp.cell_sub;
    cover_box2 = cb > 0.f ? cb * cb : 0.f;
  }
  const float cover_all2 = fmaxf(cover2, cover_box2);
  // fast pass over ALL scanned candidates (inside the radius or not): the two smallest distances with a strict '<' and the third
  // smallest value (`tie` is derived from the three at the end).  The radius test comes at the end.
  float bd = kInf, bd2 = kInf, b3 = kInf;
  int bpos = -1, bpos2 = -1;
  bool tie = false;
  if (x0 <= x1) {
    for (int cz = z0; cz <= z1; ++cz) {
      for (int cy = y0; cy <= y1; ++cy) {
        const size_t row = ((size_t)cz * qr.D[1] + (size_t)cy) * qr.D[0];
        const unsigned s0 = S[row + (size_t)x0], s1 = S[row + (size_t)x1 + 1];
        for (unsigned p = s0 + (unsigned)sub; p < s1; p += LPQ) {
          const float4 c = Gtgt[p];
          const float d2 = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
          const bool lt1 = d2 < bd, lt2 = d2 < bd2;
          b3 = fminf(b3, fmaxf(d2, bd2));                 // the displaced runner-up, or this candidate
          bd2 = fminf(fmaxf(d2, bd), bd2);
          bpos2 = lt1 ? bpos : (lt2 ? (int)p : bpos2);
          bd = fminf(d2, bd);
          bpos = lt1 ? (int)p : bpos;
        }
      }
    }
  }
  // merge the quad's partial lists into lane 0's (the other lanes' best and second best enter like candidates, their third
  // smallest value only bounds)
#pragma unroll
  for (int l = 1; l < LPQ; ++l) {
    const float obd = __shfl(bd, l, LPQ), obd2 = __shfl(bd2, l, LPQ), ob3 = __shfl(b3, l, LPQ);
    const int opos = __shfl(bpos, l, LPQ), opos2 = __shfl(bpos2, l, LPQ);
    if (sub == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float d2 = t == 0 ? obd : obd2;
        const int p = t == 0 ? opos : opos2;
        if (p >= 0) {
          const bool lt1 = d2 < bd, lt2 = d2 < bd2;
          b3 = fminf(b3, fmaxf(d2, bd2));
          bd2 = fminf(fmaxf(d2, bd), bd2);
          bpos2 = lt1 ? bpos : (lt2 ? p : bpos2);
          bd = fminf(d2, bd);
          bpos = lt1 ? p : bpos;
        }
      }
      b3 = fminf(b3, ob3);
    }
  }
  if (sub != 0) return;
  // The three smallest distances are an exact multiset (strict '<' updates keep equal values side by side), so an exact f32
  // equality that could change the winner or the runner-up shows as bd == bd2 or bd2 == b3 at the end -- no per-candidate test.
  tie = (bd == bd2 && bd < kInf) || (bd2 == b3 && bd2 < kInf);
  if (tie) {
    // rare (lattices, duplicates): the same cells again with the full (d2, original index) order
    bd = kInf; bd2 = kInf; b3 = kInf; bpos = -1; bpos2 = -1;
    unsigned boi = 0u, boi2 = 0u;
    for (int cz = z0; cz <= z1; ++cz) {
      for (int cy = y0; cy <= y1; ++cy) {
        const size_t row = ((size_t)cz * qr.D[1] + (size_t)cy) * qr.D[0];
        const unsigned s0 = S[row + (size_t)x0], s1 = S[row + (size_t)x1 + 1];
        for (unsigned p = s0; p < s1; ++p) {
          const float4 c = Gtgt[p];
          const float d2 = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
          const unsigned oi = __float_as_uint(c.w);
          if (d2 < bd || (d2 == bd && oi < boi)) {
            b3 = fminf(b3, bd2); bd2 = bd; boi2 = boi; bpos2 = bpos; bd = d2; boi = oi; bpos = (int)p;
          } else if (d2 < bd2 || (d2 == bd2 && oi < boi2)) {
            b3 = fminf(b3, bd2); bd2 = d2; boi2 = oi; bpos2 = (int)p;
          } else {
            b3 = fminf(b3, d2);
          }
        }
      }
    }
  }
  const bool has1 = bd < r2, has2 = bd2 < r2;            // NaN distances compare false: no partner
  match[j] = has1 ? bpos : -1;
  match2[j] = has2 ? bpos2 : -1;
  match_d2[j] = has1 ? bd : r2;
  // everything but the remembered candidates is at least this far: the third smallest distance (two remembered), the second
  // (partner only), the smallest (no partner within the radius) -- and nothing is known beyond what the scanned box covers
  const float others2 = has2 ? b3 : (has1 ? bd2 : bd);
  lbe[j] = sqrtf(fminf(others2, cover_all2)) * 0.999999f + motion_lo(q, bp.lo);
}

// -------------------------------------------------------------------------------------------------
// The bounded search over HALF cells (targets with a half-cell directory: dense scans).  Same contract as k_nn_bounded, one lane
// per query.  The box [l - rho, l + rho] is taken in half-cell coordinates; the half cells [fxa, fxb] of sub-row (fy, fz) of one
// grid cell are ONE run, found from the cell's start and its 8 prefix bytes (one 4-byte and one 8-byte load per grid cell for
// all of its sub-rows), then the candidates -- no chain of dependent lookups.  ~55 candidates instead of ~180.
// -------------------------------------------------------------------------------------------------
constexpr int kHalfRuns = 18;     // run-list slots per lane in LDS: the 3 x 3 x 2 sub-rows a box of the unrolled path can touch
// BATCH = candidate gathers in flight per lane.  PACK: a run is its 4-byte start plus ONE length byte (runs of sub-cells come from
// prefix bytes, so they hold at most 255 points; a longer run -- a cell without prefixes -- sends the query to the whole-cell rows):
// 23 instead of 36 KB of LDS per block, six instead of four blocks per CU for a search that waits on memory.  Measured on the
// 2 x 50 M bench, launches of 100 / 82 / 54 / 47 M listed queries (profiles/round3_nn_half_variants.txt): <4, false> 10.5 / 8.1 /
// 5.4 / 4.7 ms, <4, true> 8.5 / 6.5 / 4.5 / 4.0, <8, true> 8.1 / 6.3 / 4.3 / 3.8 (the choice), <12, true> (87 VGPRs, five blocks)
// 8.8 / 6.8 / 4.5 / 4.0; a 14-slot list at 64 VGPRs (eight blocks, 7 - 13 spilled registers) 11.6 / 8.9 / 6.2 / 5.4: the queries
// whose box touches more sub-rows than slots pay whole-cell rows.
template <int BATCH>
__device__ __forceinline__ void nn_bounded_half_body(const unsigned bx, const float4* __restrict__ Gsrc, const unsigned* __restrict__ list,
                                                     unsigned n_list, const float4* __restrict__ Gtgt, const unsigned* __restrict__ S,
                                                     const unsigned long long* __restrict__ H8,
                                                     const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2, const BoundParams& bp,
                                                     int* __restrict__ match, int* __restrict__ match2,
                                                     float* __restrict__ match_d2, float* __restrict__ lbe) {
  // per lane: the non-empty candidate runs of its box, a 4-byte start and ONE length byte each
  __shared__ unsigned s_runs[kHalfRuns][kBlock];
  __shared__ unsigned char s_len[kHalfRuns][kBlock];
  const unsigned i = bx * blockDim.x + threadIdx.x;
  if (i >= n_list) return;
  const unsigned j = list[i] & kListIndexMask;
  const float4 q = Gsrc[j];
  const int m = match[j];
  float d1 = r2;                                     // no old partner
  if (m >= 0) {
    const float4 pc = Gtgt[m];
    d1 = sqdist_l2(q.x, q.y, q.z, pc.x, pc.y, pc.z);
    const int m2 = match2[j];
    if (m2 >= 0) { const float4 pc2 = Gtgt[m2]; d1 = fminf(d1, sqdist_l2(q.x, q.y, q.z, pc2.x, pc2.y, pc2.z)); }
  }
  const float dr = sqrtf(fminf(d1, r2)) + ((m < 0) ? bp.np_extra : bp.margin);
  const float cover2 = (d1 < r2) ? fminf(dr * dr * 1.000001f, r2) : ((m < 0) ? dr * dr * 1.000001f : r2);      // as k_nn_bounded
  const float rho = sqrtf(cover2) * bp.rho_scale + bp.rho_pad;
  const float dx = q.x - im.t[0], dy = q.y - im.t[1], dz = q.z - im.t[2];
  const float lx = im.Linv[0] * dx + im.Linv[1] * dy + im.Linv[2] * dz;
  const float ly = im.Linv[3] * dx + im.Linv[4] * dy + im.Linv[5] * dz;
  const float lz = im.Linv[6] * dx + im.Linv[7] * dy + im.Linv[8] * dz;
  const float inv_h = 2.f * g.inv_cell;
  const int NX = 2 * (int)qr.D[0], NY = 2 * (int)qr.D[1], NZ = 2 * (int)qr.D[2];
  const int FX0 = max(cell_coord(lx - rho, g.origin[0], inv_h) - 2 * qr.lo[0], 0), FX1 = min(cell_coord(lx + rho, g.origin[0], inv_h) - 2 * qr.lo[0], NX - 1);
  const int FY0 = max(cell_coord(ly - rho, g.origin[1], inv_h) - 2 * qr.lo[1], 0), FY1 = min(cell_coord(ly + rho, g.origin[1], inv_h) - 2 * qr.lo[1], NY - 1);
  const int FZ0 = max(cell_coord(lz - rho, g.origin[2], inv_h) - 2 * qr.lo[2], 0), FZ1 = min(cell_coord(lz + rho, g.origin[2], inv_h) - 2 * qr.lo[2], NZ - 1);
  const float kInf = __uint_as_float(0x7f800000u);
  // what the scanned box covers: every target point outside it is at least `fd` HALF cells away in the target's local frame
  // (faces at the edge of the dense grid do not count), i.e. at global distance >= fd * cell_scale / 2 - cell_sub
  float cover_box2 = 0.f;
  const bool any = FX0 <= FX1 && FY0 <= FY1 && FZ0 <= FZ1;
  if (any) {
    const float ux = (lx - g.origin[0]) * inv_h - (float)(2 * qr.lo[0]), uy = (ly - g.origin[1]) * inv_h - (float)(2 * qr.lo[1]),
                uz = (lz - g.origin[2]) * inv_h - (float)(2 * qr.lo[2]);
    float fd = kInf;
    if (FX0 > 0) fd = fminf(fd, ux - (float)FX0);
    if (FX1 < NX - 1) fd = fminf(fd, (float)(FX1 + 1) - ux);
    if (FY0 > 0) fd = fminf(fd, uy - (float)FY0);
    if (FY1 < NY - 1) fd = fminf(fd, (float)(FY1 + 1) - uy);
    if (FZ0 > 0) fd = fminf(fd, uz - (float)FZ0);
    if (FZ1 < NZ - 1) fd = fminf(fd, (float)(FZ1 + 1) - uz);
    fd = fminf(fd, 8.0f);                                   // (a box that spans the whole grid: keep the bound finite)
    const float cb = fd * (bp.cell_scale * 0.5f) - bp.cell_sub;
    cover_box2 = cb > 0.f ? cb * cb : 0.f;
  }
  const float cover_all2 = fmaxf(cover2, cover_box2);
  float bd = kInf, bd2 = kInf, b3 = kInf;
  int bpos = -1, bpos2 = -1;
  // The search is latency bound, not arithmetic bound, once the candidates are few: (1) ALL directory words of the box are
  // requested before any is used -- the usual box is at most 3 x 3 half-cell rows by 2 grid cells in x, unrolled with clamped
  // addresses (larger boxes, rare, loop) -- and the non-empty runs go to a per-lane list in LDS; (2) the candidates of all runs
  // are then walked as ONE sequence, BATCH gathers in flight.
  const int cx0 = FX0 >> 1, cx1 = FX1 >> 1;
  int nr = 0;
#define RUN_S(i) s_runs[(i)][threadIdx.x]
#define RUN_LEN(i) s_len[(i)][threadIdx.x]
#define RUN_E(i) (RUN_S(i) + (unsigned)RUN_LEN(i))
  if (any) {
    const int nz = FZ1 - FZ0 + 1, ny = FY1 - FY0 + 1, nx = cx1 - cx0 + 1;
    if (nz <= 3 && ny <= 3 && nx <= 2) {
      // the (at most 2 x 2 x 2) grid cells of the box: cell start and the 8 prefix bytes, all requested at once
      const int gz0 = FZ0 >> 1, gy0 = FY0 >> 1;
      unsigned cs[8];
      unsigned long long ce[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int gz = min(gz0 + (t >> 2), FZ1 >> 1), gy = min(gy0 + ((t >> 1) & 1), FY1 >> 1), cx = min(cx0 + (t & 1), cx1);
        const size_t lin = ((size_t)gz * qr.D[1] + (size_t)gy) * qr.D[0] + (size_t)cx;
        cs[t] = S[lin];
        ce[t] = H8[lin];
      }
      // Cell by cell (compile-time cell index: no selection among the loaded words), the up to four sub-rows (z half, y half) of
      // a cell inside the box: bytes of the prefix word at compile-time positions, chosen by the x range.  A cell of more than
      // 255 points has no prefixes (all ones): the whole-cell rows below take the query.
      const int dz = (FZ1 >> 1) - gz0, dy = (FY1 >> 1) - gy0, dxc = cx1 - cx0;
      bool dense = false;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        dense = dense || ((t >> 2) <= dz && ((t >> 1) & 1) <= dy && (t & 1) <= dxc && ce[t] == ~0ull);
      if (dense) {
        nr = -1;
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if ((t >> 2) <= dz && ((t >> 1) & 1) <= dy && (t & 1) <= dxc) {
            const int cz2 = 2 * (gz0 + (t >> 2)), cy2 = 2 * (gy0 + ((t >> 1) & 1)), cx2 = 2 * (cx0 + (t & 1));
            const bool xa = FX0 > cx2, xb = FX1 > cx2;               // the x halves [xa, xb] of this cell lie in the box
            const unsigned st = cs[t], lo = (unsigned)ce[t], hi = (unsigned)(ce[t] >> 32);
#pragma unroll
            for (int zz = 0; zz < 2; ++zz)
#pragma unroll
              for (int yy = 0; yy < 2; ++yy) {
                if (cz2 + zz >= FZ0 && cz2 + zz <= FZ1 && cy2 + yy >= FY0 && cy2 + yy <= FY1) {
                  // codes (zz, yy, 0) = kb and kb + 1; byte k of the prefix word = points of the cell with a code <= k
                  const int kb = zz * 4 + yy * 2;
                  const unsigned w = zz ? hi : lo;
                  const unsigned e_prev = kb == 0 ? 0u : (kb == 4 ? (lo >> 24) : ((w >> (8 * ((kb & 3) - 1))) & 0xFFu));
                  const unsigned e_x0 = (w >> (8 * (kb & 3))) & 0xFFu, e_x1 = (w >> (8 * ((kb & 3) + 1))) & 0xFFu;
                  const unsigned e0 = xa ? e_x0 : e_prev, e1 = xb ? e_x1 : e_x0;
                  if (e0 < e1) {                             // at most 3 x 3 x 2 sub-rows lie in the box: the list cannot overflow
                    RUN_S(nr) = st + e0;
                    RUN_LEN(nr) = (unsigned char)(e1 - e0);
                    ++nr;
                  }
                }
              }
          }
        }
      }
    } else {
      nr = -1;                                             // a large box: the plain loops below
    }
  }
  if (nr > 0) {
    // One flat walk over the candidates of all runs, BATCH gathers in flight, without a branch: the search is bound by the
    // vector instructions it issues.  Squared distances are compared as bit patterns -- non-negative floats and +inf order like
    // unsigned integers, a NaN sorts above +inf and is never taken (as with '<' on floats), and v_min_u32 / v_max_u32 need no
    // canonicalisation of their operands; a slot past the last candidate holds +inf, which changes nothing.
    const unsigned uinf = 0x7f800000u;
    unsigned ud = uinf, ud2 = uinf, u3 = uinf;
    int r = 0;
    const int last = nr - 1;
    unsigned cur = RUN_S(0), end = RUN_E(0);
    while (r < nr) {
      unsigned p[BATCH];
      bool ok[BATCH];
#pragma unroll
      for (int t = 0; t < BATCH; ++t) {
        ok[t] = r < nr;
        p[t] = cur;                                        // (beyond the last candidate: a valid address, result ignored)
        const unsigned nx = cur + 1u;
        const bool sw = ok[t] && nx == end;                // the run ends here: the next one (or, after the last, its start again)
        r += sw ? 1 : 0;
        const int rc = min(r, last);
        const unsigned ns = RUN_S(rc), ne = ns + (unsigned)RUN_LEN(rc);
        cur = sw ? ns : (ok[t] ? nx : cur);
        end = sw ? ne : end;
      }
      float4 c4[BATCH];
#pragma unroll
      for (int t = 0; t < BATCH; ++t) c4[t] = Gtgt[p[t]];
#pragma unroll
      for (int t = 0; t < BATCH; ++t) {
        const unsigned u = ok[t] ? __float_as_uint(sqdist_l2(q.x, q.y, q.z, c4[t].x, c4[t].y, c4[t].z)) : uinf;
        const bool lt1 = u < ud, lt2 = u < ud2;
        u3 = min(u3, max(u, ud2));                         // the displaced runner-up, or this candidate
        ud2 = min(max(u, ud), ud2);
        bpos2 = lt1 ? bpos : (lt2 ? (int)p[t] : bpos2);
        ud = min(u, ud);
        bpos = lt1 ? (int)p[t] : bpos;
      }
    }
    bd = __uint_as_float(ud); bd2 = __uint_as_float(ud2); b3 = __uint_as_float(u3);
  } else if (nr < 0) {
    // a large box (a query without a partner looks np_extra beyond the radius; a partner that moved far): whole grid cells, row by
    // row -- the cells [x0, x1] of one (y, z) row are ONE run of the cell directory, and an empty row costs two words.  A superset
    // of the half cells of the box: nothing is missed, and the covered region only grows.
    for (int gz = FZ0 >> 1; gz <= (FZ1 >> 1); ++gz)
      for (int gy = FY0 >> 1; gy <= (FY1 >> 1); ++gy) {
        const size_t row = ((size_t)gz * qr.D[1] + (size_t)gy) * qr.D[0];
        const unsigned s0 = S[row + (size_t)cx0], s1 = S[row + (size_t)cx1 + 1];
        for (unsigned p = s0; p < s1; ++p) {
          const float4 c = Gtgt[p];
          const float d2 = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
          const bool lt1 = d2 < bd, lt2 = d2 < bd2;
          b3 = fminf(b3, fmaxf(d2, bd2));
          bd2 = fminf(fmaxf(d2, bd), bd2);
          bpos2 = lt1 ? bpos : (lt2 ? (int)p : bpos2);
          bd = fminf(d2, bd);
          bpos = lt1 ? (int)p : bpos;
        }
      }
  }
  // exact f32 equalities among the three smallest: the same candidates again with the full (d2, original index) order (rare)
  if (any && ((bd == bd2 && bd < kInf) || (bd2 == b3 && bd2 < kInf))) {
    bd = kInf; bd2 = kInf; b3 = kInf; bpos = -1; bpos2 = -1;
    unsigned boi = 0u, boi2 = 0u;
    auto exact_run = [&](unsigned s0, unsigned s1) {
      for (unsigned p = s0; p < s1; ++p) {
        const float4 c = Gtgt[p];
        const float d2 = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
        const unsigned oi = __float_as_uint(c.w);
        if (d2 < bd || (d2 == bd && oi < boi)) {
          b3 = fminf(b3, bd2); bd2 = bd; boi2 = boi; bpos2 = bpos; bd = d2; boi = oi; bpos = (int)p;
        } else if (d2 < bd2 || (d2 == bd2 && oi < boi2)) {
          b3 = fminf(b3, bd2); bd2 = d2; boi2 = oi; bpos2 = (int)p;
        } else {
          b3 = fminf(b3, d2);
        }
      }
    };
    if (nr >= 0) {
      for (int r = 0; r < nr; ++r) exact_run(RUN_S(r), RUN_E(r));
    } else {
      for (int gz = FZ0 >> 1; gz <= (FZ1 >> 1); ++gz)
        for (int gy = FY0 >> 1; gy <= (FY1 >> 1); ++gy) {
          const size_t row = ((size_t)gz * qr.D[1] + (size_t)gy) * qr.D[0];
          exact_run(S[row + (size_t)cx0], S[row + (size_t)cx1 + 1]);
        }
    }
  }
#undef RUN_S
#undef RUN_E
#undef RUN_LEN
  const bool has1 = bd < r2, has2 = bd2 < r2;            // NaN distances compare false: no partner
  match[j] = has1 ? bpos : -1;
  match2[j] = has2 ? bpos2 : -1;
  match_d2[j] = has1 ? bd : r2;
  const float others2 = has2 ? b3 : (has1 ? bd2 : bd);
  lbe[j] = sqrtf(fminf(others2, cover_all2)) * 0.999999f + motion_lo(q, bp.lo);
}

template <int BATCH>
__global__ __launch_bounds__(kBlock) void k_nn_bounded_half(const float4* __restrict__ Gsrc, const unsigned* __restrict__ list,
                                                            unsigned n_list, const float4* __restrict__ Gtgt, const unsigned* __restrict__ S,
                                                            const unsigned long long* __restrict__ H8,
                                                            GridDesc g, InvMap im, QueryRange qr, float r2, BoundParams bp,
                                                            int* __restrict__ match, int* __restrict__ match2,
                                                            float* __restrict__ match_d2, float* __restrict__ lbe) {
  nn_bounded_half_body<BATCH>(blockIdx.x, Gsrc, list, n_list, Gtgt, S, H8, g, im, qr, r2, bp, match, match2, match_d2, lbe);
}

// the listed queries of a batch of pairs: one job per (pair, list) -- the near list and, when it is short, the far list of a pair
// are disjoint sets of queries and run side by side
__global__ __launch_bounds__(kBlock) void k_nn_bounded_half_multi(const NnBatchDev* __restrict__ B, float r2) {
  const int jb = nn_find_range(B->job_end, B->n_jobs, blockIdx.x);
  const unsigned bx = blockIdx.x - (jb ? B->job_end[jb - 1] : 0u);
  const int p = __builtin_amdgcn_readfirstlane(B->job_pair[jb]);
  const NnPairDev& P = B->pair[p];
  const GridDesc g = P.g; const InvMap im = P.im; const QueryRange qr = P.qr; const BoundParams bp = P.bp;      // (block-uniform: scalar registers)
  nn_bounded_half_body<8>(bx, P.Gsrc, B->job_list[jb], B->job_n[jb], P.Gtgt, P.S, P.H8, g, im, qr, r2, bp, P.match, P.match2,
                          P.match_d2, P.lbe);
}

// The FAR lists of a batch of pairs (round 6): queries without a near partner, too many for the bounded search -- the first outer
// iterations of every pair.  Round 5 keyed, sorted and searched them pair by pair: per pair a key kernel, a read-back of the kept
// count, a radix sort (six launches) and k_nn_rows -- 240 times per outer iteration of a 16-scan job, ~0.3 ms each however short a
// rank's slices are.  Now ONE key kernel walks the pairs' lists (a block finds its pair from the ends of the block ranges) and
// writes (key, query) pairs of all of them into one array, the pair's index above the cell key; ONE radix sort orders them by
// (pair, cell); ONE k_nn_rows launch walks the pairs' stretches of the sorted array.  The bodies are the one-pair kernels': the same
// results bit for bit (a query's result does not depend on its neighbours in the sorted order).
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_query_keys_multi(const NnBatchDev* __restrict__ B, float r2, KeyT* __restrict__ keys, unsigned* __restrict__ vals,
                                                             unsigned* __restrict__ counts) {
  const int p = nn_find_range(B->far_end, B->n_pairs, blockIdx.x);
  const unsigned bx = blockIdx.x - (p ? B->far_end[p - 1] : 0u);
  const NnPairDev& P = B->pair[p];
  const GridDesc g = P.g; const InvMap im = P.im; const QueryRange qr = P.qr;
  const CertParams cert = {P.bp.cell_scale, P.bp.cell_sub, P.bp.lo};
  const int shift = B->key_shift;
  const KeyT mask = (KeyT)(((KeyT)1 << shift) - (KeyT)1), hi = (KeyT)((KeyT)p << shift);
  const int flags = __builtin_amdgcn_readfirstlane(P.far_flags);
  query_keys_prune_body<KeyT>(bx, P.Gsrc, P.far_list, (size_t)P.far_n, P.occ, P.occ_stride, g, im, qr, r2, cert, keys, vals, counts, counts + 1 + p, mask, hi,
                              (flags & 2) != 0, P.match, P.match2, P.match_d2, P.lbe, flags & 1);
}

// The far lists' key kernel WITH SEEDS (round 6).  While two scans meet, every query changes its partner: the far lists hold all
// queries, k_nn_rows scores each against the ~10^3 candidates of its row segments (12 ms per 100 M queries, after a 3.5 ms sort).
// But most of those queries have a target point a few millimetres away -- and any target point at distance s bounds the search to
// the ball of radius s, which is what the bounded search (k_nn_bounded_half) does around an OLD partner at a third of the cost.
// This kernel gives a query a partner to start from: after the occupancy test of k_query_keys_prune it probes up to eight
// points of the query's own half cell (the cell's prefix bytes; an empty half cell: points spread over the whole cell) and the old
// partner, if any.  A query with a probe (or old partner) nearer than sqrt(seed2) gets that point as match[j] and goes to the
// pair's SEEDED list -- a job of the bounded search, which treats it like any near-list query (the ball around a real target point
// at its exact f32 distance: the exact search inside it finds the nearest neighbour, ties included); the others are keyed for the
// sort and k_nn_rows as before.  Results are those of either exact search: identical.
constexpr int kSeedPerThread = 4;
static_assert((unsigned)(kBlock * kSeedPerThread) == kQuerySeedBlock, "queries per block of the seeding key kernel");
template <typename KeyT, int kSeedProbes>
__global__ __launch_bounds__(kBlock) void k_query_seed_multi(const NnBatchDev* __restrict__ B, float r2, KeyT* __restrict__ keys, unsigned* __restrict__ vals,
                                                             unsigned* __restrict__ counts) {
  __shared__ unsigned s_cnt[2][kBlock / kWave][kSeedPerThread];
  __shared__ unsigned s_base[2];
  const int p = nn_find_range(B->far_end, B->n_pairs, blockIdx.x);
  const unsigned bx = blockIdx.x - (p ? B->far_end[p - 1] : 0u);
  const NnPairDev& P = B->pair[p];
  const GridDesc g = P.g; const InvMap im = P.im; const QueryRange qr = P.qr;
  const CertParams cert = {P.bp.cell_scale, P.bp.cell_sub, P.bp.lo};
  const int shift = B->key_shift;
  const KeyT key_mask = (KeyT)(((KeyT)1 << shift) - (KeyT)1), key_or = (KeyT)((KeyT)p << shift);
  const int flags = __builtin_amdgcn_readfirstlane(P.far_flags);
  const bool from_state = (flags & 1) != 0, prune = (flags & 2) != 0, seed_on = (flags & 4) != 0;
  const float4* __restrict__ Gsrc = P.Gsrc; const float4* __restrict__ Gtgt = P.Gtgt;
  const unsigned* __restrict__ list = P.far_list; const unsigned* __restrict__ occ = P.occ; const unsigned* __restrict__ S = P.S;
  const unsigned long long* __restrict__ H8 = P.H8;
  int* __restrict__ match = P.match; int* __restrict__ match2 = P.match2; float* __restrict__ match_d2 = P.match_d2; float* __restrict__ lbe = P.lbe;
  const size_t n = (size_t)P.far_n;
  const unsigned stride_w = P.occ_stride;
  const float seed2 = P.seed2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t i0 = (size_t)bx * (kBlock * kSeedPerThread) + (size_t)w * kWave + (size_t)lane;      // step u: + u * kBlock
  KeyT kk[kSeedPerThread];
  unsigned jf[kSeedPerThread];
  float4 qq[kSeedPerThread];
  unsigned long long keyed_mask[kSeedPerThread], seeded_mask[kSeedPerThread];       // wave-uniform ballots
#pragma unroll
  for (int u = 0; u < kSeedPerThread; ++u) {
    const size_t i = i0 + (size_t)u * kBlock;
    jf[u] = (i < n) ? (list ? list[i] : ((unsigned)i | ((from_state && match[i] < 0) ? kListNoPartner : 0u))) : 0u;
  }
#pragma unroll
  for (int u = 0; u < kSeedPerThread; ++u) qq[u] = Gsrc[jf[u] & kListIndexMask];
  unsigned wordv[kSeedPerThread];
  int bitv[kSeedPerThread];
  float bdist[kSeedPerThread];
  unsigned long long lin[kSeedPerThread];
  unsigned code[kSeedPerThread];
#pragma unroll
  for (int u = 0; u < kSeedPerThread; ++u) {
    const float4 q = qq[u];
    const float dx = q.x - im.t[0], dy = q.y - im.t[1], dz = q.z - im.t[2];
    const float lx = im.Linv[0] * dx + im.Linv[1] * dy + im.Linv[2] * dz;
    const float ly = im.Linv[3] * dx + im.Linv[4] * dy + im.Linv[5] * dz;
    const float lz = im.Linv[6] * dx + im.Linv[7] * dy + im.Linv[8] * dz;
    int cx = 0, cy = 0, cz = 0;
    bdist[u] = 2.0f;
    lin[u] = query_cell_key(q, im, g, qr, cx, cy, cz, &bdist[u]);
    code[u] = half_code(lx, ly, lz, g);
    kk[u] = ((KeyT)lin[u] & key_mask) | key_or;
    wordv[u] = 0u; bitv[u] = -1;
    if (lin[u] != kEmptyKey) {
      if (prune) {
        const int kx = cx - qr.lo[0], ky = cy - qr.lo[1], kz = cz - qr.lo[2];
        wordv[u] = occ[((size_t)kz * qr.D[1] + (size_t)ky) * stride_w + (size_t)(kx >> 5)];
        bitv[u] = kx & 31;
      }
    } else {
      bdist[u] = 2.0f;                                   // outside the directory range: two empty cells all around (k_nn_rows)
    }
  }
  // the own cell's run and prefix bytes, the old partner: requested for all steps before any is used
  bool keep[kSeedPerThread];
  unsigned s0[kSeedPerThread], s1[kSeedPerThread];
  unsigned long long h8[kSeedPerThread];
  int mm[kSeedPerThread];
#pragma unroll
  for (int u = 0; u < kSeedPerThread; ++u) {
    const size_t i = i0 + (size_t)u * kBlock;
    const bool valid = i < n;
    keep[u] = valid && (prune ? (bitv[u] >= 0 && ((wordv[u] >> bitv[u]) & 1u)) : true);
    const bool probe = keep[u] && seed_on && lin[u] != kEmptyKey;
    const size_t l = probe ? (size_t)lin[u] : 0;
    s0[u] = S[l]; s1[u] = S[l + 1];
    h8[u] = H8[l];
    mm[u] = probe ? match[jf[u] & kListIndexMask] : -1;
    if (!probe) { s0[u] = 0u; s1[u] = 0u; }
  }
#pragma unroll
  for (int u = 0; u < kSeedPerThread; ++u) {
    const size_t i = i0 + (size_t)u * kBlock;
    const bool valid = i < n;
    const unsigned j = jf[u] & kListIndexMask;
    const float4 q = qq[u];
    if (valid && from_state && (jf[u] & kListNoPartner)) match_d2[j] = r2;
    if (valid && !keep[u]) {
      if (!(jf[u] & kListNoPartner)) {                   // (see k_nn_rows: a query that had no partner holds these values already)
        match[j] = -1; match_d2[j] = r2;
        if (match2) match2[j] = -1;
      }
      const float kInf = __uint_as_float(0x7f800000u);
      const float lb_out = bdist[u] * cert.cell_scale - cert.cell_sub;
      lbe[j] = fmaxf(fminf(sqrtf(kInf), lb_out), 0.0f) * 0.999999f + motion_lo(q, cert.lo);
    }
    // probes: the own half cell's run, else the whole cell
    unsigned a = s0[u], b = s1[u];
    if (h8[u] != ~0ull && b > a) {
      const unsigned c = code[u];
      const unsigned e_hi = (unsigned)(h8[u] >> (8 * c)) & 0xFFu, e_lo = c ? ((unsigned)(h8[u] >> (8 * (c - 1))) & 0xFFu) : 0u;
      if (e_hi > e_lo) { b = a + e_hi; a = a + e_lo; }
    }
    const unsigned len = b - a;
    float4 c4[kSeedProbes];
    unsigned pp[kSeedProbes];
#pragma unroll
    for (int t = 0; t < kSeedProbes; ++t) {
      const unsigned o = len > (unsigned)kSeedProbes ? ((unsigned)t * len) / (unsigned)kSeedProbes : min((unsigned)t, len ? len - 1u : 0u);
      pp[t] = a + o;                                     // (no point in the cell: len = 0, position a = 0: a valid address, result ignored)
      c4[t] = Gtgt[pp[t]];
    }
    const float4 co = Gtgt[mm[u] >= 0 ? mm[u] : 0];
    const unsigned uinf = 0x7f800000u;
    unsigned ub = uinf;
    int bpos = -1;
#pragma unroll
    for (int t = 0; t < kSeedProbes; ++t) {
      const unsigned ud = len ? __float_as_uint(sqdist_l2(q.x, q.y, q.z, c4[t].x, c4[t].y, c4[t].z)) : uinf;     // (bit patterns: as k_nn_bounded_half)
      bpos = ud < ub ? (int)pp[t] : bpos;
      ub = min(ud, ub);
    }
    const unsigned uo = mm[u] >= 0 ? __float_as_uint(sqdist_l2(q.x, q.y, q.z, co.x, co.y, co.z)) : uinf;
    const unsigned us = __float_as_uint(seed2);
    const bool by_old = uo < us && uo <= ub;              // the old partner (and its runner-up) stay: the bounded search reads both
    const bool by_probe = !by_old && ub < us;
    if (by_probe) { match[j] = bpos; if (match2) match2[j] = -1; }
    const bool seeded = keep[u] && (by_old || by_probe);
    keyed_mask[u] = __ballot(keep[u] && !seeded);
    seeded_mask[u] = __ballot(seeded);
    if (lane == 0) { s_cnt[0][w][u] = (unsigned)__popcll(keyed_mask[u]); s_cnt[1][w][u] = (unsigned)__popcll(seeded_mask[u]); }
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    unsigned tot = 0u;
    for (int ww = 0; ww < kBlock / kWave; ++ww)
      for (int u = 0; u < kSeedPerThread; ++u) tot += s_cnt[k][ww][u];
    // counts[0]: keyed pairs of the batch (the sort's size), counts[1 + p]: of this pair, counts[1 + kNnBatchPairs + p]: seeded
    s_base[k] = 0u;
    if (tot) {
      if (k == 0) { s_base[0] = atomicAdd(counts, tot); atomicAdd(counts + 1 + p, tot); }
      else s_base[1] = atomicAdd(counts + 1 + kNnBatchPairs + p, tot);
    }
  }
  __syncthreads();
  unsigned offk = s_base[0], offs = s_base[1];
  for (int ww = 0; ww < w; ++ww)
#pragma unroll
    for (int u = 0; u < kSeedPerThread; ++u) { offk += s_cnt[0][ww][u]; offs += s_cnt[1][ww][u]; }
  unsigned* __restrict__ seed_list = P.seed_list;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int u = 0; u < kSeedPerThread; ++u) {
    if ((keyed_mask[u] >> lane) & 1ull) {
      const unsigned slot = offk + (unsigned)__popcll(keyed_mask[u] & below);
      keys[slot] = kk[u];
      vals[slot] = jf[u];
    }
    if ((seeded_mask[u] >> lane) & 1ull) seed_list[offs + (unsigned)__popcll(seeded_mask[u] & below)] = jf[u] & kListIndexMask;
    offk += (unsigned)__popcll(keyed_mask[u]);
    offs += (unsigned)__popcll(seeded_mask[u]);
  }
}

__global__ __launch_bounds__(kBlock, 6) void k_nn_rows_multi(const NnBatchDev* __restrict__ B, const unsigned* __restrict__ order, float r2, int row_span) {
  const int p = nn_find_range(B->rows_end, B->n_pairs, blockIdx.x);
  const unsigned bx = blockIdx.x - (p ? B->rows_end[p - 1] : 0u);
  const NnPairDev& P = B->pair[p];
  const GridDesc g = P.g; const InvMap im = P.im; const QueryRange qr = P.qr;
  const CertParams cert = {P.bp.cell_scale, P.bp.cell_sub, P.bp.lo};
  nn_rows_body(bx, P.Gsrc, order + P.rows_off, (size_t)P.rows_n, P.Gtgt, P.S, g, im, qr, r2, row_span, cert, P.match, P.match_d2, P.lbe, P.match2);
}

// flags -> per-block counts (first stage of the order-preserving compaction)
__global__ __launch_bounds__(kBlock) void k_match_block_counts(const int* __restrict__ match_pos, size_t n,
                                                               unsigned* __restrict__ block_counts,
                                                               double* __restrict__ block_d2,
                                                               const float* __restrict__ match_d2) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool f = (j < n) && (ld_stream(match_pos + j) >= 0);
  const unsigned long long b = __ballot(f);
  double d = (f && match_d2) ? (double)ld_stream(match_d2 + j) : 0.0;          // match_d2 == nullptr: counts only
  d = wave_sum(d);
  __shared__ unsigned sc[kBlock / kWave];
  __shared__ double sd[kBlock / kWave];
  if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = (unsigned)__popcll(b); sd[threadIdx.x >> 6] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned c = 0; double t = 0;
    for (int k = 0; k < kBlock / kWave; ++k) { c += sc[k]; t += sd[k]; }
    block_counts[blockIdx.x] = c; block_d2[blockIdx.x] = t;
  }
}

// exclusive scan of the per-block counts in three small steps: chunk sums (kScanChunk entries per chunk),
// single-block scan of the chunk sums, per-chunk scan with its base.
constexpr int kScanChunk = 256;
__device__ __forceinline__ void scan_chunk_sums_body(const unsigned bx, const unsigned* __restrict__ counts, int nblocks,
                                                     const double* __restrict__ block_d2,
                                                     unsigned long long* __restrict__ chunk_sum,
                                                     double* __restrict__ chunk_d2,
                                                     const unsigned* __restrict__ groups,
                                                     unsigned* __restrict__ chunk_groups,
                                                     unsigned* __restrict__ chunk_rewritten) {
  const int b = (int)bx * kScanChunk + threadIdx.x;
  unsigned long long c = (b < nblocks) ? counts[b] : 0ull;
  double d = (b < nblocks) ? block_d2[b] : 0.0;
  const unsigned gw = (groups && b < nblocks) ? groups[b] : 0u;
  unsigned g = gw & 0xFFu, rw = gw >> 16;      // resident rows: active 64-row groups of the block, rows rewritten
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); g += __shfl_xor(g, o, 64); rw += __shfl_xor(rw, o, 64); }
  __shared__ unsigned long long sc[kScanChunk / kWave];
  __shared__ double sd[kScanChunk / kWave];
  __shared__ unsigned sg[kScanChunk / kWave], sr[kScanChunk / kWave];
  if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = c; sd[threadIdx.x >> 6] = d; sg[threadIdx.x >> 6] = g; sr[threadIdx.x >> 6] = rw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0; double td = 0; unsigned tg = 0, tr = 0;
    for (int k = 0; k < kScanChunk / kWave; ++k) { t += sc[k]; td += sd[k]; tg += sg[k]; tr += sr[k]; }
    chunk_sum[bx] = t; chunk_d2[bx] = td;
    if (chunk_groups) { chunk_groups[bx] = tg; chunk_rewritten[bx] = tr; }
  }
}

__global__ __launch_bounds__(kScanChunk) void k_scan_chunk_sums(const unsigned* __restrict__ counts, int nblocks,
                                                                const double* __restrict__ block_d2,
                                                                unsigned long long* __restrict__ chunk_sum,
                                                                double* __restrict__ chunk_d2,
                                                                const unsigned* __restrict__ groups = nullptr,
                                                                unsigned* __restrict__ chunk_groups = nullptr,
                                                                unsigned* __restrict__ chunk_rewritten = nullptr) {
  scan_chunk_sums_body(blockIdx.x, counts, nblocks, block_d2, chunk_sum, chunk_d2, groups, chunk_groups, chunk_rewritten);
}

// totals (and the exclusive scan of the chunk sums); chunk_groups (resident rows): scanned in place as well, total[1] = their sum
__device__ __forceinline__ void scan_chunks_body(unsigned long long* __restrict__ chunk_sum, int nchunks, const double* __restrict__ chunk_d2,
                                                 unsigned long long* __restrict__ total, double* __restrict__ total_d2,
                                                 unsigned* __restrict__ chunk_groups, const unsigned* __restrict__ chunk_rewritten) {
  __shared__ unsigned long long s[1024];
  __shared__ double sd[1024];
  __shared__ unsigned sg[1024];
  __shared__ unsigned long long sr[1024];
  const int t = threadIdx.x, T = blockDim.x;
  const int per = (nchunks + T - 1) / T;
  const int b0 = min(nchunks, t * per), b1 = min(nchunks, b0 + per);
  unsigned long long sum = 0, rs = 0; double d = 0; unsigned gs = 0;
  for (int b = b0; b < b1; ++b) { sum += chunk_sum[b]; d += chunk_d2[b]; if (chunk_groups) { gs += chunk_groups[b]; rs += chunk_rewritten[b]; } }
  s[t] = sum; sd[t] = d; sg[t] = gs; sr[t] = rs;
  __syncthreads();
  if (t == 0) {
    unsigned long long run = 0, rr = 0; double dr = 0; unsigned gr = 0;
    for (int k = 0; k < T; ++k) {
      const unsigned long long v = s[k]; s[k] = run; run += v; dr += sd[k];
      const unsigned gv = sg[k]; sg[k] = gr; gr += gv; rr += sr[k];
    }
    *total = run; *total_d2 = dr;
    if (chunk_groups) { total[1] = gr; total[2] = rr; }
  }
  __syncthreads();
  unsigned long long run = s[t];
  unsigned grun = sg[t];
  for (int b = b0; b < b1; ++b) {
    const unsigned long long v = chunk_sum[b]; chunk_sum[b] = run; run += v;
    if (chunk_groups) { const unsigned gv = chunk_groups[b]; chunk_groups[b] = grun; grun += gv; }
  }
}

__global__ void k_scan_chunks(unsigned long long* __restrict__ chunk_sum, int nchunks, const double* __restrict__ chunk_d2,
                              unsigned long long* __restrict__ total, double* __restrict__ total_d2,
                              unsigned* __restrict__ chunk_groups = nullptr, const unsigned* __restrict__ chunk_rewritten = nullptr) {
  scan_chunks_body(chunk_sum, nchunks, chunk_d2, total, total_d2, chunk_groups, chunk_rewritten);
}

__global__ __launch_bounds__(kScanChunk) void k_scan_within_chunks(const unsigned* __restrict__ counts, int nblocks,
                                                                   const unsigned long long* __restrict__ chunk_base,
                                                                   unsigned* __restrict__ offsets) {
  const int b = blockIdx.x * kScanChunk + threadIdx.x;
  const unsigned c = (b < nblocks) ? counts[b] : 0u;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __shared__ unsigned ws[kScanChunk / kWave];
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  unsigned wbase = 0;
  for (int k = 0; k < w; ++k) wbase += ws[k];
  if (b < nblocks) offsets[b] = (unsigned)chunk_base[blockIdx.x] + wbase + (inc - c);
}

// second stage: write the correspondence planes in source (cell) order
__global__ __launch_bounds__(kBlock) void k_compact_corr(const int* __restrict__ match_pos,
                                                         const unsigned* __restrict__ order, size_t n,
                                                         const unsigned* __restrict__ block_offsets,
                                                         const float4* __restrict__ Gsrc, const float4* __restrict__ LNsrc,
                                                         Affine Tsrc, const float4* __restrict__ Gtgt,
                                                         const float4* __restrict__ LNtgt, Affine Ttgt,
                                                         float4* __restrict__ A, float4* __restrict__ B,
                                                         float4* __restrict__ C, size_t out_base) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int m = (j < n) ? ld_stream(match_pos + j) : -1;
  const bool f = m >= 0;
  const unsigned long long b = __ballot(f);
  __shared__ unsigned wbase[kBlock / kWave];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wbase[w] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned base = block_offsets[blockIdx.x];
  for (int k = 0; k < w; ++k) base += wbase[k];
  if (!f) return;
  const unsigned rank = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
  const size_t o = out_base + base + rank;
  const size_t js = order ? (size_t)ld_stream(order + j) : j;     // query j of the (sorted) search order -> source position
  const float4 sp = order ? Gsrc[js] : ld_stream(Gsrc + js);
  const float4 ln = order ? LNsrc[js] : ld_stream(LNsrc + js);
  const float3 sn = pcl_so3(Tsrc, ln.x, ln.y, ln.z);
  const float4 tp = Gtgt[m];
  const float4 tl = LNtgt[m];
  const float3 tn = pcl_so3(Ttgt, tl.x, tl.y, tl.z);
  st_stream(A + o, make_float4(sp.x, sp.y, sp.z, sn.x));
  st_stream(B + o, make_float4(sn.y, sn.z, tp.x, tp.y));
  st_stream(C + o, make_float4(tp.z, tn.x, tn.y, tn.z));
}

// Resident rows (LmSet): one row per query of the pair, at its source position.  Rewrites the rows whose partner differs from the
// one the row encodes -- in the settled state of an alignment a handful per launch, where k_compact_corr gathered and rewrote all
// of them every outer iteration -- and produces what the progress line and the LM passes need from the match list: per-block
// match counts and squared-distance sums (the arithmetic of k_match_block_counts) and the active 64-row groups of the block
// (count in bits 0..7, mask in bits 8..11; bits 16..24: rows rewritten, a statistic -- summed without atomics: an atomic per
// wave on one counter made a launch that rewrites every row 3.5 times slower than k_compact_corr).  Halves of clouds that never move (impl cloud 0, fixed clouds) are stored in the
// global frame exactly as k_compact_corr writes them; halves of movable clouds in the cloud's local frame (row_to_global).
__device__ __forceinline__ void corr_update_body(const unsigned bx, const int* __restrict__ match, int* __restrict__ plane_match,
                                                 const float* __restrict__ match_d2, size_t n,
                                                 const float4* __restrict__ Psrc, const float4* __restrict__ LNsrc,
                                                 const int src_global, const Affine& Tsrc, const float4* __restrict__ Ptgt,
                                                 const float4* __restrict__ LNtgt, const int tgt_global, const Affine& Ttgt,
                                                 float4* __restrict__ A, float4* __restrict__ B, float4* __restrict__ C,
                                                 unsigned* __restrict__ block_counts, double* __restrict__ block_d2,
                                                 unsigned* __restrict__ block_groups, const unsigned char* __restrict__ done = nullptr) {
  if (done && done[bx]) return;      // every query settled by its certificate: nn_certify_body wrote this block's results
  const size_t j = (size_t)bx * blockDim.x + threadIdx.x;
  const bool in = j < n;
  const int m = in ? ld_stream(match + j) : -1;
  const int pm = in ? ld_stream(plane_match + j) : -1;
  const bool f = m >= 0;
  // Rows are rewritten in whole 128-byte lines (8 rows of a plane): a lone 16-byte store is a partial line write all the way to
  // HBM -- measured: 3.5 ms per launch with 5 % of the rows rewritten one by one, against 1.0 ms for ALL rows in full lines.
  const unsigned long long chg = __ballot(in && m != pm);
  const bool wr = in && (((chg >> (threadIdx.x & 56)) & 0xFFull) != 0ull);
  const unsigned long long wrb = __ballot(wr);
  if (wr) {
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    if (f) {
      const float4 sp = Psrc[j];
      const float4 ln = LNsrc[j];
      float3 sn = make_float3(ln.x, ln.y, ln.z);
      if (src_global) sn = pcl_so3(Tsrc, ln.x, ln.y, ln.z);
      const float4 tp = Ptgt[m];
      const float4 tl = LNtgt[m];
      float3 tn = make_float3(tl.x, tl.y, tl.z);
      if (tgt_global) tn = pcl_so3(Ttgt, tl.x, tl.y, tl.z);
      ra = make_float4(sp.x, sp.y, sp.z, sn.x);
      rb = make_float4(sn.y, sn.z, tp.x, tp.y);
      rc = make_float4(tp.z, tn.x, tn.y, tn.z);
    }
    st_stream(A + j, ra); st_stream(B + j, rb); st_stream(C + j, rc);
    if (m != pm) plane_match[j] = m;
  }
  const unsigned long long b = __ballot(f);
  double d = f ? (double)ld_stream(match_d2 + j) : 0.0;
  d = wave_sum(d);
  __shared__ unsigned sc[kBlock / kWave], sw[kBlock / kWave];
  __shared__ double sd[kBlock / kWave];
  if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = (unsigned)__popcll(b); sw[threadIdx.x >> 6] = (unsigned)__popcll(wrb); sd[threadIdx.x >> 6] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned c = 0, g = 0, gm = 0, rw = 0; double t = 0;
    for (int k = 0; k < kBlock / kWave; ++k) { c += sc[k]; t += sd[k]; rw += sw[k]; if (sc[k]) { ++g; gm |= 1u << k; } }
    block_counts[bx] = c; block_d2[bx] = t; block_groups[bx] = g | (gm << 8) | (rw << 16);    // (rw <= 256)
  }
}

__global__ __launch_bounds__(kBlock) void k_corr_update(const int* __restrict__ match, int* __restrict__ plane_match,
                                                        const float* __restrict__ match_d2, size_t n,
                                                        const float4* __restrict__ Psrc, const float4* __restrict__ LNsrc,
                                                        const int src_global, Affine Tsrc, const float4* __restrict__ Ptgt,
                                                        const float4* __restrict__ LNtgt, const int tgt_global, Affine Ttgt,
                                                        float4* __restrict__ A, float4* __restrict__ B, float4* __restrict__ C,
                                                        unsigned* __restrict__ block_counts, double* __restrict__ block_d2,
                                                        unsigned* __restrict__ block_groups) {
  corr_update_body(blockIdx.x, match, plane_match, match_d2, n, Psrc, LNsrc, src_global, Tsrc, Ptgt, LNtgt, tgt_global, Ttgt, A, B, C,
                   block_counts, block_d2, block_groups);
}

// the resident rows of a batch of pairs: the per-block results of pair p go to entries [blk0(p), blk0(p) + blocks of p) of the batch's arrays
__global__ __launch_bounds__(kBlock) void k_corr_update_multi(const NnBatchDev* __restrict__ Bt, unsigned* __restrict__ block_counts,
                                                              double* __restrict__ block_d2, unsigned* __restrict__ block_groups) {
  const int p = nn_find_range(Bt->upd_end, Bt->n_pairs, blockIdx.x);
  const unsigned b0 = p ? Bt->upd_end[p - 1] : 0u;
  const NnPairDev& P = Bt->pair[p];
  const Affine Ts = P.Tsrc, Tt = P.Ttgt;
  corr_update_body(blockIdx.x - b0, P.match, P.plane_match, P.match_d2, (size_t)P.n, P.Psrc, P.LNsrc, P.src_global, Ts, P.Ptgt, P.LNtgt,
                   P.tgt_global, Tt, P.A, P.B, P.C, block_counts + b0, block_d2 + b0, block_groups + b0, P.upd_done);
}

// ascending list of the active 64-row groups: one thread per query block (4 groups), scan inside the chunk + the chunk's base
__device__ __forceinline__ void group_list_body(const unsigned bx, const unsigned* __restrict__ block_groups, int nblocks,
                                                const unsigned* __restrict__ chunk_gbase, unsigned* __restrict__ glist) {
  const int b = (int)bx * kScanChunk + threadIdx.x;
  const unsigned w = (b < nblocks) ? block_groups[b] : 0u;
  const unsigned c = w & 0xFFu;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __shared__ unsigned ws[kScanChunk / kWave];
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  unsigned base = chunk_gbase[bx] + (inc - c);
  for (int k = 0; k < wv; ++k) base += ws[k];
  const unsigned mask = w >> 8;
#pragma unroll
  for (int k = 0; k < kBlock / kWave; ++k)
    if (mask & (1u << k)) glist[base++] = (unsigned)b * (kBlock / kWave) + (unsigned)k;
}

__global__ __launch_bounds__(kScanChunk) void k_group_list(const unsigned* __restrict__ block_groups, int nblocks,
                                                           const unsigned* __restrict__ chunk_gbase,
                                                           unsigned* __restrict__ glist) {
  group_list_body(blockIdx.x, block_groups, nblocks, chunk_gbase, glist);
}

// totals and group lists of a batch of pairs (launch_corr_totals for each of them, three launches in all): chunk c of the batch
// belongs to the pair whose chunk range holds it; pair p's totals go to totals[3 p ..], total_d2[p]
__global__ __launch_bounds__(kScanChunk) void k_scan_chunk_sums_multi(const NnBatchDev* __restrict__ Bt, const unsigned* __restrict__ block_counts,
                                                                      const double* __restrict__ block_d2, const unsigned* __restrict__ block_groups,
                                                                      unsigned long long* __restrict__ chunk_sum, double* __restrict__ chunk_d2,
                                                                      unsigned* __restrict__ chunk_groups, unsigned* __restrict__ chunk_rewritten) {
  const int p = nn_find_range(Bt->chunk_end, Bt->n_pairs, blockIdx.x);
  const unsigned c0 = p ? Bt->chunk_end[p - 1] : 0u, b0 = p ? Bt->upd_end[p - 1] : 0u;
  scan_chunk_sums_body(blockIdx.x - c0, block_counts + b0, (int)(Bt->upd_end[p] - b0), block_d2 + b0, chunk_sum + c0, chunk_d2 + c0, block_groups + b0,
                       chunk_groups + c0, chunk_rewritten + c0);
}
__global__ __launch_bounds__(1024) void k_scan_chunks_multi(const NnBatchDev* __restrict__ Bt, unsigned long long* __restrict__ chunk_sum,
                                                            const double* __restrict__ chunk_d2, unsigned long long* __restrict__ totals,
                                                            double* __restrict__ total_d2, unsigned* __restrict__ chunk_groups,
                                                            const unsigned* __restrict__ chunk_rewritten) {
  const int p = blockIdx.x;
  const unsigned c0 = p ? Bt->chunk_end[p - 1] : 0u;
  scan_chunks_body(chunk_sum + c0, (int)(Bt->chunk_end[p] - c0), chunk_d2 + c0, totals + 3 * p, total_d2 + p, chunk_groups + c0, chunk_rewritten + c0);
}
__global__ __launch_bounds__(kScanChunk) void k_group_list_multi(const NnBatchDev* __restrict__ Bt, const unsigned* __restrict__ block_groups,
                                                                 const unsigned* __restrict__ chunk_gbase) {
  const int p = nn_find_range(Bt->chunk_end, Bt->n_pairs, blockIdx.x);
  const unsigned c0 = p ? Bt->chunk_end[p - 1] : 0u, b0 = p ? Bt->upd_end[p - 1] : 0u;
  group_list_body(blockIdx.x - c0, block_groups + b0, (int)(Bt->upd_end[p] - b0), chunk_gbase + c0, Bt->pair[p].glist);
}

// gather variant for explicit (index_query, index_match) lists on unsorted AoS clouds
// (stand-alone e3d_icp_pair_system entry point)
__global__ __launch_bounds__(kBlock) void k_gather_corr(const float* __restrict__ sxyz, const float* __restrict__ snrm,
                                                        const float* __restrict__ txyz, const float* __restrict__ tnrm,
                                                        const int* __restrict__ iq, const int* __restrict__ im, size_t n,
                                                        float4* __restrict__ A, float4* __restrict__ B,
                                                        float4* __restrict__ C) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const size_t s = (size_t)iq[c], t = (size_t)im[c];
  A[c] = make_float4(sxyz[3 * s], sxyz[3 * s + 1], sxyz[3 * s + 2], snrm[3 * s]);
  B[c] = make_float4(snrm[3 * s + 1], snrm[3 * s + 2], txyz[3 * t], txyz[3 * t + 1]);
  C[c] = make_float4(txyz[3 * t + 2], tnrm[3 * t], tnrm[3 * t + 1], tnrm[3 * t + 2]);
}

// squared NN distances by ORIGINAL source index (-1 = no partner): the order the reference sums them in for its progress line
// (icp_point_to_plane.cc:226-229); e3d_icp_set_sequential_distance_sum
__global__ __launch_bounds__(kBlock) void k_match_d2_by_original(const int* __restrict__ match_pos, const float* __restrict__ match_d2,
                                                                 const unsigned* __restrict__ order, size_t n, const float4* __restrict__ Gsrc,
                                                                 float* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t js = order ? (size_t)order[j] : j;
  out[__float_as_uint(Gsrc[js].w)] = (match_pos[j] >= 0) ? match_d2[j] : -1.f;
}

// un-permute NN results to original source order / original target indices
__global__ __launch_bounds__(kBlock) void k_unpermute_matches(const int* __restrict__ match_pos,
                                                              const float* __restrict__ match_d2,
                                                              const unsigned* __restrict__ order, size_t n,
                                                              const float4* __restrict__ Gsrc,
                                                              const float4* __restrict__ Gtgt,
                                                              int* __restrict__ out_idx, float* __restrict__ out_d2) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t js = order ? (size_t)order[j] : j;
  const unsigned oi = __float_as_uint(Gsrc[js].w);
  const int m = match_pos[j];
  out_idx[oi] = (m >= 0) ? (int)__float_as_uint(Gtgt[m].w) : -1;
  out_d2[oi] = (m >= 0) ? match_d2[j] : 0.f;
}

// =================================================================================================
// a7 + a8: PointToPlaneICPImpl::compute accumulate / cost passes
//          (icp_point_to_plane_impl.h:119-211 and :240-266), fused: one pass over the
//          correspondence planes yields cost and (mode-dependent) the Gramian blocks.
// =================================================================================================
// f32 residual / Jacobian rows, literally as written in the reference (left-to-right).  (Templated on the scalar type only so that
// tools/micro/lm_variants.hip can instantiate experimental forms; the product uses T = float.  A form that pushed TWO
// correspondences through the packed v_pk_mul_f32 / v_pk_add_f32 instructions was measured and dropped: the pose entries have to
// be splatted into VGPR pairs and the pair interleaved with ~20 moves per trip, and it never beat the scalar loop -- DESIGN 4.2.)
//
// The two 12-entry rows [js ; jt] of a correspondence only hold 9 different numbers each: the translation parts are
// js[0..2] = -jt[0..2] (impl.h:162-164 / 170-172 and 188-190 / 197-199).  With m = jt[0..2], a = js[3..5], b = jt[3..5]:
//   row 1: m = sn,  a = impl.h:173-177, b = impl.h:165-168;     row 2: m = -tn, a = impl.h:200-204, b = impl.h:191-195.
template <typename T>
struct CorrRowsT {
  T r1, r2;
  T m1[3], a1[3], b1[3];
  T m2[3], a2[3], b2[3];
};

template <typename T>
__device__ __forceinline__ T dot3t(const T a0, const T a1, const T a2, const T b0, const T b1, const T b2) {
  const T e0 = a0 * b0, e1 = a1 * b1, e2 = a2 * b2;       // Eigen 3-term inner product: e0 + (e1 + e2)
  return e0 + (e1 + e2);
}
template <typename T>
__device__ __forceinline__ T rdot(const float r0, const float r1, const float r2, const T x, const T y, const T z) {
  const T e0 = r0 * x, e1 = r1 * y, e2 = r2 * z;          // row of R (block-uniform scalars) times a point
  return e0 + (e1 + e2);
}

template <typename T>
struct CorrPts { T spx, spy, spz, snx, sny, snz, tpx, tpy, tpz, tnx, tny, tnz; };

// inner poses applied with Eigen's R*p + t order (impl.h:144-151)
template <typename T>
__device__ __forceinline__ void corr_src(const float* Rs, const float* ts, const T lsx, const T lsy, const T lsz, const T lnx,
                                         const T lny, const T lnz, CorrPts<T>& P) {
  P.spx = rdot<T>(Rs[0], Rs[1], Rs[2], lsx, lsy, lsz) + ts[0];
  P.spy = rdot<T>(Rs[3], Rs[4], Rs[5], lsx, lsy, lsz) + ts[1];
  P.spz = rdot<T>(Rs[6], Rs[7], Rs[8], lsx, lsy, lsz) + ts[2];
  P.snx = rdot<T>(Rs[0], Rs[1], Rs[2], lnx, lny, lnz);
  P.sny = rdot<T>(Rs[3], Rs[4], Rs[5], lnx, lny, lnz);
  P.snz = rdot<T>(Rs[6], Rs[7], Rs[8], lnx, lny, lnz);
}
template <typename T>
__device__ __forceinline__ void corr_tgt(const float* Rt, const float* tt, const T ltx, const T lty, const T ltz, const T lmx,
                                         const T lmy, const T lmz, CorrPts<T>& P) {
  P.tpx = rdot<T>(Rt[0], Rt[1], Rt[2], ltx, lty, ltz) + tt[0];
  P.tpy = rdot<T>(Rt[3], Rt[4], Rt[5], ltx, lty, ltz) + tt[1];
  P.tpz = rdot<T>(Rt[6], Rt[7], Rt[8], ltx, lty, ltz) + tt[2];
  P.tnx = rdot<T>(Rt[0], Rt[1], Rt[2], lmx, lmy, lmz);
  P.tny = rdot<T>(Rt[3], Rt[4], Rt[5], lmx, lmy, lmz);
  P.tnz = rdot<T>(Rt[6], Rt[7], Rt[8], lmx, lmy, lmz);
}

template <bool NEED_SRC, bool NEED_TGT, typename T>
__device__ __forceinline__ void corr_rows_pts(const CorrPts<T>& P, CorrRowsT<T>& o) {
  const T spx = P.spx, spy = P.spy, spz = P.spz, snx = P.snx, sny = P.sny, snz = P.snz;
  const T tpx = P.tpx, tpy = P.tpy, tpz = P.tpz, tnx = P.tnx, tny = P.tny, tnz = P.tnz;
  o.r1 = dot3t<T>(snx, sny, snz, tpx - spx, tpy - spy, tpz - spz);                // impl.h:158
  o.r2 = dot3t<T>(tnx, tny, tnz, spx - tpx, spy - tpy, spz - tpz);                // impl.h:185
  if (NEED_SRC || NEED_TGT) {
    o.m1[0] = snx; o.m1[1] = sny; o.m1[2] = snz;                                  // impl.h:162-164 (j1s[0..2] = -m1)
    o.m2[0] = -tnx; o.m2[1] = -tny; o.m2[2] = -tnz;                               // impl.h:188-190 (j2s[0..2] = -m2 = tn)
  }
  if (NEED_TGT) {
    o.b1[0] = -sny * tpz + snz * tpy;                                             // impl.h:165-168
    o.b1[1] = snx * tpz - snz * tpx;
    o.b1[2] = -snx * tpy + sny * tpx;
    o.b2[0] = tny * tpz - tny * (tpz - spz) - tnz * tpy + tnz * (tpy - spy);      // impl.h:191-195
    o.b2[1] = -tnx * tpz + tnx * (tpz - spz) + tnz * tpx - tnz * (tpx - spx);
    o.b2[2] = tnx * tpy - tnx * (tpy - spy) - tny * tpx + tny * (tpx - spx);
  }
  if (NEED_SRC) {
    o.a1[0] = sny * spz - sny * (spz - tpz) - snz * spy + snz * (spy - tpy);      // impl.h:173-177
    o.a1[1] = -snx * spz + snx * (spz - tpz) + snz * spx - snz * (spx - tpx);
    o.a1[2] = snx * spy - snx * (spy - tpy) - sny * spx + sny * (spx - tpx);
    o.a2[0] = -tny * spz + tnz * spy;                                             // impl.h:200-204
    o.a2[1] = tnx * spz - tnz * spx;
    o.a2[2] = -tnx * spy + tny * spx;
  }
}

template <bool NEED_SRC, bool NEED_TGT, typename T, typename V4>
__device__ __forceinline__ void corr_rows(const LmSet& S, const V4& a, const V4& b, const V4& c, CorrRowsT<T>& o) {
  CorrPts<T> P;
  corr_src<T>(S.Rs, S.ts, a.x, a.y, a.z, a.w, b.x, b.y, P);
  corr_tgt<T>(S.Rt, S.tt, b.z, b.w, c.x, c.y, c.z, c.w, P);
  corr_rows_pts<NEED_SRC, NEED_TGT, T>(P, o);
}

template <int H> __device__ __forceinline__ float half_of(const float v) { return v; }

// accumulate upper triangle of J J^T (21) and r*J (6) in f64 from f32 rows cast to f64 first
// (icp_point_to_plane_impl.h:91-112,179-182: .cast<double>() before the product).  The f64 product of two numbers that
// came from f32 is exact (48 significant bits), so the fused multiply-add rounds exactly as the reference's separate
// multiplication and addition do: one v_fma_f64 instead of v_mul_f64 + v_add_f64 with the translation unit's contraction off.
__device__ __forceinline__ void acc_diag(double* H21, double* b6, const float* j, float r) {
  double J[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) J[i] = (double)j[i];
  const double R = (double)r;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int l = i; l < 6; ++l) { H21[k] = __builtin_fma(J[i], J[l], H21[k]); ++k; }
    b6[i] = __builtin_fma(R, J[i], b6[i]);
  }
}

// Both sides have variables (modes kModeTwo / kModeTwoCross).  The Gramian of the 12-entry row [js ; jt] needs 78 (+12 for
// r*J) products per row; because js[0..2] = -jt[0..2] it is determined by the Gramian of the 10 numbers u = [m a b r]:
// 45 (kModeTwo) or 54 (kModeTwoCross) products.  The f64 product of two f32 values is exact and negation commutes with
// every rounding, so sum((-m_i) a_l) = -sum(m_i a_l) bit for bit: the blocks written out below are the ones a direct
// accumulation of SS, TT, ST, bs, bt in the same order yields (icp_point_to_plane_impl.h:91-112,179-182,206-209).
//   acc layout: [0] cost, [1..6] m m^T (upper), [7..15] m a^T, [16..21] a a^T, [22..30] m b^T, [31..36] b b^T,
//               [37..39] r m, [40..42] r a, [43..45] r b, [46..54] a b^T (kModeTwoCross only)
constexpr int kAccTwo = 46, kAccCross = 55;
template <bool CROSS>
__device__ __forceinline__ void acc_reduced(double* acc, const float m0, const float m1, const float m2, const float a0,
                                            const float a1, const float a2, const float b0, const float b1, const float b2,
                                            const float r) {
  const double M[3] = {(double)m0, (double)m1, (double)m2};
  const double A[3] = {(double)a0, (double)a1, (double)a2};
  const double Bv[3] = {(double)b0, (double)b1, (double)b2};
  const double R = (double)r;
  int k = 1;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int l = i; l < 3; ++l) { acc[k] = __builtin_fma(M[i], M[l], acc[k]); ++k; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int l = 0; l < 3; ++l) acc[7 + 3 * i + l] = __builtin_fma(M[i], A[l], acc[7 + 3 * i + l]);
  k = 16;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int l = i; l < 3; ++l) { acc[k] = __builtin_fma(A[i], A[l], acc[k]); ++k; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int l = 0; l < 3; ++l) acc[22 + 3 * i + l] = __builtin_fma(M[i], Bv[l], acc[22 + 3 * i + l]);
  k = 31;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int l = i; l < 3; ++l) { acc[k] = __builtin_fma(Bv[i], Bv[l], acc[k]); ++k; }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    acc[37 + i] = __builtin_fma(R, M[i], acc[37 + i]); acc[40 + i] = __builtin_fma(R, A[i], acc[40 + i]); acc[43 + i] = __builtin_fma(R, Bv[i], acc[43 + i]);
  }
  if (CROSS) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int l = 0; l < 3; ++l) acc[46 + 3 * i + l] = __builtin_fma(A[i], Bv[l], acc[46 + 3 * i + l]);
  }
}

// output slot t of a block partial (layout below) = sgn[t] * acc[idx[t]] of the reduced accumulators
struct LmSlotMap { short idx[91]; signed char sgn[91]; };
constexpr int lm_tri3(int i, int l) { return i <= l ? (i == 0 ? l : i == 1 ? 2 + l : 5) : lm_tri3(l, i); }   // 00 01 02 11 12 22
constexpr LmSlotMap make_lm_slot_map() {
  LmSlotMap m{};
  m.idx[0] = 0; m.sgn[0] = 1;
  int k = 1;
  for (int i = 0; i < 6; ++i)
    for (int l = i; l < 6; ++l, ++k) {                    // SS = js js^T, js = [-m a]
      if (l < 3) { m.idx[k] = (short)(1 + lm_tri3(i, l)); m.sgn[k] = 1; }
      else if (i < 3) { m.idx[k] = (short)(7 + 3 * i + (l - 3)); m.sgn[k] = -1; }
      else { m.idx[k] = (short)(16 + lm_tri3(i - 3, l - 3)); m.sgn[k] = 1; }
    }
  for (int i = 0; i < 6; ++i) {                           // bs = r js
    if (i < 3) { m.idx[22 + i] = (short)(37 + i); m.sgn[22 + i] = -1; }
    else { m.idx[22 + i] = (short)(40 + i - 3); m.sgn[22 + i] = 1; }
  }
  k = 28;
  for (int i = 0; i < 6; ++i)
    for (int l = i; l < 6; ++l, ++k) {                    // TT = jt jt^T, jt = [m b]
      if (l < 3) { m.idx[k] = (short)(1 + lm_tri3(i, l)); m.sgn[k] = 1; }
      else if (i < 3) { m.idx[k] = (short)(22 + 3 * i + (l - 3)); m.sgn[k] = 1; }
      else { m.idx[k] = (short)(31 + lm_tri3(i - 3, l - 3)); m.sgn[k] = 1; }
    }
  for (int i = 0; i < 6; ++i) {                           // bt = r jt
    m.idx[49 + i] = (short)(i < 3 ? 37 + i : 43 + i - 3); m.sgn[49 + i] = 1;
  }
  for (int i = 0; i < 6; ++i)
    for (int l = 0; l < 6; ++l) {                         // ST = js jt^T
      const int t = 55 + 6 * i + l;
      if (i < 3 && l < 3) { m.idx[t] = (short)(1 + lm_tri3(i, l)); m.sgn[t] = -1; }
      else if (i < 3) { m.idx[t] = (short)(22 + 3 * i + (l - 3)); m.sgn[t] = -1; }
      else if (l < 3) { m.idx[t] = (short)(7 + 3 * l + (i - 3)); m.sgn[t] = 1; }
      else { m.idx[t] = (short)(46 + 3 * (i - 3) + (l - 3)); m.sgn[t] = 1; }
    }
  return m;
}
__device__ const LmSlotMap kLmSlotMap = make_lm_slot_map();

// one correspondence (T = float, H = 0) or one half of a packed pair (T = f2_t, H = 0 / 1) into the accumulators
template <int MODE, int H, typename T>
__device__ __forceinline__ void lm_accumulate(double* acc, const CorrRowsT<T>& R, const int side) {
  constexpr bool kOne = (MODE == kModeOne);
  constexpr bool kCross = (MODE == kModeTwoCross);
  const float r1 = half_of<H>(R.r1), r2 = half_of<H>(R.r2);
  if (MODE == kModeCost) {
    acc[0] += (double)(r1 * r1);
    acc[0] += (double)(r2 * r2);
  } else if (kOne) {
    // only one side of the pair has variables (the other is impl cloud 0); block-uniform branch
    float j1[6], j2[6];
    if (side == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { j1[i] = -half_of<H>(R.m1[i]); j1[3 + i] = half_of<H>(R.a1[i]); j2[i] = -half_of<H>(R.m2[i]); j2[3 + i] = half_of<H>(R.a2[i]); }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) { j1[i] = half_of<H>(R.m1[i]); j1[3 + i] = half_of<H>(R.b1[i]); j2[i] = half_of<H>(R.m2[i]); j2[3 + i] = half_of<H>(R.b2[i]); }
    }
    acc[0] += (double)(r1 * r1);
    acc_diag(acc + 1, acc + 22, j1, r1);
    acc[0] += (double)(r2 * r2);
    acc_diag(acc + 1, acc + 22, j2, r2);
  } else {
    acc[0] += (double)(r1 * r1);
    acc_reduced<kCross>(acc, half_of<H>(R.m1[0]), half_of<H>(R.m1[1]), half_of<H>(R.m1[2]), half_of<H>(R.a1[0]), half_of<H>(R.a1[1]),
                        half_of<H>(R.a1[2]), half_of<H>(R.b1[0]), half_of<H>(R.b1[1]), half_of<H>(R.b1[2]), r1);
    acc[0] += (double)(r2 * r2);
    acc_reduced<kCross>(acc, half_of<H>(R.m2[0]), half_of<H>(R.m2[1]), half_of<H>(R.m2[2]), half_of<H>(R.a2[0]), half_of<H>(R.a2[1]),
                        half_of<H>(R.a2[2]), half_of<H>(R.b2[0]), half_of<H>(R.b2[1]), half_of<H>(R.b2[2]), r2);
  }
}

template <int MODE, typename T, typename V4>
__device__ __forceinline__ void lm_rows(const LmSet& S, const V4& a, const V4& b, const V4& c, CorrRowsT<T>& R) {
  if (MODE == kModeCost) corr_rows<false, false, T>(S, a, b, c, R);
  else if (MODE == kModeOne) {
    if (S.side == 0) corr_rows<true, false, T>(S, a, b, c, R); else corr_rows<false, true, T>(S, a, b, c, R);
  } else corr_rows<true, true, T>(S, a, b, c, R);
}

// The planes and the group list of a set are reached through pointers stored in the LmSet, which the compiler has to treat as
// generic (flat) addresses: flat loads cost an aperture check and count against both wait counters, and a flat address cannot be
// read with a scalar load.  They are HBM allocations: say so.
typedef float lm_v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) lm_v4f* lm_rows_ptr;       // global
typedef const __attribute__((address_space(4))) unsigned* lm_glist_ptr;    // constant for the lifetime of the launch: s_load
__device__ __forceinline__ float4 ld_row(const lm_rows_ptr p, const long long r) {
  lm_v4f v;
  if constexpr (E3D_NT >= 1) v = __builtin_nontemporal_load(p + r); else v = p[r];
  return make_float4(v.x, v.y, v.z, v.w);
}

// A resident row's local halves into the global frame of the outer iteration: pcl::transformPointCloudWithNormals' operation
// order (icp_point_to_plane.cc:192-195), i.e. the very roundings k_transform_bbox / k_compact_corr apply -- the pass then sees
// the numbers a compacted row would hold.  Block-uniform branches (S sits in SGPRs).
__device__ __forceinline__ void row_to_global(const LmSet& S, float4& a, float4& b, float4& c) {
  if (S.outer & 1) {
    const float3 p = pcl_se3(S.Tos, a.x, a.y, a.z), n = pcl_so3(S.Tos, a.w, b.x, b.y);
    a = make_float4(p.x, p.y, p.z, n.x); b.x = n.y; b.y = n.z;
  }
  if (S.outer & 2) {
    const float3 p = pcl_se3(S.Tot, b.z, b.w, c.x), n = pcl_so3(S.Tot, c.y, c.z, c.w);
    b.z = p.x; b.w = p.y; c = make_float4(p.z, n.x, n.y, n.z);
  }
}
template <typename V4>
__device__ __forceinline__ void row_to_global(const LmSet&, V4&, V4&, V4&) {}     // (packed experimental row types of tools/micro: compacted rows only)

// Output slot layout per block / per set (kLmSlot doubles):
//   [0] cost, [1..21] SS upper, [22..27] bs, [28..48] TT upper, [49..54] bt, [55..90] ST (6x6)
// Every thread walks its correspondences c, c + stride, c + 2 stride, ... in this order and adds row 1 then row 2 of each: the
// per-thread sums, and with the fixed reduction tree the block partials, are the same numbers whatever the loop shape.
// Loop shapes (tools/micro/lm_variants.hip measures them; LmCfg holds the choice per mode):
//   UNR  correspondences per trip (1 or 2);
//   PF   the next trip's float4 loads are issued before the current trip's arithmetic (at 2 - 3 waves per SIMD -- the f64
//        accumulators -- the loads in flight per lane, not the occupancy, have to cover the HBM latency).
template <int MODE, int UNR, bool PF>
__device__ __forceinline__ void lm_pass_body(const LmSet* __restrict__ sets, const int* __restrict__ block_set, const int block_base,
                                             double* __restrict__ partial) {
  constexpr bool kOne = (MODE == kModeOne);
  constexpr bool kCross = (MODE == kModeTwoCross);
  constexpr int NACC = (MODE == kModeCost) ? 1 : kOne ? 28 : kCross ? kAccCross : kAccTwo;
  const int gb = block_base + blockIdx.x;
  const int si = block_set[gb];
  const LmSet S = sets[si];
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

  const long long stride = (long long)S.nblocks * kBlock;
  const lm_rows_ptr pa = (lm_rows_ptr)S.A, pb = (lm_rows_ptr)S.B, pc = (lm_rows_ptr)S.C;
  const lm_glist_ptr gl = (lm_glist_ptr)S.glist;
  const long long lane64 = threadIdx.x & 63;
  long long c = (long long)(gb - S.block_begin) * kBlock + threadIdx.x;
  if (UNR == 2) {
    // (two rows per trip: only the micro-benchmark instantiates this shape, on compacted rows)
    float4 a0, b0, c0, a1, b1, c1;
    if (PF && c + stride < S.n) { a0 = ld_row(pa, c); b0 = ld_row(pb, c); c0 = ld_row(pc, c); a1 = ld_row(pa, c + stride); b1 = ld_row(pb, c + stride); c1 = ld_row(pc, c + stride); }
    while (c + stride < S.n) {
      if (!PF) { a0 = ld_row(pa, c); b0 = ld_row(pb, c); c0 = ld_row(pc, c); a1 = ld_row(pa, c + stride); b1 = ld_row(pb, c + stride); c1 = ld_row(pc, c + stride); }
      const float4 ua = a0, ub = b0, uc = c0, va = a1, vb = b1, vc = c1;
      c += 2 * stride;
      if (PF && c + stride < S.n) { a0 = ld_row(pa, c); b0 = ld_row(pb, c); c0 = ld_row(pc, c); a1 = ld_row(pa, c + stride); b1 = ld_row(pb, c + stride); c1 = ld_row(pc, c + stride); }
      CorrRowsT<float> R;
      lm_rows<MODE, float>(S, ua, ub, uc, R);
      lm_accumulate<MODE, 0, float>(acc, R, S.side);
      lm_rows<MODE, float>(S, va, vb, vc, R);
      lm_accumulate<MODE, 0, float>(acc, R, S.side);
    }
    if (c < S.n) {
      const float4 a = ld_row(pa, c), b = ld_row(pb, c), cc = ld_row(pc, c);
      CorrRowsT<float> R;
      lm_rows<MODE, float>(S, a, b, cc, R);
      lm_accumulate<MODE, 0, float>(acc, R, S.side);
    }
  } else {
    auto trip = [&](float4 ua, float4 ub, float4 uc) {
      row_to_global(S, ua, ub, uc);
      CorrRowsT<float> R;
      lm_rows<MODE, float>(S, ua, ub, uc, R);
      lm_accumulate<MODE, 0, float>(acc, R, S.side);
    };
    float4 a0, b0, c0;
    if (gl) {
      // resident rows: a wave walks whole 64-row groups, so everything that steers the walk is wave-uniform (scalar loads and
      // branches); the group id of the trip after next is requested one trip ahead of the rows it addresses, behind the current
      // trip's arithmetic like the rows themselves
      const int ng = (int)(S.n >> 6), gstride = S.nblocks * (kBlock / kWave);
      int gw = __builtin_amdgcn_readfirstlane((gb - S.block_begin) * (kBlock / kWave) + (int)(threadIdx.x >> 6));
      unsigned g0 = (gw < ng) ? gl[gw] : 0u, g1 = (gw + gstride < ng) ? gl[gw + gstride] : 0u;
      if (gw < ng) { const long long r = ((long long)g0 << 6) | lane64; a0 = ld_row(pa, r); b0 = ld_row(pb, r); c0 = ld_row(pc, r); }
      while (gw < ng) {
        const float4 ua = a0, ub = b0, uc = c0;
        gw += gstride;
        g0 = g1;
        if (gw + gstride < ng) g1 = gl[gw + gstride];
        if (gw < ng) { const long long r = ((long long)g0 << 6) | lane64; a0 = ld_row(pa, r); b0 = ld_row(pb, r); c0 = ld_row(pc, r); }
        trip(ua, ub, uc);
      }
    } else {
      if (PF && c < S.n) { a0 = ld_row(pa, c); b0 = ld_row(pb, c); c0 = ld_row(pc, c); }
      while (c < S.n) {
        if (!PF) { a0 = ld_row(pa, c); b0 = ld_row(pb, c); c0 = ld_row(pc, c); }
        const float4 ua = a0, ub = b0, uc = c0;
        c += stride;
        if (PF && c < S.n) { a0 = ld_row(pa, c); b0 = ld_row(pb, c); c0 = ld_row(pc, c); }
        trip(ua, ub, uc);
      }
    }
  }
  // wave reduce -> LDS -> fixed-order block sum
  __shared__ double s[kBlock / kWave][NACC];
  __shared__ double red[NACC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s[w][i] = v;
  }
  __syncthreads();
  if (MODE == kModeTwo || MODE == kModeTwoCross) {
    if (threadIdx.x < NACC) {
      double v = s[0][threadIdx.x];
      for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
      red[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x < kLmSlot) {
      const int t = threadIdx.x;
      double v = 0.0;
      if (t < 55 || kCross) { v = red[kLmSlotMap.idx[t]]; if (kLmSlotMap.sgn[t] < 0) v = -v; }
      partial[(size_t)gb * kLmSlot + t] = v;             // kModeTwo: the ST block is never read [QUIRK], zeros
    }
  } else if (threadIdx.x < kLmSlot) {
    double v = 0.0;
    if (threadIdx.x < NACC) {
      v = s[0][threadIdx.x];
      for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    }
    partial[(size_t)gb * kLmSlot + threadIdx.x] = v;   // unused slots of cheaper modes are zero
  }
}

// the loop shape each mode runs with.  Measured on one MI355X (tools/micro/lm_variants.hip, 1e8 correspondences in 8 sets,
// profiles/round3_lm_variants.txt): one correspondence per trip with the next one's loads in flight wins in every mode; three
// waves per SIMD (168 VGPRs) for the two-sided modes, which the 55 / 46 f64 accumulators allow.  The translation unit is built
// with -fno-slp-vectorize: left alone, the compiler packs adjacent scalar f32 operations of these expression trees into
// v_pk_mul_f32 / v_pk_add_f32 and pays for it in moves (k_lm_cost_multi 2.00 -> 1.67 ms, mode 3 0.97 -> 0.91 ms without).
template <int MODE> struct LmCfg { static constexpr int unr = 1; static constexpr bool pf = true; static constexpr int minw = 3; };
template <> struct LmCfg<kModeCost> { static constexpr int unr = 1; static constexpr bool pf = true; static constexpr int minw = 1; };
template <> struct LmCfg<kModeOne> { static constexpr int unr = 1; static constexpr bool pf = true; static constexpr int minw = 2; };

template <int MODE>
__global__ __launch_bounds__(kBlock, LmCfg<MODE>::minw) void k_lm_pass(const LmSet* __restrict__ sets, const int* __restrict__ block_set,
                                                                       int block_base, double* __restrict__ partial) {
  lm_pass_body<MODE, LmCfg<MODE>::unr, LmCfg<MODE>::pf>(sets, block_set, block_base, partial);
}

// a8, batched: the LM tries 1..9 of one inner iteration (lambda doubled each time, icp_point_to_plane_impl.h:216-283)
// only differ in the candidate poses, so their costs are evaluated in ONE pass over the correspondence planes: each
// correspondence is loaded once and pushed through up to kLmMaxPoses pose sets.  Per pose the arithmetic, the per-thread
// order of the correspondences and the reduction tree are exactly those of k_lm_pass, so every cost is bit-identical
// to the one a separate pass would return; the host then takes the first try that lowers the cost, as the
// reference's sequential loop does.  (The usual call is the last one of an outer iteration, where all nine tries fail: an
// early-out after the first few tries would not save it.)  Nine poses are 9 x 80 f32 operations per correspondence, VALU
// bound; the side of a kModeOne pair that has no variables (impl cloud 0, the same inner pose in every candidate) is transformed
// once instead of nine times.
// Round 4: the body takes U rows per trip with the POSES in the outer loop (a pose set is 24 scalar words, x 9 far more than the
// scalar registers hold, so the poses are fetched again for every trip).  Measured (tools/micro/lm_variants.hip, 1e8 rows, nine
// poses): U = 1 1.41 ms, 2 1.49 - 1.55, 3 1.52, 4 1.60 - 1.66 -- the pass is bound by its f32 instructions (0.12 ms per pose and
// 1e8 rows = the chip's f32 issue rate for the ~55 operations of a pose), not by the scalar fetches, and more rows per trip only
// cost occupancy (102 / 139 / 176 VGPRs).  U = 1 it stays.  Per pose a thread adds its rows in the order c, c + stride, ... and row
// by row r1^2 then r2^2: the same sums as k_lm_pass<kModeCost>.
constexpr int kLmCmRows = 1;

template <bool PF, int U = kLmCmRows>
__device__ __forceinline__ void lm_cost_multi_body(const LmSet* __restrict__ sets, const LmPose* __restrict__ poses, int n_sets,
                                                   int n_poses, const int* __restrict__ block_set, double* __restrict__ partial) {
  const int gb = blockIdx.x;
  const int si = block_set[gb];
  const LmSet S = sets[si];
  double acc[kLmMaxPoses];
#pragma unroll
  for (int k = 0; k < kLmMaxPoses; ++k) acc[k] = 0.0;
  const long long stride = (long long)S.nblocks * kBlock;
  const lm_rows_ptr pa = (lm_rows_ptr)S.A, pb = (lm_rows_ptr)S.B, pc = (lm_rows_ptr)S.C;
  const lm_glist_ptr gl = (lm_glist_ptr)S.glist;
  const long long lane64 = threadIdx.x & 63;
  const bool fixed_tgt = (S.mode == kModeOne && S.side == 0), fixed_src = (S.mode == kModeOne && S.side == 1);
  // U rows (valid[u]: the row exists; the others were loaded from a clamped address and add +0.0, which leaves a sum of squares
  // unchanged bit for bit) through every pose
  auto trip = [&](float4* a, float4* b, float4* cc, const bool* valid) {
    CorrPts<float> F[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row_to_global(S, a[u], b[u], cc[u]);
      if (fixed_tgt) corr_tgt<float>(S.Rt, S.tt, b[u].z, b[u].w, cc[u].x, cc[u].y, cc[u].z, cc[u].w, F[u]);
      if (fixed_src) corr_src<float>(S.Rs, S.ts, a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, F[u]);
    }
#pragma unroll
    for (int k = 0; k < kLmMaxPoses; ++k) {
      if (k < n_poses) {
        const LmPose& P = poses[(size_t)k * n_sets + si];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          CorrPts<float> Q = F[u];
          if (!fixed_src) corr_src<float>(P.Rs, P.ts, a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, Q);
          if (!fixed_tgt) corr_tgt<float>(P.Rt, P.tt, b[u].z, b[u].w, cc[u].x, cc[u].y, cc[u].z, cc[u].w, Q);
          CorrRowsT<float> R;
          corr_rows_pts<false, false, float>(Q, R);
          acc[k] += valid[u] ? (double)(R.r1 * R.r1) : 0.0;
          acc[k] += valid[u] ? (double)(R.r2 * R.r2) : 0.0;
        }
      }
    }
  };
  float4 a[U], b[U], cc[U];
  bool valid[U];
  if (gl) {      // resident rows: whole 64-row groups per wave, wave-uniform control (lm_pass_body)
    const int ng = (int)(S.n >> 6), gstride = S.nblocks * (kBlock / kWave);
    int gw = __builtin_amdgcn_readfirstlane((gb - S.block_begin) * (kBlock / kWave) + (int)(threadIdx.x >> 6));
    while (gw < ng) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int gi = gw + u * gstride;
        valid[u] = gi < ng;
        const long long r = ((long long)gl[valid[u] ? gi : gw] << 6) | lane64;
        a[u] = ld_row(pa, r); b[u] = ld_row(pb, r); cc[u] = ld_row(pc, r);
      }
      trip(a, b, cc, valid);
      gw += U * gstride;
    }
  } else {
    long long c = (long long)(gb - S.block_begin) * kBlock + threadIdx.x;
    while (c < S.n) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long ci = c + u * stride;
        valid[u] = ci < S.n;
        const long long r = valid[u] ? ci : c;
        a[u] = ld_row(pa, r); b[u] = ld_row(pb, r); cc[u] = ld_row(pc, r);
      }
      trip(a, b, cc, valid);
      c += U * stride;
    }
  }
  __shared__ double s[kBlock / kWave][kLmMaxPoses];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kLmMaxPoses; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s[w][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLmSlot) {
    double v = 0.0;
    if (threadIdx.x < kLmMaxPoses) {
      v = s[0][threadIdx.x];
      for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    }
    partial[(size_t)gb * kLmSlot + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(kBlock) void k_lm_cost_multi(const LmSet* __restrict__ sets, const LmPose* __restrict__ poses,
                                                          int n_sets, int n_poses, const int* __restrict__ block_set,
                                                          double* __restrict__ partial) {
  lm_cost_multi_body<true>(sets, poses, n_sets, n_poses, block_set, partial);
}

// one block per set: sum the set's block partials in a fixed order.  kRedParts threads share each of the
// kLmSlot accumulators (interleaved block ranges, 4 independent loads in flight), then a fixed-order
// LDS sum -- deterministic, and no 2048-long chain of dependent HBM loads.
constexpr int kRedParts = 8;
__global__ __launch_bounds__(kLmSlot * kRedParts) void k_lm_reduce(const double* __restrict__ partial,
                                                                   const LmSet* __restrict__ sets, int nacc,
                                                                   double* __restrict__ out) {
  const LmSet S = sets[blockIdx.x];
  const int slot = threadIdx.x % kLmSlot, part = threadIdx.x / kLmSlot;
  __shared__ double sh[kRedParts][kLmSlot];
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  const double* base = partial + (size_t)S.block_begin * kLmSlot + slot;
  int b = part;
  for (; b + 3 * kRedParts < S.nblocks; b += 4 * kRedParts) {
    const double a0 = base[(size_t)b * kLmSlot], a1 = base[(size_t)(b + kRedParts) * kLmSlot];
    const double a2 = base[(size_t)(b + 2 * kRedParts) * kLmSlot], a3 = base[(size_t)(b + 3 * kRedParts) * kLmSlot];
    v0 += a0; v1 += a1; v2 += a2; v3 += a3;
  }
  for (; b < S.nblocks; b += kRedParts) v0 += base[(size_t)b * kLmSlot];
  sh[part][slot] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (part == 0 && slot < nacc) {
    double v = sh[0][slot];
#pragma unroll
    for (int k = 1; k < kRedParts; ++k) v += sh[k][slot];
    out[(size_t)blockIdx.x * kLmSlot + slot] = v;
  }
}

// =================================================================================================
// host-side launch helpers
// =================================================================================================
static inline int grid_for(size_t n, int cap = 4096) {
  size_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = (size_t)cap;
  return (int)b;
}

int launch_transform_aos(const float* xyz, const float* nrm, size_t n, const Affine& T, float* oxyz, float* onrm,
                         float* bbox_partial, float* bbox_out, hipStream_t s) {
  const int nb = grid_for(n, kMaxBboxBlocks);
  hipLaunchKernelGGL(k_transform_aos, dim3(nb), dim3(kBlock), 0, s, xyz, nrm, n, T, oxyz, onrm, bbox_partial);
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, bbox_partial, nb, bbox_out);
  return nb;
}

int launch_transform_bbox(const float4* L4, size_t n, const Affine& T, float4* G4, float* bbox_partial,
                          float* bbox_out, hipStream_t s) {
  const int nb = grid_for(n, kMaxBboxBlocks);
  hipLaunchKernelGGL(k_transform_bbox, dim3(nb), dim3(kBlock), 0, s, L4, n, T, G4, bbox_partial);
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, bbox_partial, nb, bbox_out);
  return nb;
}

void launch_bbox_aos(const float* xyz, size_t n, float* bbox_partial, float* bbox_out, hipStream_t s) {
  const int nb = grid_for(n, kMaxBboxBlocks);
  hipLaunchKernelGGL(k_bbox_aos, dim3(nb), dim3(kBlock), 0, s, xyz, n, bbox_partial);
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, bbox_partial, nb, bbox_out);
}

void launch_cell_keys(const float* xyz, size_t n, const GridDesc& g, unsigned long long* keys, unsigned* vals,
                      hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_cell_keys, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, xyz, n, g, keys, vals);
}

void launch_permute(const float* xyz, const float* nrm, const unsigned* order, size_t n, float4* L4, float4* LN,
                    hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_permute, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, xyz, nrm, order, n, L4, LN);
}

void launch_count_cells(const unsigned long long* keys, size_t n, unsigned* counter, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_count_cells, dim3(grid_for(n, 2048)), dim3(kBlock), 0, s, keys, n, counter);
}

void launch_build_table(const unsigned long long* keys, size_t n, HashEntry* table, unsigned mask, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_build_table, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, keys, n, table, mask);
}

void launch_nn_query(const float4* Gsrc, size_t n_src, const float4* Gtgt, const HashEntry* table, const GridDesc& g,
                     const InvMap& im, float r2, int* match_pos, float* match_d2, hipStream_t s) {
  if (!n_src) return;
  hipLaunchKernelGGL(k_nn_query, dim3((unsigned)div_up(n_src, kBlock)), dim3(kBlock), 0, s, Gsrc, n_src, Gtgt, table,
                     g, im, r2, match_pos, match_d2);
}

void launch_match_scan(const int* match_pos, const float* match_d2, size_t n, unsigned* block_counts,
                       unsigned* block_offsets, double* block_d2, unsigned long long* chunk_sum, double* chunk_d2,
                       unsigned long long* total, double* total_d2, hipStream_t s) {
  const int nb = (int)div_up(n ? n : 1, kBlock);
  // match_pos == nullptr: block_counts / block_d2 hold the per-block sums already (their producer wrote them: k_obs_eval)
  if (match_pos) hipLaunchKernelGGL(k_match_block_counts, dim3(nb), dim3(kBlock), 0, s, match_pos, n, block_counts, block_d2,
                                    match_d2);
  const int nch = (nb + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(k_scan_chunk_sums, dim3(nch), dim3(kScanChunk), 0, s, block_counts, nb, block_d2, chunk_sum, chunk_d2, (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr);
  hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, chunk_sum, nch, chunk_d2, total, total_d2, (unsigned*)nullptr, (const unsigned*)nullptr);
  hipLaunchKernelGGL(k_scan_within_chunks, dim3(nch), dim3(kScanChunk), 0, s, block_counts, nb, chunk_sum, block_offsets);
}

void launch_query_keys(const float4* Gsrc, size_t n, const GridDesc& g, const InvMap& im, const QueryRange& qr,
                       unsigned long long* keys, unsigned* vals, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_query_keys, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, n, g, im, qr, keys, vals);
}

void launch_query_keys32(const float4* Gsrc, size_t n, const GridDesc& g, const InvMap& im, const QueryRange& qr,
                         unsigned* keys, unsigned* vals, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_query_keys32, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, n, g, im, qr, keys, vals);
}

void launch_nn_cells(const float4* Gsrc, const unsigned* order, size_t n, const float4* Gtgt, const HashEntry* table,
                     const unsigned* dense_start, const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2,
                     int* match_pos, float* match_d2, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_nn_cells, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, order, n, Gtgt, table,
                     dense_start, g, im, qr, r2, match_pos, match_d2);
}

static int row_span_setting() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("E3D_ROW_SPAN");      // tuning knob: x-extent (cells) of a row segment
    v = e ? atoi(e) : kRowSpan;
    if (v < 0) v = 0;
    if (v > 62) v = 62;
  }
  return v;
}

void launch_nn_rows(const float4* Gsrc, const unsigned* order, size_t n, const float4* Gtgt, const unsigned* dense_start,
                    const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2, const CertParams& cert, int* match_pos,
                    float* match_d2, float* lbe, int* match2, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_nn_rows, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, order, n, Gtgt, dense_start,
                     g, im, qr, r2, row_span_setting(), cert, match_pos, match_d2, lbe, match2);
}

void launch_query_keys_list(const float4* Gsrc, const unsigned* list, size_t n, const GridDesc& g, const InvMap& im,
                            const QueryRange& qr, unsigned long long* keys, unsigned* vals, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_query_keys_list, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, list, n, g, im, qr, keys, vals);
}
void launch_query_keys32_list(const float4* Gsrc, const unsigned* list, size_t n, const GridDesc& g, const InvMap& im,
                              const QueryRange& qr, unsigned* keys, unsigned* vals, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_query_keys32_list, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, list, n, g, im, qr, keys, vals);
}

void launch_block_occupancy(const unsigned* dense_start, const QueryRange& qr, unsigned stride_w, unsigned* tmp, unsigned* occ, hipStream_t s) {
  const size_t rows = (size_t)qr.D[1] * qr.D[2];
  if (!rows || !stride_w) return;
  hipLaunchKernelGGL(k_occ_cells, dim3((unsigned)div_up(rows * stride_w * 32u, kBlock)), dim3(kBlock), 0, s, dense_start, qr, stride_w, tmp);
  hipLaunchKernelGGL(k_occ_dilate, dim3((unsigned)div_up(rows * stride_w, kBlock)), dim3(kBlock), 0, s, (const unsigned*)tmp, qr.D[1], qr.D[2], stride_w, occ);
}

void launch_query_keys_prune(bool keys32, const float4* Gsrc, const unsigned* list, size_t n, const unsigned* occ, unsigned stride_w, const GridDesc& g,
                             const InvMap& im, const QueryRange& qr, float r2, const CertParams& cert, void* keys, unsigned* vals, unsigned* count,
                             int* match, int* match2, float* match_d2, float* lbe, bool from_state, hipStream_t s) {
  if (!n) return;
  const dim3 grid((unsigned)div_up(n, (size_t)kPruneBlock));
  const int fs = (from_state && !list) ? 1 : 0;
  if (keys32)
    hipLaunchKernelGGL(k_query_keys_prune<unsigned>, grid, dim3(kBlock), 0, s, Gsrc, list, n, occ, stride_w, g, im, qr, r2, cert,
                       (unsigned*)keys, vals, count, match, match2, match_d2, lbe, fs);
  else
    hipLaunchKernelGGL(k_query_keys_prune<unsigned long long>, grid, dim3(kBlock), 0, s, Gsrc, list, n, occ, stride_w, g, im, qr, r2, cert,
                       (unsigned long long*)keys, vals, count, match, match2, match_d2, lbe, fs);
}

void launch_nn_certify(const float4* Gsrc, size_t n, const float4* Gtgt, const MotionBound& cum_up, float r2, float near2, bool none_near, int* match, int* match2,
                       const float* lbe, float* match_d2, unsigned* todo_near, unsigned* todo_far, unsigned* counts, hipStream_t s) {
  if (!n) return;
  // four query chains in flight per lane: 0.58 / 0.56 / 0.53 ms per 50 M queries with 1 / 2 / 4
  hipLaunchKernelGGL(k_nn_certify<4>, dim3((unsigned)div_up(n, (size_t)kCertPerWave * (kBlock / kWave))), dim3(kBlock), 0, s, Gsrc, n, Gtgt,
                     cum_up, r2, near2, none_near ? 1 : 0, match, match2, lbe, match_d2, todo_near, todo_far, counts);
}

void launch_nn_bounded(const float4* Gsrc, const unsigned* list, size_t n_list, const float4* Gtgt, const unsigned* dense_start,
                       const unsigned long long* half_prefix, bool half_always, const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2,
                       const BoundParams& bp, int* match, int* match2, float* match_d2, float* lbe, hipStream_t s) {
  if (!n_list) return;
  static const size_t quad_limit = [] { const char* e = getenv("E3D_NN_QUAD_LIMIT"); return e ? (size_t)atoll(e) : (size_t)12000000; }();
  static const size_t half_min = [] { const char* e = getenv("E3D_NN_HALF_MIN"); return e ? (size_t)atoll(e) : (size_t)200000; }();
  if (half_prefix && (half_always || n_list >= half_min)) {
    // long lists are bound by the candidates they evaluate: the half-cell directory cuts those to a third
    hipLaunchKernelGGL((k_nn_bounded_half<8>), dim3((unsigned)div_up(n_list, kBlock)), dim3(kBlock), 0, s, Gsrc, list, (unsigned)n_list,
                       Gtgt, dense_start, half_prefix, g, im, qr, r2, bp, match, match2, match_d2, lbe);
    return;
  }
  if (n_list <= quad_limit)
    hipLaunchKernelGGL(k_nn_bounded<4>, dim3((unsigned)div_up(4 * n_list, kBlock)), dim3(kBlock), 0, s, Gsrc, list, (unsigned)n_list, Gtgt,
                       dense_start, g, im, qr, r2, bp, match, match2, match_d2, lbe);
  else
    hipLaunchKernelGGL(k_nn_bounded<1>, dim3((unsigned)div_up(n_list, kBlock)), dim3(kBlock), 0, s, Gsrc, list, (unsigned)n_list, Gtgt,
                       dense_start, g, im, qr, r2, bp, match, match2, match_d2, lbe);
}

void launch_half_keys(const float* xyz, size_t n, const GridDesc& g, unsigned* keys, unsigned* vals, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_half_keys, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, xyz, n, g, keys, vals);
}
void launch_cell_keys_ordered(const float* xyz, const unsigned* order, size_t n, const GridDesc& g, unsigned long long* keys, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_cell_keys_ordered, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, xyz, order, n, g, keys);
}
void launch_half_prefix(const unsigned long long* keys, const float4* L4, size_t n, const GridDesc& g, const QueryRange& qr,
                        const unsigned* dense_start, unsigned long long* half_prefix, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_half_ends, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, keys, L4, n, g, qr, dense_start,
                     reinterpret_cast<unsigned char*>(half_prefix));
  hipLaunchKernelGGL(k_half_fix, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, keys, n, qr, dense_start, half_prefix);
}

// Filter constants of k_nn_mfma for a target grid (cell = local cell size, sigma_max = largest singular value of the
// target's local -> global linear part).  Returns false if no valid scale exists (the caller uses k_nn_rows).
// Geometry (segment origin = midpoint of the first and last query; rs = row span in cells): a candidate lies within
// sqrt((rs/2 + 2)^2 + 8) cells of the origin, a query within sqrt((rs/2 + 1)^2 + 2) cells (2 % margin for the f32 cell mapping).
bool mfma_filter_params(double cell, double sigma_max, int row_span, float r2, MfParams* P) {
  const double rs = (double)row_span;
  const double ext_c = 1.02 * sigma_max * cell * std::sqrt((0.5 * rs + 2.0) * (0.5 * rs + 2.0) + 8.0);
  const double ext_q = 1.02 * sigma_max * cell * std::sqrt((0.5 * rs + 1.0) * (0.5 * rs + 1.0) + 2.0);
  if (!(ext_c > 0) || !std::isfinite(ext_c)) return false;
  const int e = (int)std::floor(std::log2(31.0 / ext_c));
  if (e < -100 || e > 100) return false;
  const double S = std::ldexp(1.0, e);
  const double Mc = ext_c * S, Mq = ext_q * S;                     // <= 31: f16 ulp <= 2^-6 for the hi parts
  const double r2s = (double)r2 * S * S;
  if (!(r2s < 3.0e4) || !(r2s > 1e-30)) return false;
  auto ulp16 = [](double v) { return std::ldexp(1.0, (int)std::floor(std::log2(std::max(v, 1e-30))) - 10); };
  const double u = std::ldexp(1.0, -24);
  const double sum_abs = 2.0 * Mq * Mc * (1.0 + std::ldexp(1.0, -8)) + Mc * Mc + Mq * Mq;
  // accumulation of 16 products in f32 + rounding of the two norms (5 ops each) + residual of their hi/lo split
  // (rtz: below one ulp of the lo part) + f32 rounding of the exact scaled d2; factor 3 for the unspecified internal
  // rounding of the matrix core
  const double eta = 3.0 * (16.0 * u * sum_abs + 6.0 * u * (Mc * Mc + Mq * Mq) + std::ldexp(1.0, -10) * (ulp16(Mc * Mc) + ulp16(Mq * Mq)) +
                            4.0 * u * (r2s + Mc * Mc)) + 1e-6;
  // coordinate representation: f32 rounding of (p - o) (<= 2^-24 * 32) + residual of the rtz hi/lo split, at worst a
  // flushed f16 denormal (2^-14); per point sqrt(3) x that, two points
  const double delta = 2.0 * std::sqrt(3.0) * (std::ldexp(1.0, -14) + 32.0 * u + std::ldexp(1.0, -16));
  P->S = (float)S;
  P->r2s = (float)(r2s * (1.0 + 1e-6));
  P->eta2 = (float)(2.0 * eta);
  P->delta4 = (float)(4.0 * delta);
  P->delta4sq = (float)(4.0 * delta * delta);
  return true;
}

void launch_nn_mfma(const float4* Gsrc, const unsigned* order, size_t n, const float4* Gtgt, const unsigned* dense_start,
                    const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2, const MfParams& P, int* match_pos,
                    float* match_d2, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_nn_mfma, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, Gsrc, order, n, Gtgt, dense_start,
                     g, im, qr, r2, row_span_setting(), P, match_pos, match_d2);
}
int nn_row_span() { return row_span_setting(); }

void launch_dense_counts(const unsigned long long* keys, size_t n, const QueryRange& qr, unsigned* counts, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_dense_counts, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, keys, n, qr, counts);
}

void launch_compact_corr(const int* match_pos, const unsigned* order, size_t n, const unsigned* block_offsets, const float4* Gsrc,
                         const float4* LNsrc, const Affine& Tsrc, const float4* Gtgt, const float4* LNtgt,
                         const Affine& Ttgt, float4* A, float4* B, float4* C, size_t out_base, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_compact_corr, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, match_pos, order, n,
                     block_offsets, Gsrc, LNsrc, Tsrc, Gtgt, LNtgt, Ttgt, A, B, C, out_base);
}

void launch_corr_update(const int* match, int* plane_match, const float* match_d2, size_t n, const float4* Psrc, const float4* LNsrc,
                        bool src_global, const Affine& Tsrc, const float4* Ptgt, const float4* LNtgt, bool tgt_global, const Affine& Ttgt,
                        float4* A, float4* B, float4* C, unsigned* block_counts, double* block_d2, unsigned* block_groups, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_corr_update, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, match, plane_match, match_d2, n, Psrc,
                     LNsrc, src_global ? 1 : 0, Tsrc, Ptgt, LNtgt, tgt_global ? 1 : 0, Ttgt, A, B, C, block_counts, block_d2, block_groups);
}

void launch_corr_totals(size_t n, const unsigned* block_counts, const double* block_d2, const unsigned* block_groups,
                        unsigned long long* chunk_sum, double* chunk_d2, unsigned* chunk_groups, unsigned long long* totals,
                        double* total_d2, unsigned* glist, hipStream_t s) {
  const int nb = (int)div_up(n ? n : 1, kBlock);
  const int nch = (nb + kScanChunk - 1) / kScanChunk;
  unsigned* chunk_rewritten = chunk_groups + nch;        // (chunk_groups holds 2 * nch entries)
  hipLaunchKernelGGL(k_scan_chunk_sums, dim3(nch), dim3(kScanChunk), 0, s, block_counts, nb, block_d2, chunk_sum, chunk_d2, block_groups, chunk_groups, chunk_rewritten);
  hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, chunk_sum, nch, chunk_d2, totals, total_d2, chunk_groups, (const unsigned*)chunk_rewritten);
  hipLaunchKernelGGL(k_group_list, dim3(nch), dim3(kScanChunk), 0, s, block_groups, nb, chunk_groups, glist);
}

void launch_nn_certify_multi(const NnBatchDev* batch, unsigned n_blocks, float r2, hipStream_t s) {
  static_assert(kNnCertBlockQueries == kCertPerWave * (kBlock / kWave), "queries per certificate block");
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_nn_certify_multi, dim3(n_blocks), dim3(kBlock), 0, s, batch, r2);
}
void launch_nn_bounded_half_multi(const NnBatchDev* batch, unsigned n_blocks, float r2, hipStream_t s) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_nn_bounded_half_multi, dim3(n_blocks), dim3(kBlock), 0, s, batch, r2);
}
void launch_query_keys_multi(bool keys32, const NnBatchDev* batch, unsigned n_blocks, float r2, void* keys, unsigned* vals, unsigned* counts, hipStream_t s) {
  if (!n_blocks) return;
  if (keys32) hipLaunchKernelGGL(k_query_keys_multi<unsigned>, dim3(n_blocks), dim3(kBlock), 0, s, batch, r2, (unsigned*)keys, vals, counts);
  else hipLaunchKernelGGL(k_query_keys_multi<unsigned long long>, dim3(n_blocks), dim3(kBlock), 0, s, batch, r2, (unsigned long long*)keys, vals, counts);
}
void launch_query_seed_multi(bool keys32, const NnBatchDev* batch, unsigned n_blocks, float r2, void* keys, unsigned* vals, unsigned* counts, hipStream_t s) {
  if (!n_blocks) return;
  static const int probes = [] { const char* e = getenv("E3D_NN_SEED_PROBES"); return e ? atoi(e) : 8; }();      // (4 or 8 probes per query: experiments)
  if (probes == 4) {
    if (keys32) hipLaunchKernelGGL((k_query_seed_multi<unsigned, 4>), dim3(n_blocks), dim3(kBlock), 0, s, batch, r2, (unsigned*)keys, vals, counts);
    else hipLaunchKernelGGL((k_query_seed_multi<unsigned long long, 4>), dim3(n_blocks), dim3(kBlock), 0, s, batch, r2, (unsigned long long*)keys, vals, counts);
  } else {
    if (keys32) hipLaunchKernelGGL((k_query_seed_multi<unsigned, 8>), dim3(n_blocks), dim3(kBlock), 0, s, batch, r2, (unsigned*)keys, vals, counts);
    else hipLaunchKernelGGL((k_query_seed_multi<unsigned long long, 8>), dim3(n_blocks), dim3(kBlock), 0, s, batch, r2, (unsigned long long*)keys, vals, counts);
  }
}
void launch_nn_rows_multi(const NnBatchDev* batch, unsigned n_blocks, const unsigned* order, float r2, hipStream_t s) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_nn_rows_multi, dim3(n_blocks), dim3(kBlock), 0, s, batch, order, r2, row_span_setting());
}
void launch_corr_update_multi(const NnBatchDev* batch, unsigned n_blocks, unsigned* block_counts, double* block_d2, unsigned* block_groups, hipStream_t s) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_corr_update_multi, dim3(n_blocks), dim3(kBlock), 0, s, batch, block_counts, block_d2, block_groups);
}
void launch_corr_totals_multi(const NnBatchDev* batch, int n_pairs, unsigned n_chunks, const unsigned* block_counts, const double* block_d2,
                              const unsigned* block_groups, unsigned long long* chunk_sum, double* chunk_d2, unsigned* chunk_groups,
                              unsigned* chunk_rewritten, unsigned long long* totals, double* total_d2, hipStream_t s) {
  static_assert(kNnScanChunk == kScanChunk, "blocks per chunk");
  if (!n_pairs || !n_chunks) return;
  hipLaunchKernelGGL(k_scan_chunk_sums_multi, dim3(n_chunks), dim3(kScanChunk), 0, s, batch, block_counts, block_d2, block_groups, chunk_sum, chunk_d2,
                     chunk_groups, chunk_rewritten);
  hipLaunchKernelGGL(k_scan_chunks_multi, dim3((unsigned)n_pairs), dim3(1024), 0, s, batch, chunk_sum, (const double*)chunk_d2, totals, total_d2, chunk_groups,
                     (const unsigned*)chunk_rewritten);
  hipLaunchKernelGGL(k_group_list_multi, dim3(n_chunks), dim3(kScanChunk), 0, s, batch, block_groups, (const unsigned*)chunk_groups);
}

void launch_gather_corr(const float* sxyz, const float* snrm, const float* txyz, const float* tnrm, const int* iq,
                        const int* im, size_t n, float4* A, float4* B, float4* C, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_gather_corr, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, sxyz, snrm, txyz, tnrm, iq,
                     im, n, A, B, C);
}

void launch_unpermute_matches(const int* match_pos, const float* match_d2, const unsigned* order, size_t n, const float4* Gsrc,
                              const float4* Gtgt, int* out_idx, float* out_d2, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_unpermute_matches, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, match_pos, match_d2,
                     order, n, Gsrc, Gtgt, out_idx, out_d2);
}

void launch_match_d2_by_original(const int* match_pos, const float* match_d2, const unsigned* order, size_t n, const float4* Gsrc, float* out,
                                 hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_match_d2_by_original, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, match_pos, match_d2, order, n, Gsrc, out);
}

void launch_lm_pass(int mode, const LmSet* sets, const int* block_set, int block_base, int nblocks, double* partial, hipStream_t s) {
  if (nblocks <= 0) return;
  switch (mode) {
    case kModeCost:
      hipLaunchKernelGGL(k_lm_pass<kModeCost>, dim3(nblocks), dim3(kBlock), 0, s, sets, block_set, block_base, partial);
      break;
    case kModeOne:
      hipLaunchKernelGGL(k_lm_pass<kModeOne>, dim3(nblocks), dim3(kBlock), 0, s, sets, block_set, block_base, partial);
      break;
    case kModeTwo:
      hipLaunchKernelGGL(k_lm_pass<kModeTwo>, dim3(nblocks), dim3(kBlock), 0, s, sets, block_set, block_base, partial);
      break;
    default:
      hipLaunchKernelGGL(k_lm_pass<kModeTwoCross>, dim3(nblocks), dim3(kBlock), 0, s, sets, block_set, block_base, partial);
      break;
  }
}

void launch_lm_cost_multi(const LmSet* sets, const LmPose* poses, int n_sets, int n_poses, const int* block_set, int nblocks,
                          double* partial, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(k_lm_cost_multi, dim3(nblocks), dim3(kBlock), 0, s, sets, poses, n_sets, n_poses, block_set, partial);
}

void launch_lm_reduce(const double* partial, const LmSet* sets, int n_sets, int nacc, double* out, hipStream_t s) {
  if (n_sets <= 0) return;
  hipLaunchKernelGGL(k_lm_reduce, dim3(n_sets), dim3(kLmSlot * kRedParts), 0, s, partial, sets, nacc, out);
}

}  // namespace e3d
