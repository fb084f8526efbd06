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
    const float4 l = L4[i];
    const float3 p = pcl_se3(T, l.x, l.y, l.z);
    G4[i] = make_float4(p.x, p.y, p.z, l.w);
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
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool start = (j < n) && (j == 0 || keys[j] != keys[j - 1]);
  const unsigned long long b = __ballot(start);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(counter, (unsigned)__popcll(b));
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

// flags -> per-block counts (first stage of the order-preserving compaction)
__global__ __launch_bounds__(kBlock) void k_match_block_counts(const int* __restrict__ match_pos, size_t n,
                                                               unsigned* __restrict__ block_counts,
                                                               double* __restrict__ block_d2,
                                                               const float* __restrict__ match_d2) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool f = (j < n) && (match_pos[j] >= 0);
  const unsigned long long b = __ballot(f);
  double d = f ? (double)match_d2[j] : 0.0;
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

// single-block exclusive scan of the per-block counts (nblocks <= a few hundred thousand)
__global__ void k_scan_block_counts(const unsigned* __restrict__ counts, int nblocks,
                                    unsigned* __restrict__ offsets, unsigned long long* __restrict__ total,
                                    const double* __restrict__ block_d2, double* __restrict__ total_d2) {
  __shared__ unsigned long long s[1024];
  __shared__ double sd[1024];
  const int t = threadIdx.x, T = blockDim.x;
  const int per = (nblocks + T - 1) / T;
  const int b0 = t * per, b1 = min(nblocks, b0 + per);
  unsigned long long sum = 0; double d = 0;
  for (int b = b0; b < b1; ++b) { sum += counts[b]; d += block_d2[b]; }
  s[t] = sum; sd[t] = d;
  __syncthreads();
  if (t == 0) {
    unsigned long long run = 0; double dr = 0;
    for (int k = 0; k < T; ++k) { const unsigned long long v = s[k]; s[k] = run; run += v; dr += sd[k]; }
    *total = run; *total_d2 = dr;
  }
  __syncthreads();
  unsigned long long run = s[t];
  for (int b = b0; b < b1; ++b) { offsets[b] = (unsigned)run; run += counts[b]; }
}

// second stage: write the correspondence planes in source (cell) order
__global__ __launch_bounds__(kBlock) void k_compact_corr(const int* __restrict__ match_pos, size_t n,
                                                         const unsigned* __restrict__ block_offsets,
                                                         const float4* __restrict__ Gsrc, const float4* __restrict__ LNsrc,
                                                         Affine Tsrc, const float4* __restrict__ Gtgt,
                                                         const float4* __restrict__ LNtgt, Affine Ttgt,
                                                         float4* __restrict__ A, float4* __restrict__ B,
                                                         float4* __restrict__ C, size_t out_base) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int m = (j < n) ? match_pos[j] : -1;
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
  const float4 sp = Gsrc[j];
  const float4 ln = LNsrc[j];
  const float3 sn = pcl_so3(Tsrc, ln.x, ln.y, ln.z);
  const float4 tp = Gtgt[m];
  const float4 tl = LNtgt[m];
  const float3 tn = pcl_so3(Ttgt, tl.x, tl.y, tl.z);
  A[o] = make_float4(sp.x, sp.y, sp.z, sn.x);
  B[o] = make_float4(sn.y, sn.z, tp.x, tp.y);
  C[o] = make_float4(tp.z, tn.x, tn.y, tn.z);
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

// un-permute NN results to original source order / original target indices
__global__ __launch_bounds__(kBlock) void k_unpermute_matches(const int* __restrict__ match_pos,
                                                              const float* __restrict__ match_d2, size_t n,
                                                              const float4* __restrict__ Gsrc,
                                                              const float4* __restrict__ Gtgt,
                                                              int* __restrict__ out_idx, float* __restrict__ out_d2) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned oi = __float_as_uint(Gsrc[j].w);
  const int m = match_pos[j];
  out_idx[oi] = (m >= 0) ? (int)__float_as_uint(Gtgt[m].w) : -1;
  out_d2[oi] = (m >= 0) ? match_d2[j] : 0.f;
}

// =================================================================================================
// a7 + a8: PointToPlaneICPImpl::compute accumulate / cost passes
//          (icp_point_to_plane_impl.h:119-211 and :240-266), fused: one pass over the
//          correspondence planes yields cost and (mode-dependent) the Gramian blocks.
// =================================================================================================
// f32 residual / Jacobian rows, literally as written in the reference (left-to-right).
struct CorrRows {
  float r1, r2;
  float j1s[6], j1t[6], j2s[6], j2t[6];
};

template <bool NEED_SRC, bool NEED_TGT>
__device__ __forceinline__ void corr_rows(const LmSet& S, const float4 a, const float4 b, const float4 c, CorrRows& o) {
  // inner poses applied with Eigen's R*p + t order (impl.h:144-151)
  const float lsx = a.x, lsy = a.y, lsz = a.z, lnx = a.w, lny = b.x, lnz = b.y;
  const float ltx = b.z, lty = b.w, ltz = c.x, lmx = c.y, lmy = c.z, lmz = c.w;
  const float spx = dot3e(S.Rs[0], S.Rs[1], S.Rs[2], lsx, lsy, lsz) + S.ts[0];
  const float spy = dot3e(S.Rs[3], S.Rs[4], S.Rs[5], lsx, lsy, lsz) + S.ts[1];
  const float spz = dot3e(S.Rs[6], S.Rs[7], S.Rs[8], lsx, lsy, lsz) + S.ts[2];
  const float snx = dot3e(S.Rs[0], S.Rs[1], S.Rs[2], lnx, lny, lnz);
  const float sny = dot3e(S.Rs[3], S.Rs[4], S.Rs[5], lnx, lny, lnz);
  const float snz = dot3e(S.Rs[6], S.Rs[7], S.Rs[8], lnx, lny, lnz);
  const float tpx = dot3e(S.Rt[0], S.Rt[1], S.Rt[2], ltx, lty, ltz) + S.tt[0];
  const float tpy = dot3e(S.Rt[3], S.Rt[4], S.Rt[5], ltx, lty, ltz) + S.tt[1];
  const float tpz = dot3e(S.Rt[6], S.Rt[7], S.Rt[8], ltx, lty, ltz) + S.tt[2];
  const float tnx = dot3e(S.Rt[0], S.Rt[1], S.Rt[2], lmx, lmy, lmz);
  const float tny = dot3e(S.Rt[3], S.Rt[4], S.Rt[5], lmx, lmy, lmz);
  const float tnz = dot3e(S.Rt[6], S.Rt[7], S.Rt[8], lmx, lmy, lmz);

  o.r1 = dot3e(snx, sny, snz, tpx - spx, tpy - spy, tpz - spz);                // impl.h:158
  o.r2 = dot3e(tnx, tny, tnz, spx - tpx, spy - tpy, spz - tpz);                // impl.h:185
  if (NEED_TGT) {
    o.j1t[0] = snx; o.j1t[1] = sny; o.j1t[2] = snz;                            // impl.h:162-168
    o.j1t[3] = -sny * tpz + snz * tpy;
    o.j1t[4] = snx * tpz - snz * tpx;
    o.j1t[5] = -snx * tpy + sny * tpx;
    o.j2t[0] = -tnx; o.j2t[1] = -tny; o.j2t[2] = -tnz;                         // impl.h:188-195
    o.j2t[3] = tny * tpz - tny * (tpz - spz) - tnz * tpy + tnz * (tpy - spy);
    o.j2t[4] = -tnx * tpz + tnx * (tpz - spz) + tnz * tpx - tnz * (tpx - spx);
    o.j2t[5] = tnx * tpy - tnx * (tpy - spy) - tny * tpx + tny * (tpx - spx);
  }
  if (NEED_SRC) {
    o.j1s[0] = -snx; o.j1s[1] = -sny; o.j1s[2] = -snz;                         // impl.h:170-177
    o.j1s[3] = sny * spz - sny * (spz - tpz) - snz * spy + snz * (spy - tpy);
    o.j1s[4] = -snx * spz + snx * (spz - tpz) + snz * spx - snz * (spx - tpx);
    o.j1s[5] = snx * spy - snx * (spy - tpy) - sny * spx + sny * (spx - tpx);
    o.j2s[0] = tnx; o.j2s[1] = tny; o.j2s[2] = tnz;                            // impl.h:197-204
    o.j2s[3] = -tny * spz + tnz * spy;
    o.j2s[4] = tnx * spz - tnz * spx;
    o.j2s[5] = -tnx * spy + tny * spx;
  }
}

// accumulate upper triangle of J J^T (21) and r*J (6) in f64 from f32 rows cast to f64 first
// (icp_point_to_plane_impl.h:91-112,179-182: .cast<double>() before the product)
__device__ __forceinline__ void acc_diag(double* H21, double* b6, const float* j, float r) {
  double J[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) J[i] = (double)j[i];
  const double R = (double)r;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int l = i; l < 6; ++l) { H21[k] += J[i] * J[l]; ++k; }
    b6[i] += R * J[i];
  }
}
__device__ __forceinline__ void acc_cross(double* H36, const float* js, const float* jt) {
  double A[6], Bv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { A[i] = (double)js[i]; Bv[i] = (double)jt[i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int l = 0; l < 6; ++l) H36[6 * i + l] += A[i] * Bv[l];
}

// Output slot layout per block / per set (kLmSlot doubles):
//   [0] cost, [1..21] SS upper, [22..27] bs, [28..48] TT upper, [49..54] bt, [55..90] ST (6x6)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_lm_pass(const float4* __restrict__ A, const float4* __restrict__ B,
                                                    const float4* __restrict__ C, const LmSet* __restrict__ sets,
                                                    const int* __restrict__ block_set, int block_base,
                                                    double* __restrict__ partial) {
  constexpr bool kOne = (MODE == kModeOne);
  constexpr bool kTwo = (MODE == kModeTwo || MODE == kModeTwoCross);
  constexpr bool kCross = (MODE == kModeTwoCross);
  constexpr int NACC = (MODE == kModeCost) ? 1 : kOne ? 28 : kCross ? 91 : 55;
  const int gb = block_base + blockIdx.x;
  const int si = block_set[gb];
  const LmSet S = sets[si];
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

  const long long stride = (long long)S.nblocks * kBlock;
  for (long long c = (long long)(gb - S.block_begin) * kBlock + threadIdx.x; c < S.n; c += stride) {
    const float4 a = A[S.off + c], b = B[S.off + c], cc = C[S.off + c];
    CorrRows R;
    if (MODE == kModeCost) {
      corr_rows<false, false>(S, a, b, cc, R);
      acc[0] += (double)(R.r1 * R.r1);
      acc[0] += (double)(R.r2 * R.r2);
    } else if (kOne) {
      // only one side of the pair has variables (the other is impl cloud 0); block-uniform branch
      if (S.side == 0) {
        corr_rows<true, false>(S, a, b, cc, R);
        acc[0] += (double)(R.r1 * R.r1);
        acc_diag(acc + 1, acc + 22, R.j1s, R.r1);
        acc[0] += (double)(R.r2 * R.r2);
        acc_diag(acc + 1, acc + 22, R.j2s, R.r2);
      } else {
        corr_rows<false, true>(S, a, b, cc, R);
        acc[0] += (double)(R.r1 * R.r1);
        acc_diag(acc + 1, acc + 22, R.j1t, R.r1);
        acc[0] += (double)(R.r2 * R.r2);
        acc_diag(acc + 1, acc + 22, R.j2t, R.r2);
      }
    } else if (kTwo) {
      corr_rows<true, true>(S, a, b, cc, R);
      acc[0] += (double)(R.r1 * R.r1);
      acc_diag(acc + 1, acc + 22, R.j1s, R.r1);
      acc_diag(acc + 28, acc + 49, R.j1t, R.r1);
      if (kCross) acc_cross(acc + 55, R.j1s, R.j1t);
      acc[0] += (double)(R.r2 * R.r2);
      acc_diag(acc + 1, acc + 22, R.j2s, R.r2);
      acc_diag(acc + 28, acc + 49, R.j2t, R.r2);
      if (kCross) acc_cross(acc + 55, R.j2s, R.j2t);
    }
  }
  // wave reduce -> LDS -> fixed-order block sum
  __shared__ double s[kBlock / kWave][NACC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s[w][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLmSlot) {
    double v = 0.0;
    if (threadIdx.x < NACC) {
      v = s[0][threadIdx.x];
      for (int k = 1; k < kBlock / kWave; ++k) v += s[k][threadIdx.x];
    }
    partial[(size_t)gb * kLmSlot + threadIdx.x] = v;   // unused slots of cheaper modes are zero
  }
}

// one block per set: sum the set's block partials in block order
__global__ void k_lm_reduce(const double* __restrict__ partial, const LmSet* __restrict__ sets, int nacc,
                            double* __restrict__ out) {
  const LmSet S = sets[blockIdx.x];
  if ((int)threadIdx.x >= nacc) return;
  double v = 0.0;
  for (int b = 0; b < S.nblocks; ++b) v += partial[(size_t)(S.block_begin + b) * kLmSlot + threadIdx.x];
  out[(size_t)blockIdx.x * kLmSlot + threadIdx.x] = v;
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
  hipLaunchKernelGGL(k_count_cells, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, keys, n, counter);
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
                       unsigned* block_offsets, double* block_d2, unsigned long long* total, double* total_d2,
                       hipStream_t s) {
  const int nb = (int)div_up(n ? n : 1, kBlock);
  hipLaunchKernelGGL(k_match_block_counts, dim3(nb), dim3(kBlock), 0, s, match_pos, n, block_counts, block_d2,
                     match_d2);
  hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, s, block_counts, nb, block_offsets, total,
                     block_d2, total_d2);
}

void launch_compact_corr(const int* match_pos, size_t n, const unsigned* block_offsets, const float4* Gsrc,
                         const float4* LNsrc, const Affine& Tsrc, const float4* Gtgt, const float4* LNtgt,
                         const Affine& Ttgt, float4* A, float4* B, float4* C, size_t out_base, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_compact_corr, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, match_pos, n,
                     block_offsets, Gsrc, LNsrc, Tsrc, Gtgt, LNtgt, Ttgt, A, B, C, out_base);
}

void launch_gather_corr(const float* sxyz, const float* snrm, const float* txyz, const float* tnrm, const int* iq,
                        const int* im, size_t n, float4* A, float4* B, float4* C, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_gather_corr, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, sxyz, snrm, txyz, tnrm, iq,
                     im, n, A, B, C);
}

void launch_unpermute_matches(const int* match_pos, const float* match_d2, size_t n, const float4* Gsrc,
                              const float4* Gtgt, int* out_idx, float* out_d2, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_unpermute_matches, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, match_pos, match_d2,
                     n, Gsrc, Gtgt, out_idx, out_d2);
}

void launch_lm_pass(int mode, const float4* A, const float4* B, const float4* C, const LmSet* sets,
                    const int* block_set, int block_base, int nblocks, double* partial, hipStream_t s) {
  if (nblocks <= 0) return;
  switch (mode) {
    case kModeCost:
      hipLaunchKernelGGL(k_lm_pass<kModeCost>, dim3(nblocks), dim3(kBlock), 0, s, A, B, C, sets, block_set, block_base, partial);
      break;
    case kModeOne:
      hipLaunchKernelGGL(k_lm_pass<kModeOne>, dim3(nblocks), dim3(kBlock), 0, s, A, B, C, sets, block_set, block_base, partial);
      break;
    case kModeTwo:
      hipLaunchKernelGGL(k_lm_pass<kModeTwo>, dim3(nblocks), dim3(kBlock), 0, s, A, B, C, sets, block_set, block_base, partial);
      break;
    default:
      hipLaunchKernelGGL(k_lm_pass<kModeTwoCross>, dim3(nblocks), dim3(kBlock), 0, s, A, B, C, sets, block_set, block_base, partial);
      break;
  }
}

void launch_lm_reduce(const double* partial, const LmSet* sets, int n_sets, int nacc, double* out, hipStream_t s) {
  if (n_sets <= 0) return;
  hipLaunchKernelGGL(k_lm_reduce, dim3(n_sets), dim3(128), 0, s, partial, sets, nacc, out);
}

}  // namespace e3d
