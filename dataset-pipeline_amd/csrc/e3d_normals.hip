// e3d_normals.hip -- k-nearest-neighbour normal estimation on gfx950 (path A' of SURVEY.md section 8):
//   pcl::NormalEstimationTwoPassOMP::computeFeature      src/geometry/two_pass_normal_3d_omp.hpp:48-119
//   pcl::computePointNormalTwoPass (indices)             src/geometry/two_pass_normal_3d.h:92-109
//   pcl::computeMeanAndCovarianceMatrixTwoPass           src/geometry/two_pass_centroid.hpp:155-259
//   pcl::solvePlaneParameters / pcl::eigen33 / flipNormalTowardsViewpoint   (PCL 1.10, recalled)
//
// Exact kNN without a tree: points are bucketed in a uniform grid (hash table); a query scans its 27-cell block
// into a per-thread max-heap kept in LDS (column layout, bank-conflict free) and is RESOLVED when its k-th
// neighbour is provably closer than the nearest face of the block (nothing outside can beat or tie it).
// Unresolved queries (sparse regions, outliers) are retried on a 4x coarser grid, and so on until the block
// covers the whole cloud -- so every neighbour list is exact, ordered by (f32 squared distance, original index),
// and includes the query point itself, as FLANN's nearestKSearch returns it.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/e3d_hip.h"
#include "../../include/e3d_libm.h"   // bit-defined atan2f / cosf / sinf: the same bits on host and device
#include "e3d_icp_kernels.hpp"

#pragma clang fp contract(off)

namespace e3d {

constexpr int kKnnBlock = 128;   // threads per block; LDS = 128 * k * 8 bytes
constexpr int kKnnMaxK = 128;

struct KnnGrid {
  GridDesc g;
  float cell;
  float dmin[3], dmax[3];   // data bbox
  float slack;
  // dense cell-start directory (exclusive max-scan of the run ends, as in the ICP grid): the cells [xa, xb] of one (y, z) row are
  // ONE run [S[row + xa], S[row + xb + 1]) found with two independent loads; nullptr: the hash table, cell by cell
  const unsigned* S;
  unsigned D[3];
  unsigned xcd_map;         // G > 0: the scan kernels hand an XCD G consecutive blocks of every 8 G (knn_block)
};
constexpr int kKnnSelFallback = 253;

// Workgroup b runs on XCD b % 8 (observed placement, not a contract -- another placement is slower, not wrong), and every XCD has
// its own 4 MB L2.  Queries are in cell order and a block's candidates are the rows of the 27 cells around them: the ~100 blocks
// that follow share most of those rows.  Dealt round-robin, those blocks land on all eight XCDs and every L2 fetches the same rows
// from HBM (round 4's counters: 2.7 GB fetched per scan of a 0.32 GB point array, 4 GB by the sampled histogram).  The logical
// block index below hands an XCD G CONSECUTIVE blocks of every stretch of 8 G: the blocks that share rows share an L2 (fetch of
// the scan kernel and of the sampled histogram halved, profiles/round5_pmc_normals_k8_fetch_xcd*.txt), and every XCD still gets
// its pieces from all over the cloud -- one contiguous EIGHTH of the query order per XCD (round 4) cost a scanner-sampled scan
// 8 - 25 %, because an eighth of such a scan is not an eighth of the work.  A bijection of [0, nb): every query is handled
// exactly once.  xcd_map = G (E3D_KNN_XCD, default 64 blocks = 8192 queries; 0: off).
__device__ __forceinline__ unsigned knn_block(const KnnGrid& G, unsigned b, unsigned nb) {
  constexpr unsigned kXcds = 8;
  const unsigned g = G.xcd_map, super = kXcds * g;
  if (!g || b >= (nb / super) * super) return b;              // (the last, incomplete stretch keeps its order)
  const unsigned s0 = (b / super) * super, r = b - s0;
  return s0 + (r % kXcds) * g + r / kXcds;
}

// sqdist_l2 (query - candidate, (dx^2 + dy^2) + dz^2, every operation rounded on its own) with x and y as ONE packed operation each:
// a candidate's x and y arrive in consecutive registers, so v_pk_add_f32 / v_pk_mul_f32 take them as they are (the compiler's own
// pairing -- x with z -- needs two register moves per candidate, which it places right behind the loads and so waits for them)
typedef float knn_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float knn_sqdist(const knn_f2 qxy, float qz, const float4 c) {
#ifdef E3D_KNN_FORCE_W   // experiment: 16 instead of 12 bytes per candidate and lane through the vector L1
  asm volatile("" :: "v"(c.w));
#endif
  const knn_f2 cxy = {c.x, c.y};
  const knn_f2 d = qxy - cxy;
  const knn_f2 sq = d * d;
  const float dz = qz - c.z;
  float acc = sq.x + sq.y;
  acc = acc + dz * dz;
  return acc;
}

// runs of the cells [x0, x1] of row (y, z): with the dense directory one run, with the hash table one per cell.  body(begin, end)
template <class F>
__device__ __forceinline__ void knn_row(const KnnGrid& G, const HashEntry* __restrict__ table, int x0, int x1, int y, int z, F body) {
  if (G.S) {
    if (y < 0 || z < 0 || y >= (int)G.D[1] || z >= (int)G.D[2]) return;
    x0 = max(x0, 0); x1 = min(x1, (int)G.D[0] - 1);
    if (x0 > x1) return;
    const size_t row = ((size_t)z * G.D[1] + (size_t)y) * G.D[0];
    const unsigned s0 = G.S[row + (size_t)x0], e0 = G.S[row + (size_t)x1 + 1];
    body(s0, e0);
    return;
  }
  constexpr int kMaxC = (1 << 21) - 1;
  if (y < 0 || z < 0 || y > kMaxC || z > kMaxC) return;
  for (int x = max(x0, 0); x <= min(x1, kMaxC); ++x) {
    const unsigned long long key = cell_key(x, y, z);
    unsigned h = hash_key(key) & G.g.mask;
    for (;;) {
      const HashEntry en = table[h];
      if (en.key == key) { body(en.start, en.end); break; }
      if (en.key == kEmptyKey) break;
      h = (h + 1) & G.g.mask;
    }
  }
}

// linear index of a point's cell in the dense bounding grid (the grid's origin lies two cells below the bounding box and D covers the
// box plus two cells, so every point falls inside; the clamp only guards against a NaN coordinate)
__global__ __launch_bounds__(kBlock) void k_cell_keys_dense(const float* __restrict__ xyz, size_t n, GridDesc g, unsigned Dx, unsigned Dy,
                                                            unsigned Dz, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned cx = (unsigned)min(max(cell_coord(xyz[3 * i], g.origin[0], g.inv_cell), 0), (int)Dx - 1);
  const unsigned cy = (unsigned)min(max(cell_coord(xyz[3 * i + 1], g.origin[1], g.inv_cell), 0), (int)Dy - 1);
  const unsigned cz = (unsigned)min(max(cell_coord(xyz[3 * i + 2], g.origin[2], g.inv_cell), 0), (int)Dz - 1);
  keys[i] = (cz * Dy + cy) * Dx + cx;
  vals[i] = (unsigned)i;
}

// The dense directory S[c] = number of points with a key below c (= the start of cell c's run, or of the next occupied cell's) over
// the ncell + 2 entries of the bounding grid.  Less than 1 % of a scan's cells hold points, so clearing the array, marking the run
// ends and an exclusive max scan over all of it (12 B of traffic per cell: 1.3 ms at 318 M cells) is replaced by ONE write of each
// word: a coarse directory C[t] = number of points with key < 4096 t comes from that recipe on a 4096 times smaller array, then
// one pass over the points writes the occupied cells' starts and an occupancy bit per cell, and block t fills the 4096 cells of its
// tile: every empty cell takes the next occupied one's start (wave ballots, no scan; behind the tile's last one C[t + 1]), the
// tile leaves as full lines.  Tiles without points (most) are one coalesced fill.
constexpr unsigned kDirTileLog2 = 12, kDirTile = 1u << kDirTileLog2;
// over the sorted points: the last point of a tile's run writes the coarse end; the first point of a cell's run writes the cell's
// start into S and sets the cell's bit (a tile of a dense region holds a million points: its block must not walk them)
__global__ __launch_bounds__(kBlock) void k_dir_mark(const unsigned* __restrict__ keys, size_t n, unsigned* __restrict__ coarse_ends,
                                                     unsigned* __restrict__ S, unsigned* __restrict__ occupied) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned key = keys[j];
  if (j == 0 || keys[j - 1] != key) {
    S[key] = (unsigned)j;
    atomicOr(&occupied[key >> 5], 1u << (key & 31u));
  }
  const unsigned t = key >> kDirTileLog2;
  if (j + 1 < n && (keys[j + 1] >> kDirTileLog2) == t) return;
  coarse_ends[t] = (unsigned)(j + 1);
}

__global__ __launch_bounds__(256) void k_dense_directory(const unsigned* __restrict__ keys, const unsigned* __restrict__ occupied,
                                                         const unsigned* __restrict__ coarse, unsigned* __restrict__ S, size_t n_entries) {
  __shared__ unsigned sl[kDirTile];
  __shared__ unsigned first_of_wave[4], tail_of_wave[4];
  constexpr unsigned kUnset = 0xFFFFFFFFu;
  const unsigned tid = threadIdx.x;
  const size_t c0 = (size_t)blockIdx.x << kDirTileLog2;
  const unsigned lb0 = coarse[blockIdx.x], lb1 = coarse[blockIdx.x + 1];
  const unsigned cnt = (unsigned)min((size_t)kDirTile, n_entries - c0);
  if (lb0 == lb1) {                                                      // no point in the tile (most tiles)
    if (cnt == kDirTile) {                                               // whole tile: 16-byte stores (the tile starts on a 16 KB boundary of S)
      typedef unsigned u4_t __attribute__((ext_vector_type(4)));
      u4_t* __restrict__ S4 = reinterpret_cast<u4_t*>(S + c0);
      const u4_t v = {lb1, lb1, lb1, lb1};
#pragma unroll
      for (unsigned i = tid; i < kDirTile / 4; i += 256) S4[i] = v;
    } else {
      for (unsigned i = tid; i < cnt; i += 256) S[c0 + i] = lb1;
    }
    return;
  }
  // the occupied cells' starts: a tile with few points (a scan's surfaces: a few hundred) finds them in its stretch of the sorted
  // keys; one with many (the floor under a scanner: a million) takes what k_dir_mark wrote, through the occupancy bits.  Then each
  // wave owns a quarter of the tile and walks it from the end, 64 cells a round: an empty cell takes the nearest start at or after it
  if (lb1 - lb0 <= 2u * kDirTile) {
    for (unsigned i = tid; i < kDirTile; i += 256) sl[i] = kUnset;
    __syncthreads();
    for (unsigned j = lb0 + tid; j < lb1; j += 256) {
      const unsigned key = keys[j];
      if (j == lb0 || keys[j - 1] != key) sl[key - (unsigned)c0] = j;
    }
  } else {
    const unsigned* __restrict__ occ = occupied + (c0 >> 5);
    for (unsigned i = tid; i < kDirTile; i += 256) {
      const bool has = i < cnt && ((occ[i >> 5] >> (i & 31u)) & 1u) != 0u;
      sl[i] = has ? S[c0 + i] : kUnset;
    }
  }
  __syncthreads();
  const unsigned w = tid >> 6, lane = tid & 63;
  unsigned carry = kUnset;                                               // the nearest start behind the cells seen so far (wave-uniform)
  for (int r = (int)(kDirTile / 4 / 64) - 1; r >= 0; --r) {
    const unsigned i = w * (kDirTile / 4) + 64u * (unsigned)r + lane;
    const unsigned v = sl[i];
    const unsigned long long mask = __ballot(v != kUnset);
    const unsigned long long at_or_after = mask & (~0ull << lane);
    const int src = at_or_after ? __builtin_ctzll(at_or_after) : (int)lane;
    const unsigned got = __shfl(v, src);
    sl[i] = at_or_after ? got : carry;
    if (mask) carry = __shfl(v, __builtin_ctzll(mask));
  }
  if (lane == 0) first_of_wave[w] = carry;
  __syncthreads();
  if (tid < 4) {
    unsigned t = lb1;                                                    // behind the tile's last start: the next tile's first point
    for (int ww = 3; ww > (int)tid; --ww) if (first_of_wave[ww] != kUnset) t = first_of_wave[ww];
    tail_of_wave[tid] = t;
  }
  __syncthreads();
  for (unsigned i = tid; i < cnt; i += 256) {
    const unsigned v = sl[i];
    S[c0 + i] = v != kUnset ? v : tail_of_wave[i >> (kDirTileLog2 - 2)];
  }
}

// -DE3D_KNN_PROF=1 (E3D_EXTRA_HIPCC_FLAGS): shader-clock stop-watch of the sections of the two scan kernels, summed over the waves
// (lane 0 of each wave adds its differences); E3D_KNN_STATS=1 prints them after a call.  Diagnostics only.
#ifdef E3D_KNN_PROF
__device__ unsigned long long g_knn_prof[32];
#define KP_DECL(N) unsigned long long kp_t[N] = {}
#define KP_MARK(i) (kp_t[i] = __builtin_readcyclecounter())
#define KP_FLUSH(base, N) do { if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int kp_i = 1; kp_i < (N); ++kp_i) if (kp_t[kp_i] && kp_t[kp_i - 1]) atomicAdd(&g_knn_prof[(base) + kp_i - 1], kp_t[kp_i] - kp_t[kp_i - 1]); atomicAdd(&g_knn_prof[(base) + 15], 1ull); } } while (0)
#else
#define KP_DECL(N) do { } while (0)
#define KP_MARK(i) do { } while (0)
#define KP_FLUSH(base, N) do { } while (0)
#endif

constexpr int kKnnBins = 64;           // histogram bins of the two-pass variant: cell^2 / 32 wide, [0, 2 cell^2)
constexpr int kKnnHistBlock = 256;

// Pass A of the two-pass variant (see k_knn_normals<3>): per query the first bin of the squared-distance histogram of its 27
// cells at which the count reaches k (255: none -- the list-maintaining variant takes the query).  Only 64 bytes of LDS per thread:
// the kernel runs at full occupancy, which is what this latency-bound scan needs.
// stride > 1 (todo == nullptr): the SAMPLED form for the single-pass variant k_knn_normals<4> -- thread gi looks at query gi * stride
// only, and k is the count the threshold of that variant aims at (n_todo = number of sampled queries).
__global__ __launch_bounds__(kKnnHistBlock) void k_knn_hist(const float4* __restrict__ P4, const unsigned* __restrict__ todo, size_t n_todo,
                                                            const HashEntry* __restrict__ table, KnnGrid G, int k,
                                                            const float4* __restrict__ Q4, unsigned char* __restrict__ sel_bin, unsigned stride,
                                                            float estimate) {
  __shared__ unsigned hw[kKnnBins / 4][kKnnHistBlock];
  const int tid = threadIdx.x;
  const size_t gi = (size_t)knn_block(G, blockIdx.x, gridDim.x) * blockDim.x + tid;
  if (gi >= n_todo) return;
  const unsigned qid = todo ? todo[gi] : (unsigned)gi * stride;
  const float4 q = Q4[qid];
  KP_DECL(6); KP_MARK(0);
#pragma unroll
  for (int wv = 0; wv < kKnnBins / 4; ++wv) hw[wv][tid] = 0u;
  const int cx = cell_coord(q.x, G.g.origin[0], G.g.inv_cell);
  const int cy = cell_coord(q.y, G.g.origin[1], G.g.inv_cell);
  const int cz = cell_coord(q.z, G.g.origin[2], G.g.inv_cell);
  // A scan samples its surfaces with a density that falls with the squared range, so a 27-cell block of a grid sized for the mean
  // density holds anything from a few to thousands of points.  The directory words of the block's nine rows give its population
  // n27 before any point is touched, and the histogram's bin width follows it (round 4): cell^2 / (32 f) with f = 1, 2, 4 for
  // n27 <= 12 k, 24 k, more.  The k-th neighbour of a dense block is far inside the block, and with bins sized for the mean density
  // most of the block's candidates fell into the first bins, more than the k + 4 list slots of pass B hold (the query then fell
  // back to the list-maintaining variant: 1.7 M of 20 M queries of a scanner-sampled scan).  Pass B reads f from the selected bin's
  // byte.  (Handing the densest blocks to a grid of half the cell size instead was built as well: 33.9 instead of 23.9 ms on that
  // scan -- a second sort and a 1.2 G-cell directory for 0.7 M queries; removed.)
  int fexp = 0;
  if (G.S) {
    unsigned n27 = 0;
    for (int oz = -1; oz <= 1; ++oz)
      for (int oy = -1; oy <= 1; ++oy)
        knn_row(G, table, cx - 1, cx + 1, cy + oy, cz + oz, [&](unsigned m, unsigned e) { n27 += e - m; });
    fexp = n27 <= 12u * (unsigned)k ? 0 : (n27 <= 24u * (unsigned)k ? 1 : 2);
    if (estimate > 0.f) {
      // sampled form, small k: the threshold from the block's population alone -- a surface patch of (3 cells)^2 holding n27 points
      // has k of them within r^2 = 9 k / (pi n27) cell^2 -- without touching a candidate.  Any threshold gives the exact result; a
      // poor one sends the query to the two-pass variant.
      const float edge = estimate * 9.0f * (float)k / (3.14159265f * (float)max(n27, 1u));
      if (!(edge < 2.0f)) { sel_bin[gi] = 255; return; }
      const int fe = edge < 0.5f ? 2 : (edge < 1.0f ? 1 : 0);
      const int sel = min(63, (int)(edge * (float)(32 << fe)));
      sel_bin[gi] = (unsigned char)((fe << 6) | sel);
      return;
    }
  }
  const float inv_w2 = (float)(32 << fexp) * G.g.inv_cell * G.g.inv_cell;
  const knn_f2 qxy = {q.x, q.y};
  KP_MARK(1);
  auto add = [&](float d2) {
    const float bb = d2 * inv_w2;
    if (bb < (float)kKnnBins) { const int b = (int)bb; atomicAdd(&hw[b >> 2][tid], 1u << (8 * (b & 3))); }
  };
  for (int oz = -1; oz <= 1; ++oz)
    for (int oy = -1; oy <= 1; ++oy)
      knn_row(G, table, cx - 1, cx + 1, cy + oy, cz + oz, [&](unsigned m, unsigned e) {
        // eight candidates in flight per lane, addressed from one pointer (the loads differ by immediate offsets); the last
        // batch of a row reads up to seven entries past its end (P4 is padded by eight) and masks them
        for (; m + 8 <= e; m += 8) {
          const float4* __restrict__ pm = P4 + m;
          float4 cb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) cb[j] = pm[j];
#pragma unroll
          for (int j = 0; j < 8; ++j) add(knn_sqdist(qxy, q.z, cb[j]));
        }
        if (m < e) {
          const float4* __restrict__ pm = P4 + m;
          const unsigned nvalid = e - m;
          float4 cb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) cb[j] = pm[j];
#pragma unroll
          for (int j = 0; j < 8; ++j) if ((unsigned)j < nvalid) add(knn_sqdist(qxy, q.z, cb[j]));
        }
      });
  KP_MARK(2);
  // first bin at which the running count reaches k (a byte that wrapped only makes the count too small: checked after pass B)
  int sel = 255, cum = 0;
  for (int wv = 0; wv < kKnnBins / 4 && sel == 255; ++wv) {
    const unsigned word = hw[wv][tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cum += (int)((word >> (8 * j)) & 0xFFu);
      if (sel == 255 && cum >= k) sel = 4 * wv + j;
    }
  }
  // [2 bits: bin scale | 6 bits: bin]; k not reached: 255 with the full range (the k-th neighbour is at least sqrt(2) cells away),
  // kKnnSelFallback with a narrowed one (it may still lie inside the block: the list-maintaining variant looks)
  sel_bin[gi] = (unsigned char)(sel == 255 ? (fexp == 0 ? 255 : kKnnSelFallback) : ((fexp << 6) | sel));
  KP_MARK(3); KP_FLUSH(0, 4);
}

__device__ __forceinline__ bool knn_less(float d1, unsigned p1, float d2, unsigned p2, const float4* __restrict__ P4) {
  if (d1 != d2) return d1 < d2;
  return __float_as_uint(P4[p1].w) < __float_as_uint(P4[p2].w);     // tie: lower ORIGINAL index first (rare path)
}

// pcl::computeRoots2 / computeRoots / eigen33 (PCL 1.10 common/eigen.hpp, recalled)
__device__ __forceinline__ void compute_roots2(float b, float c, float* roots) {
  roots[0] = 0.f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

__device__ __forceinline__ void compute_roots(const float* m, float* roots) {
#define M(i, j) m[3 * (i) + (j)]
  const float c0 = M(0, 0) * M(1, 1) * M(2, 2) + 2.f * M(0, 1) * M(0, 2) * M(1, 2) - M(0, 0) * M(1, 2) * M(1, 2) -
                   M(1, 1) * M(0, 2) * M(0, 2) - M(2, 2) * M(0, 1) * M(0, 1);
  const float c1 = M(0, 0) * M(1, 1) - M(0, 1) * M(0, 1) + M(0, 0) * M(2, 2) - M(0, 2) * M(0, 2) + M(1, 1) * M(2, 2) -
                   M(1, 2) * M(1, 2);
  const float c2 = M(0, 0) + M(1, 1) + M(2, 2);
#undef M
  if (fabsf(c0) < FLT_EPSILON) { compute_roots2(c2, c1, roots); return; }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  const float rho = sqrtf(-a_over_3);
  const float theta = e3d_atan2f(sqrtf(-q), half_b) * s_inv3;
  const float cos_theta = e3d_cosf(theta);
  const float sin_theta = e3d_sinf(theta);
  roots[0] = c2_over_3 + 2.f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  if (roots[1] >= roots[2]) {
    t = roots[1]; roots[1] = roots[2]; roots[2] = t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  }
  if (roots[0] <= 0.f) compute_roots2(c2, c1, roots);
}

__device__ __forceinline__ void eigen33_smallest(const float* cov, float& eigenvalue, float* v) {
  float scale = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(cov[i]));
  if (scale <= FLT_MIN) scale = 1.0f;
  float s[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) s[i] = cov[i] / scale;
  float roots[3];
  compute_roots(s, roots);
  eigenvalue = roots[0] * scale;
  s[0] -= roots[0]; s[4] -= roots[0]; s[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  v1[0] = s[1] * s[5] - s[2] * s[4]; v1[1] = s[2] * s[3] - s[0] * s[5]; v1[2] = s[0] * s[4] - s[1] * s[3];
  v2[0] = s[1] * s[8] - s[2] * s[7]; v2[1] = s[2] * s[6] - s[0] * s[8]; v2[2] = s[0] * s[7] - s[1] * s[6];
  v3[0] = s[4] * s[8] - s[5] * s[7]; v3[1] = s[5] * s[6] - s[3] * s[8]; v3[2] = s[3] * s[7] - s[4] * s[6];
  const float l1 = v1[0] * v1[0] + (v1[1] * v1[1] + v1[2] * v1[2]);
  const float l2 = v2[0] * v2[0] + (v2[1] * v2[1] + v2[2] * v2[2]);
  const float l3 = v3[0] * v3[0] + (v3[1] * v3[1] + v3[2] * v3[2]);
  const float* best; float len;
  if (l1 >= l2 && l1 >= l3) { best = v1; len = l1; }
  else if (l2 >= l1 && l2 >= l3) { best = v2; len = l2; }
  else { best = v3; len = l3; }
  const float sl = sqrtf(len);
  v[0] = best[0] / sl; v[1] = best[1] / sl; v[2] = best[2] / sl;
}

// Sort the first cnt words of a query's LDS list ascending (words = [16-bit key | 16-bit tag]); N = list capacity.  Returns false
// if neighbours with equal keys could not be put into exact order within three passes (the caller then runs its exact LDS sort;
// the list is left key-sorted).  pos_of(word) -> position, dist_of(position) -> f32 squared distance.
template <int N, class PosOf, class DistOf>
__device__ __forceinline__ bool knn_sort_regs(unsigned (&a)[N], int cnt, PosOf pos_of, DistOf dist_of, const float4* __restrict__ P4) {
  // Batcher's merge exchange for arbitrary N (Knuth 5.2.2 M): all indices are compile-time constants after unrolling
#pragma unroll
  for (int p = 1; p < N; p <<= 1)
#pragma unroll
    for (int k = p; k >= 1; k >>= 1)
#pragma unroll
      for (int j = k % p; j <= N - 1 - k; j += 2 * k)
#pragma unroll
        for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i)
          if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
            const unsigned lo = min(a[i + j], a[i + j + k]), hi = max(a[i + j], a[i + j + k]);
            a[i + j] = lo; a[i + j + k] = hi;
          }
  bool dirty = true;
#pragma unroll 1
  for (int pass = 0; pass < 4 && dirty; ++pass) {
    dirty = false;
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) {
      if (i + 1 < cnt && ((a[i] ^ a[i + 1]) >> 16) == 0u) {           // equal keys: the exact order decides (rare)
        const unsigned p0 = pos_of(a[i]), p1 = pos_of(a[i + 1]);
        if (knn_less(dist_of(p1), p1, dist_of(p0), p0, P4)) { const unsigned t = a[i]; a[i] = a[i + 1]; a[i + 1] = t; dirty = true; }
      }
    }
  }
  return !dirty;
}

template <int N, class PosOf, class DistOf>
__device__ __forceinline__ bool knn_sort_words(unsigned* __restrict__ hp, int tid, int cnt, PosOf pos_of, DistOf dist_of,
                                               const float4* __restrict__ P4) {
  unsigned a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = (i < cnt) ? hp[(size_t)i * kKnnBlock + tid] : 0xFFFFFFFFu;
  const bool settled = knn_sort_regs<N>(a, cnt, pos_of, dist_of, P4);
#pragma unroll
  for (int i = 0; i < N; ++i) if (i < cnt) hp[(size_t)i * kKnnBlock + tid] = a[i];
  return settled;
}

// The single-pass variant with 16-bit list entries (k_knn_normals<5>): LDS holds the candidates' [row | offset] tags only, two per
// dword; here the tags come back, each candidate is fetched again for its distance key, the [key | tag] words are sorted in
// registers and the first k of them leave as positions (dword i of the list = neighbour i).  rs = the nine row starts in LDS.
// Returns false if equal keys could not be ordered exactly (the caller hands the query to the two-pass variant).
constexpr int kKnnTagSlots = 64;
template <class DistOf>
__device__ __forceinline__ bool knn_sort_tags(unsigned* __restrict__ list, const unsigned* __restrict__ rs, unsigned rs_centre, int tid, int cnt, int k,
                                              const knn_f2 qxy, float qz, float key_scale, DistOf dist_of, const float4* __restrict__ P4) {
  constexpr int N = kKnnTagSlots;
  unsigned a[N];
  // (eight of the nine row starts in LDS, the centre row's in a register: 8 + N / 2 = 40 dwords per lane are 20 KB per block, eight
  // blocks -- four waves per SIMD -- where nine would leave seven)
  auto pos_of = [&](unsigned w) {
    const unsigned row = (w >> 12) & 15u;
    const unsigned from_lds = rs[(size_t)(row - (row > 4u ? 1u : 0u)) * kKnnBlock + tid];
    return (row == 4u ? rs_centre : from_lds) + (w & 0xFFFu);
  };
#pragma unroll
  for (int i0 = 0; i0 < N; i0 += 8) {
    if (i0 < cnt) {
      unsigned tg[8]; float4 c[8];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const unsigned w2 = list[(size_t)((i0 + j) >> 1) * kKnnBlock + tid];
        tg[j] = w2 & 0xFFFFu; tg[j + 1] = w2 >> 16;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) if (i0 + j >= cnt) tg[j] = tg[0];          // (slots past the count hold nothing: any valid tag)
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = P4[pos_of(tg[j])];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d2 = knn_sqdist(qxy, qz, c[j]);
        a[i0 + j] = (i0 + j < cnt) ? (((unsigned)(d2 * key_scale) << 16) | tg[j]) : 0xFFFFFFFFu;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[i0 + j] = 0xFFFFFFFFu;
    }
  }
  const bool settled = knn_sort_regs<N>(a, cnt, pos_of, dist_of, P4);
#pragma unroll
  for (int i = 0; i < N / 2; ++i) if (i < k) list[(size_t)i * kKnnBlock + tid] = pos_of(a[i]);      // (k <= N / 2: the list's dwords)
  return settled;
}

// One pass over the queries listed in `todo` (or all points when todo == nullptr) on one grid level.
// kSel selects how the k best so far are kept in LDS -- all three keep the same set, so results are equal:
//   0  a max-heap (k > 32);
//   1  an unsorted list with the worst entry cached in registers (k <= 16: the re-scan after a replacement is k independent LDS
//      reads, cheaper than a sift-down's dependent chain while k is small -- 20 M points: k = 8 20.7 vs 23.1 ms, k = 32 121 vs 83 ms);
//   2  the unsorted list in groups of eight with each group's worst entry in registers (16 < k <= 32): a replacement re-scans
//      one group (8 independent reads) and picks the worst of at most four group maxima.
//   3  (k <= 60, dense cell directory) TWO PASSES without any per-candidate list maintenance -- with 64 lanes some lane replaces
//      an entry at almost every candidate, so the variants above walk their replacement path for nearly every candidate
//      (rocprofv3: ~130 instructions per candidate).  Pass A only counts (its own kernel, k_knn_hist: 64 B of LDS per query,
//      full occupancy): a 64-bin histogram of the squared distances (bin width = cell^2 / 32, one byte per bin, one ds_add per
//      candidate) gives the first bin b at which the count reaches k.  Pass B appends the candidates of bins <= b (k plus a
//      handful) to the query's LDS list as one word each -- [16-bit distance key | row | offset in the row] -- and skips the cells
//      farther away than that distance.  The collected set contains the k nearest: it holds EVERY candidate up to a distance with
//      at least k of them.  The words are sorted by a sorting network in registers (knn_sort_words), equal keys by the recomputed
//      f32 distances and the original indices, so the list comes out in (distance, original index) order like the other
//      variants'; one directory word per slot then turns [row | offset] into positions.  4 B of LDS per list slot: 16 waves per CU
//      at k = 32.
//   4  (k <= 36, dense directory, level 0, all queries) ONE PASS: pass A of variant 3 costs as much as the collection itself, and
//      it is run for every query to learn one number -- the distance that holds k candidates -- that neighbouring queries
//      share up to Poisson noise.  Here k_knn_hist looks at every rep_stride-th query of the cell order only and finds the
//      distance that holds T candidates (T between k and the list's capacity); every query collects the candidates closer than
//      the mean of its rep_avg nearest samples' distances.  The collected set is exact for ANY threshold -- it holds every
//      candidate up to that distance -- so the sample decides only how many queries land inside [k, capacity]; the others
//      (a few per cent) run through variant 3 afterwards.  The sort is the 64-word network, only the first k words are kept.
//   Variant 3: whatever does not fit (more than `cap` candidates in those bins: duplicates, dense clusters; no bin reaches k
//      within 2 cell^2; a row of 4096+ points) goes to `fb_todo`, which the host runs through the list-maintaining variant on the
//      same grid: results are exact either way.  (20 M points, k = 32: 64 ms with variant 2 -> 11.6 ms; k = 8: 18 -> 8.5 ms.)
// reach: 1 = the 27 cells around the query's cell; 2 = the 125 cells (list-maintaining variants only), the retry pass for the
// ~1 % of queries whose k-th neighbour lies outside the 27 cells, instead of a second grid of twice the cell size.
template <int kSel>
__global__ __launch_bounds__(kKnnBlock) void k_knn_normals(const float4* __restrict__ P4, size_t n,
                                                           const unsigned* __restrict__ todo, size_t n_todo,
                                                           const HashEntry* __restrict__ table, KnnGrid G, int k, int cap,
                                                           float vpx, float vpy, float vpz,
                                                           const float4* __restrict__ Q4 /* queries by sorted pos of level 0 */,
                                                           float* __restrict__ out_n, float* __restrict__ out_c,
                                                           int* __restrict__ out_knn, float* __restrict__ out_mean,
                                                           unsigned* __restrict__ next_todo,
                                                           unsigned* __restrict__ next_count,
                                                           unsigned* __restrict__ fb_todo, unsigned* __restrict__ fb_count,
                                                           const unsigned char* __restrict__ sel_bins,
                                                           int reach, int rep_stride, int rep_avg, float tau_ratio,
                                                           unsigned* __restrict__ seed_pos, unsigned char* __restrict__ seed_flag, unsigned seed_cap) {
  extern __shared__ unsigned char smem[];
  // variants 0..2: [cap = k distances][cap positions]; variant 3: [cap words: key | row | offset, later the positions]
  constexpr int kOffWords = 0;
  float* hd = reinterpret_cast<float*>(smem) + (size_t)kOffWords * kKnnBlock;
  // variant 5: [8 row starts (the centre row's stays in a register)][cap / 2 dwords: two 16-bit tags each, later the k positions]
  unsigned* hp = reinterpret_cast<unsigned*>(smem) + (size_t)(kOffWords + (kSel == 5 ? 8 : (kSel >= 3 ? 0 : cap))) * kKnnBlock;
  unsigned* rs_lds = reinterpret_cast<unsigned*>(smem);
  const int tid = threadIdx.x;
  size_t gi = (size_t)(kSel >= 3 ? knn_block(G, blockIdx.x, gridDim.x) : blockIdx.x) * blockDim.x + tid;
  if constexpr (kSel < 3) {
    // The 125-cell pass (reach 2) takes what a level left over: a few ten thousand unrelated queries, one wave per SIMD at
    // best, and a wave walks the UNION of the cells its lanes need, one memory round trip after the other -- its duration is
    // that chain's, not the list's.  With every rep_stride-th lane holding a query (the others leave) a wave's union is
    // smaller and there are rep_stride times as many waves to overlap the round trips.
    if (rep_stride > 1) { if (gi % (size_t)rep_stride) return; gi /= (size_t)rep_stride; }
  }
  if (gi >= n_todo) return;
  const unsigned qid = todo ? todo[gi] : (unsigned)gi;              // index into Q4
  const float4 q = Q4[qid];
  const unsigned q_oi = __float_as_uint(q.w);
  KP_DECL(12); KP_MARK(0);
#define HD(i) hd[(size_t)(i) * kKnnBlock + tid]
#define HP(i) hp[(size_t)(i) * kKnnBlock + tid]
  // !kHeap: the k best so far live UNSORTED in LDS; the worst of them (by (distance, original index), the order of the result
  // list) is cached in registers with its slot.  A candidate costs one compare against that register; one that enters overwrites
  // the worst slot and re-scans the k distances for the new worst.
  constexpr bool kHeap = kSel == 0;
  int cnt = 0;
  float td = 0.f; unsigned tp = 0; int tpos = 0;
  float gd[4] = {0.f, 0.f, 0.f, 0.f};            // kSel == 2: worst entry of each group of eight slots, and its slot
  int gs[4] = {0, 0, 0, 0};
  auto scan_group = [&](int g, float& od, int& oi) {
    const int b = 8 * g, e = (k < b + 8) ? k : b + 8;
    float bd = HD(b); int bi = b;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      const int i = b + j;
      if (i < e) {
        const float x = HD(i);
        if (x > bd) { bd = x; bi = i; }
        else if (x == bd && knn_less(bd, HP(bi), x, HP(i), P4)) { bi = i; }
      }
    }
    od = bd; oi = bi;
  };
  auto pick_group = [&]() {
    float bd = gd[0]; int bi = gs[0];
#pragma unroll
    for (int g = 1; g < 4; ++g)
      if (8 * g < k) {
        if (gd[g] > bd) { bd = gd[g]; bi = gs[g]; }
        else if (gd[g] == bd && knn_less(bd, HP(bi), gd[g], HP(gs[g]), P4)) { bi = gs[g]; }
      }
    td = bd; tpos = bi; tp = HP(bi);
  };
  auto find_worst = [&]() {
    if constexpr (kSel == 2) {
      if (cnt == k && tp == 0xFFFFFFFFu) {          // first call, right after the list filled: every group
#pragma unroll
        for (int g = 0; g < 4; ++g) if (8 * g < k) scan_group(g, gd[g], gs[g]);
      } else {                                     // after a replacement: the group of the replaced slot
        const int g = tpos >> 3;
        float nd; int ni;
        scan_group(g, nd, ni);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) if (gg == g) { gd[gg] = nd; gs[gg] = ni; }
      }
      pick_group();
      return;
    }
    float bd = HD(0); int bi = 0;
#pragma unroll 8
    for (int i = 1; i < k; ++i) {
      const float x = HD(i);
      if (x > bd) { bd = x; bi = i; }
      else if (x == bd && knn_less(bd, HP(bi), x, HP(i), P4)) { bi = i; }
    }
    td = bd; tpos = bi; tp = HP(bi);
  };
  auto consider = [&](unsigned m, const float4 c) {
    const float d2 = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z);
    if constexpr (kHeap) {
      if (cnt < k) {
        int i = cnt++;                                                // sift up
        HD(i) = d2; HP(i) = m;
        while (i > 0) {
          const int p = (i - 1) >> 1;
          const float pd = HD(p); const unsigned pp = HP(p);
          if (knn_less(pd, pp, d2, m, P4)) { HD(i) = pd; HP(i) = pp; HD(p) = d2; HP(p) = m; i = p; } else break;
        }
        return;
      }
      const float rd = HD(0); const unsigned rp = HP(0);
      if (d2 > rd) return;                                            // common case: one compare
      if (!knn_less(d2, m, rd, rp, P4)) return;
      int i = 0;                                                      // replace the root, sift down
      for (;;) {
        const int l = 2 * i + 1, r = l + 1;
        int big = -1; float bd = d2; unsigned bp = m;
        if (l < k) { const float ld = HD(l); const unsigned lp = HP(l); if (knn_less(bd, bp, ld, lp, P4)) { big = l; bd = ld; bp = lp; } }
        if (r < k) { const float xd = HD(r); const unsigned xp = HP(r); if (knn_less(bd, bp, xd, xp, P4)) { big = r; bd = xd; bp = xp; } }
        if (big < 0) break;
        HD(i) = bd; HP(i) = bp; i = big;
      }
      HD(i) = d2; HP(i) = m;
      return;
    }
    if (cnt < k) {
      HD(cnt) = d2; HP(cnt) = m;
      if (++cnt == k) { tp = 0xFFFFFFFFu; find_worst(); }
      return;
    }
    if (d2 > td) return;                                            // common case: one compare
    if (!knn_less(d2, m, td, tp, P4)) return;
    HD(tpos) = d2; HP(tpos) = m;
    find_worst();
  };
  const int cx = cell_coord(q.x, G.g.origin[0], G.g.inv_cell);
  const int cy = cell_coord(q.y, G.g.origin[1], G.g.inv_cell);
  const int cz = cell_coord(q.z, G.g.origin[2], G.g.inv_cell);
  constexpr int kMaxC = (1 << 21) - 1;
  // squared distance from the query to the nearest face of neighbour cell (ox, oy, oz), minus the rounding slack of cell_coord
  auto face2 = [&](int ox, int oy, int oz) -> float {
    const float fx = ox < 0 ? q.x - (G.g.origin[0] + (float)cx * G.cell) : (ox > 0 ? (G.g.origin[0] + (float)(cx + 1) * G.cell) - q.x : 0.f);
    const float fy = oy < 0 ? q.y - (G.g.origin[1] + (float)cy * G.cell) : (oy > 0 ? (G.g.origin[1] + (float)(cy + 1) * G.cell) - q.y : 0.f);
    const float fz = oz < 0 ? q.z - (G.g.origin[2] + (float)cz * G.cell) : (oz > 0 ? (G.g.origin[2] + (float)(cz + 1) * G.cell) - q.z : 0.f);
    const float ax = fmaxf(fx, 0.f), ay = fmaxf(fy, 0.f), az = fmaxf(fz, 0.f);
    const float face = sqrtf(ax * ax + ay * ay + az * az) - G.slack;
    return face > 0.f ? face * face * 0.99999f : 0.f;
  };
  bool fallback = false;
  if constexpr (kSel >= 3) {
    // ---- the candidate set: every point of the 27 cells closer than a threshold that holds at least k of them ----
    //   3: the threshold is the upper edge of the bin pass A (k_knn_hist) selected for THIS query: k .. k + 4 candidates;
    //   4: the threshold comes from the histograms of a SAMPLE of the queries (every rep_stride-th in cell order, k_knn_hist with
    //      stride; the mean of rep_avg of them): it aims at a count between k and the list's capacity.  A count outside that
    //      window sends the query to the two-pass variant (fb_todo) -- any threshold gives the exact set, the sample only decides
    //      how often it fits.
    float tau2 = 0.f;             // squared threshold
    float key_scale = 0.f;        // 16-bit key = (unsigned)(d2 * key_scale), monotone in d2
    int sel_bin = 0;
    float inv_w2 = 0.f;
    if constexpr (kSel == 3) {
      const int sel_byte = (int)sel_bins[gi];                         // pass A (k_knn_hist): [bin scale | bin], or a verdict
      bool block_is_everything = true;                                // the 27 cells cover the whole cloud (tiny clouds, k > n)
      {
        const int c3[3] = {cx, cy, cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (G.g.origin[a] + (float)(c3[a] - 1) * G.cell > G.dmin[a]) block_is_everything = false;
          if (G.g.origin[a] + (float)(c3[a] + 2) * G.cell <= G.dmax[a]) block_is_everything = false;
        }
      }
      if (sel_byte == 255 && !block_is_everything) {
        // fewer than k points within the histogram's range (sqrt(2) cells at the least): the k-th neighbour cannot be certified
        // inside this block of 27 cells, so the query goes to the next level as it is -- not through the list-maintaining variant
        // first (2 M queries of a scanner-sampled scan took that detour)
        const unsigned slot = atomicAdd(next_count, 1u);
        next_todo[slot] = qid;
        return;
      }
      sel_bin = (sel_byte == kKnnSelFallback || sel_byte == 255) ? kKnnBins : (sel_byte & 63);
      inv_w2 = (float)(32 << ((sel_byte >> 6) & 3)) * G.g.inv_cell * G.g.inv_cell;       // bins of cell^2 / (32 f): [0, 2 cell^2 / f)
      if (sel_bin >= kKnnBins) fallback = true;                       // the k-th neighbour lies beyond the histogram's range
      tau2 = (float)(sel_bin + 1) * (1.0f / inv_w2) * 1.00001f;        // (only the cells' face test uses it)
    } else {
      const size_t n_reps = (n_todo + (size_t)rep_stride - 1) / (size_t)rep_stride;
      const size_t r0 = (gi / (size_t)rep_stride) & ~(size_t)(rep_avg - 1);
      float edge_sum = 0.f; int have = 0;
      for (int i = 0; i < rep_avg; ++i)
        if (r0 + (size_t)i < n_reps) {
          const int bsel = (int)sel_bins[r0 + (size_t)i];
          if (bsel < kKnnSelFallback) { edge_sum += (float)((bsel & 63) + 1) / (float)(32 << ((bsel >> 6) & 3)); ++have; }
        }
      if (have == 0) fallback = true;                                 // no sampled query near by found its count inside the block
      else {
        tau2 = edge_sum / (float)have * G.cell * G.cell * tau_ratio;
        key_scale = 65000.0f / tau2;
      }
    }
    if (!fallback) {
      {
        // every candidate below the threshold is appended to the query's LDS list in scan order; the sorting network below
        // does not care about the order (a bucket placement by bin pairs with LDS fetch-and-adds was measured: the 32 B of slot
        // offsets per query cost more occupancy than the pre-sorting saved)
        int placed = 0;
        bool long_row = false;
        KP_MARK(1);
        // A collected candidate is stored as ONE word: [16-bit distance key | 4-bit row | 12-bit offset], row = 3 (oz + 1) + (oy + 1),
        // offset = its position relative to the start of cell (cx - 1, cy + oy, cz + oz) in the dense directory.  Positions are
        // recovered after the sort from the nine row starts -- 4 instead of 6 bytes of LDS per slot (14 instead of 10 waves
        // per CU for this latency-bound scan).  A 3-cell row of 4096 points or more: the list-maintaining variant takes the query.
        // (These variants run with the dense directory only.)
        unsigned rstart[9], rfirst[9], rend[9];
        const knn_f2 qxy = {q.x, q.y};
        auto place4 = [&](unsigned tag0, const float4* cb, unsigned nvalid) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d2 = knn_sqdist(qxy, q.z, cb[j]);
            bool take; unsigned key;
            if constexpr (kSel == 3) {
              const float bb = d2 * inv_w2;
              take = bb < (float)kKnnBins && (int)bb <= sel_bin;
              key = (unsigned)(bb * 1024.0f);
            } else if constexpr (kSel == 4) {
              take = d2 < tau2;
              key = (unsigned)(d2 * key_scale);
            } else {
              take = d2 < tau2;
              key = 0u;                                               // (variant 5 computes the keys after the collection)
            }
            if ((unsigned)j < nvalid && take) {
              if constexpr (kSel == 5) {
                if (placed < cap)
                  reinterpret_cast<unsigned short*>(hp)[((size_t)(placed >> 1) * kKnnBlock + tid) * 2 + (size_t)(placed & 1)] = (unsigned short)(tag0 + (unsigned)j);
              } else {
                if (placed < cap) HP(placed) = (key << 16) | (tag0 + (unsigned)j);
              }
              ++placed;
            }
          }
        };
        // the nine rows' directory words first (27 independent loads); a row the threshold cannot reach, an empty one or a long
        // one has first == end
#pragma unroll
        for (int oz = -1; oz <= 1; ++oz)
#pragma unroll
          for (int oy = -1; oy <= 1; ++oy) {
            const int r = 3 * (oz + 1) + (oy + 1);
            rstart[r] = 0u; rfirst[r] = 0u; rend[r] = 0u;
            if ((oy | oz) != 0 && face2(0, oy, oz) > tau2) continue;
            const int y = cy + oy, z = cz + oz;
            if (y < 0 || z < 0 || y >= (int)G.D[1] || z >= (int)G.D[2]) continue;
            const int xlo = max((face2(-1, oy, oz) > tau2) ? cx : cx - 1, 0), xhi = min((face2(1, oy, oz) > tau2) ? cx : cx + 1, (int)G.D[0] - 1);
            const size_t row = ((size_t)z * G.D[1] + (size_t)y) * G.D[0];
            const unsigned r0 = G.S[row + (size_t)max(cx - 1, 0)];
            const unsigned m0 = G.S[row + (size_t)xlo], e = G.S[row + (size_t)xhi + 1];
            rstart[r] = r0;
            if (e - r0 > 4096u) { long_row = true; continue; }
            rfirst[r] = m0; rend[r] = e;
          }
        // where the stream goes on after row r: the first candidate of the next row that has any (its batch is requested while
        // row r's last one is evaluated)
        unsigned rnext[9];
        {
          unsigned nx = 0u;
#pragma unroll
          for (int r = 8; r >= 0; --r) { rnext[r] = nx; if (rfirst[r] < rend[r]) nx = rfirst[r]; }
          // (nx: the first candidate of all)
          // Candidates stream through two half batches of four that are refilled as soon as they are evaluated: four to eight
          // loads stay in flight across batch and row boundaries (this kernel runs at two to three waves per SIMD -- its LDS list
          // -- so the latency has to be covered inside the wave).  A half batch may read past its row's end (P4 is padded by
          // eight); those entries are masked.
          float4 ha[4], hb[4];
          {
            const float4* __restrict__ pm = P4 + nx;
#pragma unroll
            for (int j = 0; j < 4; ++j) ha[j] = pm[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) hb[j] = pm[4 + j];
          }
#pragma unroll
          for (int r = 0; r < 9; ++r) {
            const unsigned e = rend[r], r0 = rstart[r], rtag = (unsigned)r << 12;
            for (unsigned m = rfirst[r]; m < e;) {
              const unsigned nm = m + 8u;
              const float4* __restrict__ pn = P4 + (nm < e ? nm : rnext[r]);
              const unsigned nvalid = e - m;
              place4(rtag | (m - r0), ha, nvalid);
#pragma unroll
              for (int j = 0; j < 4; ++j) ha[j] = pn[j];
              place4(rtag | (m + 4u - r0), hb, nvalid > 4u ? nvalid - 4u : 0u);
#pragma unroll
              for (int j = 0; j < 4; ++j) hb[j] = pn[4 + j];
              m = nm;
            }
          }
        }
        KP_MARK(2);
        // more candidates than list slots (duplicates, a dense cluster in one bin), or a long row: the other variant takes over
        if (placed > cap || placed < k || long_row) {
          fallback = true;
        } else {
          cnt = placed;
          auto row_start = [&](int r) -> unsigned {
            const int oy = r % 3 - 1, oz = r / 3 - 1;
            const size_t row = ((size_t)(cz + oz) * G.D[1] + (size_t)(cy + oy)) * G.D[0];
            return G.S[row + (size_t)max(cx - 1, 0)];
          };
          auto pos_of = [&](unsigned w) { return row_start((int)((w >> 12) & 15u)) + (w & 0xFFFu); };
          auto dist_of = [&](unsigned m) { const float4 c = P4[m]; return sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z); };
          if constexpr (kSel == 5) {
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r != 4) rs_lds[(size_t)(r - (r > 4 ? 1 : 0)) * kKnnBlock + tid] = rstart[r];
            const bool settled = knn_sort_tags(hp, rs_lds, rstart[4], tid, cnt, k, qxy, q.z, key_scale, dist_of, P4);
            KP_MARK(3);
            if (!settled) fallback = true;              // (a long run of equal keys: the two-pass variant has room to sort it in LDS)
            cnt = k;
          } else {
          // exact (d2, original index) order.  The words go through a sorting network in registers (Batcher's merge exchange, no
          // LDS traffic and no divergence), then neighbours with EQUAL 16-bit keys are put into exact order by up to three bubble
          // passes over the recomputed f32 distances and the original indices; a longer run of equal keys (lattices, duplicates)
          // falls back to the exact insertion sort in LDS.  Variant 4 keeps the first k + 1 .. of the sorted list only.
          bool settled;
          if constexpr (kSel == 4) {
            // (variant 4 serves k <= 10 with at most 36 slots: without the larger networks the kernel needs fewer registers)
            if (cap <= 20) settled = knn_sort_words<20>(hp, tid, cnt, pos_of, dist_of, P4);
            else if (cap <= 28) settled = knn_sort_words<28>(hp, tid, cnt, pos_of, dist_of, P4);
            else settled = knn_sort_words<36>(hp, tid, cnt, pos_of, dist_of, P4);
          } else {
            if (cap <= 12) settled = knn_sort_words<12>(hp, tid, cnt, pos_of, dist_of, P4);
            else if (cap <= 20) settled = knn_sort_words<20>(hp, tid, cnt, pos_of, dist_of, P4);
            else if (cap <= 36) settled = knn_sort_words<36>(hp, tid, cnt, pos_of, dist_of, P4);
            else settled = knn_sort_words<64>(hp, tid, cnt, pos_of, dist_of, P4);
          }
          if (!settled) {
            for (int i = 1; i < cnt; ++i) {
              const unsigned kw = HP(i), kk = kw >> 16;
              unsigned kp = 0; float kd = -1.f;
              int j = i - 1;
              while (j >= 0) {
                const unsigned jw = HP(j), jk = jw >> 16;
                if (jk < kk) break;
                if (jk == kk) {
                  if (kd < 0.f) { kp = pos_of(kw); kd = dist_of(kp); }
                  const unsigned jp = pos_of(jw);
                  if (!knn_less(kd, kp, dist_of(jp), jp, P4)) break;
                }
                HP(j + 1) = jw;
                --j;
              }
              HP(j + 1) = kw;
            }
          }
          KP_MARK(3);
          // [row | offset] -> positions.  Only the first k entries are neighbours (variant 4 collects up to the capacity); the
          // nine row starts go through the LDS slots behind them when there are nine free ones (dynamic index), else through
          // the directory again.
          if (cnt > k) cnt = k;
          if (k + 9 <= cap) {
#pragma unroll
            for (int r = 0; r < 9; ++r) HP(k + r) = rstart[r];
            for (int i0 = 0; i0 < cnt; i0 += 8) {
              unsigned w[8], st[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) { w[j] = HP(min(i0 + j, cnt - 1)); st[j] = HP(k + (int)((w[j] >> 12) & 15u)); }
#pragma unroll
              for (int j = 0; j < 8; ++j) if (i0 + j < cnt) HP(i0 + j) = st[j] + (w[j] & 0xFFFu);
            }
          } else {
            for (int i0 = 0; i0 < cnt; i0 += 8) {
              unsigned w[8], st[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) { w[j] = HP(min(i0 + j, cnt - 1)); st[j] = row_start((int)((w[j] >> 12) & 15u)); }
#pragma unroll
              for (int j = 0; j < 8; ++j) if (i0 + j < cnt) HP(i0 + j) = st[j] + (w[j] & 0xFFFu);
            }
          }
          }
          if (!fallback) td = dist_of(HP(cnt - 1));                             // the k-th nearest (>= it for a cloud of fewer than k points)
          KP_MARK(4);
        }
      }
    }
  } else {
  // own cell first, then face, edge and corner neighbours: the list fills with near points early, so fewer of the later candidates
    // replace an entry (the result does not depend on the order)
    // reach 2 (the retry pass on the same grid): the 27 inner cells as above, then the shell of 98 cells around them.  A query that
    // arrives with the k nearest of its 27 cells (seed_pos: it came from the scan variants of this level) starts from those.
    bool seeded = false;
    if constexpr (kSel < 3) {
      if (reach == 2 && seed_flag && gi < (size_t)seed_cap && seed_flag[gi]) {
        seeded = true;
        for (int i = 0; i < k; ++i) { const unsigned m = seed_pos[gi * (size_t)k + (size_t)i]; consider(m, P4[m]); }
      }
    }
    // (rings 0 .. 3: the 27 inner cells; ring 4: the shell.  Measured for the 125-cell pass, whose 64 lanes are unrelated queries:
    // every lane walking its OWN list of the shell cells it needs (bit mask from the face tests) instead of the wave walking the
    // union in step -- 1.5 instead of 0.87 ms: the longest list of 64 unrelated queries is nearly the union, and its visits no
    // longer coalesce.  The seeds save the 27 inner cells' round trips only.)
    for (int ring = seeded ? 4 : 0; ring <= 2 + reach; ++ring)
    for (int oz = -reach; oz <= reach; ++oz)
      for (int oy = -reach; oy <= reach; ++oy)
        for (int ox = -reach; ox <= reach; ++ox) {
          const int cheb = max(max(abs(ox), abs(oy)), abs(oz));
          if ((cheb <= 1 ? (ox != 0) + (oy != 0) + (oz != 0) : 2 + cheb) != ring) continue;
          const int x = cx + ox, y = cy + oy, z = cz + oz;
          if (x < 0 || y < 0 || z < 0 || x > kMaxC || y > kMaxC || z > kMaxC) continue;
          if (ring > 0 && cnt == k) {
            // the list is full: a neighbour cell whose nearest face is farther than the current worst entry cannot contribute
            // (every point of the cell is at least that far; `slack` covers the rounding of cell_coord as in the test below).
            // Queries of a wave share their cell, so whole waves skip the same cells.
            const float fx = ox < 0 ? q.x - (G.g.origin[0] + (float)(cx + ox + 1) * G.cell) : (ox > 0 ? (G.g.origin[0] + (float)(cx + ox) * G.cell) - q.x : 0.f);
            const float fy = oy < 0 ? q.y - (G.g.origin[1] + (float)(cy + oy + 1) * G.cell) : (oy > 0 ? (G.g.origin[1] + (float)(cy + oy) * G.cell) - q.y : 0.f);
            const float fz = oz < 0 ? q.z - (G.g.origin[2] + (float)(cz + oz + 1) * G.cell) : (oz > 0 ? (G.g.origin[2] + (float)(cz + oz) * G.cell) - q.z : 0.f);
            const float ax = fmaxf(fx, 0.f), ay = fmaxf(fy, 0.f), az = fmaxf(fz, 0.f);
            const float face = sqrtf(ax * ax + ay * ay + az * az) - G.slack;
            if (face > 0.f && face * face * 0.99999f > (kHeap ? HD(0) : td)) continue;
          }
          knn_row(G, table, x, x, y, z, [&](unsigned m, unsigned e) {     // dense directory: two words; hash table otherwise
            for (; m + 4 <= e; m += 4) {                                  // four candidates in flight per lane
              const float4 c0 = P4[m], c1 = P4[m + 1], c2 = P4[m + 2], c3 = P4[m + 3];
              consider(m, c0); consider(m + 1, c1); consider(m + 2, c2); consider(m + 3, c3);
            }
            for (; m < e; ++m) consider(m, P4[m]);
          });
        }
}
  if (fallback) {
    const unsigned slot = atomicAdd(fb_count, 1u);
    fb_todo[slot] = qid;
    return;
  }
  // resolved?  the k-th neighbour must be strictly closer than the nearest face of the 27-cell block that still
  // has data behind it
  bool resolved;
  {
    float safe = 3.402823466e+38f;
    const int c[3] = {cx, cy, cz};
    const float qq[3] = {q.x, q.y, q.z};
    bool covers_all = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo = G.g.origin[a] + (float)(c[a] - reach) * G.cell;
      const float hi = G.g.origin[a] + (float)(c[a] + reach + 1) * G.cell;
      if (lo > G.dmin[a]) { safe = fminf(safe, qq[a] - lo); covers_all = false; }
      if (hi <= G.dmax[a]) { safe = fminf(safe, hi - qq[a]); covers_all = false; }
    }
    if (covers_all) resolved = true;
    else {
      safe -= G.slack;
      const float kth = kHeap ? HD(0) : td;      // variant 3: the farthest collected candidate (>= the k-th nearest)
      resolved = (cnt >= k) && safe > 0.f && kth < safe * safe * 0.99999f;
    }
  }
  if (!resolved) {
    const unsigned slot = atomicAdd(next_count, 1u);
    next_todo[slot] = qid;
    if constexpr (kSel >= 3) {
      // the k nearest of the 27 cells are known: the 125-cell pass that takes this query starts from them and looks at the shell
      // only (it is one wave deep -- its duration is the latency of the cells it visits one after the other)
      if (seed_pos && slot < seed_cap && cnt == k) {
        for (int i = 0; i < k; ++i) seed_pos[(size_t)slot * (size_t)k + (size_t)i] = HP(i);
        seed_flag[slot] = 1;
      }
    }
    return;
  }
  if constexpr (kSel >= 3) {
    // sorted already (sorting network + exact order of equal keys)
  } else {
  // heap sort in place -> ascending (d2, original index): Floyd's heap construction, then repeated extraction of the maximum
  for (int start = kHeap ? -1 : cnt / 2 - 1; start >= 0; --start) {     // (variant 3: cnt may exceed k here, all of it is sorted)
    const float ld = HD(start); const unsigned lp = HP(start);
    int i = start;
    for (;;) {
      const int l = 2 * i + 1, r = l + 1;
      int big = -1; float bd = ld; unsigned bp = lp;
      if (l < cnt) { const float xd = HD(l); const unsigned xp = HP(l); if (knn_less(bd, bp, xd, xp, P4)) { big = l; bd = xd; bp = xp; } }
      if (r < cnt) { const float xd = HD(r); const unsigned xp = HP(r); if (knn_less(bd, bp, xd, xp, P4)) { big = r; bd = xd; bp = xp; } }
      if (big < 0) break;
      HD(i) = bd; HP(i) = bp; i = big;
    }
    HD(i) = ld; HP(i) = lp;
  }
  for (int end = cnt - 1; end > 0; --end) {
    const float ld = HD(end); const unsigned lp = HP(end);
    HD(end) = HD(0); HP(end) = HP(0);
    int i = 0;
    for (;;) {
      const int l = 2 * i + 1, r = l + 1;
      int big = -1; float bd = ld; unsigned bp = lp;
      if (l < end) { const float xd = HD(l); const unsigned xp = HP(l); if (knn_less(bd, bp, xd, xp, P4)) { big = l; bd = xd; bp = xp; } }
      if (r < end) { const float xd = HD(r); const unsigned xp = HP(r); if (knn_less(bd, bp, xd, xp, P4)) { big = r; bd = xd; bp = xp; } }
      if (big < 0) break;
      HD(i) = bd; HP(i) = bp; i = big;
    }
    HD(i) = ld; HP(i) = lp;
  }
  }
  if (cnt > k) cnt = k;            // variant 3 collected a few more than k: the sorted list's first k are the neighbours
  if (out_knn) {
    for (int i = 0; i < k; ++i) out_knn[(size_t)q_oi * k + i] = (i < cnt) ? (int)__float_as_uint(P4[HP(i)].w) : -1;
  }
  if (out_mean) {
    // LocalStatisticalOutlierRemoval, first pass (local_statistical_outlier_removal.hpp:104-109): mean distance to the
    // k - 1 nearest neighbours (entry 0 is the query point), f64 sum of the f32 roots in neighbour order
    double dist_sum = 0.0;
    for (int i = 1; i < cnt; ++i) {
      float d2i;
      if constexpr (kSel >= 3) { const float4 c = P4[HP(i)]; d2i = sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z); }
      else d2i = HD(i);
      dist_sum += (double)sqrtf(d2i);
    }
    out_mean[q_oi] = (float)(dist_sum / (double)(k - 1));
  }
  if (!out_n) return;
  float nx, ny, nz, curv;
  const float qnan = __uint_as_float(0x7fc00000u);
  if (cnt < 3) {                                                      // two_pass_normal_3d.h:97-103
    nx = ny = nz = curv = qnan;
  } else {
    // two_pass_centroid.hpp:176-192 (dense branch), f32, neighbour order
    // the neighbours are gathered eight at a time (eight independent loads in flight); the sums keep the neighbour order
    float a6 = 0.f, a7 = 0.f, a8 = 0.f;
    KP_MARK(5);
    for (int i0 = 0; i0 < cnt; i0 += 8) {
      float4 pb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = P4[HP(min(i0 + j, cnt - 1))];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + j < cnt) { a6 += pb[j].x; a7 += pb[j].y; a8 += pb[j].z; }
    }
    const float fc = (float)cnt;
    a6 = a6 / fc; a7 = a7 / fc; a8 = a8 / fc;
    float a0 = 0.f / fc, a1 = 0.f / fc, a2 = 0.f / fc, a3 = 0.f / fc, a4 = 0.f / fc, a5 = 0.f / fc;
    for (int i0 = 0; i0 < cnt; i0 += 8) {
      float4 pb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = P4[HP(min(i0 + j, cnt - 1))];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + j < cnt) {
          const float4 p = pb[j];
          a0 += (p.x - a6) * (p.x - a6);
          a1 += (p.x - a6) * (p.y - a7);
          a2 += (p.x - a6) * (p.z - a8);
          a3 += (p.y - a7) * (p.y - a7);
          a4 += (p.y - a7) * (p.z - a8);
          a5 += (p.z - a8) * (p.z - a8);
        }
    }
    KP_MARK(6);
    float cov[9];
    cov[0] = a0 / fc; cov[1] = a1 / fc; cov[2] = a2 / fc; cov[4] = a3 / fc; cov[5] = a4 / fc; cov[8] = a5 / fc;
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, v[3];
    eigen33_smallest(cov, ev, v);
    nx = v[0]; ny = v[1]; nz = v[2];
    const float eig_sum = cov[0] + cov[4] + cov[8];
    curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
    // flipNormalTowardsViewpoint
    const float vx = vpx - q.x, vy = vpy - q.y, vz = vpz - q.z;
    const float cos_theta = (vx * nx + vy * ny + vz * nz);
    if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
    KP_MARK(7);
  }
  out_n[3 * (size_t)q_oi] = nx; out_n[3 * (size_t)q_oi + 1] = ny; out_n[3 * (size_t)q_oi + 2] = nz;
  out_c[q_oi] = curv;
  if constexpr (kSel >= 3) { KP_MARK(8); KP_FLUSH(16, 9); }
#undef HD
#undef HP
}

// The 125-cell pass, ONE WAVE PER QUERY (dense directory, k <= 64).  What a level leaves over is a few ten thousand unrelated
// queries; with a lane per query a wave walks the union of its lanes' cells one memory round trip after the other and there is
// one wave per SIMD at best (0.65 - 0.9 ms for 32 k queries).  Here the 64 lanes of a wave share ONE query:
//   * lanes 0 .. 24 fetch the directory words of the 25 rows (y, z) of the 5 x 5 x 5 block, a wave scan turns the row lengths into
//     offsets, and the candidates are dealt to the lanes round-robin (up to kWideSlots each, all loads of a lane in flight);
//   * the k-th smallest squared distance is found bit by bit (31 counting steps: v_cmp into a lane mask, s_bcnt1 -- the counting is
//     scalar work), everything up to it (ties included) is compacted into LDS, one entry per lane, and a 64-lane bitonic network
//     sorts by (distance, original index): the order of every other variant;
//   * lane i holds neighbour i: index list written as it lies; the mean distance, the two-pass covariance and the eigenvector are
//     summed by lane 0 in neighbour order from LDS -- the same f32 / f64 sums as k_knn_normals.
// More candidates than 64 * kWideSlots, or more ties at the k-th distance than lanes: the query goes to fb_todo (the lane-per-query
// kernel takes it).  Unresolved queries (k-th neighbour beyond the block) go to next_todo as before.
constexpr int kWideSlots = 16;
constexpr int kWideWaves = 4;                                          // queries per block
__global__ __launch_bounds__(64 * kWideWaves) void k_knn_wide_wave(const float4* __restrict__ P4, const unsigned* __restrict__ todo, size_t n_todo,
                                                                  KnnGrid G, int k, float vpx, float vpy, float vpz, const float4* __restrict__ Q4,
                                                                  float* __restrict__ out_n, float* __restrict__ out_c, int* __restrict__ out_knn,
                                                                  float* __restrict__ out_mean, unsigned* __restrict__ next_todo,
                                                                  unsigned* __restrict__ next_count, unsigned* __restrict__ fb_todo,
                                                                  unsigned* __restrict__ fb_count) {
  __shared__ unsigned s_pre[kWideWaves][28], s_start[kWideWaves][28];
  __shared__ unsigned s_u[kWideWaves][64], s_oi[kWideWaves][64], s_pos[kWideWaves][64];
  __shared__ float s_x[kWideWaves][64], s_y[kWideWaves][64], s_z[kWideWaves][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t gi = (size_t)blockIdx.x * kWideWaves + (size_t)wv;
  const bool active = gi < n_todo;                                     // (whole waves; every wave reaches the barriers)
  const unsigned qid = active ? todo[gi] : 0u;
  const float4 q = Q4[qid];
  const unsigned q_oi = __float_as_uint(q.w);
  const int cx = cell_coord(q.x, G.g.origin[0], G.g.inv_cell);
  const int cy = cell_coord(q.y, G.g.origin[1], G.g.inv_cell);
  const int cz = cell_coord(q.z, G.g.origin[2], G.g.inv_cell);
  // the 25 rows of the block
  unsigned rs = 0u, re = 0u;
  if (active && lane < 25) {
    const int y = cy + lane % 5 - 2, z = cz + lane / 5 - 2;
    const int x0 = max(cx - 2, 0), x1 = min(cx + 2, (int)G.D[0] - 1);
    if (y >= 0 && z >= 0 && y < (int)G.D[1] && z < (int)G.D[2] && x0 <= x1) {
      const size_t row = ((size_t)z * G.D[1] + (size_t)y) * G.D[0];
      rs = G.S[row + (size_t)x0]; re = G.S[row + (size_t)x1 + 1];
    }
  }
  const unsigned len = re - rs;
  unsigned incl = len;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  const unsigned T = __shfl(incl, 63, 64);
  if (lane < 25) { s_pre[wv][lane] = incl - len; s_start[wv][lane] = rs; }
  if (lane == 25) s_pre[wv][25] = T;
  __syncthreads();
  bool overflow = active && T > 64u * (unsigned)kWideSlots;
  // candidates, round-robin: candidate i of the block's sequence goes to lane i % 64
  unsigned u[kWideSlots], oi[kWideSlots], ps[kWideSlots];
  {
    unsigned p[kWideSlots];
#pragma unroll
    for (int j = 0; j < kWideSlots; ++j) {
      const unsigned i = (unsigned)lane + 64u * (unsigned)j;
      p[j] = 0xFFFFFFFFu;
      if (active && !overflow && i < T) {
        int lo = 0, hi = 24;                                         // the row with pre[row] <= i < pre[row + 1]
#pragma unroll
        for (int it = 0; it < 5; ++it) { const int mid = (lo + hi + 1) >> 1; if (s_pre[wv][mid] <= i) lo = mid; else hi = mid - 1; }
        p[j] = s_start[wv][lo] + (i - s_pre[wv][lo]);
      }
    }
    float4 c[kWideSlots];
#pragma unroll
    for (int j = 0; j < kWideSlots; ++j) c[j] = P4[p[j] != 0xFFFFFFFFu ? p[j] : 0u];
#pragma unroll
    for (int j = 0; j < kWideSlots; ++j) {
      const bool ok = p[j] != 0xFFFFFFFFu;
      const unsigned bits = __float_as_uint(sqdist_l2(q.x, q.y, q.z, c[j].x, c[j].y, c[j].z));
      u[j] = ok ? bits : 0xFFFFFFFFu;                                // (a NaN distance sorts behind every number, like the unused slots)
      oi[j] = __float_as_uint(c[j].w); ps[j] = p[j];
    }
  }
  const unsigned kk = min((unsigned)k, T);
  // the kk-th smallest distance, bit by bit: after the loop thr = that value
  unsigned thr = 0u;
  if (active && !overflow && kk > 0u) {
#pragma unroll 1
    for (int b = 30; b >= 0; --b) {
      const unsigned m = thr | ((1u << b) - 1u);
      unsigned c_le = 0u;
#pragma unroll
      for (int j = 0; j < kWideSlots; ++j) c_le += (unsigned)__popcll(__ballot(u[j] <= m));
      if (c_le < kk) thr |= 1u << b;
    }
  }
  unsigned n_sel = 0u;                                                 // everything up to thr, ties included
#pragma unroll
  for (int j = 0; j < kWideSlots; ++j) n_sel += (unsigned)__popcll(__ballot(active && !overflow && u[j] <= thr && u[j] != 0xFFFFFFFFu));
  if (active && !overflow && n_sel > 64u) overflow = true;            // more ties at the k-th distance than lanes
  if (overflow) {
    if (lane == 0) { const unsigned slot = atomicAdd(fb_count, 1u); fb_todo[slot] = qid; }
  }
  const bool work = active && !overflow;
  // compaction: lane l's selected slots go to [offset of l, ...) in (lane, slot) order -- any order, the network sorts
  {
    unsigned mine = 0u;
#pragma unroll
    for (int j = 0; j < kWideSlots; ++j) mine += (work && u[j] <= thr && u[j] != 0xFFFFFFFFu) ? 1u : 0u;
    unsigned inc2 = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc2, o, 64); if (lane >= o) inc2 += t; }
    unsigned at = inc2 - mine;
    s_u[wv][lane] = 0xFFFFFFFFu; s_oi[wv][lane] = 0xFFFFFFFFu; s_pos[wv][lane] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kWideSlots; ++j)
      if (work && u[j] <= thr && u[j] != 0xFFFFFFFFu) { s_u[wv][at] = u[j]; s_oi[wv][at] = oi[j]; s_pos[wv][at] = ps[j]; ++at; }
    __syncthreads();
  }
  // one entry per lane, bitonic network on (distance bits, original index)
  unsigned eu = s_u[wv][lane], eo = s_oi[wv][lane], ep = s_pos[wv][lane];
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      const unsigned pu = __shfl_xor(eu, stride, 64), po = __shfl_xor(eo, stride, 64), pp = __shfl_xor(ep, stride, 64);
      const bool up = (lane & size) == 0;                              // this block of `size` lanes sorts ascending
      const bool lower = (lane & stride) == 0;                         // this lane keeps the smaller of the pair when ascending
      const bool mine_less = eu < pu || (eu == pu && eo < po);
      const bool keep_mine = (lower == up) ? mine_less : !mine_less;
      if (!keep_mine) { eu = pu; eo = po; ep = pp; }
    }
  const int cnt = (int)kk;
  // resolved?  (as k_knn_normals with reach 2)
  bool resolved = true;
  if (work) {
    float safe = 3.402823466e+38f;
    const int c3[3] = {cx, cy, cz};
    const float qq[3] = {q.x, q.y, q.z};
    bool covers_all = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo = G.g.origin[a] + (float)(c3[a] - 2) * G.cell;
      const float hi = G.g.origin[a] + (float)(c3[a] + 3) * G.cell;
      if (lo > G.dmin[a]) { safe = fminf(safe, qq[a] - lo); covers_all = false; }
      if (hi <= G.dmax[a]) { safe = fminf(safe, hi - qq[a]); covers_all = false; }
    }
    if (!covers_all) {
      safe -= G.slack;
      const float kth = __uint_as_float(__shfl(eu, max(cnt - 1, 0), 64));
      resolved = (cnt >= k) && safe > 0.f && kth < safe * safe * 0.99999f;
    }
    if (!resolved && lane == 0) { const unsigned slot = atomicAdd(next_count, 1u); next_todo[slot] = qid; }
  }
  const bool emit = work && resolved;
  if (emit && out_knn && lane < k) out_knn[(size_t)q_oi * k + lane] = (lane < cnt) ? (int)eo : -1;
  // neighbour i's coordinates and distance to LDS; lane 0 sums in neighbour order
  {
    const float4 c = P4[(emit && lane < cnt) ? ep : 0u];
    s_x[wv][lane] = c.x; s_y[wv][lane] = c.y; s_z[wv][lane] = c.z;
    s_u[wv][lane] = eu;
  }
  __syncthreads();
  if (!emit || lane != 0) return;
  if (out_mean) {
    double dist_sum = 0.0;
    for (int i = 1; i < cnt; ++i) dist_sum += (double)sqrtf(__uint_as_float(s_u[wv][i]));
    out_mean[q_oi] = (float)(dist_sum / (double)(k - 1));
  }
  if (!out_n) return;
  float nx, ny, nz, curv;
  const float qnan = __uint_as_float(0x7fc00000u);
  if (cnt < 3) {
    nx = ny = nz = curv = qnan;
  } else {
    float a6 = 0.f, a7 = 0.f, a8 = 0.f;
    for (int i = 0; i < cnt; ++i) { a6 += s_x[wv][i]; a7 += s_y[wv][i]; a8 += s_z[wv][i]; }
    const float fc = (float)cnt;
    a6 = a6 / fc; a7 = a7 / fc; a8 = a8 / fc;
    float a0 = 0.f / fc, a1 = 0.f / fc, a2 = 0.f / fc, a3 = 0.f / fc, a4 = 0.f / fc, a5 = 0.f / fc;
    for (int i = 0; i < cnt; ++i) {
      const float px = s_x[wv][i], py = s_y[wv][i], pz = s_z[wv][i];
      a0 += (px - a6) * (px - a6);
      a1 += (px - a6) * (py - a7);
      a2 += (px - a6) * (pz - a8);
      a3 += (py - a7) * (py - a7);
      a4 += (py - a7) * (pz - a8);
      a5 += (pz - a8) * (pz - a8);
    }
    float cov[9];
    cov[0] = a0 / fc; cov[1] = a1 / fc; cov[2] = a2 / fc; cov[4] = a3 / fc; cov[5] = a4 / fc; cov[8] = a5 / fc;
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, v[3];
    eigen33_smallest(cov, ev, v);
    nx = v[0]; ny = v[1]; nz = v[2];
    const float eig_sum = cov[0] + cov[4] + cov[8];
    curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
    const float vx = vpx - q.x, vy = vpy - q.y, vz = vpz - q.z;
    const float cos_theta = (vx * nx + vy * ny + vz * nz);
    if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
  }
  out_n[3 * (size_t)q_oi] = nx; out_n[3 * (size_t)q_oi + 1] = ny; out_n[3 * (size_t)q_oi + 2] = nz;
  out_c[q_oi] = curv;
}

// Radius-search variant (pcl::Feature::searchForNeighbors with setRadiusSearch, two_pass_normal_3d_omp.hpp:66): every
// point within the radius (FLANN: squared distance strictly below (float)((double)r * r)) takes part; one thread per
// point walks its 27 grid cells twice -- centroid, then centred products -- so nothing is stored per neighbour and the
// neighbour count is unbounded.  The f32 sums run in grid order, not in the distance order PCL's sorted radius search
// returns, so results agree with the reference to summation round-off, not bit for bit.
__global__ __launch_bounds__(kKnnBlock) void k_radius_normals(const float4* __restrict__ P4, size_t n,
                                                              const HashEntry* __restrict__ table, KnnGrid G, float r2,
                                                              float vpx, float vpy, float vpz, float* __restrict__ out_n,
                                                              float* __restrict__ out_c, int* __restrict__ out_count) {
  const size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= n) return;
  const float4 q = P4[gi];
  const unsigned q_oi = __float_as_uint(q.w);
  const int cx = cell_coord(q.x, G.g.origin[0], G.g.inv_cell);
  const int cy = cell_coord(q.y, G.g.origin[1], G.g.inv_cell);
  const int cz = cell_coord(q.z, G.g.origin[2], G.g.inv_cell);
  constexpr int kMaxC = (1 << 21) - 1;
  unsigned rs[27], re[27];
  {
    int r = 0;
    for (int oz = -1; oz <= 1; ++oz)
      for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox, ++r) {
          rs[r] = 0; re[r] = 0;
          const int x = cx + ox, y = cy + oy, z = cz + oz;
          if (x < 0 || y < 0 || z < 0 || x > kMaxC || y > kMaxC || z > kMaxC) continue;
          const unsigned long long key = cell_key(x, y, z);
          unsigned h = hash_key(key) & G.g.mask;
          for (;;) {
            const HashEntry en = table[h];
            if (en.key == key) { rs[r] = en.start; re[r] = en.end; break; }
            if (en.key == kEmptyKey) break;
            h = (h + 1) & G.g.mask;
          }
        }
  }
  int cnt = 0;
  float a6 = 0.f, a7 = 0.f, a8 = 0.f;
  for (int r = 0; r < 27; ++r)
    for (unsigned m = rs[r]; m < re[r]; ++m) {
      const float4 c = P4[m];
      if (sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z) < r2) { ++cnt; a6 += c.x; a7 += c.y; a8 += c.z; }
    }
  float nx, ny, nz, curv;
  const float qnan = __uint_as_float(0x7fc00000u);
  if (cnt < 3) {                                                      // two_pass_normal_3d.h:97-103 (0 neighbours: NaN as well, :68-71)
    nx = ny = nz = curv = qnan;
  } else {
    const float fc = (float)cnt;
    a6 = a6 / fc; a7 = a7 / fc; a8 = a8 / fc;
    float a0 = 0.f / fc, a1 = 0.f / fc, a2 = 0.f / fc, a3 = 0.f / fc, a4 = 0.f / fc, a5 = 0.f / fc;
    for (int r = 0; r < 27; ++r)
      for (unsigned m = rs[r]; m < re[r]; ++m) {
        const float4 p = P4[m];
        if (!(sqdist_l2(q.x, q.y, q.z, p.x, p.y, p.z) < r2)) continue;
        a0 += (p.x - a6) * (p.x - a6);
        a1 += (p.x - a6) * (p.y - a7);
        a2 += (p.x - a6) * (p.z - a8);
        a3 += (p.y - a7) * (p.y - a7);
        a4 += (p.y - a7) * (p.z - a8);
        a5 += (p.z - a8) * (p.z - a8);
      }
    float cov[9];
    cov[0] = a0 / fc; cov[1] = a1 / fc; cov[2] = a2 / fc; cov[4] = a3 / fc; cov[5] = a4 / fc; cov[8] = a5 / fc;
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, v[3];
    eigen33_smallest(cov, ev, v);
    nx = v[0]; ny = v[1]; nz = v[2];
    const float eig_sum = cov[0] + cov[4] + cov[8];
    curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
    const float vx = vpx - q.x, vy = vpy - q.y, vz = vpz - q.z;      // flipNormalTowardsViewpoint
    const float cos_theta = (vx * nx + vy * ny + vz * nz);
    if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
  }
  out_n[3 * (size_t)q_oi] = nx; out_n[3 * (size_t)q_oi + 1] = ny; out_n[3 * (size_t)q_oi + 2] = nz;
  out_c[q_oi] = curv;
  if (out_count) out_count[q_oi] = cnt;
}

// grid keys for an arbitrary cell size (points taken from AoS xyz)
struct LevelBuffers {
  DevBuf<unsigned long long> ka, kb;
  DevBuf<unsigned> va, vb, counter;
  DevBuf<char> temp;
  DevBuf<float4> P4, LN;
  DevBuf<HashEntry> table;
  DevBuf<unsigned> dense;          // dense cell-start directory of the level (when the bounding grid is small enough)
  DevBuf<unsigned> coarse;         // its coarse form (one word per 4096 cells), from which the directory is written
  DevBuf<unsigned char> sel_bin;   // pass A results of the two-pass variant: the selected bin per query
};


// Device buffers of one kNN call.  A 20 M point call touches ~5 GB in ~20 buffers; allocating and freeing them costs several
// milliseconds per call (hipMalloc maps pages, hipFree synchronises the device), so finished calls park their workspace in a
// per-process pool and the next call on the same device takes it over.  e3d_release_workspaces() empties the pool; workspaces
// above E3D_WORKSPACE_KEEP_GB (default 32) are freed right away.
struct KnnWorkspace {
  int device = -1;
  hipStream_t stream = nullptr;
  DevBuf<float> raw, bbox_partial, bbox_out, d_on, d_oc, d_mean;
  DevBuf<int> d_knn;
  DevBuf<unsigned char> d_in;
  LevelBuffers L;
  DevBuf<float4> Q4;
  DevBuf<unsigned> todo_a, todo_b, fb_todo;
  DevBuf<unsigned> seed_pos;           // the k nearest of the 27 cells of queries that go on to the 125-cell pass (positions), and
  DevBuf<unsigned char> seed_flag;     // which entries of that pass's list have them
  PinBuf<unsigned> mailbox;            // the few words a call reads back (bounding box, list lengths): pinned, not the pageable staging path
  // a few words back from the device and the stream synchronised
  void read_back(void* dst, const void* src_dev, size_t bytes) {
    static const bool pinned = [] { const char* e = getenv("E3D_KNN_PINNED"); return !(e && e[0] == '0'); }();     // (0: as rounds 1 - 5, for A / B timing)
    if (!pinned) { E3D_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream)); E3D_HIP(hipStreamSynchronize(stream)); return; }
    mailbox.reserve(64);
    E3D_HIP(hipMemcpyAsync(mailbox.p, src_dev, bytes, hipMemcpyDeviceToHost, stream));
    E3D_HIP(hipStreamSynchronize(stream));
    memcpy(dst, mailbox.p, bytes);
  }
  size_t bytes() const {
    return raw.cap * 4 + bbox_partial.cap * 4 + bbox_out.cap * 4 + d_on.cap * 4 + d_oc.cap * 4 + d_mean.cap * 4 + d_knn.cap * 4 + d_in.cap +
           L.ka.cap * 8 + L.kb.cap * 8 + L.va.cap * 4 + L.vb.cap * 4 + L.counter.cap * 4 + L.temp.cap + L.P4.cap * 16 + L.LN.cap * 16 +
           L.table.cap * sizeof(HashEntry) + L.dense.cap * 4 + L.coarse.cap * 4 + L.sel_bin.cap + Q4.cap * 16 + todo_a.cap * 4 +
           todo_b.cap * 4 + fb_todo.cap * 4 + seed_pos.cap * 4 + seed_flag.cap;
  }
  ~KnnWorkspace() { if (stream) (void)hipStreamDestroy(stream); }
};

static std::mutex& workspace_mutex() { static std::mutex* m = new std::mutex; return *m; }
static std::vector<KnnWorkspace*>& workspace_pool() { static auto* v = new std::vector<KnnWorkspace*>; return *v; }   // never destroyed: no HIP calls at exit

struct WorkspaceLease {
  KnnWorkspace* ws = nullptr;
  WorkspaceLease() {
    int dev = 0;
    E3D_HIP(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(workspace_mutex());
      auto& pool = workspace_pool();
      for (size_t i = 0; i < pool.size(); ++i)
        if (pool[i]->device == dev) { ws = pool[i]; pool.erase(pool.begin() + (long)i); break; }
    }
    if (!ws) {
      ws = new KnnWorkspace;
      ws->device = dev;
      if (hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking) != hipSuccess) { delete ws; ws = nullptr; throw Error(E3D_ERR_HIP, "hipStreamCreate failed"); }
    }
  }
  ~WorkspaceLease() {
    if (!ws) return;
    static const double keep_gb = [] { const char* e = getenv("E3D_WORKSPACE_KEEP_GB"); return e ? atof(e) : 32.0; }();
    if ((double)ws->bytes() > keep_gb * 1073741824.0) { delete ws; return; }
    std::lock_guard<std::mutex> lock(workspace_mutex());
    workspace_pool().push_back(ws);
  }
};

}  // namespace e3d

using namespace e3d;

static size_t release_workspace_pool() {
  std::vector<KnnWorkspace*> take;
  {
    std::lock_guard<std::mutex> lock(workspace_mutex());
    take.swap(workspace_pool());
  }
  int prev = 0;
  const bool have_dev = hipGetDevice(&prev) == hipSuccess;
  size_t bytes = 0;
  for (KnnWorkspace* w : take) { bytes += w->bytes(); (void)hipSetDevice(w->device); delete w; }
  if (have_dev) (void)hipSetDevice(prev);
  return bytes;
}

// e3d_common.hpp: the pool is a cache -- an allocation anywhere in the library that runs out of HBM gives it back and retries
bool e3d::release_cached_device_memory() { return release_workspace_pool() > 0; }

extern "C" int e3d_release_workspaces(void) {
  (void)release_workspace_pool();
  return 0;
}

namespace e3d {
// A pointer the kernels can use as it is (device memory of the current device)?  Host pointers go through a staging copy.
static bool is_device_pointer(const void* ptr) {
  if (!ptr) return false;
  hipPointerAttribute_t attr{};
  if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  return attr.type == hipMemoryTypeDevice && attr.device == dev;
}

// The exact kNN pass over a host cloud.  Results stay on the device: normals + curvature (if want_normals), the neighbour
// index lists (if d_knn) and the mean neighbour distance (if d_mean), all in input order.
// direct_n / direct_c: device buffers that take the normals / curvatures as they are computed (no staging in W.d_on / W.d_oc);
// a cloud that already lies in device memory is read in place.
static void knn_pass(KnnWorkspace& W, const float* xyz, size_t n, int k, const float* viewpoint, bool want_normals, bool want_knn,
                     bool want_mean, float* direct_n = nullptr, float* direct_c = nullptr) {
    hipStream_t s = W.stream;
    DevBuf<float>&bbox_partial = W.bbox_partial, &bbox_out = W.bbox_out;
    DevBuf<int>* d_knn_out = want_knn ? &W.d_knn : nullptr;
    DevBuf<float>* d_mean_out = want_mean ? &W.d_mean : nullptr;
    const bool in_place = is_device_pointer(xyz);
    if (!in_place) W.raw.reserve(3 * n);
    struct { const float* p; } raw{in_place ? xyz : W.raw.p};
    struct { float* p; } d_on{nullptr}, d_oc{nullptr};
    if (want_normals) {
      if (direct_n && direct_c) { d_on.p = direct_n; d_oc.p = direct_c; }
      else { W.d_on.reserve(3 * n); W.d_oc.reserve(n); d_on.p = W.d_on.p; d_oc.p = W.d_oc.p; }
    }
    if (d_knn_out) d_knn_out->reserve(n * (size_t)k);
    if (d_mean_out) d_mean_out->reserve(n);
    const bool knn_indices = d_knn_out != nullptr;
    DevBuf<int> no_knn;
    DevBuf<int>& d_knn = d_knn_out ? *d_knn_out : no_knn;
    if (!in_place) copy_in(W.raw.p, xyz, sizeof(float) * 3 * n, s);
    bbox_partial.reserve(6 * (size_t)kMaxBboxBlocks); bbox_out.reserve(6);
    launch_bbox_aos(raw.p, n, bbox_partial.p, bbox_out.p, s);
    float bb[6];
    W.read_back(bb, bbox_out.p, sizeof bb);
    double ext[3], extent = 0, vol = 1;
    for (int a = 0; a < 3; ++a) { ext[a] = (double)bb[3 + a] - (double)bb[a]; extent = std::max(extent, ext[a]); }
    if (!(extent > 0) || !std::isfinite(extent)) extent = 1.0;
    // starting cell size: k points inside ~1.5 cells^2 of a surface-like cloud whose area is estimated from the
    // bounding box faces; the multi-level retry makes any choice exact, this only sets the speed
    double area = 2.0 * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2]);
    if (!(area > 0)) area = extent * extent;
    (void)vol;
    // (swept at 20 M points of the synthetic room scan, tools/bench_normals.py: the estimate is low in the dense part of a scan,
    // where most points are; with the two-pass variant 1.2 - 1.5 is the optimum for k = 32 and 0.3 - 0.5 for k = 8; E3D_KNN_CELL_FACTOR overrides)
    static const double cell_factor_env = [] { const char* e = getenv("E3D_KNN_CELL_FACTOR"); const double v = e ? atof(e) : 0.0; return v > 0 ? v : 0.0; }();
    const double cell_factor = cell_factor_env > 0 ? cell_factor_env : (k > 16 ? 1.3 : 0.45);
    double cell = std::sqrt((double)k * area / (cell_factor * M_PI * (double)n));
    cell = std::max(cell, extent / 1.0e6);
    double magnitude = 0;
    for (int a = 0; a < 6; ++a) magnitude = std::max(magnitude, std::fabs((double)bb[a]));

    LevelBuffers& L = W.L;
    L.ka.reserve(n); L.kb.reserve(n); L.va.reserve(n); L.vb.reserve(n); L.counter.reserve(4);
    L.P4.reserve(n + 8);
    // queries in level-0 cell order (spatially coherent for every level): level 0's sorted points themselves; a further level sorts
    // into the other buffer (the two are swapped then)
    struct { const float4* p; } Q4{nullptr};
    DevBuf<unsigned>&todo_a = W.todo_a, &todo_b = W.todo_b;
    todo_a.reserve(n); todo_b.reserve(n);
    unsigned* todo = nullptr;
    size_t n_todo = n;
    static const int forced_sel = [] { const char* e = getenv("E3D_KNN_SELECT"); return e ? atoi(e) : -1; }();     // experiments: 0 heap, 1 flat, 2 grouped, 3 two-pass
    const int sel_list = k <= 16 ? 1 : (k <= 32 ? 2 : 0);                   // list-maintaining variant (also variant 3's fallback)
    constexpr int kTwoPassMaxK = 60;                                        // k + 4 list slots <= the largest sorting network (64)
    const int sel = (forced_sel >= 0 && forced_sel <= 3 && (forced_sel < 2 || (forced_sel == 2 && k <= 32) || (forced_sel == 3 && k <= kTwoPassMaxK))) ? forced_sel : (k <= kTwoPassMaxK ? 3 : 0);
    // spare list slots of the two-pass variant (it serves the single scan's leftovers, so its occupancy matters little): with 4 a
    // fifth of a scanner-sampled scan's leftovers overflowed into the list-maintaining variant (20 M points, scanner-sampled:
    // k = 32 18.8 -> 17.5 ms with 20, k = 8 11.3 -> 11.0 with 12; uniform scan unchanged; profiles/round5_normals_cap_extra.txt)
    static const int cap_extra_env = [] { const char* e = getenv("E3D_KNN_CAP_EXTRA"); return e ? atoi(e) : -1; }();
    // (k > 32: the two-pass variant is the main kernel there, its LDS list decides the occupancy: 4 as before)
    const int cap_extra = cap_extra_env >= 0 ? cap_extra_env : (k <= 16 ? 12 : (k <= 32 ? 20 : 4));
    const int cap = sel == 3 ? std::min(((std::max(k + cap_extra, 12) + 1) & ~1), 64) : k;   // list entries per thread in LDS (<= the largest sorting network)
    const size_t lds = sel == 3 ? (size_t)cap * kKnnBlock * 4 : (size_t)cap * kKnnBlock * 8, lds_list = (size_t)k * kKnnBlock * 8;
    auto kernel_of = [](int v) {
      return v == 0 ? k_knn_normals<0> : (v == 1 ? k_knn_normals<1> : (v == 2 ? k_knn_normals<2> : (v == 3 ? k_knn_normals<3> : (v == 4 ? k_knn_normals<4> : k_knn_normals<5>))));
    };
    // the single-pass variant (see k_knn_normals): sampled thresholds, level 0 only; its leftovers take the two-pass variant
    static const int single_env = [] { const char* e = getenv("E3D_KNN_SINGLE"); return e ? atoi(e) : 1; }();
    static const int rep_stride_env = [] { const char* e = getenv("E3D_KNN_REP_STRIDE"); return e ? atoi(e) : 0; }();
    static const int rep_avg_env = [] { const char* e = getenv("E3D_KNN_REP_AVG"); return e ? atoi(e) : 0; }();
    static const double rep_target_env = [] { const char* e = getenv("E3D_KNN_REP_TARGET"); return e ? atof(e) : 0.0; }();
    constexpr int kSinglePassMaxK = kKnnTagSlots / 2;                        // variant 5 hands back k positions in its cap / 2 dwords
    const bool single = sel == 3 && single_env != 0 && k <= kSinglePassMaxK && k >= 3;
    // small k: 32-bit [key | tag] entries (variant 4, 36 slots: the occupancy of the two-pass variant at k = 32); beyond that
    // the 64 slots would leave 10 waves per CU, so the list holds 16-bit tags and the keys are computed afterwards (variant 5)
    const int single_variant = k <= 10 ? 4 : 5;
    static const int cap1_env = [] { const char* e = getenv("E3D_KNN_CAP1"); return e ? atoi(e) : 0; }();
    const int cap1 = single_variant == 5 ? kKnnTagSlots : ((cap1_env >= k + 9 && cap1_env <= 36) ? cap1_env : 32);   // list slots of the single-pass variant (32: 16 KB of LDS per block, five waves per SIMD with variant 4's 86 registers)
    const int rep_stride = rep_stride_env > 0 ? rep_stride_env : 8;
    const int rep_avg = (rep_avg_env == 1 || rep_avg_env == 2 || rep_avg_env == 4) ? rep_avg_env : 2;
    // where the sampled thresholds come from: the block population of the sampled query alone (k_knn_hist's `estimate`, the default
    // with a dense directory: the sampling kernel touches no candidate, 0.51 -> 0.1 ms at 20 M points; 5.70 -> 5.33 ms at k = 8,
    // 8.38 -> 7.98 at k = 32, scanner-sampled 17.5 -> 16.8 / 11.2 -> 10.6) or its distance histogram (E3D_KNN_EST=0).  The scale
    // (E3D_KNN_EST_SCALE) widens the estimate: 1.05 measured best between 0.8 and 1.25.
    static const int rep_estimate_env = [] { const char* e = getenv("E3D_KNN_EST"); return e ? atoi(e) : 64; }();
    const bool rep_estimate = rep_estimate_env != 0 && k <= rep_estimate_env;
    static const float rep_est_scale = [] { const char* e = getenv("E3D_KNN_EST_SCALE"); const double v = e ? atof(e) : 1.05; return (float)(v > 0 ? v : 1.05); }();
    // the count the threshold aims at: the middle of [k, capacity] (k = 32: 48 of 64), a little below it for small k where the
    // relative Poisson noise of the count is larger on the low side
    const int rep_target = rep_target_env > 0 ? (int)rep_target_env : std::min((k + cap1) / 2, 2 * k + 6);
    const size_t lds1 = single_variant == 5 ? (size_t)(8 + cap1 / 2) * kKnnBlock * 4 : (size_t)cap1 * kKnnBlock * 4;
    if (single) E3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_of(single_variant)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    E3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_of(sel)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    E3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_of(sel_list)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_list));
    DevBuf<unsigned>& fb_todo = W.fb_todo;
    if (sel == 3) { fb_todo.reserve(n); L.counter.reserve(4); }
    static const double level_step = [] { const char* e = getenv("E3D_KNN_LEVEL_STEP"); const double v = e ? atof(e) : 0.0; return v > 1 ? v : 2.0; }();   // cell growth per retry level (4 -> 2: -7 % at k = 32)
    static const bool wide_pass = [] { const char* e = getenv("E3D_KNN_WIDE"); return e ? atoi(e) != 0 : true; }();
    // seeds of the 125-cell pass (it takes lists of at most n / 64 queries)
    static const bool seed_env = [] { const char* e = getenv("E3D_KNN_SEED"); return e ? atoi(e) != 0 : true; }();
    // (only the lane-per-query wide pass reads them: with the wave-per-query pass -- the default for k <= 64 -- nothing would, and the
    // scan kernels would write k positions per unresolved query plus a memset per level for nothing)
    static const bool wide_wave = [] { const char* e = getenv("E3D_KNN_WIDE_WAVE"); return e ? atoi(e) != 0 : true; }();
    const bool wave_per_query = wide_wave && sel == 3 && k <= 64;
    const unsigned seed_cap = (wide_pass && seed_env && sel == 3 && !wave_per_query) ? (unsigned)(n / 64 + 1) : 0u;
    if (seed_cap) { W.seed_pos.reserve((size_t)seed_cap * (size_t)k); W.seed_flag.reserve(seed_cap); }
    unsigned* const seed_pos_p = seed_cap ? W.seed_pos.p : nullptr;
    unsigned char* const seed_flag_p = seed_cap ? W.seed_flag.p : nullptr;
    // Grid of cell size `cell_size` over all points into LB: sorted points, dense directory when the bounding grid has at most
    // 2^dir_log2 cells (else the hash table); false if the extent does not fit 21-bit cell coordinates.
    static const int dense_log2 = [] { const char* e = getenv("E3D_KNN_DENSE_LOG2"); return e ? std::min(atoi(e), 31) : 30; }();
    auto build_level = [&](LevelBuffers& LB, double cell_size, int dir_log2, KnnGrid& G) -> bool {
      G = KnnGrid{};
      static const unsigned xcd_map = [] { const char* e = getenv("E3D_KNN_XCD"); const int v = e ? atoi(e) : 64; return (unsigned)std::min(std::max(v, 0), 4096); }();
      G.xcd_map = xcd_map;
      G.cell = (float)cell_size;
      G.g.inv_cell = (float)(1.0 / (double)G.cell);
      for (int a = 0; a < 3; ++a) { G.g.origin[a] = (float)((double)bb[a] - 2.0 * cell_size); G.dmin[a] = bb[a]; G.dmax[a] = bb[3 + a]; }
      G.slack = (float)(16.0 * FLT_EPSILON * (magnitude + 4.0 * cell_size) + 1e-4 * cell_size);
      if (extent / cell_size > (double)((1 << 21) - 8)) return false;
      LB.ka.reserve(n); LB.kb.reserve(n); LB.va.reserve(n); LB.vb.reserve(n); LB.counter.reserve(4); LB.P4.reserve(n + 8);   // (+ 8: the scan kernels read whole batches)
      // dense directory over the bounding grid (cells 0 .. cell of the bbox maximum + 2 per axis) unless it would be huge.  With it
      // the sort key is the 32-bit linear cell index (same (z, y, x) order as the 63-bit key: 4 radix passes instead of 8).
      QueryRange qr{};
      double prod = 1.0;
      for (int a = 0; a < 3; ++a) {
        qr.lo[a] = 0;
        const double cmax = std::floor(((double)bb[3 + a] - (double)G.g.origin[a]) * (double)G.g.inv_cell);
        qr.D[a] = (unsigned)std::max(1.0, std::min(cmax + 4.0, 2097152.0));
        prod *= (double)qr.D[a];
      }
      G.S = nullptr;
      if (prod <= (double)((size_t)1 << dir_log2)) {
        const size_t ncell = (size_t)prod;
        int bits = 1;
        while (((size_t)1 << bits) < ncell) ++bits;
        unsigned* k32_in = reinterpret_cast<unsigned*>(LB.ka.p);
        unsigned* k32_out = reinterpret_cast<unsigned*>(LB.kb.p);
        hipLaunchKernelGGL(k_cell_keys_dense, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, raw.p, n, G.g, qr.D[0], qr.D[1], qr.D[2], k32_in, LB.va.p);
        sort_pairs_u32_u32(k32_in, k32_out, LB.va.p, LB.vb.p, n, bits, LB.temp, s);
        launch_permute(raw.p, nullptr, LB.vb.p, n, LB.P4.p, nullptr, s);
        LB.dense.reserve(ncell + 2);
        const size_t n_tiles = div_up(ncell + 2, (size_t)kDirTile);
        const size_t n_words = n_tiles * (kDirTile / 32);                 // one bit per cell, whole tiles
        LB.coarse.reserve(n_tiles + 1 + n_words);
        unsigned* const occupied = LB.coarse.p + n_tiles + 1;
        E3D_HIP(hipMemsetAsync(LB.coarse.p, 0, sizeof(unsigned) * (n_tiles + 1 + n_words), s));
        hipLaunchKernelGGL(k_dir_mark, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, k32_out, n, LB.coarse.p, LB.dense.p, occupied);
        exclusive_max_scan_u32(LB.coarse.p, n_tiles + 1, LB.temp, s);
        hipLaunchKernelGGL(k_dense_directory, dim3((unsigned)n_tiles), dim3(256), 0, s, k32_out, occupied, LB.coarse.p, LB.dense.p, ncell + 2);
        G.S = LB.dense.p;
        for (int a = 0; a < 3; ++a) G.D[a] = qr.D[a];
      } else {
        launch_cell_keys(raw.p, n, G.g, LB.ka.p, LB.va.p, s);
        sort_pairs_u64_u32(LB.ka.p, LB.kb.p, LB.va.p, LB.vb.p, n, 63, LB.temp, s);
        launch_permute(raw.p, nullptr, LB.vb.p, n, LB.P4.p, nullptr, s);
        // no dense directory (the bounding grid is too large): hash table of the occupied cells
        E3D_HIP(hipMemsetAsync(LB.counter.p, 0, 2 * sizeof(unsigned), s));
        launch_count_cells(LB.kb.p, n, LB.counter.p, s);
        unsigned n_cells = 0;
        E3D_HIP(hipMemcpyAsync(&n_cells, LB.counter.p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        E3D_HIP(hipStreamSynchronize(s));
        size_t tsize = 64;
        while (tsize < 2 * (size_t)n_cells) tsize <<= 1;
        LB.table.reserve(tsize);
        G.g.mask = (unsigned)(tsize - 1);
        E3D_HIP(hipMemsetAsync(LB.table.p, 0xFF, sizeof(HashEntry) * tsize, s));
        launch_build_table(LB.kb.p, n, LB.table.p, G.g.mask, s);
      }
      return true;
    };
    // One search of the listed queries (todo_list == nullptr: all) on a built level.  Results of resolved queries are written; the
    // others are appended to `next_list` (device counter LB.counter[1]: k-th neighbour beyond the 27 cells).  Fallback queries of
    // the two-pass variant (LB.counter[2]) join that list for the wide pass or run through the list-maintaining variant on the same
    // grid right away.  n_next_out: length of next_list; returns the number of fallback queries.
    auto search_level = [&](LevelBuffers& LB, const KnnGrid& G, const unsigned* todo_list, size_t n_list, unsigned* next_list,
                            unsigned& n_next_out, bool merge_fb_into_next, unsigned* single_list) {
      E3D_HIP(hipMemsetAsync(LB.counter.p + 1, 0, 3 * sizeof(unsigned), s));
      if (seed_cap) E3D_HIP(hipMemsetAsync(seed_flag_p, 0, seed_cap, s));
      const int lsel = (sel == 3 && !G.S) ? sel_list : sel;            // the two-pass variant needs the dense directory
      if (lsel == 3 && single && todo_list == nullptr && n_list == n && single_list != nullptr) {
        // single pass over all queries with sampled thresholds; those whose count missed the window are listed in single_list and
        // take the two passes below (the next-level list keeps growing: same counter)
        const size_t n_reps = div_up(n_list, (size_t)rep_stride);
        LB.sel_bin.reserve(n_list);
        hipLaunchKernelGGL(k_knn_hist, dim3((unsigned)div_up(n_reps, kKnnHistBlock)), dim3(kKnnHistBlock), 0, s, LB.P4.p, (const unsigned*)nullptr, n_reps,
                           LB.table.p, G, rep_target, Q4.p, LB.sel_bin.p, (unsigned)rep_stride, (rep_estimate && G.S) ? rep_est_scale : 0.f);
        hipLaunchKernelGGL(kernel_of(single_variant), dim3((unsigned)div_up(n_list, kKnnBlock)), dim3(kKnnBlock), lds1, s, LB.P4.p, n, (const unsigned*)nullptr, n_list, LB.table.p, G, k, cap1,
                           viewpoint[0], viewpoint[1], viewpoint[2], Q4.p, want_normals ? d_on.p : nullptr, want_normals ? d_oc.p : nullptr,
                           knn_indices ? d_knn.p : nullptr, d_mean_out ? d_mean_out->p : nullptr, next_list, LB.counter.p + 1,
                           single_list, LB.counter.p + 3, LB.sel_bin.p, 1, rep_stride, rep_avg, 1.0f, seed_pos_p, seed_flag_p, seed_cap);
        unsigned n_single_fb = 0;
        W.read_back(&n_single_fb, LB.counter.p + 3, sizeof(unsigned));
        E3D_HIP(hipGetLastError());
        if (getenv("E3D_KNN_STATS")) fprintf(stderr, "[knn] single pass (variant %d): %zu queries, target %d of %d slots, %u to the two-pass variant\n", single_variant, n_list, rep_target, cap1, n_single_fb);
        todo_list = single_list;
        n_list = n_single_fb;
      }
      const unsigned nblk = (unsigned)div_up(n_list, kKnnBlock);
      if (lsel == 3 && n_list > 0) {
        LB.sel_bin.reserve(n_list);
        hipLaunchKernelGGL(k_knn_hist, dim3((unsigned)div_up(n_list, kKnnHistBlock)), dim3(kKnnHistBlock), 0, s, LB.P4.p, todo_list, n_list,
                           LB.table.p, G, k, Q4.p, LB.sel_bin.p, 1u, 0.f);
      }
      if (n_list > 0)
      hipLaunchKernelGGL(kernel_of(lsel), dim3(nblk), dim3(kKnnBlock), lsel == 3 ? lds : lds_list, s, LB.P4.p, n, todo_list, n_list, LB.table.p, G, k, lsel == 3 ? cap : k,
                         viewpoint[0], viewpoint[1], viewpoint[2], Q4.p, want_normals ? d_on.p : nullptr, want_normals ? d_oc.p : nullptr,
                         knn_indices ? d_knn.p : nullptr, d_mean_out ? d_mean_out->p : nullptr, next_list, LB.counter.p + 1,
                         fb_todo.p, LB.counter.p + 2, LB.sel_bin.p, 1, 1, 1, 1.0f, lsel == 3 ? seed_pos_p : nullptr, lsel == 3 ? seed_flag_p : nullptr, seed_cap);
      unsigned cnts[2] = {0, 0};
      W.read_back(cnts, LB.counter.p + 1, 2 * sizeof(unsigned));
      E3D_HIP(hipGetLastError());
      const unsigned n_fb = cnts[1];
      if (merge_fb_into_next && n_fb > 0 && ((size_t)cnts[0] + n_fb) * 64 <= n) {
        // the two-pass variant's leftovers join the wide pass that follows (its first 27 cells are this level's block, the shell
        // beyond them is skipped by the face test once the list is full): one launch instead of two
        E3D_HIP(hipMemcpyAsync(next_list + cnts[0], fb_todo.p, sizeof(unsigned) * n_fb, hipMemcpyDeviceToDevice, s));
        cnts[0] += n_fb;
      } else if (n_fb > 0) {
        // queries the two-pass variant could not settle on this level (see its comment): the list-maintaining variant, same
        // grid, appending its unresolved ones to the same next-level list
        hipLaunchKernelGGL(kernel_of(sel_list), dim3((unsigned)div_up((size_t)n_fb, kKnnBlock)), dim3(kKnnBlock), lds_list, s, LB.P4.p, n,
                           fb_todo.p, (size_t)n_fb, LB.table.p, G, k, k, viewpoint[0], viewpoint[1], viewpoint[2], Q4.p,
                           want_normals ? d_on.p : nullptr, want_normals ? d_oc.p : nullptr, knn_indices ? d_knn.p : nullptr,
                           d_mean_out ? d_mean_out->p : nullptr, next_list, LB.counter.p + 1, nullptr, nullptr, nullptr, 1, 1, 1, 1.0f, nullptr, nullptr, 0u);
        W.read_back(cnts, LB.counter.p + 1, sizeof(unsigned));
        E3D_HIP(hipGetLastError());
      }
      n_next_out = cnts[0];
      return n_fb;
    };
    bool queries_kept = false;
    for (int level = 0; level < 64 && n_todo > 0; ++level) {
      KnnGrid G{};
      if (level >= 1 && !queries_kept) { std::swap(W.Q4.p, L.P4.p); std::swap(W.Q4.cap, L.P4.cap); queries_kept = true; }   // (level 0's points stay the queries)
      if (!build_level(L, cell, dense_log2, G)) { cell *= 4.0; --level; continue; }        // too many cells for 21-bit coordinates: coarsen
      if (level == 0) Q4.p = L.P4.p;
      unsigned* next = (todo == todo_a.p) ? todo_b.p : todo_a.p;
      unsigned n_next = 0;
      // (level 0: no list yet, so the other list buffer is free for the single-pass variant's leftovers)
      const unsigned n_fb = search_level(L, G, todo, n_todo, next, n_next, wide_pass, todo == nullptr ? todo_b.p : nullptr);
      if (getenv("E3D_KNN_STATS")) fprintf(stderr, "[knn] level %d cell %g todo %zu fallback %u next %u\n", level, (double)cell, n_todo, n_fb, n_next);
#ifdef E3D_KNN_PROF
      if (getenv("E3D_KNN_STATS")) {
        unsigned long long hp[32], zero[32] = {};
        E3D_HIP(hipStreamSynchronize(s));
        E3D_HIP(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_knn_prof), sizeof hp));
        E3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_prof), zero, sizeof zero));
        fprintf(stderr, "[knn prof] hist: waves %llu cycles/wave:", hp[15]);
        for (int i = 0; i < 5; ++i) fprintf(stderr, " %.0f", hp[15] ? (double)hp[i] / (double)hp[15] : 0.0);
        fprintf(stderr, "\n[knn prof] normals<3>: waves %llu cycles/wave:", hp[31]);
        for (int i = 0; i < 10; ++i) fprintf(stderr, " %.0f", hp[31] ? (double)hp[16 + i] / (double)hp[31] : 0.0);
        fprintf(stderr, "\n");
      }
#endif
      // the few that need a wider look (the k-th neighbour lies outside the 27 cells: sparse regions, outliers): the list-maintaining
      // variant over the 125 cells of the same grid, which reaches as far as a grid of twice the cell size would -- no second grid
      // build for ~1 % of the queries
      // (a long list is cheaper on a grid of twice the cell size with the two-pass kernels: the list-maintaining variant over 125 cells
      // took 20 ms for the 2 M far-field queries of a scanner-sampled scan, a grid build is 1.7 ms)
      if (n_next > 0 && wide_pass && (size_t)n_next * 64 <= n) {
        unsigned* wide_out = (next == todo_a.p) ? todo_b.p : todo_a.p;
        unsigned cw[1] = {0};
        E3D_HIP(hipMemsetAsync(L.counter.p + 1, 0, sizeof(unsigned), s));
        static const int wide_spread = [] { const char* e = getenv("E3D_KNN_WIDE_SPREAD"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 4; }();
        auto lane_per_query = [&](const unsigned* list, size_t n_list, bool seeded) {
          hipLaunchKernelGGL(kernel_of(sel_list), dim3((unsigned)div_up(n_list * (size_t)wide_spread, kKnnBlock)), dim3(kKnnBlock), lds_list, s, L.P4.p, n,
                             list, n_list, L.table.p, G, k, k, viewpoint[0], viewpoint[1], viewpoint[2], Q4.p,
                             want_normals ? d_on.p : nullptr, want_normals ? d_oc.p : nullptr, knn_indices ? d_knn.p : nullptr,
                             d_mean_out ? d_mean_out->p : nullptr, wide_out, L.counter.p + 1, nullptr, nullptr, nullptr, 2, wide_spread, 1, 1.0f,
                             seeded ? seed_pos_p : nullptr, seeded ? seed_flag_p : nullptr, seed_cap);
        };
        if (wave_per_query && G.S) {
          // one wave per query (k_knn_wide_wave); what it hands back (more candidates than its lanes hold, long tie runs) takes the
          // lane-per-query kernel
          E3D_HIP(hipMemsetAsync(L.counter.p + 2, 0, sizeof(unsigned), s));
          hipLaunchKernelGGL(k_knn_wide_wave, dim3((unsigned)div_up((size_t)n_next, (size_t)kWideWaves)), dim3(64 * kWideWaves), 0, s, L.P4.p, next, (size_t)n_next, G, k,
                             viewpoint[0], viewpoint[1], viewpoint[2], Q4.p, want_normals ? d_on.p : nullptr, want_normals ? d_oc.p : nullptr,
                             knn_indices ? d_knn.p : nullptr, d_mean_out ? d_mean_out->p : nullptr, wide_out, L.counter.p + 1, fb_todo.p, L.counter.p + 2);
          unsigned n_back = 0;
          W.read_back(&n_back, L.counter.p + 2, sizeof(unsigned));
          E3D_HIP(hipGetLastError());
          if (getenv("E3D_KNN_STATS")) fprintf(stderr, "[knn] level %d wave-per-query pass: %u queries, %u handed to the lane-per-query kernel\n", level, n_next, n_back);
          if (n_back > 0) lane_per_query(fb_todo.p, (size_t)n_back, false);
        } else {
          lane_per_query(next, (size_t)n_next, true);
        }
        W.read_back(cw, L.counter.p + 1, sizeof(unsigned));
        E3D_HIP(hipGetLastError());
        if (getenv("E3D_KNN_STATS")) fprintf(stderr, "[knn] level %d wide pass todo %u next %u\n", level, n_next, cw[0]);
        next = wide_out;
        n_next = cw[0];
        cell *= 2.0;
      }
      todo = next;
      n_todo = n_next;
      cell *= level_step;
    }
    if (n_todo != 0) throw Error(E3D_ERR_INVALID, "e3d_normals_knn: internal error, unresolved queries remain");
}

static void require_device() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    throw Error(E3D_ERR_NO_DEVICE, "no HIP device visible (libe3dhip needs an MI355X / gfx950 GPU)");
}

// LocalStatisticalOutlierRemoval, second pass (local_statistical_outlier_removal.hpp:113-160)
__global__ __launch_bounds__(256) void k_outlier_classify(const int* __restrict__ knn, const float* __restrict__ mean_dist, size_t n,
                                                          int k, double factor, int negative, unsigned char* __restrict__ inlier) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int valid = 0;
  double sum = 0;
  for (int j = 1; j < k; ++j) {
    const int nb = knn[i * k + j];
    if (nb < 0) break;                                   // cloud smaller than k: the list ends here (nn_indices.size (), :129)
    const double d = (double)mean_dist[nb];
    if (d > 0) { ++valid; sum += d; }
  }
  const double mean = sum / (double)valid;              // 0 / 0 = NaN: every comparison below is false, as in the reference
  const double threshold = mean * factor;
  const double own = (double)mean_dist[i];
  const bool removed = (!negative && own > threshold) || (negative && own <= threshold);
  inlier[i] = removed ? 0 : 1;
}
}  // namespace e3d

namespace e3d {
__global__ __launch_bounds__(kBlock) void k_libm_eval(int fn, const float* __restrict__ x, const float* __restrict__ y, size_t n,
                                                      float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  switch (fn) {
    case 0: r = e3d_atanf(x[i]); break;
    case 1: r = e3d_atan2f(x[i], y[i]); break;
    case 2: r = e3d_sinf(x[i]); break;
    case 3: r = e3d_cosf(x[i]); break;
    case 4: r = e3d_tanf(x[i]); break;
    default: r = e3d_log2f(x[i]); break;
  }
  out[i] = r;
}
}  // namespace e3d

extern "C" int e3d_libm_eval(int fn, const float* x, const float* y, size_t n, float* out) {
  try {
    if ((!x || !y || !out) && n) throw Error(E3D_ERR_INVALID, "e3d_libm_eval: null argument");
    if (fn < 0 || fn > 5) throw Error(E3D_ERR_INVALID, "e3d_libm_eval: fn must be 0..5");
    if (!n) return 0;
    DevBuf<float> dx, dy, dout;
    dx.reserve(n); dy.reserve(n); dout.reserve(n);
    hipStream_t s = nullptr;
    copy_in(dx.p, x, sizeof(float) * n, s); copy_in(dy.p, y, sizeof(float) * n, s);
    hipLaunchKernelGGL(k_libm_eval, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, s, fn, dx.p, dy.p, n, dout.p);
    copy_out(out, dout.p, sizeof(float) * n, s);
    E3D_HIP(hipStreamSynchronize(s));
    return 0;
  } catch (const e3d::Error& e) { e3d::set_last_error(e.what()); return e.code; }
  catch (const std::exception& e) { e3d::set_last_error(e.what()); return E3D_ERR_INVALID; }
}

extern "C" int e3d_normals_knn(const float* xyz, size_t n, int k, const float* viewpoint, float* out_normals,
                               float* out_curvature, int32_t* knn_indices) {
  try {
    if ((!xyz && n) || !viewpoint || (!out_normals && n) || (!out_curvature && n))
      throw Error(E3D_ERR_INVALID, "e3d_normals_knn: null argument");
    if (k < 1 || k > kKnnMaxK) throw Error(E3D_ERR_INVALID, fmt("e3d_normals_knn: k = %d outside [1, %d]", k, kKnnMaxK));
    if (n >= (size_t)1 << 31) throw Error(E3D_ERR_INVALID, "e3d_normals_knn: more than 2^31-1 points");
    require_device();
    if (n == 0) return 0;
    WorkspaceLease lease;
    KnnWorkspace& W = *lease.ws;
    hipStream_t s = W.stream;
    const bool direct = is_device_pointer(out_normals) && is_device_pointer(out_curvature);   // results straight into the caller's device buffers
    knn_pass(W, xyz, n, k, viewpoint, true, knn_indices != nullptr, false, direct ? out_normals : nullptr, direct ? out_curvature : nullptr);
    DevBuf<int>& d_knn = W.d_knn;
    if (!direct) {
      copy_out(out_normals, W.d_on.p, sizeof(float) * 3 * n, s);
      copy_out(out_curvature, W.d_oc.p, sizeof(float) * n, s);
    }
    if (knn_indices) copy_out(knn_indices, d_knn.p, sizeof(int) * n * (size_t)k, s);
    E3D_HIP(hipStreamSynchronize(s));
    return 0;
  } catch (const e3d::Error& e) {
    e3d::set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    e3d::set_last_error(e.what());
    return E3D_ERR_INVALID;
  }
}

// pcl::LocalStatisticalOutlierRemoval<PointT>::applyFilterIndices (src/geometry/local_statistical_outlier_removal.hpp:71-172)
extern "C" int e3d_local_outlier_removal(const float* xyz, size_t n, int mean_k, double distance_factor_threshold, int negative,
                                         uint8_t* inlier, float* mean_distances) {
  try {
    if ((!xyz && n) || (!inlier && n)) throw Error(E3D_ERR_INVALID, "e3d_local_outlier_removal: null argument");
    if (mean_k < 1 || mean_k + 1 > kKnnMaxK) throw Error(E3D_ERR_INVALID, fmt("e3d_local_outlier_removal: mean_k = %d outside [1, %d]", mean_k, kKnnMaxK - 1));
    if (n >= (size_t)1 << 31) throw Error(E3D_ERR_INVALID, "e3d_local_outlier_removal: more than 2^31-1 points");
    require_device();
    if (n == 0) return 0;
    // non-finite points: distance 0 in the first pass, "problematic" (removed unless negative) in the second (:88-95, :115-125)
    std::vector<float> finite;
    std::vector<size_t> origin;
    bool all_finite = true;
    for (size_t i = 0; i < n && all_finite; ++i)
      all_finite = std::isfinite(xyz[3 * i]) && std::isfinite(xyz[3 * i + 1]) && std::isfinite(xyz[3 * i + 2]);
    const float* pts = xyz;
    size_t m = n;
    if (!all_finite) {
      for (size_t i = 0; i < n; ++i)
        if (std::isfinite(xyz[3 * i]) && std::isfinite(xyz[3 * i + 1]) && std::isfinite(xyz[3 * i + 2])) {
          finite.insert(finite.end(), xyz + 3 * i, xyz + 3 * i + 3);
          origin.push_back(i);
        }
      pts = finite.data(); m = origin.size();
    }
    // m <= mean_k: the search returns the m points there are (pcl::KdTreeFLANN clamps k), the mean still divides by mean_k and
    // the second pass walks the shorter lists -- the neighbour lists are padded with -1 and the kernels stop there
    if (m == 0) { for (size_t i = 0; i < n; ++i) { inlier[i] = 0; if (mean_distances) mean_distances[i] = 0.f; } return 0; }
    WorkspaceLease lease;
    KnnWorkspace& W = *lease.ws;
    hipStream_t s = W.stream;
    DevBuf<float>& d_mean = W.d_mean;
    DevBuf<int>& d_knn = W.d_knn;
    DevBuf<unsigned char>& d_in = W.d_in;
    const float vp[3] = {0.f, 0.f, 0.f};
    const int k = mean_k + 1;
    knn_pass(W, pts, m, k, vp, false, true, true);
    d_in.reserve(m);
    hipLaunchKernelGGL(k_outlier_classify, dim3((unsigned)div_up(m, 256)), dim3(256), 0, s, d_knn.p, d_mean.p, m, k,
                       distance_factor_threshold, negative, d_in.p);
    if (all_finite) {
      copy_out(inlier, d_in.p, m, s);
      if (mean_distances) copy_out(mean_distances, d_mean.p, sizeof(float) * m, s);
      E3D_HIP(hipStreamSynchronize(s));
    } else {
      std::vector<unsigned char> in(m);
      std::vector<float> md(m);
      copy_out(in.data(), d_in.p, m, s);
      copy_out(md.data(), d_mean.p, sizeof(float) * m, s);
      E3D_HIP(hipStreamSynchronize(s));
      // a non-finite point is never an inlier: removed when !negative (:120-125); with negative it reaches the comparison
      // with distance 0 <= threshold -> removed as well unless the threshold is NaN
      for (size_t i = 0; i < n; ++i) { inlier[i] = 0; if (mean_distances) mean_distances[i] = 0.f; }
      for (size_t j = 0; j < m; ++j) { inlier[origin[j]] = in[j]; if (mean_distances) mean_distances[origin[j]] = md[j]; }
    }
    return 0;
  } catch (const e3d::Error& e) {
    e3d::set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    e3d::set_last_error(e.what());
    return E3D_ERR_INVALID;
  }
}

extern "C" int e3d_normals_radius(const float* xyz, size_t n, float radius, const float* viewpoint, float* out_normals,
                                  float* out_curvature, int32_t* neighbor_counts) {
  try {
    if ((!xyz && n) || !viewpoint || (!out_normals && n) || (!out_curvature && n))
      throw Error(E3D_ERR_INVALID, "e3d_normals_radius: null argument");
    if (!(radius > 0.f) || !std::isfinite(radius)) throw Error(E3D_ERR_INVALID, "e3d_normals_radius: radius must be positive");
    if (n >= (size_t)1 << 31) throw Error(E3D_ERR_INVALID, "e3d_normals_radius: more than 2^31-1 points");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw Error(E3D_ERR_NO_DEVICE, "no HIP device visible (libe3dhip needs an MI355X / gfx950 GPU)");
    if (n == 0) return 0;
    hipStream_t s = nullptr;
    E3D_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{s};
    DevBuf<float> raw, d_on, d_oc, bbox_partial, bbox_out;
    DevBuf<int> d_cnt;
    raw.reserve(3 * n); d_on.reserve(3 * n); d_oc.reserve(n);
    if (neighbor_counts) d_cnt.reserve(n);
    copy_in(raw.p, xyz, sizeof(float) * 3 * n, s);
    bbox_partial.reserve(6 * (size_t)kMaxBboxBlocks); bbox_out.reserve(6);
    launch_bbox_aos(raw.p, n, bbox_partial.p, bbox_out.p, s);
    float bb[6];
    copy_out(bb, bbox_out.p, sizeof bb, s);
    E3D_HIP(hipStreamSynchronize(s));
    double extent = 0, magnitude = 0;
    for (int a = 0; a < 3; ++a) extent = std::max(extent, (double)bb[3 + a] - (double)bb[a]);
    for (int a = 0; a < 6; ++a) magnitude = std::max(magnitude, std::fabs((double)bb[a]));
    // cell = radius plus slack for the f32 cell-index computation: every point within the radius lies in the 27 cells
    double cell = (double)radius * (1.0 + 1e-4) + 16.0 * FLT_EPSILON * (magnitude + 4.0 * (double)radius);
    if (extent / cell > (double)((1 << 21) - 8)) throw Error(E3D_ERR_INVALID, "e3d_normals_radius: radius too small for the extent of the cloud");
    KnnGrid G{};
    G.cell = (float)cell;
    G.g.inv_cell = (float)(1.0 / (double)G.cell);
    for (int a = 0; a < 3; ++a) { G.g.origin[a] = (float)((double)bb[a] - 2.0 * cell); G.dmin[a] = bb[a]; G.dmax[a] = bb[3 + a]; }
    LevelBuffers L;
    L.ka.reserve(n); L.kb.reserve(n); L.va.reserve(n); L.vb.reserve(n); L.counter.reserve(4); L.P4.reserve(n);
    launch_cell_keys(raw.p, n, G.g, L.ka.p, L.va.p, s);
    sort_pairs_u64_u32(L.ka.p, L.kb.p, L.va.p, L.vb.p, n, 63, L.temp, s);
    launch_permute(raw.p, nullptr, L.vb.p, n, L.P4.p, nullptr, s);
    E3D_HIP(hipMemsetAsync(L.counter.p, 0, 2 * sizeof(unsigned), s));
    launch_count_cells(L.kb.p, n, L.counter.p, s);
    unsigned n_cells = 0;
    E3D_HIP(hipMemcpyAsync(&n_cells, L.counter.p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    E3D_HIP(hipStreamSynchronize(s));
    size_t tsize = 64;
    while (tsize < 2 * (size_t)n_cells) tsize <<= 1;
    L.table.reserve(tsize);
    G.g.mask = (unsigned)(tsize - 1);
    E3D_HIP(hipMemsetAsync(L.table.p, 0xFF, sizeof(HashEntry) * tsize, s));
    launch_build_table(L.kb.p, n, L.table.p, G.g.mask, s);
    const double rr = (double)radius;
    const float r2 = (float)(rr * rr);      // pcl::KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius)
    hipLaunchKernelGGL(k_radius_normals, dim3((unsigned)div_up(n, kKnnBlock)), dim3(kKnnBlock), 0, s, L.P4.p, n, L.table.p, G, r2,
                       viewpoint[0], viewpoint[1], viewpoint[2], d_on.p, d_oc.p, neighbor_counts ? d_cnt.p : nullptr);
    copy_out(out_normals, d_on.p, sizeof(float) * 3 * n, s);
    copy_out(out_curvature, d_oc.p, sizeof(float) * n, s);
    if (neighbor_counts) copy_out(neighbor_counts, d_cnt.p, sizeof(int) * n, s);
    E3D_HIP(hipStreamSynchronize(s));
    E3D_HIP(hipGetLastError());
    return 0;
  } catch (const e3d::Error& e) {
    e3d::set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    e3d::set_last_error(e.what());
    return E3D_ERR_INVALID;
  }
}
