// e3d_multires.hip -- MergeClosePoints of the multi-resolution point cloud construction (SURVEY f1;
// src/opt/multi_scale_point_cloud.cc:44-124) on the MI355X.
//
// The reference is a sequential greedy loop: in point order, every point that has not been absorbed yet becomes a centre and
// absorbs ALL points within the merge distance (absorbed ones included).  The set of centres is therefore the
// lexicographically first maximal independent set of the "closer than the merge distance" graph, and each output point
// depends only on its centre's neighbourhood.  Both parts are computed in parallel and exactly:
//
//  k_merge_decide  one thread per point, tickets handed out in POINT ORDER by an atomic counter.  A thread gathers its
//                  lower-index neighbours once and then waits for their decisions: covered as soon as one of them is a
//                  centre, centre once all of them are covered.  Every thread only ever waits for LOWER tickets, which
//                  belong to threads that are already running (or done), so the waits cannot deadlock; the greedy loop's
//                  dependency chains become a pipelined wavefront instead of n sequential steps.
//  k_merge_output  one thread per centre (compacted in point order): mean position, per-scan counts / colour sums,
//                  majority scan, maximum of max_radius over its neighbourhood.
//
// Order-dependent details of the reference follow FLANN's unsorted radius-search order, which is not pinned by anything:
// f32 sums are formed in grid order here (round-off differs), and the majority scan's tie rule is taken for neighbours
// visited in increasing index (the scan whose LAST merged point has the lowest index wins a tie).
#include <cfloat>
#include <cmath>
#include <vector>

#include "../../include/e3d_hip.h"
#include "e3d_icp_kernels.hpp"

#pragma clang fp contract(off)

namespace e3d {

constexpr int kMergeBlock = 256;
constexpr int kMergeList = 40;        // lower-index neighbours kept per thread (LDS); more -> cells are re-scanned while waiting
constexpr int kMergeMaxScans = 16;

enum : unsigned char { kUndecided = 0, kCentre = 1, kCovered = 2 };

struct MergeGrid { GridDesc g; };

__device__ __forceinline__ void merge_cell_range(const HashEntry* __restrict__ table, const GridDesc& g, int x, int y, int z,
                                                 unsigned& s, unsigned& e) {
  s = 0; e = 0;
  constexpr int kMaxC = (1 << 21) - 1;
  if (x < 0 || y < 0 || z < 0 || x > kMaxC || y > kMaxC || z > kMaxC) return;
  const unsigned long long key = cell_key(x, y, z);
  unsigned h = hash_key(key) & g.mask;
  for (;;) {
    const HashEntry en = table[h];
    if (en.key == key) { s = en.start; e = en.end; return; }
    if (en.key == kEmptyKey) return;
    h = (h + 1) & g.mask;
  }
}

__device__ __forceinline__ unsigned char load_state(const unsigned char* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_state(unsigned char* p, unsigned char v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// pos_of[i] = position of original point i in the cell-ordered array P4 (P4[pos].w = original index)
__global__ __launch_bounds__(kMergeBlock) void k_merge_positions(const float4* __restrict__ P4, size_t n, unsigned* __restrict__ pos_of) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) pos_of[__float_as_uint(P4[j].w)] = (unsigned)j;
}

__global__ __launch_bounds__(kMergeBlock) void k_merge_decide(const float4* __restrict__ P4, const unsigned* __restrict__ pos_of,
                                                              size_t n, const HashEntry* __restrict__ table, GridDesc g, float r2,
                                                              unsigned* __restrict__ ticket, unsigned char* __restrict__ state) {
  __shared__ unsigned nb[kMergeList][kMergeBlock];
  __shared__ unsigned wave_base[kMergeBlock / kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_base[wv] = atomicAdd(ticket, (unsigned)kWave);
  __builtin_amdgcn_wave_barrier();
  const size_t i = (size_t)wave_base[wv] + lane;        // point index = ticket: lower indices always start first
  if (i >= n) return;
  const float4 q = P4[pos_of[i]];
  const int cx = cell_coord(q.x, g.origin[0], g.inv_cell), cy = cell_coord(q.y, g.origin[1], g.inv_cell),
            cz = cell_coord(q.z, g.origin[2], g.inv_cell);
  // lower-index neighbours within the merge distance
  int cnt = 0;
  for (int oz = -1; oz <= 1; ++oz)
    for (int oy = -1; oy <= 1; ++oy)
      for (int ox = -1; ox <= 1; ++ox) {
        unsigned s, e;
        merge_cell_range(table, g, cx + ox, cy + oy, cz + oz, s, e);
        for (unsigned m = s; m < e; ++m) {
          const float4 c = P4[m];
          const unsigned j = __float_as_uint(c.w);
          if (j < (unsigned)i && sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z) < r2) {
            if (cnt < kMergeList) nb[cnt][threadIdx.x] = j;
            ++cnt;
          }
        }
      }
  // The wait loop has a WAVE-UNIFORM exit (ballot) and the decision is stored inside the loop body: with a divergent
  // `store; return` the compiler may run the exit block only after the whole wave has left the loop, and a lane waiting for
  // another lane of its own wave would never see the decision.
  bool done = false;
  if (cnt == 0) { store_state(&state[i], kCentre); done = true; }
  while (__ballot(!done)) {
    if (!done) {
      bool centre_seen = false, all_covered = true;
      if (cnt <= kMergeList) {
        for (int k = 0; k < cnt; ++k) {
          const unsigned char st = load_state(&state[nb[k][threadIdx.x]]);
          if (st == kCentre) centre_seen = true;
          if (st == kUndecided) all_covered = false;
        }
      } else {     // rare: very dense neighbourhood -> walk the cells again
        for (int oz = -1; oz <= 1; ++oz)
          for (int oy = -1; oy <= 1; ++oy)
            for (int ox = -1; ox <= 1; ++ox) {
              unsigned s, e;
              merge_cell_range(table, g, cx + ox, cy + oy, cz + oz, s, e);
              for (unsigned m = s; m < e; ++m) {
                const float4 c = P4[m];
                const unsigned j = __float_as_uint(c.w);
                if (j < (unsigned)i && sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z) < r2) {
                  const unsigned char st = load_state(&state[j]);
                  if (st == kCentre) centre_seen = true;
                  if (st == kUndecided) all_covered = false;
                }
              }
            }
      }
      if (centre_seen) { store_state(&state[i], kCovered); done = true; }
      else if (all_covered) { store_state(&state[i], kCentre); done = true; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// flags (centre = 1) -> exclusive positions, in point order (single-block scan over per-block counts)
__global__ __launch_bounds__(kMergeBlock) void k_merge_block_counts(const unsigned char* __restrict__ state, size_t n,
                                                                    unsigned* __restrict__ block_counts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool f = i < n && state[i] == kCentre;
  const unsigned long long b = __ballot(f);
  __shared__ unsigned sc[kMergeBlock / kWave];
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = (unsigned)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) { unsigned c = 0; for (int k = 0; k < kMergeBlock / kWave; ++k) c += sc[k]; block_counts[blockIdx.x] = c; }
}

__global__ __launch_bounds__(kMergeBlock) void k_merge_output(const float4* __restrict__ P4, const unsigned* __restrict__ pos_of, size_t n,
                                                              const HashEntry* __restrict__ table, GridDesc g, float r2,
                                                              const unsigned char* __restrict__ state,
                                                              const unsigned* __restrict__ block_offsets, const float* __restrict__ colors,
                                                              const unsigned char* __restrict__ scan_idx, const float* __restrict__ max_radius,
                                                              int num_scans, float* __restrict__ out_xyz, float* __restrict__ out_color,
                                                              unsigned char* __restrict__ out_scan, float* __restrict__ out_max_radius) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool f = i < n && state[i] == kCentre;
  const unsigned long long b = __ballot(f);
  __shared__ unsigned sc[kMergeBlock / kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sc[wv] = (unsigned)__popcll(b);
  __syncthreads();
  if (!f) return;
  unsigned o = block_offsets[blockIdx.x];
  for (int k = 0; k < wv; ++k) o += sc[k];
  o += (unsigned)__popcll(b & ((1ull << lane) - 1ull));

  const float4 q = P4[pos_of[i]];
  const int cx = cell_coord(q.x, g.origin[0], g.inv_cell), cy = cell_coord(q.y, g.origin[1], g.inv_cell),
            cz = cell_coord(q.z, g.origin[2], g.inv_cell);
  int count[kMergeMaxScans];
  unsigned last[kMergeMaxScans];
  float csum[kMergeMaxScans];
#pragma unroll
  for (int s2 = 0; s2 < kMergeMaxScans; ++s2) { count[s2] = 0; last[s2] = 0; csum[s2] = 0.f; }
  float ax = 0.f, ay = 0.f, az = 0.f, mr = -1.f;
  int total = 0;
  for (int oz = -1; oz <= 1; ++oz)
    for (int oy = -1; oy <= 1; ++oy)
      for (int ox = -1; ox <= 1; ++ox) {
        unsigned s, e;
        merge_cell_range(table, g, cx + ox, cy + oy, cz + oz, s, e);
        for (unsigned m = s; m < e; ++m) {
          const float4 c = P4[m];
          if (!(sqdist_l2(q.x, q.y, q.z, c.x, c.y, c.z) < r2)) continue;
          const unsigned j = __float_as_uint(c.w);
          const int sidx = scan_idx[j];
          ax += c.x; ay += c.y; az += c.z;
          const float col = colors[j];
          const float mrj = max_radius[j];
          if (mrj > mr) mr = mrj;
#pragma unroll
          for (int s2 = 0; s2 < kMergeMaxScans; ++s2)
            if (s2 == sidx) { count[s2] += 1; csum[s2] += col; last[s2] = max(last[s2], j); }
          ++total;
        }
      }
  // majority scan; ties: the scan that reaches the maximum first when neighbours are visited in index order
  int best = 0;
#pragma unroll
  for (int s2 = 1; s2 < kMergeMaxScans; ++s2)
    if (s2 < num_scans && (count[s2] > count[best] || (count[s2] == count[best] && count[s2] > 0 && last[s2] < last[best]))) best = s2;
  int bc = 1; float bs = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < kMergeMaxScans; ++s2) if (s2 == best) { bc = count[s2]; bs = csum[s2]; }
  const float ft = (float)total;
  out_xyz[3 * (size_t)o] = ax / ft; out_xyz[3 * (size_t)o + 1] = ay / ft; out_xyz[3 * (size_t)o + 2] = az / ft;
  out_color[o] = bs / (float)bc;
  out_scan[o] = (unsigned char)best;
  out_max_radius[o] = mr;
}

}  // namespace e3d

using namespace e3d;

extern "C" int64_t e3d_merge_close_points(float merge_distance, int num_scans, const float* xyz, const float* colors,
                                          const uint8_t* scan_indices, const float* max_radius, size_t n, float* out_xyz,
                                          float* out_colors, uint8_t* out_scan_indices, float* out_max_radius) {
  try {
    if (n && (!xyz || !colors || !scan_indices || !max_radius || !out_xyz || !out_colors || !out_scan_indices || !out_max_radius))
      throw Error(E3D_ERR_INVALID, "e3d_merge_close_points: null argument");
    if (!(merge_distance > 0.f) || !std::isfinite(merge_distance)) throw Error(E3D_ERR_INVALID, "e3d_merge_close_points: merge distance must be positive");
    if (num_scans < 1 || num_scans > kMergeMaxScans) throw Error(E3D_ERR_INVALID, fmt("e3d_merge_close_points: 1..%d scans supported", kMergeMaxScans));
    if (n >= (size_t)1 << 31) throw Error(E3D_ERR_INVALID, "e3d_merge_close_points: more than 2^31-1 points");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(E3D_ERR_NO_DEVICE, "no HIP device visible (libe3dhip needs an MI355X / gfx950 GPU)");
    if (n == 0) return 0;
    hipStream_t s = nullptr;
    E3D_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{s};

    DevBuf<float> raw, d_col, d_mr, bbox_partial, bbox_out, o_xyz, o_col, o_mr;
    DevBuf<unsigned char> d_scan, state, o_scan;
    raw.reserve(3 * n); d_col.reserve(n); d_mr.reserve(n); d_scan.reserve(n); state.reserve(n);
    copy_in(raw.p, xyz, sizeof(float) * 3 * n, s);
    copy_in(d_col.p, colors, sizeof(float) * n, s);
    copy_in(d_mr.p, max_radius, sizeof(float) * n, s);
    copy_in(d_scan.p, scan_indices, n, s);
    bbox_partial.reserve(6 * (size_t)kMaxBboxBlocks); bbox_out.reserve(6);
    launch_bbox_aos(raw.p, n, bbox_partial.p, bbox_out.p, s);
    float bb[6];
    copy_out(bb, bbox_out.p, sizeof bb, s);
    E3D_HIP(hipStreamSynchronize(s));
    double extent = 0, magnitude = 0;
    for (int a = 0; a < 3; ++a) extent = std::max(extent, (double)bb[3 + a] - (double)bb[a]);
    for (int a = 0; a < 6; ++a) magnitude = std::max(magnitude, std::fabs((double)bb[a]));
    double cell = (double)merge_distance * (1.0 + 1e-4) + 16.0 * FLT_EPSILON * (magnitude + 4.0 * (double)merge_distance);
    while (extent / cell > (double)((1 << 21) - 8)) cell *= 2.0;      // cells larger than the radius are fine, only slower
    GridDesc g{};
    g.inv_cell = (float)(1.0 / cell);
    for (int a = 0; a < 3; ++a) g.origin[a] = (float)((double)bb[a] - 2.0 * cell);
    DevBuf<unsigned long long> ka, kb;
    DevBuf<unsigned> va, vb, counter, pos_of, ticket, block_counts, block_offsets;
    DevBuf<char> temp;
    DevBuf<float4> P4;
    DevBuf<HashEntry> table;
    ka.reserve(n); kb.reserve(n); va.reserve(n); vb.reserve(n); counter.reserve(2); P4.reserve(n); pos_of.reserve(n); ticket.reserve(1);
    launch_cell_keys(raw.p, n, g, ka.p, va.p, s);
    sort_pairs_u64_u32(ka.p, kb.p, va.p, vb.p, n, 63, temp, s);
    launch_permute(raw.p, nullptr, vb.p, n, P4.p, nullptr, s);
    E3D_HIP(hipMemsetAsync(counter.p, 0, 2 * sizeof(unsigned), s));
    launch_count_cells(kb.p, n, counter.p, s);
    unsigned n_cells = 0;
    E3D_HIP(hipMemcpyAsync(&n_cells, counter.p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    E3D_HIP(hipStreamSynchronize(s));
    size_t tsize = 64;
    while (tsize < 2 * (size_t)n_cells) tsize <<= 1;
    table.reserve(tsize);
    g.mask = (unsigned)(tsize - 1);
    E3D_HIP(hipMemsetAsync(table.p, 0xFF, sizeof(HashEntry) * tsize, s));
    launch_build_table(kb.p, n, table.p, g.mask, s);
    const double md = (double)merge_distance;
    const float r2 = (float)(md * md);        // pcl::KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius), strict <
    const unsigned nblk = (unsigned)div_up(n, kMergeBlock);
    hipLaunchKernelGGL(k_merge_positions, dim3(nblk), dim3(kMergeBlock), 0, s, P4.p, n, pos_of.p);
    E3D_HIP(hipMemsetAsync(state.p, 0, n, s));
    E3D_HIP(hipMemsetAsync(ticket.p, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_merge_decide, dim3(nblk), dim3(kMergeBlock), 0, s, P4.p, pos_of.p, n, table.p, g, r2, ticket.p, state.p);
    block_counts.reserve(nblk); block_offsets.reserve(nblk);
    hipLaunchKernelGGL(k_merge_block_counts, dim3(nblk), dim3(kMergeBlock), 0, s, state.p, n, block_counts.p);
    std::vector<unsigned> hc(nblk), ho(nblk);
    copy_out(hc.data(), block_counts.p, sizeof(unsigned) * nblk, s);
    E3D_HIP(hipStreamSynchronize(s));
    E3D_HIP(hipGetLastError());
    size_t total = 0;
    for (unsigned b = 0; b < nblk; ++b) { ho[b] = (unsigned)total; total += hc[b]; }
    copy_in(block_offsets.p, ho.data(), sizeof(unsigned) * nblk, s);
    o_xyz.reserve(3 * total); o_col.reserve(total); o_mr.reserve(total); o_scan.reserve(total);
    hipLaunchKernelGGL(k_merge_output, dim3(nblk), dim3(kMergeBlock), 0, s, P4.p, pos_of.p, n, table.p, g, r2, state.p, block_offsets.p,
                       d_col.p, d_scan.p, d_mr.p, num_scans, o_xyz.p, o_col.p, o_scan.p, o_mr.p);
    copy_out(out_xyz, o_xyz.p, sizeof(float) * 3 * total, s);
    copy_out(out_colors, o_col.p, sizeof(float) * total, s);
    copy_out(out_scan_indices, o_scan.p, total, s);
    copy_out(out_max_radius, o_mr.p, sizeof(float) * total, s);
    E3D_HIP(hipStreamSynchronize(s));
    E3D_HIP(hipGetLastError());
    return (int64_t)total;
  } catch (const e3d::Error& e) {
    e3d::set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    e3d::set_last_error(e.what());
    return E3D_ERR_INVALID;
  }
}
