// e3d_icp.hip -- host driver of the point-to-plane ICP path and its C-ABI (include/e3d_hip.h).
//
// Mirrors icp::PointToPlaneICP (src/icp/icp_point_to_plane.{h,cc}) and PointToPlaneICPImpl
// (src/icp/icp_point_to_plane_impl.h) of the reference; see DESIGN.md for the MI355X-side design:
//   * every cloud gets a STATIC hash grid in its own local frame (cell >= search radius), built once
//     per search radius; queries are mapped into the target's local frame to pick the 27 candidate
//     cells, distances are evaluated on the global-frame f32 coordinates exactly like the reference
//     (so no index is rebuilt per outer iteration, unlike the reference's per-pair kd-trees);
//   * correspondences are materialised once per outer iteration as three float4 planes; every LM
//     pass streams them (48 B/correspondence) and reduces cost + Gramian blocks in f64;
//   * the accumulate pass and the cost pass of consecutive LM steps are fused (same numbers, half the
//     passes); the tiny LDL^T solve and the SE3 update run on the host in the reference's precisions.
#include <atomic>
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/e3d_hip.h"
#include "e3d_comm.hpp"
#include "e3d_icp_kernels.hpp"
#include "e3d_math.hpp"

#pragma clang fp contract(off)

namespace e3d {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

static int g_device = 0;
// E3D_NN_MODE / e3d_set_nn_mode: 0 auto, 1 per-query, 2 hash-table buckets, 3 dense rows, 4 dense rows + MFMA filter,
// 5 dense rows with the half-cell directory of the bounded search built whatever the density and used for every list (tests)
static int g_nn_mode = [] { const char* e = getenv("E3D_NN_MODE"); const int v = e ? atoi(e) : 0; return (v >= 0 && v <= 5) ? v : 0; }();
static std::atomic<unsigned long long> g_grid_generation{0};   // handles run on one host thread per GPU (--gpus N)

// -------------------------------------------------------------------------------------------------
struct Cloud {
  size_t n = 0;
  bool fixed = false;
  DevBuf<float> raw_xyz, raw_nrm;   // as given (fixed: already in the global frame), AoS
  float T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};   // global_T_cloud
  // static grid (valid for grid_radius)
  bool grid_valid = false;
  float grid_radius = -1.f;
  float grid_T[12];                 // pose the grid's slack was sized for (only its linear part matters)
  DevBuf<float4> L4, LN, G4;
  DevBuf<HashEntry> table;
  DevBuf<unsigned> dense_start;     // dense cell-start directory over qrange (empty if the grid is too large)
  bool has_dense = false;
  DevBuf<unsigned long long> half_prefix;   // half-cell directory (8 prefix bytes per cell) of the bounded search: dense clouds
  bool has_half = false;
  // one bit per cell of the dense directory's range: does any of the 27 cells around it hold a point (k_query_keys_prune)
  DevBuf<unsigned> occ27;
  unsigned occ_stride = 0;                  // 32-bit words per (y, z) row
  bool has_occ = false;
  GridDesc grid{};
  QueryRange qrange{};              // target-cell range a query needs to hit to have candidates
  unsigned n_cells = 0;             // occupied cells
  int key_bits = 0;                 // bits of the dense query keys
  unsigned long long generation = 0;   // changes whenever the sorted order changes (invalidates the certificates' partners)
  // bound of how far any point of this cloud has moved in the global frame since the grid was built (sum over the pose
  // updates of ||dL|| R + ||dt||), and of the f32 rounding of one local -> global evaluation (running maximum)
  double cum_motion = 0.0, last_motion = 0.0, err_max = 0.0;
  double build_slack = 0.0;         // slack of the global -> local mapping the cell size was derived with
  float lmin[3], lmax[3];           // local bbox
  float bmin[3], bmax[3];           // global bbox of the current outer iteration
  int cloud_index = -1;             // impl index in the current AlignMeshes
  // G4 / bmin / bmax are those of pose G4_T (while G4_valid): a cloud whose pose did not change since the last outer iteration --
  // impl cloud 0 never moves, fixed clouds never do -- is not transformed again
  bool G4_valid = false;
  float G4_T[12];
  // per-pose quantities every directed pair with this cloud as target asks for (an all-pairs job asks 15 times per outer iteration:
  // the Jacobi sweeps and the inversion were 0.1 ms of host time per batch of 32 pairs)
  mutable bool pose_cache_valid = false;
  mutable float pose_cache_T[12];
  mutable double pose_cache_smin = 0.0;
  mutable InvMap pose_cache_im;
};

// per-query state of a directed pair, kept from one outer iteration to the next (source order): partner position, certificate
// bound (see k_nn_certify), and the list of queries the certificate did not settle
struct PairState {
  DevBuf<int> match, match2;       // partner, runner-up of the last search (or -1)
  DevBuf<float> lbe;
  DevBuf<unsigned> todo_count;     // todo_count[0..1] = lengths of the two lists of the current search (lists: handle scratch)
  size_t n = 0;
  long long jbase = -1;
  unsigned long long src_gen = 0, tgt_gen = 0;
  bool fresh = true;           // no search has filled the state yet
  bool prune = true;           // the key kernel settles the far list's queries with an empty block (until that stops paying: sort_query_keys_pruned)
  double settled_frac = 1.0;   // share of the queries the last k_nn_certify settled, and the clouds' motion bound of that iteration (certify_now)
  double certify_motion = 0.0;
  double matched_frac = 1.0;   // share of the queries the last search found a partner for
  // motion of the source's queries relative to the target since the state was created (MotionBound, e3d_icp_kernels.hpp): a query
  // at distance rho from the source's bounding-box centre has moved at most mA * rho + mB in the target's frame
  double mA = 0.0, mB = 0.0;
  // resident correspondence rows (LmSet): one row per query at its source position, rewritten only where the partner changed
  DevBuf<float4> pA, pB, pC;
  DevBuf<int> plane_match;         // the partner each row encodes (-1: zero row)
  DevBuf<unsigned> glist;          // active 64-row groups of the last update
  bool rows_valid = false;         // plane_match describes the planes (same sorted orders, same frames)
  bool src_global = false, tgt_global = false;   // the half is stored in the global frame (its cloud never moves) at pose *_T
  float src_T[12], tgt_T[12];
};

struct PairJob {
  int src, tgt;            // indices into the handle's cloud table (fixed = clouds.size())
  int impl_src, impl_tgt;
  long long count = 0;      // this rank's correspondences
  long long gcount = 0;     // all ranks' (what the reference prints; decides which pairs enter the LM system)
  double dsum = 0.0;
  bool dsum_f32 = false;    // dsum holds the reference's sequential f32 sum (e3d_icp_set_sequential_distance_sum)
  size_t corr_off = 0;
  bool mine = true;
  PairState* resident = nullptr;   // the pair's rows are the resident ones of this state (else: compacted into cA / cB / cC)
  long long vrows = 0;             // resident: 64 * active groups, the rows an LM pass walks
};

// The ten damped solves of an LM step (lm_compute) are independent: a few host threads per handle take them side by side once the
// system is large enough to matter (90 unknowns: 10 x 0.09 ms one after the other).  Workers sleep between steps.
class SolvePool {
 public:
  explicit SolvePool(int workers) {
    for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { run(); });
  }
  ~SolvePool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // fn(i) for i in [0, n), the caller's thread included; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    { std::lock_guard<std::mutex> g(m_); fn_ = &fn; next_ = 0; n_ = n; pending_ = n; ++epoch_; }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  void work() {
    for (;;) {
      int i;
      const std::function<void(int)>* fn;
      { std::lock_guard<std::mutex> g(m_); if (!fn_ || next_ >= n_) return; i = next_++; fn = fn_; }
      (*fn)(i);
      { std::lock_guard<std::mutex> g(m_); if (--pending_ == 0) done_.notify_all(); }
    }
  }
  void run() {
    unsigned long long seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> g(m_); cv_.wait(g, [&] { return stop_ || epoch_ != seen; }); if (stop_) return; seen = epoch_; }
      work();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int next_ = 0, n_ = 0, pending_ = 0;
  unsigned long long epoch_ = 0;
  bool stop_ = false;
};

}  // namespace e3d

using namespace e3d;

struct e3d_icp {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<std::unique_ptr<Cloud>> clouds;   // movable clouds
  std::unique_ptr<Cloud> fixed;                 // merged fixed cloud (global frame)
  int max_inner = 150;
  // the progress line's "avg. distance" from the reference's own sum: f32, sequential, in original source order (one device ->
  // host copy of n floats and a host loop per pair; off: f64 sum on the device, which does not stagnate for large clouds)
  bool sequential_dsum = [] { const char* e = getenv("E3D_ICP_SEQUENTIAL_DISTANCE_SUM"); return e && e[0] == '1'; }();
  DevBuf<float> d2_by_orig;
  std::vector<float> h_d2_by_orig;
  int nn_mode = 0;                              // 0 auto, 1 per-query kernel, 2 hash-table bucket kernel, 3 dense-directory row kernel, 4 row kernel with MFMA filter
  // max cells of a dense directory (4 B each; default 2^33 = 32 GB of the 288 GB per cloud, E3D_DENSE_CELLS overrides);
  // hash table beyond
  size_t dense_cell_budget = [] { const char* e = getenv("E3D_DENSE_CELLS"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 33; }();
  int rank = 0, world = 1;
  e3d_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  e3d_comm* comm = nullptr;                     // native RCCL collectives (e3d_icp_set_comm); not owned
  DevBuf<double> d_red;                         // staging of small host buffers for the collectives

  // scratch
  DevBuf<float> bbox_partial, bbox_out;
  PinBuf<float> h_bbox;
  DevBuf<unsigned long long> keys_a, keys_b;
  DevBuf<unsigned> vals_a, vals_b, counter;
  DevBuf<char> sort_temp;
  DevBuf<int> match_pos;
  DevBuf<float> match_d2;
  DevBuf<unsigned> block_counts, block_offsets, block_groups, chunk_groups;
  DevBuf<unsigned char> block_done;          // row-update blocks the certificate kernel settled whole (find_pairs_multi)
  DevBuf<double> block_d2, chunk_d2;
  DevBuf<unsigned long long> d_total, chunk_sum;
  DevBuf<double> d_total_d2;
  PinBuf<unsigned long long> h_total;
  PinBuf<double> h_total_d2;
  DevBuf<float4> cA, cB, cC;
  size_t corr_used = 0;
  // e3d_icp_set_resident_rows / E3D_ICP_RESIDENT=0: compacted planes rewritten every outer iteration for every pair (the round-3
  // data flow; A/B timing, tests)
  bool resident_rows = [] { const char* e = getenv("E3D_ICP_RESIDENT"); return !(e && e[0] == '0'); }();
  bool resident_now = false;                    // this outer iteration keeps resident rows for the pairs of the certificate path
  int lm_prev_end_step = -1;                    // LM step at which the previous outer iteration's LM ended with ten rejections (-1: none yet)
  std::unique_ptr<SolvePool> solve_pool;        // host threads of the LM step's damped solves (systems of >= 30 unknowns)
  DevBuf<LmSet> d_sets;
  PinBuf<LmSet> h_sets;
  DevBuf<LmPose> d_poses;
  PinBuf<LmPose> h_poses;
  DevBuf<int> d_block_set;
  DevBuf<double> d_partial, d_setsum;
  PinBuf<double> h_setsum;
  std::unique_ptr<EventTimer> lm_timer, nn_timer, nn_timer_c;
  // stop-watch of a kernel group whose readings are taken at the end of the NN phase: every start / stop pair gets its own events,
  // so timing a kernel adds no host wait (the all-pairs job launches hundreds of each per iteration, round 4)
  struct LazyTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double acc = 0.0;
    void start(hipStream_t s) {
      if (used == ev.size()) { hipEvent_t a, b; E3D_HIP(hipEventCreate(&a)); E3D_HIP(hipEventCreate(&b)); ev.emplace_back(a, b); }
      E3D_HIP(hipEventRecord(ev[used].first, s));
    }
    void stop(hipStream_t s) { E3D_HIP(hipEventRecord(ev[used].second, s)); ++used; }
    double take() {
      for (size_t i = 0; i < used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(ev[i].second) == hipSuccess && hipEventElapsedTime(&t, ev[i].first, ev[i].second) == hipSuccess) acc += (double)t;
      }
      used = 0;
      const double v = acc; acc = 0.0; return v;
    }
    ~LazyTimer() { for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } }
  } tm_sort, tm_scan, tm_compact, tm_bounded, tm_certify, tm_search;
  // per-pair scratch of a batch of directed pairs (find_pairs_batched): squared distances and the two todo lists of the certificate
  // search -- what one pair's kernels hand to the next kernel of the same pair
  struct PairSlot { DevBuf<float> match_d2; DevBuf<unsigned> todo_near, todo_far; };
  std::vector<std::unique_ptr<PairSlot>> slots;
  PinBuf<unsigned> h_todo_all;
  DevBuf<unsigned> d_todo_all;                  // list lengths of a batch (two words per pair), cleared once per batch
  PinBuf<NnBatchDev> h_batch;                   // the pair table of find_pairs_multi
  DevBuf<NnBatchDev> d_batch;
  DevBuf<unsigned> chunk_rewritten;
  DevBuf<unsigned long long> d_totals_all;
  DevBuf<double> d_d2_all;
  PinBuf<unsigned long long> h_totals_all;
  PinBuf<double> h_d2_all;

  std::map<std::pair<int, int>, std::unique_ptr<PairState>> pair_state;
  PinBuf<unsigned> h_todo;
  DevBuf<unsigned> todo_near, todo_far;         // queries the certificates did not settle (per search; shared by all pairs)
  DevBuf<unsigned> prune_count;                 // number of (key, query) pairs k_query_keys_prune kept
  PinBuf<unsigned> h_prune_count;
  size_t last_corr_total = 0;                   // correspondences of the previous outer iteration (sizes the planes)
  long long nn_global_bound_updates = 0;        // pose updates of a pair that fell back to the clouds' global motion bound (not near-rigid poses)
  DevBuf<unsigned long long> nn_stats;
  DevBuf<float> lbe_scratch;

  std::vector<e3d_icp_pair_record> pair_records;
  std::vector<e3d_icp_iter_record> iter_records;

  ~e3d_icp() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace e3d {

static Affine to_affine(const float* T) {
  Affine a;
  for (int i = 0; i < 12; ++i) a.m[i] = T[i];
  return a;
}

static void sync(e3d_icp* h) { E3D_HIP(hipStreamSynchronize(h->stream)); }

static void upload_cloud(e3d_icp* h, Cloud& c, const float* xyz, const float* nrm, size_t n) {
  c.n = n;
  c.raw_xyz.reserve(3 * n);
  c.raw_nrm.reserve(3 * n);
  copy_in(c.raw_xyz.p, xyz, sizeof(float) * 3 * n, h->stream);
  copy_in(c.raw_nrm.p, nrm, sizeof(float) * 3 * n, h->stream);
  sync(h);
}

static void ensure_bbox_scratch(e3d_icp* h) {
  h->bbox_partial.reserve(6 * (size_t)kMaxBboxBlocks);
  h->bbox_out.reserve(6);
  h->h_bbox.reserve(6);
}

// E3D_NN_HALF: 0 = no half-cell directory (A/B timing), 1 = for dense clouds (default), 2 = always (tests)
static int half_mode() { static const int v = [] { const char* e = getenv("E3D_NN_HALF"); return e ? atoi(e) : 1; }(); return v; }

// Build the static local-frame grid of a cloud for search radius d (global frame).
static void build_grid(e3d_icp* h, Cloud& c, float d) {
  hipStream_t s = h->stream;
  ensure_bbox_scratch(h);
  const size_t n = c.n;
  // local bbox
  if (n > 0) {
    launch_bbox_aos(c.raw_xyz.p, n, h->bbox_partial.p, h->bbox_out.p, s);
    copy_out(h->h_bbox.p, h->bbox_out.p, sizeof(float) * 6, s);
    sync(h);
    for (int k = 0; k < 3; ++k) { c.lmin[k] = h->h_bbox.p[k]; c.lmax[k] = h->h_bbox.p[3 + k]; }
  } else {
    for (int k = 0; k < 3; ++k) { c.lmin[k] = 0.f; c.lmax[k] = 0.f; }
  }
  // cell size: local search radius (r / sigma_min) plus slack for the f32 global->local mapping and the
  // f32 cell-index computation (derivation in DESIGN.md "NN search: exactness of the candidate set").
  const double r = (double)d;
  double smin = min_singular_value_3x3(c.T);
  if (!(smin > 1e-12)) smin = 1e-12;
  double Linv[9];
  double ninv = 1.0;
  if (invert_3x3(c.T, Linv)) {
    ninv = 0;
    for (int i = 0; i < 3; ++i) ninv = std::max(ninv, std::fabs(Linv[3 * i]) + std::fabs(Linv[3 * i + 1]) + std::fabs(Linv[3 * i + 2]));
  }
  double m_local = 0, nL = 0, m_t = 0;
  for (int k = 0; k < 3; ++k) {
    m_local = std::max(m_local, std::max(std::fabs((double)c.lmin[k]), std::fabs((double)c.lmax[k])));
    nL = std::max(nL, std::fabs((double)c.T[4 * k]) + std::fabs((double)c.T[4 * k + 1]) + std::fabs((double)c.T[4 * k + 2]));
    m_t = std::max(m_t, std::fabs((double)c.T[4 * k + 3]));
  }
  const double m_global = nL * m_local + m_t + r;
  const double r_local = r / smin;
  const double slack = 16.0 * FLT_EPSILON * (m_global * ninv + m_local + r_local);
  c.build_slack = slack;
  double extent = 0;
  for (int k = 0; k < 3; ++k) extent = std::max(extent, (double)c.lmax[k] - (double)c.lmin[k]);
  double cell = (r_local + slack) * (1.0 + 1e-3 + 8.0 * FLT_EPSILON * (extent / std::max(r_local, 1e-30) + 4.0));
  // 21 bits per axis: enlarge the cell if the extent would not fit (keeps exactness, costs candidates)
  const double max_cells = (double)((1 << 21) - 8);
  if (extent / cell > max_cells) cell = extent / max_cells;
  if (!(cell > 0) || !std::isfinite(cell)) cell = 1.0;
  c.grid.inv_cell = (float)(1.0 / cell);
  // make sure the f32 inverse does not shrink the effective cell below the bound
  while (1.0 / (double)c.grid.inv_cell < (r_local + slack) * (1.0 + 5e-4)) c.grid.inv_cell = std::nextafter(c.grid.inv_cell, 0.f);
  for (int k = 0; k < 3; ++k) c.grid.origin[k] = (float)((double)c.lmin[k] - 2.0 * cell);

  c.L4.reserve(n); c.LN.reserve(n); c.G4.reserve(n);
  c.G4_valid = false;                                      // the sorted order changes
  c.generation = g_grid_generation.fetch_add(1, std::memory_order_relaxed) + 1;
  c.cum_motion = 0.0; c.last_motion = 0.0; c.err_max = 0.0;
  unsigned n_cells = 0;
  if (n > 0) {
    h->keys_a.reserve(n); h->keys_b.reserve(n); h->vals_a.reserve(n); h->vals_b.reserve(n);
    h->counter.reserve(1);
    if (half_mode() != 0) {
      // sub-cell order inside every cell: sort by the 3-bit half-cell code first, then (stable) by the cell key
      unsigned* fk_a = reinterpret_cast<unsigned*>(h->keys_b.p);
      unsigned* fk_b = fk_a + n;
      launch_half_keys(c.raw_xyz.p, n, c.grid, fk_a, h->vals_b.p, s);
      sort_pairs_u32_u32(fk_a, fk_b, h->vals_b.p, h->vals_a.p, n, 3, h->sort_temp, s);
      launch_cell_keys_ordered(c.raw_xyz.p, h->vals_a.p, n, c.grid, h->keys_a.p, s);
    } else {
      launch_cell_keys(c.raw_xyz.p, n, c.grid, h->keys_a.p, h->vals_a.p, s);
    }
    sort_pairs_u64_u32(h->keys_a.p, h->keys_b.p, h->vals_a.p, h->vals_b.p, n, 63, h->sort_temp, s);
    launch_permute(c.raw_xyz.p, c.raw_nrm.p, h->vals_b.p, n, c.L4.p, c.LN.p, s);
    E3D_HIP(hipMemsetAsync(h->counter.p, 0, sizeof(unsigned), s));
    launch_count_cells(h->keys_b.p, n, h->counter.p, s);
    E3D_HIP(hipMemcpyAsync(&n_cells, h->counter.p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    sync(h);
  }
  size_t tsize = 64;
  while (tsize < 2 * (size_t)n_cells) tsize <<= 1;
  c.table.reserve(tsize);
  c.grid.mask = (unsigned)(tsize - 1);
  E3D_HIP(hipMemsetAsync(c.table.p, 0xFF, sizeof(HashEntry) * tsize, s));
  if (n > 0) launch_build_table(h->keys_b.p, n, c.table.p, c.grid.mask, s);
  sync(h);
  // dense query-key range: cells of the stored points +- 2 (one for the 27-neighbourhood, one for host/device
  // rounding of the bbox corners)
  double prod = 1.0;
  for (int k = 0; k < 3; ++k) {
    const int cmin = (int)std::floor(((float)c.lmin[k] - c.grid.origin[k]) * c.grid.inv_cell);
    const int cmax = (int)std::floor(((float)c.lmax[k] - c.grid.origin[k]) * c.grid.inv_cell);
    c.qrange.lo[k] = cmin - 2;
    c.qrange.D[k] = (unsigned)(cmax - cmin + 5);
    prod *= (double)c.qrange.D[k];
  }
  c.key_bits = 1;
  while (c.key_bits < 63 && std::ldexp(1.0, c.key_bits) <= prod + 1.0) ++c.key_bits;
  c.n_cells = n_cells;
  // dense cell-start directory (4 B per cell of the bounding grid) when it fits the budget: replaces 27 random
  // hash probes per cell group by one coalesced lookup.  288 GB of HBM make this the default; huge sparse grids
  // keep the hash table.
  c.has_dense = false;
  if (n > 0 && prod + 2.0 <= (double)h->dense_cell_budget) {
    const size_t ncell = (size_t)prod;
    c.dense_start.reserve(ncell + 2);
    E3D_HIP(hipMemsetAsync(c.dense_start.p, 0, sizeof(unsigned) * (ncell + 2), s));
    launch_dense_counts(h->keys_b.p, n, c.qrange, c.dense_start.p, s);
    exclusive_max_scan_u32(c.dense_start.p, ncell + 2, h->sort_temp, s);
    sync(h);
    c.has_dense = true;
  }
  // half-cell directory for the bounded search: clouds with several points per cell (the others gain nothing from it); 8 bytes
  // per cell of the bounding grid, twice the cell directory -- so it has to fit the same budget three times over (4 + 8 B per
  // cell), and a failed allocation only costs the directory: k_nn_bounded works on whole cells without it
  c.has_half = false;
  if (c.has_dense && half_mode() != 0 && n > 0 && 3.0 * (prod + 2.0) <= (double)h->dense_cell_budget &&
      ((double)n >= 4.0 * (double)std::max(n_cells, 1u) || half_mode() == 2 || h->nn_mode == 5)) {
    const size_t ncell = (size_t)prod;
    bool ok = true;
    try { c.half_prefix.reserve(ncell + 2); }
    catch (const Error&) { ok = false; (void)hipGetLastError(); c.half_prefix.release(); }
    if (ok) {
      E3D_HIP(hipMemsetAsync(c.half_prefix.p, 0, sizeof(unsigned long long) * (ncell + 2), s));
      launch_half_prefix(h->keys_b.p, c.L4.p, n, c.grid, c.qrange, c.dense_start.p, c.half_prefix.p, s);
      sync(h);
      c.has_half = true;
    }
  }
  if (!c.has_half) c.half_prefix.release();
  // occupancy bits of the 27-cell blocks (E3D_NN_PRUNE=0: none): cells / 8 bytes, the unpadded bits of the build in the sort's key
  // buffer when it is large enough
  c.has_occ = false;
  static const bool want_occ = [] { const char* e = getenv("E3D_NN_PRUNE"); return !(e && e[0] == '0'); }();
  if (c.has_dense && want_occ && n > 0) {
    const unsigned stride_w = (c.qrange.D[0] + 31u) / 32u;
    const size_t words = (size_t)c.qrange.D[2] * c.qrange.D[1] * stride_w;
    try {
      DevBuf<unsigned> tmp;
      unsigned* t = reinterpret_cast<unsigned*>(h->keys_a.p);
      if (words * sizeof(unsigned) > h->keys_a.cap * sizeof(unsigned long long)) { tmp.reserve(words); t = tmp.p; }
      c.occ27.reserve(words);
      launch_block_occupancy(c.dense_start.p, c.qrange, stride_w, t, c.occ27.p, s);
      if (tmp.p) sync(h);                            // (the temporary is freed on return)
      c.occ_stride = stride_w;
      c.has_occ = true;
    } catch (const Error&) { (void)hipGetLastError(); }
  }
  if (!c.has_occ) c.occ27.release();
  c.grid_valid = true;
  c.grid_radius = d;
  std::memcpy(c.grid_T, c.T, sizeof c.grid_T);
}

// The grid stays valid while the search radius is unchanged and the pose's linear part has not drifted
// in scale (ICP only left-multiplies rotations, so sigma_min is constant up to rounding).
static bool grid_usable(const Cloud& c, float d) {
  if (!c.grid_valid || c.grid_radius != d) return false;
  const double s0 = min_singular_value_3x3(c.grid_T), s1 = min_singular_value_3x3(c.T);
  return std::fabs(s0 - s1) <= 1e-4 * std::max(s0, 1e-30);
}

static InvMap make_invmap_uncached(const Cloud& c);
static void refresh_pose_cache(const Cloud& c) {
  if (c.pose_cache_valid && std::memcmp(c.pose_cache_T, c.T, sizeof c.T) == 0) return;
  c.pose_cache_smin = min_singular_value_3x3(c.T);
  c.pose_cache_im = make_invmap_uncached(c);
  std::memcpy(c.pose_cache_T, c.T, sizeof c.T);
  c.pose_cache_valid = true;
}
static double cloud_smin(const Cloud& c) { refresh_pose_cache(c); return c.pose_cache_smin; }
static InvMap make_invmap(const Cloud& c) { refresh_pose_cache(c); return c.pose_cache_im; }
static InvMap make_invmap_uncached(const Cloud& c) {
  InvMap im;
  double Linv[9];
  if (!invert_3x3(c.T, Linv)) { for (int i = 0; i < 9; ++i) Linv[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int i = 0; i < 9; ++i) im.Linv[i] = (float)Linv[i];
  for (int k = 0; k < 3; ++k) im.t[k] = c.T[4 * k + 3];
  return im;
}

static void transform_cloud(e3d_icp* h, Cloud& c) {
  ensure_bbox_scratch(h);
  if (c.n == 0) {
    for (int k = 0; k < 3; ++k) { c.bmin[k] = FLT_MAX; c.bmax[k] = -FLT_MAX; }
    return;
  }
  if (c.G4_valid && std::memcmp(c.G4_T, c.T, sizeof c.G4_T) == 0) return;     // same bits in, same bits out
  launch_transform_bbox(c.L4.p, c.n, to_affine(c.T), c.G4.p, h->bbox_partial.p, h->bbox_out.p, h->stream);
  copy_out(h->h_bbox.p, h->bbox_out.p, sizeof(float) * 6, h->stream);
  sync(h);
  for (int k = 0; k < 3; ++k) { c.bmin[k] = h->h_bbox.p[k]; c.bmax[k] = h->h_bbox.p[3 + k]; }
  std::memcpy(c.G4_T, c.T, sizeof c.G4_T);
  c.G4_valid = true;
}

static bool bbox_intersects(const Cloud& a, const Cloud& b) {
  // !a.bbox.intersection(b.bbox).isEmpty()   (icp_point_to_plane.cc:214-215)
  for (int k = 0; k < 3; ++k) {
    const float lo = std::max(a.bmin[k], b.bmin[k]);
    const float hi = std::min(a.bmax[k], b.bmax[k]);
    if (lo > hi) return false;
  }
  return true;
}

static inline float radius_sq(float d) {
  const double r = (double)d;   // pcl::KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius)
  return (float)(r * r);
}

static double env_double(const char* name, double dflt) {
  const char* e = getenv(name);
  return e ? atof(e) : dflt;
}
static float round_up_f(double v) {
  float f = (float)v;
  if ((double)f < v) f = std::nextafter(f, FLT_MAX);
  return f;
}
static float round_down_f(double v) {
  float f = (float)v;
  if ((double)f > v) f = std::nextafter(f, -FLT_MAX);
  return f;
}

// State a directed pair keeps between outer iterations (both clouds' sorted orders are static while their grids are):
// reset whenever a grid was rebuilt or the slice changed.
static PairState& pair_state_for(e3d_icp* h, int src_id, int tgt_id, const Cloud& src, const Cloud& tgt, size_t j0, size_t n) {
  std::unique_ptr<PairState>& up = h->pair_state[std::make_pair(src_id, tgt_id)];
  if (!up) up.reset(new PairState());
  PairState& ps = *up;
  if (ps.n != n || ps.jbase != (long long)j0 || ps.src_gen != src.generation || ps.tgt_gen != tgt.generation) {
    ps.match.reserve(n); ps.match2.reserve(n); ps.lbe.reserve(n); ps.todo_count.reserve(2);
    ps.n = n; ps.jbase = (long long)j0; ps.src_gen = src.generation; ps.tgt_gen = tgt.generation;
    ps.fresh = true; ps.prune = true; ps.settled_frac = 1.0; ps.certify_motion = 0.0; ps.matched_frac = 1.0;
    ps.rows_valid = false;
    ps.mA = 0.0; ps.mB = 0.0;
  }
  return ps;
}

// bound of how far any point of a cloud moves when its pose changes from T0 to T1 (global frame): with c the centre of the cloud's
// local bounding box and R its half diagonal, dL p + dt = dL (p - c) + (dL c + dt), so |.| <= ||dL||_2 R + |dL c + dt| -- the
// second term is what the centre really moves (a rotation about the cloud's middle costs only the first)
static double pose_motion_bound(const Cloud& c, const float* T0, const float* T1) {
  float dT[12];
  double ctr[3], R2 = 0;
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) dT[4 * r + k] = (float)((double)T1[4 * r + k] - (double)T0[4 * r + k]);
    dT[4 * r + 3] = 0.f;
    ctr[r] = 0.5 * ((double)c.lmin[r] + (double)c.lmax[r]);
    const double hw = 0.5 * ((double)c.lmax[r] - (double)c.lmin[r]);
    R2 += hw * hw;
  }
  double dc2 = 0;
  for (int r = 0; r < 3; ++r) {
    double e = (double)T1[4 * r + 3] - (double)T0[4 * r + 3];
    for (int k = 0; k < 3; ++k) e += ((double)T1[4 * r + k] - (double)T0[4 * r + k]) * ctr[k];
    dc2 += e * e;
  }
  double fro = 0;
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) fro += (double)dT[4 * r + k] * (double)dT[4 * r + k];
  double smax = max_singular_value_3x3(dT) * (1.0 + 1e-6) + 1e-12 * std::sqrt(fro);
  if (!(smax <= std::sqrt(fro))) smax = std::sqrt(fro);      // Frobenius norm bounds the spectral norm (also the NaN fallback)
  return (smax * std::sqrt(R2) + std::sqrt(dc2)) * (1.0 + 1e-9);
}
// bound of the f32 rounding error (Euclidean) of one evaluation of pcl_se3(T, p) for a point of the cloud: three products and
// three sums per component, |error| <= 4 u (|L_row| |p| + |t_r|) with u = 2^-24 (5 u taken)
static double pose_rounding_bound(const Cloud& c, const float* T) {
  double fro = 0, tt = 0, R2 = 0;
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) fro += (double)T[4 * r + k] * (double)T[4 * r + k];
    tt += (double)T[4 * r + 3] * (double)T[4 * r + 3];
    const double m = std::max(std::fabs((double)c.lmin[r]), std::fabs((double)c.lmax[r]));
    R2 += m * m;
  }
  return 2.5 * FLT_EPSILON * (std::sqrt(fro) * std::sqrt(R2) + std::sqrt(tt));
}

// E3D_NN_PERQUERY=0: the certificates use the clouds' global motion bounds (rounds 2 - 4) instead of the bound per query
static bool per_query_motion() { static const bool on = [] { const char* e = getenv("E3D_NN_PERQUERY"); return !(e && e[0] == '0'); }(); return on; }
// largest entry of |L^T L - I|: the singular values of the pose's linear part lie within sqrt(1 -+ 3 dev).  Poses are re-composed in
// f32 every outer iteration (Tn = R * T) and never re-orthonormalised, so dev grows by ~1e-7 per iteration: the bound per query is
// scaled by what dev allows (accumulate_pair_motion) instead of being switched off at a fixed tolerance (at 4e-6 a
// long run silently fell back to the clouds' global bound).  NaN poses give NaN -> not near-rigid.
static double pose_ortho_dev(const float* T) {
  double dev = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j) {
      double d = 0;
      for (int r = 0; r < 3; ++r) d += (double)T[4 * r + i] * (double)T[4 * r + j];
      const double e = std::fabs(d - (i == j ? 1.0 : 0.0));
      if (!(e <= dev)) dev = e;
    }
  return dev;
}
constexpr double kNearRigidDev = 1e-3;
// M = Tt^-1 Ts (3 x 4, row-major): the source's local frame -> the target's
static bool relative_map(const float* Ts, const float* Tt, double M[12]) {
  double Li[9];
  if (!invert_3x3(Tt, Li)) return false;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) M[4 * r + c] = Li[3 * r] * (double)Ts[c] + Li[3 * r + 1] * (double)Ts[4 + c] + Li[3 * r + 2] * (double)Ts[8 + c];
    M[4 * r + 3] = Li[3 * r] * ((double)Ts[3] - (double)Tt[3]) + Li[3 * r + 1] * ((double)Ts[7] - (double)Tt[7]) + Li[3 * r + 2] * ((double)Ts[11] - (double)Tt[11]);
  }
  return true;
}
// what a pose update (Ts0, Tt0) -> (Ts1, Tt1) adds to a pair's accumulators: rigid poses -> the bound per query, else the
// clouds' global bounds of this update (glob) into b alone
// (returns false when the update went into the global bound: a pose that is not near-rigid -- counted, E3D_NN_STATS)
static bool accumulate_pair_motion(PairState& ps, const Cloud& src, const Cloud& tgt, const float* Ts0, const float* Ts1, const float* Tt0, const float* Tt1, double glob) {
  double M0[12], M1[12];
  const double ds = std::max(pose_ortho_dev(Ts0), pose_ortho_dev(Ts1)), dt = std::max(pose_ortho_dev(Tt0), pose_ortho_dev(Tt1));
  if (!per_query_motion()) { ps.mB += glob; return true; }
  if (!(ds <= kNearRigidDev) || !(dt <= kNearRigidDev) || !relative_map(Ts0, Tt0, M0) || !relative_map(Ts1, Tt1, M1)) {
    ps.mB += glob;
    return false;
  }
  // The motion is bounded in the target's LOCAL frame (a |p - c| + b, p - c in the source's local frame); the kernels measure
  // rho = |q - cs| and distances in the GLOBAL frame: |p - c| <= rho / sigma_min(L_src), a global length <= sigma_max(L_tgt) x the
  // local one, and a distance d between resting points reads (sigma(L_tgt at T1) - sigma(L_tgt at T0)) d differently after the
  // update -- d <= 2 x the search radius for every distance a certificate compares.
  const double f_t = std::sqrt(1.0 + 3.0 * dt), f_st = f_t / std::sqrt(1.0 - 3.0 * ds);
  const double rescale = 3.0 * dt * 2.0 * (double)std::max(tgt.grid_radius, 0.f);
  float dM[12];
  double fro = 0, dc2 = 0;
  for (int r = 0; r < 3; ++r) {
    double e = M1[4 * r + 3] - M0[4 * r + 3];
    for (int k = 0; k < 3; ++k) {
      const double d = M1[4 * r + k] - M0[4 * r + k];
      dM[4 * r + k] = (float)d; fro += d * d;
      e += d * 0.5 * ((double)src.lmin[k] + (double)src.lmax[k]);
    }
    dM[4 * r + 3] = 0.f;
    dc2 += e * e;
  }
  double smax = max_singular_value_3x3(dM) * (1.0 + 1e-6) + 1e-7 * std::sqrt(fro);      // (dM was rounded to f32: 6e-8 relative)
  if (!(smax <= std::sqrt(fro) * (1.0 + 1e-6))) smax = std::sqrt(fro) * (1.0 + 1e-6);   // Frobenius norm bounds the spectral norm (also the NaN fallback)
  ps.mA += smax * f_st * (1.0 + 1e-9);
  ps.mB += (std::sqrt(dc2) * f_t + rescale) * (1.0 + 1e-9);
  return true;
}
// the two roundings of a pair's accumulated bound, and the source centre's global position (rho is measured from it)
static void pair_motion_bounds(const PairState& ps, const Cloud& src, const Cloud& tgt, MotionBound& lo, MotionBound& up) {
  const double err = 2.0 * (src.err_max + tgt.err_max);
  if (per_query_motion()) {
    lo.a = round_down_f(ps.mA * (1.0 - 2e-6)); lo.b = round_down_f(ps.mB * (1.0 - 2e-6));
    up.a = round_up_f(ps.mA * (1.0 + 2e-6)); up.b = round_up_f((ps.mB * (1.0 + 2e-6) + err) * (1.0 + 1e-6));
    double cmax = 0;
    for (int r = 0; r < 3; ++r) {
      double v = (double)src.T[4 * r + 3];
      for (int k = 0; k < 3; ++k) v += (double)src.T[4 * r + k] * 0.5 * ((double)src.lmin[k] + (double)src.lmax[k]);
      lo.cs[r] = up.cs[r] = (float)v;
      cmax = std::max(cmax, std::fabs(v));
    }
    // error of rho = |q - cs| in a kernel: q is the f32 transform of a source point (err_max: it grows with the coordinates'
    // magnitude -- ADVICE round 5: a fixed 5e-5 m stopped covering it a few hundred metres from the origin), cs was rounded to f32
    // (half an ulp per component); the three subtractions, the squares and the root are inside the kernels' relative 2e-6
    lo.rho_err = up.rho_err = round_up_f((src.err_max + 2.0 * FLT_EPSILON * cmax) * (1.0 + 1e-6) + 1e-7);
  } else {
    const double cum_pair = src.cum_motion + tgt.cum_motion;
    lo.a = up.a = 0.f;
    lo.b = round_down_f(cum_pair * (1.0 - 2e-6));
    up.b = round_up_f((cum_pair * (1.0 + 2e-6) + err) * (1.0 + 1e-6));
    for (int r = 0; r < 3; ++r) lo.cs[r] = up.cs[r] = 0.f;
    lo.rho_err = up.rho_err = 0.f;
  }
}

// certificate constants of k_nn_rows for a target at its current pose
static CertParams make_cert_params(const Cloud& tgt, const MotionBound& lo) {
  CertParams cp;
  double smin = cloud_smin(tgt);
  if (!(smin > 0)) smin = 0;
  const double cell = 1.0 / (double)tgt.grid.inv_cell;
  double extent = 0;
  for (int k = 0; k < 3; ++k) extent = std::max(extent, (double)tgt.lmax[k] - (double)tgt.lmin[k]);
  cp.cell_scale = round_down_f(smin * cell * (1.0 - 1e-5));
  // mapping error of the query + rounding of the stored points' global coordinates (both inside the build's slack), and the
  // f32 fuzz of the cell boundaries
  cp.cell_sub = round_up_f(smin * (2.0 * tgt.build_slack + 8.0 * FLT_EPSILON * (extent + 4.0 * cell)) * (1.0 + 1e-5));
  cp.lo = lo;
  return cp;
}

// dense-directory row kernels: plain (mode 3 / auto) or with the MFMA filter (mode 4); identical results.  The plain kernel
// writes its results (and the certificate bounds) at the queries' source positions, the filtered one in the sorted order.
static bool launch_rows(int mode, const Cloud& tgt, const float4* srcG, const unsigned* order, size_t n, const InvMap& im, float r2,
                        const CertParams& cert, int* match_pos, float* match_d2, float* lbe, int* match2, hipStream_t s) {
  MfParams P;
  if (mode == 4 && mfma_filter_params(1.0 / (double)tgt.grid.inv_cell, max_singular_value_3x3(tgt.T) * (1.0 + 1e-6), nn_row_span(), r2, &P)) {
    launch_nn_mfma(srcG, order, n, tgt.G4.p, tgt.dense_start.p, tgt.grid, im, tgt.qrange, r2, P, match_pos, match_d2, s);
    return false;
  }
  launch_nn_rows(srcG, order, n, tgt.G4.p, tgt.dense_start.p, tgt.grid, im, tgt.qrange, r2, cert, match_pos, match_d2, lbe, match2, s);
  return true;
}

static void sort_query_keys(e3d_icp* h, const Cloud& tgt, const float4* srcG, const unsigned* list, size_t n, const InvMap& im) {
  hipStream_t s = h->stream;
  h->keys_a.reserve(n); h->keys_b.reserve(n); h->vals_a.reserve(n); h->vals_b.reserve(n);
  h->tm_sort.start(s);
  struct Stop { e3d_icp* h; hipStream_t s; ~Stop() { h->tm_sort.stop(s); } } stop_at_return{h, s};
  if (tgt.key_bits <= 31) {   // 8-byte (key, index) pairs through the radix passes
    unsigned* ka = reinterpret_cast<unsigned*>(h->keys_a.p);
    unsigned* kb = reinterpret_cast<unsigned*>(h->keys_b.p);
    if (list) launch_query_keys32_list(srcG, list, n, tgt.grid, im, tgt.qrange, ka, h->vals_a.p, s);
    else launch_query_keys32(srcG, n, tgt.grid, im, tgt.qrange, ka, h->vals_a.p, s);
    sort_pairs_u32_u32(ka, kb, h->vals_a.p, h->vals_b.p, n, tgt.key_bits, h->sort_temp, s);
  } else {
    if (list) launch_query_keys_list(srcG, list, n, tgt.grid, im, tgt.qrange, h->keys_a.p, h->vals_a.p, s);
    else launch_query_keys(srcG, n, tgt.grid, im, tgt.qrange, h->keys_a.p, h->vals_a.p, s);
    sort_pairs_u64_u32(h->keys_a.p, h->keys_b.p, h->vals_a.p, h->vals_b.p, n, tgt.key_bits, h->sort_temp, s);
  }
}

// Is k_nn_certify worth a pass over this pair's queries?  While two scans are centimetres apart every certificate breaks with every
// pose update (0.7 ms per outer iteration at 2 x 50 M points for nothing).  After a pass that settled less than 2 % of the queries
// the pair is searched whole -- the state of the last search still tells which queries had no partner (from_state) -- until half
// of the queries have a partner (the scans have met) or the clouds' motion bound of an iteration falls below half of what it was
// at that pass.  A wrong guess costs searches, never a
// result.  E3D_NN_CERT_SKIP=0: always test.
static bool certify_now(const PairState& ps, const Cloud& src, const Cloud& tgt) {
  static const bool allow = [] { const char* e = getenv("E3D_NN_CERT_SKIP"); return !(e && e[0] == '0'); }();
  return !allow || ps.settled_frac >= 0.02 || ps.matched_frac >= 0.5 || (src.last_motion + tgt.last_motion) < 0.5 * ps.certify_motion;
}
static void certify_ran(PairState& ps, const Cloud& src, const Cloud& tgt, size_t n, size_t unsettled) {
  ps.settled_frac = n ? 1.0 - (double)unsettled / (double)n : 1.0;
  ps.certify_motion = src.last_motion + tgt.last_motion;
}

// The certificate path's variant: the key kernel settles the queries whose 27-cell block holds no target point (k_query_keys_prune:
// one bit per query from the target's occupancy bits) and only the others are sorted; returns how many those are (h->vals_b: their
// list entries in key order).  One host round trip for the count -- the sort's size.  It pays while the scans are centimetres
// apart (most blocks empty: 2 x 50 M points, 9.4 -> 5 ms per outer iteration); once nine queries in ten have candidates the pair
// goes back to the plain key kernel (ps.prune; a fresh state starts over).  E3D_NN_PRUNE=0: never.
static size_t sort_query_keys_pruned(e3d_icp* h, PairState& ps, const Cloud& tgt, const float4* srcG, const unsigned* list, size_t n, const InvMap& im,
                                     float r2, const CertParams& cert, float* match_d2, bool from_state) {
  static const size_t min_list = (size_t)env_double("E3D_NN_PRUNE_MIN", 262144.0);
  if (!tgt.has_occ || !ps.prune || n < min_list) { sort_query_keys(h, tgt, srcG, list, n, im); return n; }
  hipStream_t s = h->stream;
  h->keys_a.reserve(n); h->keys_b.reserve(n); h->vals_a.reserve(n); h->vals_b.reserve(n);
  h->prune_count.reserve(1); h->h_prune_count.reserve(1);
  h->tm_sort.start(s);
  struct Stop { e3d_icp* h; hipStream_t s; ~Stop() { h->tm_sort.stop(s); } } stop_at_return{h, s};
  const bool k32 = tgt.key_bits <= 31;
  E3D_HIP(hipMemsetAsync(h->prune_count.p, 0, sizeof(unsigned), s));
  launch_query_keys_prune(k32, srcG, list, n, tgt.occ27.p, tgt.occ_stride, tgt.grid, im, tgt.qrange, r2, cert, h->keys_a.p, h->vals_a.p, h->prune_count.p,
                          ps.match.p, ps.match2.p, match_d2, ps.lbe.p, from_state, s);
  copy_out(h->h_prune_count.p, h->prune_count.p, sizeof(unsigned), s);
  sync(h);
  const size_t kept = h->h_prune_count.p[0];
  if ((double)kept > 0.9 * (double)n) ps.prune = false;
  if (kept == 0) return 0;
  if (k32) sort_pairs_u32_u32(reinterpret_cast<unsigned*>(h->keys_a.p), reinterpret_cast<unsigned*>(h->keys_b.p), h->vals_a.p, h->vals_b.p, kept, tgt.key_bits, h->sort_temp, s, n);
  else sort_pairs_u64_u32(h->keys_a.p, h->keys_b.p, h->vals_a.p, h->vals_b.p, kept, tgt.key_bits, h->sort_temp, s, n);
  return kept;
}

// NN search + compaction for one directed pair; appends to the correspondence planes.
// Multi-GPU: every rank holds all clouds and handles the slice [j0, j1) of the source cloud (cell order).
// E3D_NN_PROFILE=1: wall-clock split of the search of one outer iteration (synchronises between the phases; diagnostics only)
static double g_nn_prof[8];
static bool nn_profile() { static const bool on = [] { const char* e = getenv("E3D_NN_PROFILE"); return e && e[0] == '1'; }(); return on; }
struct NnPhase {
  hipStream_t s; int slot; std::chrono::steady_clock::time_point t0;
  NnPhase(hipStream_t st, int sl) : s(st), slot(sl) { if (nn_profile()) { (void)hipStreamSynchronize(s); t0 = std::chrono::steady_clock::now(); } }
  ~NnPhase() { if (nn_profile()) { (void)hipStreamSynchronize(s); g_nn_prof[slot] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } }
};

// which search a directed pair takes: the certificate path (per-pair state, results in source order) needs the target's dense
// directory; only that path keeps resident correspondence rows
static bool pair_is_dense(const e3d_icp* h, const Cloud& tgt) {
  return h->nn_mode >= 2 || (h->nn_mode == 0 && (double)tgt.n >= 4.0 * (double)std::max(tgt.n_cells, 1u));
}
static bool pair_uses_rows(const e3d_icp* h, const Cloud& tgt) {
  return pair_is_dense(h, tgt) && tgt.has_dense && (h->nn_mode == 0 || h->nn_mode == 3 || h->nn_mode == 5);
}
// the far list (queries without a near partner) goes through the bounded search instead of sort + k_nn_rows when it holds less than
// 1 / E3D_NN_FAR_DIV of the pair's queries (default 32; 0: always the bounded search -- experiments)
static bool far_list_is_short(size_t n_far, size_t n) {
  static const long long div = [] { const char* e = getenv("E3D_NN_FAR_DIV"); return e ? atoll(e) : 32ll; }();
  return n_far > 0 && (div <= 0 || n_far * (size_t)div < n);
}
static size_t resident_rows_cap(size_t n) { return div_up(n, 64) * 64; }
static size_t resident_bytes(size_t n) { return resident_rows_cap(n) * 48 + n * 4 + div_up(n, 64) * 4; }

static void find_pair(e3d_icp* h, Cloud& src, Cloud& tgt, float d, PairJob& job, size_t j0, size_t j1,
                      e3d_icp_iter_record& rec) {
  hipStream_t s = h->stream;
  const size_t n = j1 - j0;
  job.count = 0; job.dsum = 0.0; job.corr_off = h->corr_used;
  if (n == 0 || tgt.n == 0) return;
  const float4* srcG = src.G4.p + j0;
  const float4* srcLN = src.LN.p + j0;
  h->match_d2.reserve(n);
  const size_t nb = div_up(n, kBlock);
  h->block_counts.reserve(nb); h->block_offsets.reserve(nb); h->block_d2.reserve(nb);
  h->d_total.reserve(3); h->d_total_d2.reserve(1); h->h_total.reserve(3); h->h_total_d2.reserve(1);
  if (!h->nn_timer) h->nn_timer.reset(new EventTimer());
  // dense data (many points per cell): queries sorted by target cell + the LDS-bucket kernels; sparse data: one thread per
  // query.  All are exact and return identical results.  The default row kernel keeps a per-query certificate between the
  // outer iterations: a query whose partner of the last iteration is provably still its unique nearest neighbour is settled
  // by k_nn_certify (one gather), only the others are sorted and searched.
  const bool dense = pair_is_dense(h, tgt);
  const bool rows = pair_uses_rows(h, tgt);
  PairState* rstate = nullptr;
  static const bool use_cert = [] { const char* e = getenv("E3D_NN_CERT"); return !(e && e[0] == '0'); }();
  static const bool want_stats = [] { const char* e = getenv("E3D_NN_STATS"); return e && e[0] == '1'; }();
  const unsigned* order = nullptr;
  int* match_pos = nullptr;
  // the timer brackets the search kernels themselves (what rocprofv3 reports for them); keys + sort are part of t_nn_ms
  if (rows) {
    PairState& ps = pair_state_for(h, job.src, job.tgt, src, tgt, j0, n);
    if (h->resident_now) rstate = &ps;
    match_pos = ps.match.p;
    const InvMap im = make_invmap(tgt);
    const double cum_pair = src.cum_motion + tgt.cum_motion;
    MotionBound m_lo, m_up;
    pair_motion_bounds(ps, src, tgt, m_lo, m_up);
    const CertParams cert = make_cert_params(tgt, m_lo);
    size_t n_far = n, n_near = 0;
    const unsigned* list = nullptr;
    const bool test_certificates = !ps.fresh && use_cert && certify_now(ps, src, tgt);
    const bool from_state = !ps.fresh && use_cert && !test_certificates;       // searched whole; the state says who had no partner
    if (test_certificates) {
      NnPhase ph(s, 0);
      static const double margin_frac = env_double("E3D_NN_MARGIN", 0.08), near_frac = env_double("E3D_NN_NEAR", 0.4);   // of the radius
      const MotionBound& cum_up = m_up;
      const float near2 = (float)((near_frac * (double)d) * (near_frac * (double)d));
      h->h_todo.reserve(2);
      h->todo_near.reserve(n); h->todo_far.reserve(n);
      E3D_HIP(hipMemsetAsync(ps.todo_count.p, 0, 2 * sizeof(unsigned), s));
      if (!h->nn_timer_c) h->nn_timer_c.reset(new EventTimer());
      h->nn_timer_c->start(s);
      // Queries without a partner whose "nothing within the radius" bound no longer holds: k_nn_bounded searches them np_extra
      // beyond the radius (cheap where the target is absent, and the bound then survives the following pose updates) -- but only
      // once the poses move little: while the scans are still centimetres apart such a query has the other surface just outside
      // the radius, one thread would scan 64 full cells, and the bound would not survive the next update anyway.
      static const double np_frac = env_double("E3D_NN_NP_EXTRA", 0.5), np_gate = env_double("E3D_NN_NP_GATE", 0.2);   // of the radius
      const bool none_near = np_frac > 0 && (src.last_motion + tgt.last_motion) < np_gate * (double)d;
      launch_nn_certify(srcG, n, tgt.G4.p, cum_up, radius_sq(d), near2, none_near, ps.match.p, ps.match2.p, ps.lbe.p, h->match_d2.p, h->todo_near.p,
                        h->todo_far.p, ps.todo_count.p, s);
      h->nn_timer_c->stop(s);
      copy_out(h->h_todo.p, ps.todo_count.p, 2 * sizeof(unsigned), s);
      sync(h);
      { const double t = h->nn_timer_c->ms(); rec.t_nn_certify_ms += t; rec.t_nn_query_ms += t; }
      rec.nn_certify_launches++; rec.nn_certify_queries += (long long)n; rec.nn_kernel_launches++;
      n_near = h->h_todo.p[0]; n_far = h->h_todo.p[1];
      certify_ran(ps, src, tgt, n, n_near + n_far);
      h->tm_bounded.start(s);                             // (read lazily: timing the bounded search costs no synchronisation)
      list = h->todo_far.p;
      // old partner close by: only the cells its distance (+ margin) reaches, one thread per query, no sort
      double smin = cloud_smin(tgt);
      if (!(smin > 1e-12)) smin = 1e-12;
      double m_local = 0;
      for (int k = 0; k < 3; ++k) m_local = std::max(m_local, std::max(std::fabs((double)tgt.lmin[k]), std::fabs((double)tgt.lmax[k])));
      BoundParams bp;
      bp.margin = (float)(margin_frac * (double)d);
      bp.rho_scale = round_up_f((1.0 + 1e-5) / smin);
      bp.rho_pad = round_up_f(2.0 * tgt.build_slack + 8.0 * FLT_EPSILON * m_local);
      bp.lo = cert.lo;
      bp.cell_scale = cert.cell_scale; bp.cell_sub = cert.cell_sub;
      bp.np_extra = none_near ? (float)(np_frac * (double)d) : 0.f;   // gate closed: partnerless queries on the far list keep the plain radius
      launch_nn_bounded(srcG, h->todo_near.p, n_near, tgt.G4.p, tgt.dense_start.p, tgt.has_half ? tgt.half_prefix.p : nullptr, h->nn_mode == 5, tgt.grid, im, tgt.qrange, radius_sq(d), bp, ps.match.p,
                        ps.match2.p, h->match_d2.p, ps.lbe.p, s);
      if (far_list_is_short(n_far, n)) {
        // few queries without a near partner: the same kernel (whole radius for those without any) instead of sort + row kernel,
        // whose cost is per visited cell row, not per query
        launch_nn_bounded(srcG, h->todo_far.p, n_far, tgt.G4.p, tgt.dense_start.p, tgt.has_half ? tgt.half_prefix.p : nullptr, h->nn_mode == 5, tgt.grid, im, tgt.qrange, radius_sq(d), bp, ps.match.p,
                          ps.match2.p, h->match_d2.p, ps.lbe.p, s);
        n_near += n_far; n_far = 0;
      }
      h->tm_bounded.stop(s);
      if (n_near > 0) { rec.nn_bounded_launches++; rec.nn_bounded_queries += (long long)n_near; rec.nn_kernel_launches++; }
    }
    size_t n_rows = 0;                                     // far-list queries with a candidate in their 27 cells: sorted and searched
    if (n_far > 0) {
      NnPhase ph(s, 1);
      n_rows = sort_query_keys_pruned(h, ps, tgt, srcG, list, n_far, im, radius_sq(d), cert, h->match_d2.p, from_state);
      rec.nn_sort_calls++; rec.nn_kernel_launches++;
    }
    h->nn_timer->start(s);
    if (n_rows > 0)
      launch_rows(3, tgt, srcG, h->vals_b.p, n_rows, im, radius_sq(d), cert, ps.match.p, h->match_d2.p, ps.lbe.p, ps.match2.p, s);
    ps.fresh = false;
    if (n_rows > 0) { rec.nn_search_launches++; rec.nn_search_queries += (long long)n_rows; rec.nn_kernel_launches++; }
    if (want_stats)
      fprintf(stderr, "[nn %d->%d] queries %zu bounded %zu rows %zu cum %.3g (last %.3g) err %.3g\n", job.src, job.tgt, n,
              n_near, n_far, cum_pair, src.last_motion + tgt.last_motion, src.err_max + tgt.err_max);
  } else if (dense) {
    h->match_pos.reserve(n);
    match_pos = h->match_pos.p;
    const InvMap im = make_invmap(tgt);
    sort_query_keys(h, tgt, srcG, nullptr, n, im);
    h->nn_timer->start(s);
    bool source_order = false;
    if (tgt.has_dense && h->nn_mode != 2) {
      h->lbe_scratch.reserve(n);
      source_order = launch_rows(h->nn_mode, tgt, srcG, h->vals_b.p, n, im, radius_sq(d), make_cert_params(tgt, MotionBound{}), h->match_pos.p,
                                 h->match_d2.p, h->lbe_scratch.p, nullptr, s);
    } else {
      launch_nn_cells(srcG, h->vals_b.p, n, tgt.G4.p, tgt.table.p, nullptr, tgt.grid, im, tgt.qrange, radius_sq(d),
                      h->match_pos.p, h->match_d2.p, s);
    }
    order = source_order ? nullptr : h->vals_b.p;
    rec.nn_search_launches++; rec.nn_search_queries += (long long)n; rec.nn_kernel_launches += 2; rec.nn_sort_calls++;
  } else {
    h->match_pos.reserve(n);
    match_pos = h->match_pos.p;
    rec.nn_search_launches++; rec.nn_search_queries += (long long)n; rec.nn_kernel_launches++;
    h->nn_timer->start(s);
    launch_nn_query(srcG, n, tgt.G4.p, tgt.table.p, tgt.grid, make_invmap(tgt), radius_sq(d), h->match_pos.p,
                    h->match_d2.p, s);
  }
  h->nn_timer->stop(s);
  NnPhase ph3(s, 2);                                                 // (its constructor waits for the search kernels)
  h->chunk_sum.reserve(div_up(nb, 256) + 1); h->chunk_d2.reserve(div_up(nb, 256) + 1);
  if (rstate) {
    // resident rows: bring the pair's planes up to date with the match list (rows whose partner changed), count, list the groups
    PairState& ps = *rstate;
    const size_t cap = resident_rows_cap(n);
    // a cloud that never moves inside this AlignMeshes call (impl cloud 0, the fixed cloud) keeps its half in the global frame:
    // no outer transform inside the LM passes for it.  If such a cloud did move since the rows were written (another role in an
    // earlier call), or a half changes frames, every row is rewritten once.
    const bool sg = src.fixed || src.cloud_index == 0, tg = tgt.fixed || tgt.cloud_index == 0;
    bool valid = ps.rows_valid && ps.pA.cap >= cap && ps.src_global == sg && ps.tgt_global == tg;
    if (valid && sg && std::memcmp(ps.src_T, src.T, sizeof ps.src_T) != 0) valid = false;
    if (valid && tg && std::memcmp(ps.tgt_T, tgt.T, sizeof ps.tgt_T) != 0) valid = false;
    if (!valid) {
      ps.pA.reserve(cap); ps.pB.reserve(cap); ps.pC.reserve(cap); ps.plane_match.reserve(n); ps.glist.reserve(div_up(n, 64));
      E3D_HIP(hipMemsetAsync(ps.plane_match.p, 0xFE, sizeof(int) * n, s));          // "never written": every row is
      if (cap > n) {                                                                 // the rows past the last query stay zero rows
        E3D_HIP(hipMemsetAsync(ps.pA.p + n, 0, sizeof(float4) * (cap - n), s));
        E3D_HIP(hipMemsetAsync(ps.pB.p + n, 0, sizeof(float4) * (cap - n), s));
        E3D_HIP(hipMemsetAsync(ps.pC.p + n, 0, sizeof(float4) * (cap - n), s));
      }
      ps.src_global = sg; ps.tgt_global = tg;
      std::memcpy(ps.src_T, src.T, sizeof ps.src_T); std::memcpy(ps.tgt_T, tgt.T, sizeof ps.tgt_T);
      ps.rows_valid = true;
    }
    h->block_groups.reserve(nb); h->chunk_groups.reserve(2 * (div_up(nb, 256) + 1));
    h->tm_compact.start(s);
    launch_corr_update(ps.match.p, ps.plane_match.p, h->match_d2.p, n, (sg ? src.G4.p : src.L4.p) + j0, srcLN, sg, to_affine(src.T),
                       tg ? tgt.G4.p : tgt.L4.p, tgt.LN.p, tg, to_affine(tgt.T), ps.pA.p, ps.pB.p, ps.pC.p, h->block_counts.p,
                       h->block_d2.p, h->block_groups.p, s);
    h->tm_compact.stop(s);
    h->tm_scan.start(s);
    launch_corr_totals(n, h->block_counts.p, h->block_d2.p, h->block_groups.p, h->chunk_sum.p, h->chunk_d2.p, h->chunk_groups.p,
                       h->d_total.p, h->d_total_d2.p, ps.glist.p, s);
    h->tm_scan.stop(s);
    rec.nn_update_launches++; rec.nn_kernel_launches += 4;
    copy_out(h->h_total.p + 1, h->d_total.p + 1, 2 * sizeof(unsigned long long), s);
  } else {
  h->tm_scan.start(s);
  launch_match_scan(match_pos, h->match_d2.p, n, h->block_counts.p, h->block_offsets.p, h->block_d2.p,
                    h->chunk_sum.p, h->chunk_d2.p, h->d_total.p, h->d_total_d2.p, s);
  h->tm_scan.stop(s);
  rec.nn_kernel_launches += 4;
  }
  copy_out(h->h_total.p, h->d_total.p, sizeof(unsigned long long), s);
  copy_out(h->h_total_d2.p, h->d_total_d2.p, sizeof(double), s);
  sync(h);
  { const double t = h->nn_timer->ms(); rec.t_nn_query_ms += t; rec.t_nn_search_ms += t; }
  job.count = (long long)h->h_total.p[0];
  job.dsum = h->h_total_d2.p[0];
  if (rows) pair_state_for(h, job.src, job.tgt, src, tgt, j0, n).matched_frac = (double)job.count / (double)n;
  if (h->sequential_dsum && !h->comm && h->world <= 1 && n == src.n) {
    h->d2_by_orig.reserve(n); h->h_d2_by_orig.resize(n);
    launch_match_d2_by_original(match_pos, h->match_d2.p, order, n, srcG, h->d2_by_orig.p, s);
    copy_out(h->h_d2_by_orig.data(), h->d2_by_orig.p, sizeof(float) * n, s);
    sync(h);
    float distance_sum = 0.f;                                  // icp_point_to_plane.cc:226-229
    for (size_t i = 0; i < n; ++i) { const float v = h->h_d2_by_orig[i]; if (v >= 0.f) distance_sum += v; }
    job.dsum = (double)distance_sum; job.dsum_f32 = true;
  }
  if (rstate) {
    job.resident = rstate;
    job.vrows = 64 * (long long)h->h_total.p[1];
    rec.corr_rows_rewritten += (long long)h->h_total.p[2];
    rec.corr_rows_walked += job.vrows;
    return;
  }
  if (job.count == 0) return;
  const size_t need = h->corr_used + (size_t)job.count;
  if (need > h->cA.cap) {
    const size_t ncap = std::max(need, h->cA.cap + h->cA.cap / 2);
    h->cA.grow_keep(ncap, h->corr_used, s);
    h->cB.grow_keep(ncap, h->corr_used, s);
    h->cC.grow_keep(ncap, h->corr_used, s);
  }
  h->tm_compact.start(s);
  launch_compact_corr(match_pos, order, n, h->block_offsets.p, srcG, srcLN,
                      to_affine(src.T), tgt.G4.p, tgt.LN.p, to_affine(tgt.T), h->cA.p, h->cB.p, h->cC.p,
                      h->corr_used, s);   // the merged fixed cloud keeps T = identity (exact)
  h->tm_compact.stop(s);
  rec.nn_update_launches++; rec.nn_kernel_launches++;
  h->corr_used = need;
  rec.corr_rows_rewritten += job.count;
  rec.corr_rows_walked += job.count;
}

// The certificate search + resident row update of a BATCH of directed pairs (round 4).  find_pair reads two results back per pair
// (the todo-list lengths after k_nn_certify, the totals after the row update) and synchronises the stream for each: 480 host round
// trips per outer iteration of a 240-pair job, 30 - 40 ms that do not shrink when the job is spread over more GPUs.  Pairs do not
// depend on each other, so a batch runs in phases: (A) k_nn_certify of every pair, ONE read-back of all list lengths; (B) the
// searches and the row update of every pair, ONE read-back of all totals.  Per-pair scratch (distances, todo lists) comes from
// the handle's slots; what a pair's kernels share with the next pair's (sort buffers, block counts) is ordered by the stream.
// Same kernels, same launches, same results as find_pair's certificate branch with resident rows.
struct BatchItem {
  PairJob* job; Cloud* src; Cloud* tgt; size_t j0, n; PairState* ps;
  InvMap im; CertParams cert; BoundParams bp; double cum_pair;
  bool certified = false, from_state = false; size_t n_near = 0, n_far = 0;
};
static constexpr size_t kPairBatch = 32;
// queries of a batch: its scratch (squared distances, two todo lists) is 12 B per query, 3 GB at most -- whatever the clouds' size
static constexpr size_t kBatchQueries = (size_t)256 << 20;

static void find_pairs_batched(e3d_icp* h, std::vector<BatchItem>& items, float d, e3d_icp_iter_record& rec) {
  hipStream_t s = h->stream;
  const size_t B = items.size();
  static const bool use_cert = [] { const char* e = getenv("E3D_NN_CERT"); return !(e && e[0] == '0'); }();
  static const bool want_stats = [] { const char* e = getenv("E3D_NN_STATS"); return e && e[0] == '1'; }();
  static const double margin_frac = env_double("E3D_NN_MARGIN", 0.08), near_frac = env_double("E3D_NN_NEAR", 0.4);   // of the radius
  static const double np_frac = env_double("E3D_NN_NP_EXTRA", 0.5), np_gate = env_double("E3D_NN_NP_GATE", 0.2);   // of the radius
  while (h->slots.size() < B) h->slots.emplace_back(new e3d_icp::PairSlot());
  h->h_todo_all.reserve(2 * kPairBatch); h->d_totals_all.reserve(3 * kPairBatch); h->d_d2_all.reserve(kPairBatch);
  h->h_totals_all.reserve(3 * kPairBatch); h->h_d2_all.reserve(kPairBatch);
  size_t n_max = 0;
  for (BatchItem& it : items) n_max = std::max(n_max, it.n);
  const size_t nb_max = div_up(n_max, kBlock);
  h->block_counts.reserve(nb_max); h->block_d2.reserve(nb_max); h->block_groups.reserve(nb_max);
  h->chunk_sum.reserve(div_up(nb_max, 256) + 1); h->chunk_d2.reserve(div_up(nb_max, 256) + 1); h->chunk_groups.reserve(2 * (div_up(nb_max, 256) + 1));
  // ---- phase A: certificates ----------------------------------------------------------------------------------------------
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    e3d_icp::PairSlot& sl = *h->slots[i];
    Cloud& src = *it.src; Cloud& tgt = *it.tgt;
    PairState& ps = *it.ps;
    sl.match_d2.reserve(it.n);
    it.im = make_invmap(tgt);
    it.cum_pair = src.cum_motion + tgt.cum_motion;
    MotionBound m_lo, cum_up;
    pair_motion_bounds(ps, src, tgt, m_lo, cum_up);
    it.cert = make_cert_params(tgt, m_lo);
    it.n_far = it.n; it.n_near = 0;
    it.certified = !ps.fresh && use_cert && certify_now(ps, src, tgt);
    it.from_state = !ps.fresh && use_cert && !it.certified;
    if (!it.certified) continue;
    const float4* srcG = src.G4.p + it.j0;
    const float near2 = (float)((near_frac * (double)d) * (near_frac * (double)d));
    sl.todo_near.reserve(it.n); sl.todo_far.reserve(it.n);
    E3D_HIP(hipMemsetAsync(ps.todo_count.p, 0, 2 * sizeof(unsigned), s));
    // (queries without a partner and the np_extra gate: see find_pair)
    const bool none_near = np_frac > 0 && (src.last_motion + tgt.last_motion) < np_gate * (double)d;
    h->tm_certify.start(s);
    launch_nn_certify(srcG, it.n, tgt.G4.p, cum_up, radius_sq(d), near2, none_near, ps.match.p, ps.match2.p, ps.lbe.p, sl.match_d2.p, sl.todo_near.p,
                      sl.todo_far.p, ps.todo_count.p, s);
    h->tm_certify.stop(s);
    copy_out(h->h_todo_all.p + 2 * i, ps.todo_count.p, 2 * sizeof(unsigned), s);
    rec.nn_certify_launches++; rec.nn_certify_queries += (long long)it.n; rec.nn_kernel_launches++;
    double smin = cloud_smin(tgt);
    if (!(smin > 1e-12)) smin = 1e-12;
    double m_local = 0;
    for (int k = 0; k < 3; ++k) m_local = std::max(m_local, std::max(std::fabs((double)tgt.lmin[k]), std::fabs((double)tgt.lmax[k])));
    it.bp.margin = (float)(margin_frac * (double)d);
    it.bp.rho_scale = round_up_f((1.0 + 1e-5) / smin);
    it.bp.rho_pad = round_up_f(2.0 * tgt.build_slack + 8.0 * FLT_EPSILON * m_local);
    it.bp.lo = it.cert.lo;
    it.bp.cell_scale = it.cert.cell_scale; it.bp.cell_sub = it.cert.cell_sub;
    it.bp.np_extra = none_near ? (float)(np_frac * (double)d) : 0.f;
  }
  sync(h);
  // ---- phase B: searches, row update, totals ----------------------------------------------------------------------------------
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    e3d_icp::PairSlot& sl = *h->slots[i];
    Cloud& src = *it.src; Cloud& tgt = *it.tgt;
    PairState& ps = *it.ps;
    const size_t n = it.n;
    const float4* srcG = src.G4.p + it.j0;
    const float4* srcLN = src.LN.p + it.j0;
    const unsigned* list = nullptr;
    if (it.certified) {
      it.n_near = h->h_todo_all.p[2 * i]; it.n_far = h->h_todo_all.p[2 * i + 1];
      certify_ran(ps, src, tgt, it.n, it.n_near + it.n_far);
      list = sl.todo_far.p;
      const unsigned long long* half = tgt.has_half ? tgt.half_prefix.p : nullptr;
      h->tm_bounded.start(s);
      launch_nn_bounded(srcG, sl.todo_near.p, it.n_near, tgt.G4.p, tgt.dense_start.p, half, h->nn_mode == 5, tgt.grid, it.im, tgt.qrange, radius_sq(d), it.bp,
                        ps.match.p, ps.match2.p, sl.match_d2.p, ps.lbe.p, s);
      if (far_list_is_short(it.n_far, n)) {
        launch_nn_bounded(srcG, sl.todo_far.p, it.n_far, tgt.G4.p, tgt.dense_start.p, half, h->nn_mode == 5, tgt.grid, it.im, tgt.qrange, radius_sq(d), it.bp,
                          ps.match.p, ps.match2.p, sl.match_d2.p, ps.lbe.p, s);
        it.n_near += it.n_far; it.n_far = 0;
      }
      h->tm_bounded.stop(s);
      if (it.n_near > 0) { rec.nn_bounded_launches++; rec.nn_bounded_queries += (long long)it.n_near; rec.nn_kernel_launches++; }
    }
    if (it.n_far > 0) {
      rec.nn_sort_calls++; rec.nn_kernel_launches++;
      const size_t n_rows = sort_query_keys_pruned(h, ps, tgt, srcG, list, it.n_far, it.im, radius_sq(d), it.cert, sl.match_d2.p, it.from_state);
      if (n_rows > 0) {
        h->tm_search.start(s);
        launch_rows(3, tgt, srcG, h->vals_b.p, n_rows, it.im, radius_sq(d), it.cert, ps.match.p, sl.match_d2.p, ps.lbe.p, ps.match2.p, s);
        h->tm_search.stop(s);
        rec.nn_search_launches++; rec.nn_search_queries += (long long)n_rows; rec.nn_kernel_launches++;
      }
    }
    ps.fresh = false;
    if (want_stats)
      fprintf(stderr, "[nn %d->%d] queries %zu bounded %zu rows %zu cum %.3g (last %.3g) err %.3g\n", it.job->src, it.job->tgt, n, it.n_near, it.n_far,
              it.cum_pair, src.last_motion + tgt.last_motion, src.err_max + tgt.err_max);
    // resident rows (see find_pair)
    const size_t cap = resident_rows_cap(n);
    const bool sg = src.fixed || src.cloud_index == 0, tg = tgt.fixed || tgt.cloud_index == 0;
    bool valid = ps.rows_valid && ps.pA.cap >= cap && ps.src_global == sg && ps.tgt_global == tg;
    if (valid && sg && std::memcmp(ps.src_T, src.T, sizeof ps.src_T) != 0) valid = false;
    if (valid && tg && std::memcmp(ps.tgt_T, tgt.T, sizeof ps.tgt_T) != 0) valid = false;
    if (!valid) {
      ps.pA.reserve(cap); ps.pB.reserve(cap); ps.pC.reserve(cap); ps.plane_match.reserve(n); ps.glist.reserve(div_up(n, 64));
      E3D_HIP(hipMemsetAsync(ps.plane_match.p, 0xFE, sizeof(int) * n, s));
      if (cap > n) {
        E3D_HIP(hipMemsetAsync(ps.pA.p + n, 0, sizeof(float4) * (cap - n), s));
        E3D_HIP(hipMemsetAsync(ps.pB.p + n, 0, sizeof(float4) * (cap - n), s));
        E3D_HIP(hipMemsetAsync(ps.pC.p + n, 0, sizeof(float4) * (cap - n), s));
      }
      ps.src_global = sg; ps.tgt_global = tg;
      std::memcpy(ps.src_T, src.T, sizeof ps.src_T); std::memcpy(ps.tgt_T, tgt.T, sizeof ps.tgt_T);
      ps.rows_valid = true;
    }
    h->tm_compact.start(s);
    launch_corr_update(ps.match.p, ps.plane_match.p, sl.match_d2.p, n, (sg ? src.G4.p : src.L4.p) + it.j0, srcLN, sg, to_affine(src.T),
                       tg ? tgt.G4.p : tgt.L4.p, tgt.LN.p, tg, to_affine(tgt.T), ps.pA.p, ps.pB.p, ps.pC.p, h->block_counts.p,
                       h->block_d2.p, h->block_groups.p, s);
    h->tm_compact.stop(s);
    h->tm_scan.start(s);
    launch_corr_totals(n, h->block_counts.p, h->block_d2.p, h->block_groups.p, h->chunk_sum.p, h->chunk_d2.p, h->chunk_groups.p,
                       h->d_totals_all.p + 3 * i, h->d_d2_all.p + i, ps.glist.p, s);
    h->tm_scan.stop(s);
    rec.nn_update_launches++; rec.nn_kernel_launches += 4;
  }
  copy_out(h->h_totals_all.p, h->d_totals_all.p, sizeof(unsigned long long) * 3 * B, s);
  copy_out(h->h_d2_all.p, h->d_d2_all.p, sizeof(double) * B, s);
  sync(h);
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    it.job->count = (long long)h->h_totals_all.p[3 * i];
    it.job->dsum = h->h_d2_all.p[i];
    it.ps->matched_frac = it.n ? (double)it.job->count / (double)it.n : 1.0;
    it.job->resident = it.ps;
    it.job->vrows = 64 * (long long)h->h_totals_all.p[3 * i + 1];
    rec.corr_rows_rewritten += (long long)h->h_totals_all.p[3 * i + 2];
    rec.corr_rows_walked += it.job->vrows;
    rec.queries += (long long)it.n;
    rec.correspondences += it.job->count;
  }
}

// The same batch with ONE launch per kernel (round 5): the batch's pairs go into a device table (NnBatchDev) and k_nn_certify_multi,
// k_nn_bounded_half_multi, k_corr_update_multi and the three kernels of the totals walk it -- 7 launches and two host round trips per
// batch where find_pairs_batched issues ~8 launches per PAIR.  Same kernels bodies on the same data: counts, distances, rows and group
// lists are those of find_pair bit for bit (tests/test_gpu_switches.py, E3D_ICP_BATCH=1 / 0).  A pair's far list that is too long for
// the bounded search (the first outer iterations) is still sorted and searched by k_nn_rows pair by pair.  Needs the half-cell
// directory of every target of the batch (dense scans have it); returns false otherwise and the caller takes find_pairs_batched.
static bool find_pairs_multi(e3d_icp* h, std::vector<BatchItem>& items, float d, e3d_icp_iter_record& rec) {
  hipStream_t s = h->stream;
  const size_t B = items.size();
  for (const BatchItem& it : items) if (!it.tgt->has_half || h->nn_mode == 5) return false;
  static const bool use_cert = [] { const char* e = getenv("E3D_NN_CERT"); return !(e && e[0] == '0'); }();
  static const bool want_stats = [] { const char* e = getenv("E3D_NN_STATS"); return e && e[0] == '1'; }();
  static const double margin_frac = env_double("E3D_NN_MARGIN", 0.08), near_frac = env_double("E3D_NN_NEAR", 0.4);   // of the radius
  static const double np_frac = env_double("E3D_NN_NP_EXTRA", 0.5), np_gate = env_double("E3D_NN_NP_GATE", 0.2);   // of the radius
  while (h->slots.size() < B) h->slots.emplace_back(new e3d_icp::PairSlot());
  h->h_todo_all.reserve(2 * kPairBatch); h->d_todo_all.reserve(2 * kPairBatch);
  h->d_totals_all.reserve(3 * kPairBatch); h->d_d2_all.reserve(kPairBatch); h->h_totals_all.reserve(3 * kPairBatch); h->h_d2_all.reserve(kPairBatch);
  h->h_batch.reserve(1); h->d_batch.reserve(1);
  NnBatchDev& T = h->h_batch.p[0];
  static_assert(kPairBatch <= (size_t)kNnBatchPairs, "pair table too small");
  T.n_pairs = (int)B; T.n_jobs = 0;
  // ---- phase A: the table, the rows' validity, the certificates ----------------------------------------------------------------
  unsigned cert_blocks = 0, upd_blocks = 0, chunks = 0;
  long long cert_queries = 0;
  E3D_HIP(hipMemsetAsync(h->d_todo_all.p, 0, 2 * sizeof(unsigned) * B, s));
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    e3d_icp::PairSlot& sl = *h->slots[i];
    Cloud& src = *it.src; Cloud& tgt = *it.tgt;
    PairState& ps = *it.ps;
    const size_t n = it.n;
    sl.match_d2.reserve(n); sl.todo_near.reserve(n); sl.todo_far.reserve(n);
    it.im = make_invmap(tgt);
    it.cum_pair = src.cum_motion + tgt.cum_motion;
    MotionBound m_lo, m_up;
    pair_motion_bounds(ps, src, tgt, m_lo, m_up);
    it.cert = make_cert_params(tgt, m_lo);
    it.n_far = n; it.n_near = 0;
    it.certified = !ps.fresh && use_cert && certify_now(ps, src, tgt);
    it.from_state = !ps.fresh && use_cert && !it.certified;
    const bool none_near = np_frac > 0 && (src.last_motion + tgt.last_motion) < np_gate * (double)d;
    double smin = cloud_smin(tgt);
    if (!(smin > 1e-12)) smin = 1e-12;
    double m_local = 0;
    for (int k = 0; k < 3; ++k) m_local = std::max(m_local, std::max(std::fabs((double)tgt.lmin[k]), std::fabs((double)tgt.lmax[k])));
    it.bp.margin = (float)(margin_frac * (double)d);
    it.bp.rho_scale = round_up_f((1.0 + 1e-5) / smin);
    it.bp.rho_pad = round_up_f(2.0 * tgt.build_slack + 8.0 * FLT_EPSILON * m_local);
    it.bp.lo = it.cert.lo;
    it.bp.cell_scale = it.cert.cell_scale; it.bp.cell_sub = it.cert.cell_sub;
    it.bp.np_extra = none_near ? (float)(np_frac * (double)d) : 0.f;
    // resident rows (see find_pair)
    const size_t cap = resident_rows_cap(n);
    const bool sg = src.fixed || src.cloud_index == 0, tg = tgt.fixed || tgt.cloud_index == 0;
    bool valid = ps.rows_valid && ps.pA.cap >= cap && ps.src_global == sg && ps.tgt_global == tg;
    if (valid && sg && std::memcmp(ps.src_T, src.T, sizeof ps.src_T) != 0) valid = false;
    if (valid && tg && std::memcmp(ps.tgt_T, tgt.T, sizeof ps.tgt_T) != 0) valid = false;
    if (!valid) {
      ps.pA.reserve(cap); ps.pB.reserve(cap); ps.pC.reserve(cap); ps.plane_match.reserve(n); ps.glist.reserve(div_up(n, 64));
      E3D_HIP(hipMemsetAsync(ps.plane_match.p, 0xFE, sizeof(int) * n, s));
      if (cap > n) {
        E3D_HIP(hipMemsetAsync(ps.pA.p + n, 0, sizeof(float4) * (cap - n), s));
        E3D_HIP(hipMemsetAsync(ps.pB.p + n, 0, sizeof(float4) * (cap - n), s));
        E3D_HIP(hipMemsetAsync(ps.pC.p + n, 0, sizeof(float4) * (cap - n), s));
      }
      ps.src_global = sg; ps.tgt_global = tg;
      std::memcpy(ps.src_T, src.T, sizeof ps.src_T); std::memcpy(ps.tgt_T, tgt.T, sizeof ps.tgt_T);
      ps.rows_valid = true;
    }
    NnPairDev& P = T.pair[i];
    P.Gsrc = src.G4.p + it.j0; P.Gtgt = tgt.G4.p; P.S = tgt.dense_start.p; P.H8 = tgt.half_prefix.p;
    P.match = ps.match.p; P.match2 = ps.match2.p; P.lbe = ps.lbe.p; P.match_d2 = sl.match_d2.p;
    P.todo_near = sl.todo_near.p; P.todo_far = sl.todo_far.p; P.counts = h->d_todo_all.p + 2 * i;
    P.n = (unsigned)n; P.none_near = none_near ? 1 : 0;
    P.cum_up = m_up;
    P.near2 = (float)((near_frac * (double)d) * (near_frac * (double)d));
    P.g = tgt.grid; P.im = it.im; P.qr = tgt.qrange; P.bp = it.bp;
    P.Psrc = (sg ? src.G4.p : src.L4.p) + it.j0; P.LNsrc = src.LN.p + it.j0; P.Ptgt = tg ? tgt.G4.p : tgt.L4.p; P.LNtgt = tgt.LN.p;
    P.src_global = sg ? 1 : 0; P.tgt_global = tg ? 1 : 0; P.Tsrc = to_affine(src.T); P.Ttgt = to_affine(tgt.T);
    P.A = ps.pA.p; P.B = ps.pB.p; P.C = ps.pC.p; P.plane_match = ps.plane_match.p; P.glist = ps.glist.p;
    if (it.certified) { cert_blocks += (unsigned)div_up(n, (size_t)kNnCertBlockQueries); cert_queries += (long long)n; }
    T.cert_end[i] = cert_blocks;
    const unsigned nb = (unsigned)div_up(n, kBlock);
    upd_blocks += nb; T.upd_end[i] = upd_blocks;
    chunks += (unsigned)div_up((size_t)nb, (size_t)kNnScanChunk); T.chunk_end[i] = chunks;
  }
  h->block_counts.reserve(upd_blocks); h->block_d2.reserve(upd_blocks); h->block_groups.reserve(upd_blocks);
  h->chunk_sum.reserve(chunks + 1); h->chunk_d2.reserve(chunks + 1); h->chunk_groups.reserve(chunks + 1); h->chunk_rewritten.reserve(chunks + 1);
  // The certificate kernel writes the row update's per-block results for the 256-query blocks it settles whole, and the update skips
  // those on one flag (E3D_NN_FUSE_UPDATE=0: off) -- for the pairs whose LAST certificate pass left fewer than E3D_NN_FUSE_GATE
  // (default 0.3 %) of the queries unsettled: a block is settled whole with probability exp(-256 x that share) (46 % at the gate,
  // 80 % at 0.09 %, the settled all-pairs job); below that the flag's extra round trip in front of every block of the update and
  // the extra sums in the certificate kernel cost more than the skipped blocks save (measured: profiles/round6_certify_update_fusion.txt).
  static const bool fuse_update = [] { const char* e = getenv("E3D_NN_FUSE_UPDATE"); return !(e && e[0] == '0'); }();
  static const double fuse_gate = env_double("E3D_NN_FUSE_GATE", 0.003);
  {
    if (fuse_update && cert_blocks) { h->block_done.reserve(upd_blocks); E3D_HIP(hipMemsetAsync(h->block_done.p, 0, upd_blocks, s)); }
    unsigned b0 = 0;
    for (size_t i = 0; i < B; ++i) {
      NnPairDev& P = T.pair[i];
      const bool on = fuse_update && cert_blocks && items[i].certified && (1.0 - items[i].ps->settled_frac) < fuse_gate;
      P.upd_counts = on ? h->block_counts.p + b0 : nullptr; P.upd_d2 = on ? h->block_d2.p + b0 : nullptr;
      P.upd_groups = on ? h->block_groups.p + b0 : nullptr; P.upd_done = on ? h->block_done.p + b0 : nullptr;
      b0 = T.upd_end[i];
    }
  }
  E3D_HIP(hipMemcpyAsync(h->d_batch.p, h->h_batch.p, sizeof(NnBatchDev), hipMemcpyHostToDevice, s));
  if (cert_blocks) {
    h->tm_certify.start(s);
    launch_nn_certify_multi(h->d_batch.p, cert_blocks, radius_sq(d), s);
    h->tm_certify.stop(s);
    rec.nn_certify_launches++; rec.nn_certify_queries += cert_queries; rec.nn_kernel_launches++;
    copy_out(h->h_todo_all.p, h->d_todo_all.p, 2 * sizeof(unsigned) * B, s);
  }
  sync(h);
  // ---- phase B: the bounded searches as list jobs of ONE launch, long far lists pair by pair, row update, totals ----------------
  unsigned job_blocks = 0;
  long long job_queries = 0;
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    if (!it.certified) continue;
    e3d_icp::PairSlot& sl = *h->slots[i];
    it.n_near = h->h_todo_all.p[2 * i]; it.n_far = h->h_todo_all.p[2 * i + 1];
    certify_ran(*it.ps, *it.src, *it.tgt, it.n, it.n_near + it.n_far);
    auto add_job = [&](const unsigned* list, size_t n_list) {
      if (!n_list) return;
      const int jb = T.n_jobs++;
      T.job_pair[jb] = (int)i; T.job_list[jb] = list; T.job_n[jb] = (unsigned)n_list;
      job_blocks += (unsigned)div_up(n_list, kBlock); T.job_end[jb] = job_blocks;
      job_queries += (long long)n_list;
    };
    add_job(sl.todo_near.p, it.n_near);
    if (far_list_is_short(it.n_far, it.n)) {                  // (few queries without a near partner: see find_pair)
      add_job(sl.todo_far.p, it.n_far);
      it.n_near += it.n_far; it.n_far = 0;
    }
  }
  auto launch_bounded_jobs = [&] {
    if (T.n_jobs <= 0) return;
    E3D_HIP(hipMemcpyAsync(h->d_batch.p, h->h_batch.p, sizeof(NnBatchDev), hipMemcpyHostToDevice, s));
    h->tm_bounded.start(s);
    launch_nn_bounded_half_multi(h->d_batch.p, job_blocks, radius_sq(d), s);
    h->tm_bounded.stop(s);
    rec.nn_bounded_launches++; rec.nn_bounded_queries += job_queries; rec.nn_kernel_launches++;
  };
  // the far lists that are too long for the bounded search (first outer iterations): ONE key kernel, ONE sort, ONE k_nn_rows launch
  // for the batch (round 6; E3D_NN_FAR_BATCH=0: pair by pair as in round 5).  The pair's index rides above the cell key, so the
  // batch's keys need key_bits + log2(pairs) bits: 8-byte (key, query) pairs while that fits 32 bits.
  static const bool far_batch = [] { const char* e = getenv("E3D_NN_FAR_BATCH"); return !(e && e[0] == '0'); }();
  static const size_t prune_min_list = (size_t)env_double("E3D_NN_PRUNE_MIN", 262144.0);
  // SEEDS (k_query_seed_multi): once a fair share of a pair's queries found a partner in the last search (the scans are about to
  // meet, or have met), the key kernel probes the query's own half cell and the old partner; a query with a target point nearer than
  // E3D_NN_SEED_NEAR x radius takes it as its partner to start from and goes through the bounded search instead of sort + k_nn_rows.
  // E3D_NN_SEED=0: off; E3D_NN_SEED_FRAC: the matched share of the last search from which a pair's far list is seeded.
  static const bool seed_allowed = [] { const char* e = getenv("E3D_NN_SEED"); return !(e && e[0] == '0'); }();
  static const double seed_frac = env_double("E3D_NN_SEED_FRAC", 0.3), seed_near = env_double("E3D_NN_SEED_NEAR", 0.4);
  static const bool seed_fresh = env_double("E3D_NN_SEED_FRESH", 0.0) != 0.0;      // (a pair's first search: nothing is known about it)
  auto seeds_for = [&](const BatchItem& it) { return seed_allowed && it.n_far > 0 && (it.ps->fresh ? seed_fresh : it.ps->matched_frac >= seed_frac); };
  size_t far_pairs = 0, far_total = 0;
  int kb_max = 1;
  for (size_t i = 0; i < B; ++i) if (items[i].n_far > 0) { ++far_pairs; far_total += items[i].n_far; kb_max = std::max(kb_max, items[i].tgt->key_bits); }
  int pair_bits = 0;
  while (((size_t)1 << pair_bits) < B) ++pair_bits;
  // (a batch whose keys would need 12-byte pairs only because of the pair bits keeps the 8-byte pairs of the pair-by-pair path)
  const bool far_multi = far_batch && far_pairs > 0 && (kb_max + pair_bits <= 32 || kb_max > 31) && kb_max + pair_bits <= 63;
  if (!far_multi) launch_bounded_jobs();
  if (far_multi) {
    const bool k32 = kb_max + pair_bits <= 32;
    bool seed_any = false;
    for (size_t i = 0; i < B; ++i)
      if (seeds_for(items[i])) seed_any = true;
    const size_t key_block = seed_any ? (size_t)kQuerySeedBlock : (size_t)kQueryKeysBlock;
    unsigned key_blocks = 0;
    for (size_t i = 0; i < B; ++i) {
      BatchItem& it = items[i];
      NnPairDev& P = T.pair[i];
      P.far_n = 0; P.far_list = nullptr; P.far_flags = 0; P.occ = nullptr; P.occ_stride = 0; P.rows_off = 0; P.rows_n = 0; P.seed_list = nullptr; P.seed2 = 0.f;
      if (it.n_far > 0) {
        const bool prune = it.tgt->has_occ && it.ps->prune && it.n_far >= prune_min_list;       // (sort_query_keys_pruned's rule)
        const bool seed = seeds_for(it);
        P.far_n = (unsigned)it.n_far;
        P.far_list = it.certified ? h->slots[i]->todo_far.p : nullptr;
        P.far_flags = ((prune && it.from_state && !it.certified) ? 1 : 0) | (prune ? 2 : 0) | (seed ? 4 : 0);
        P.occ = it.tgt->has_occ ? it.tgt->occ27.p : nullptr; P.occ_stride = it.tgt->occ_stride;
        // (the seeded list behind the pair's near list: the two together are at most the pair's queries)
        P.seed_list = h->slots[i]->todo_near.p + (it.certified ? (size_t)h->h_todo_all.p[2 * i] : 0);
        P.seed2 = (float)((seed_near * (double)d) * (seed_near * (double)d));
        key_blocks += (unsigned)div_up(it.n_far, key_block);
      }
      T.far_end[i] = key_blocks;
    }
    T.key_shift = kb_max;
    h->keys_a.reserve(far_total); h->keys_b.reserve(far_total); h->vals_a.reserve(far_total); h->vals_b.reserve(far_total);
    constexpr size_t kCounts = 1 + 2 * (size_t)kNnBatchPairs;
    h->prune_count.reserve(kCounts); h->h_prune_count.reserve(kCounts);
    E3D_HIP(hipMemcpyAsync(h->d_batch.p, h->h_batch.p, sizeof(NnBatchDev), hipMemcpyHostToDevice, s));
    h->tm_sort.start(s);
    E3D_HIP(hipMemsetAsync(h->prune_count.p, 0, sizeof(unsigned) * kCounts, s));
    if (seed_any) launch_query_seed_multi(k32, h->d_batch.p, key_blocks, radius_sq(d), h->keys_a.p, h->vals_a.p, h->prune_count.p, s);
    else launch_query_keys_multi(k32, h->d_batch.p, key_blocks, radius_sq(d), h->keys_a.p, h->vals_a.p, h->prune_count.p, s);
    h->tm_sort.stop(s);
    copy_out(h->h_prune_count.p, h->prune_count.p, sizeof(unsigned) * kCounts, s);
    sync(h);
    const size_t kept_total = h->h_prune_count.p[0];
    unsigned row_blocks = 0, off = 0;
    for (size_t i = 0; i < B; ++i) {
      BatchItem& it = items[i];
      NnPairDev& P = T.pair[i];
      const unsigned kept = h->h_prune_count.p[1 + i], seeded = seed_any ? h->h_prune_count.p[1 + (size_t)kNnBatchPairs + i] : 0u;
      if ((P.far_flags & 2) && (double)kept + (double)seeded > 0.9 * (double)it.n_far) it.ps->prune = false;
      P.rows_off = off; P.rows_n = kept; off += kept;
      row_blocks += (unsigned)div_up((size_t)kept, kBlock);
      T.rows_end[i] = row_blocks;
      if (kept > 0) rec.nn_search_queries += (long long)kept;
      if (seeded > 0) {                                     // the seeded queries: one more list job of the bounded search
        const int jb = T.n_jobs++;
        T.job_pair[jb] = (int)i; T.job_list[jb] = P.seed_list; T.job_n[jb] = seeded;
        job_blocks += (unsigned)div_up((size_t)seeded, kBlock); T.job_end[jb] = job_blocks;
        job_queries += (long long)seeded;
        it.n_near += seeded; it.n_far -= std::min((size_t)seeded, it.n_far);
      }
    }
    launch_bounded_jobs();
    if (kept_total > 0) {
      h->tm_sort.start(s);
      // (the temporary storage for the lists' whole length: the kept count grows from one outer iteration to the next, e3d_sort.hip)
      if (k32) sort_pairs_u32_u32(reinterpret_cast<unsigned*>(h->keys_a.p), reinterpret_cast<unsigned*>(h->keys_b.p), h->vals_a.p, h->vals_b.p, kept_total, kb_max + pair_bits, h->sort_temp, s, far_total);
      else sort_pairs_u64_u32(h->keys_a.p, h->keys_b.p, h->vals_a.p, h->vals_b.p, kept_total, kb_max + pair_bits, h->sort_temp, s, far_total);
      h->tm_sort.stop(s);
    }
    rec.nn_kernel_launches++; rec.nn_sort_calls++;
    if (T.n_jobs <= 0) E3D_HIP(hipMemcpyAsync(h->d_batch.p, h->h_batch.p, sizeof(NnBatchDev), hipMemcpyHostToDevice, s));     // (the pairs' stretches of the sorted array)
    if (kept_total > 0) {
      h->tm_search.start(s);
      launch_nn_rows_multi(h->d_batch.p, row_blocks, h->vals_b.p, radius_sq(d), s);
      h->tm_search.stop(s);
      rec.nn_search_launches++; rec.nn_kernel_launches++;
    }
  }
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    PairState& ps = *it.ps;
    if (it.n_far > 0 && !far_multi) {
      e3d_icp::PairSlot& sl = *h->slots[i];
      const float4* srcG = it.src->G4.p + it.j0;
      const size_t n_rows = sort_query_keys_pruned(h, ps, *it.tgt, srcG, it.certified ? sl.todo_far.p : nullptr, it.n_far, it.im, radius_sq(d), it.cert, sl.match_d2.p, it.from_state);
      rec.nn_kernel_launches++; rec.nn_sort_calls++;
      if (n_rows > 0) {
        h->tm_search.start(s);
        launch_rows(3, *it.tgt, srcG, h->vals_b.p, n_rows, it.im, radius_sq(d), it.cert, ps.match.p, sl.match_d2.p, ps.lbe.p, ps.match2.p, s);
        h->tm_search.stop(s);
        rec.nn_search_launches++; rec.nn_search_queries += (long long)n_rows; rec.nn_kernel_launches++;
      }
    }
    ps.fresh = false;
    if (want_stats)
      fprintf(stderr, "[nn %d->%d] queries %zu bounded %zu rows %zu cum %.3g (last %.3g) err %.3g\n", it.job->src, it.job->tgt, it.n, it.n_near, it.n_far,
              it.cum_pair, it.src->last_motion + it.tgt->last_motion, it.src->err_max + it.tgt->err_max);
  }
  h->tm_compact.start(s);
  launch_corr_update_multi(h->d_batch.p, upd_blocks, h->block_counts.p, h->block_d2.p, h->block_groups.p, s);
  h->tm_compact.stop(s);
  h->tm_scan.start(s);
  launch_corr_totals_multi(h->d_batch.p, (int)B, chunks, h->block_counts.p, h->block_d2.p, h->block_groups.p, h->chunk_sum.p, h->chunk_d2.p, h->chunk_groups.p,
                           h->chunk_rewritten.p, h->d_totals_all.p, h->d_d2_all.p, s);
  h->tm_scan.stop(s);
  rec.nn_update_launches++; rec.nn_kernel_launches += 4; rec.nn_batches++;
  copy_out(h->h_totals_all.p, h->d_totals_all.p, sizeof(unsigned long long) * 3 * B, s);
  copy_out(h->h_d2_all.p, h->d_d2_all.p, sizeof(double) * B, s);
  sync(h);
  for (size_t i = 0; i < B; ++i) {
    BatchItem& it = items[i];
    it.job->count = (long long)h->h_totals_all.p[3 * i];
    it.job->dsum = h->h_d2_all.p[i];
    it.ps->matched_frac = it.n ? (double)it.job->count / (double)it.n : 1.0;
    it.job->resident = it.ps;
    it.job->vrows = 64 * (long long)h->h_totals_all.p[3 * i + 1];
    rec.corr_rows_rewritten += (long long)h->h_totals_all.p[3 * i + 2];
    rec.corr_rows_walked += it.job->vrows;
    rec.queries += (long long)it.n;
    rec.correspondences += it.job->count;
  }
  return true;
}

// number of LM blocks for a set of n correspondences in a system of n_sets sets (deterministic function of the two).  Every
// block ends with a wave / block reduction of up to 55 f64 accumulators (~1000 instructions, 2 - 3 loop trips' worth): with
// hundreds of sets (all-pairs jobs) 1024 blocks per set would leave each thread ~40 trips, so the cap shrinks with the set count
// while the whole launch keeps >= 8192 blocks (~10 rounds over the resident slots) for balance.  (On equal-sized sets far fewer
// blocks stream faster -- 0.78 instead of 0.91 ms per 1e8 correspondences with 512 blocks in all, profiles/round3_lm_blocks.txt --
// but the pairs of a real job differ in size and the nine-pose cost pass wants occupancy: 512 / 2048 blocks in all made the
// all-pairs LM 567 -> 696 ms and the 2-scan step no faster.)
static int lm_blocks_for(long long n, int n_sets = 1) {
  long long b = (n + (long long)kBlock * 8 - 1) / ((long long)kBlock * 8);
  long long cap = 8192 / std::max(n_sets, 1);
  if (cap > 1024) cap = 1024;   // 4 blocks per CU; the rest is grid-stride
  if (cap < 64) cap = 64;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// (a callback with world == 1 is a tap: the "reduction" over one rank still passes every buffer through it -- bench.py records the
// sums of a single-GPU run that way and replays them to a handle working as rank 0 of a larger world)
static bool sharded(const e3d_icp* h) { return h->comm != nullptr || h->world > 1 || h->allreduce != nullptr; }

// E3D_LM_PROFILE=1: wall-clock split of an outer iteration's LM on the host (stderr; diagnostics only)
static bool lm_profile() { static const bool on = [] { const char* e = getenv("E3D_LM_PROFILE"); return e && e[0] == '1'; }(); return on; }
struct LmProf { double solve_ms = 0, eval_ms = 0, reduce_ms = 0, prepare_ms = 0, callback_ms = 0, callback_max = 0, wait_max = 0; int solves = 0, evals = 0; };
static thread_local LmProf g_lm_prof;
struct LmTick {
  double& acc; std::chrono::steady_clock::time_point t0;
  explicit LmTick(double& a) : acc(a) { if (lm_profile()) t0 = std::chrono::steady_clock::now(); }
  ~LmTick() { if (lm_profile()) acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// sum of the per-set results (n doubles, already reduced over this rank's blocks) over the ranks: in place in HBM on the
// handle's stream with the native communicator; the host copy follows either way
static void reduce_setsums(e3d_icp* h, int ns) {
  hipStream_t s = h->stream;
  LmTick tick(g_lm_prof.reduce_ms);
  const size_t n = (size_t)kLmSlot * (size_t)ns;
  if (h->comm) comm_allreduce_f64(h->comm, h->d_setsum.p, n, s);
  copy_out(h->h_setsum.p, h->d_setsum.p, sizeof(double) * n, s);
  { double w = 0; { LmTick tw(w); sync(h); } g_lm_prof.wait_max = std::max(g_lm_prof.wait_max, w); }
  if (!h->comm && h->allreduce) {
    double c = 0;
    { LmTick tc(c); if (h->allreduce(h->h_setsum.p, n, h->allreduce_user) != 0) throw Error(E3D_ERR_INVALID, "allreduce callback failed"); }
    g_lm_prof.callback_ms += c; g_lm_prof.callback_max = std::max(g_lm_prof.callback_max, c);
  }
}

// sum of a small host buffer over the ranks (per-pair counts: once per outer iteration)
static void reduce_host(e3d_icp* h, double* buf, size_t n) {
  if (!n || !sharded(h)) return;
  if (h->comm) {
    hipStream_t s = h->stream;
    h->d_red.reserve(n);
    copy_in(h->d_red.p, buf, sizeof(double) * n, s);
    comm_allreduce_f64(h->comm, h->d_red.p, n, s);
    copy_out(buf, h->d_red.p, sizeof(double) * n, s);
    sync(h);
  } else if (h->allreduce(buf, n, h->allreduce_user) != 0) {
    throw Error(E3D_ERR_INVALID, "allreduce callback failed");
  }
}

struct LmSystem {
  int n_impl = 0, nv = 0;
  std::vector<PairJob*> sets;        // this rank's non-empty pairs, grouped by mode
  std::vector<int> mode_begin, mode_blocks, mode_block_base;   // per mode 1..3
  int total_blocks = 0;
  std::vector<double> H, b;
  double cost = 0;
};

// Evaluate cost (+ H, b when full) at the given inner poses.
static void lm_evaluate(e3d_icp* h, LmSystem& L, const std::vector<SE3f>& poses, bool full, std::vector<double>& H,
                        std::vector<double>& b, double& cost, e3d_icp_iter_record& rec) {
  hipStream_t s = h->stream;
  LmTick tick(g_lm_prof.eval_ms); g_lm_prof.evals++;
  const int ns = (int)L.sets.size();
  const int nv = L.nv;
  H.assign((size_t)nv * nv, 0.0);
  b.assign((size_t)nv, 0.0);
  cost = 0.0;
  if (ns > 0) {
    for (int i = 0; i < ns; ++i) {
      LmSet& S = h->h_sets.p[i];
      const SE3f& ps = poses[L.sets[i]->impl_src];
      const SE3f& pt = poses[L.sets[i]->impl_tgt];
      quat_to_matrix<float>(ps.q.w, ps.q.x, ps.q.y, ps.q.z, S.Rs);
      quat_to_matrix<float>(pt.q.w, pt.q.x, pt.q.y, pt.q.z, S.Rt);
      for (int k = 0; k < 3; ++k) { S.ts[k] = ps.t[k]; S.tt[k] = pt.t[k]; }
    }
    E3D_HIP(hipMemcpyAsync(h->d_sets.p, h->h_sets.p, sizeof(LmSet) * ns, hipMemcpyHostToDevice, s));
    if (!h->lm_timer) h->lm_timer.reset(new EventTimer());
    EventTimer& tm = *h->lm_timer;
    tm.start(s);
    if (full) {
      for (int m = 1; m <= 3; ++m)
        launch_lm_pass(m, h->d_sets.p, h->d_block_set.p, L.mode_block_base[m], L.mode_blocks[m], h->d_partial.p, s);
    } else {
      launch_lm_pass(kModeCost, h->d_sets.p, h->d_block_set.p, 0, L.total_blocks, h->d_partial.p, s);
    }
    tm.stop(s);
    launch_lm_reduce(h->d_partial.p, h->d_sets.p, ns, kLmSlot, h->d_setsum.p, s);
    reduce_setsums(h, ns);      // every rank holds the same sets (lm_prepare), so the per-set blocks add up across the ranks
    rec.t_lm_kernel_ms += tm.ms();
    if (full) { rec.full_passes++; rec.t_lm_full_kernel_ms += tm.ms(); } else rec.cost_passes++;
    // scatter the per-set systems into H, b  (Accumulate, icp_point_to_plane_impl.h:82-113)
    for (int i = 0; i < ns; ++i) {
      const double* r = h->h_setsum.p + (size_t)kLmSlot * i;
      cost += r[0];
      if (!full) continue;
      const PairJob& pj = *L.sets[i];
      const int si = 6 * (pj.impl_src - 1), ti = 6 * (pj.impl_tgt - 1);
      const LmSet& S = h->h_sets.p[i];
      if (S.mode == kModeOne) {
        const int vi = (S.side == 0) ? si : ti;
        int k = 1;
        for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) H[(size_t)(vi + a) * nv + vi + c] += r[k++];
        for (int a = 0; a < 6; ++a) b[vi + a] += r[22 + a];
      } else {
        int k = 1;
        for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) H[(size_t)(si + a) * nv + si + c] += r[k++];
        for (int a = 0; a < 6; ++a) b[si + a] += r[22 + a];
        k = 28;
        for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) H[(size_t)(ti + a) * nv + ti + c] += r[k++];
        for (int a = 0; a < 6; ++a) b[ti + a] += r[49 + a];
        if (S.mode == kModeTwoCross)
          for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) H[(size_t)(si + a) * nv + ti + c] += r[55 + 6 * a + c];
        // kModeTwo: the (src,tgt) block lies in the lower triangle, which the solver never reads [QUIRK]
      }
    }
  } else {
    if (full) rec.full_passes++; else rec.cost_passes++;
  }
}

// Costs of up to kLmMaxPoses candidate pose sets in one pass (k_lm_cost_multi); costs[k] for cand[k].
static void lm_evaluate_costs(e3d_icp* h, LmSystem& L, const std::vector<std::vector<SE3f>>& cand, std::vector<double>& costs,
                              e3d_icp_iter_record& rec) {
  hipStream_t s = h->stream;
  LmTick tick(g_lm_prof.eval_ms); g_lm_prof.evals++;
  const int ns = (int)L.sets.size();
  const int np = (int)cand.size();
  costs.assign(np, 0.0);
  if (ns > 0) {
    h->h_poses.reserve((size_t)np * ns);
    h->d_poses.reserve((size_t)kLmMaxPoses * ns);
    for (int k = 0; k < np; ++k)
      for (int i = 0; i < ns; ++i) {
        LmPose& P = h->h_poses.p[(size_t)k * ns + i];
        const SE3f& ps = cand[k][L.sets[i]->impl_src];
        const SE3f& pt = cand[k][L.sets[i]->impl_tgt];
        quat_to_matrix<float>(ps.q.w, ps.q.x, ps.q.y, ps.q.z, P.Rs);
        quat_to_matrix<float>(pt.q.w, pt.q.x, pt.q.y, pt.q.z, P.Rt);
        for (int c = 0; c < 3; ++c) { P.ts[c] = ps.t[c]; P.tt[c] = pt.t[c]; }
      }
    E3D_HIP(hipMemcpyAsync(h->d_poses.p, h->h_poses.p, sizeof(LmPose) * (size_t)np * ns, hipMemcpyHostToDevice, s));
    if (!h->lm_timer) h->lm_timer.reset(new EventTimer());
    EventTimer& tm = *h->lm_timer;
    tm.start(s);
    launch_lm_cost_multi(h->d_sets.p, h->d_poses.p, ns, np, h->d_block_set.p, L.total_blocks, h->d_partial.p, s);
    tm.stop(s);
    launch_lm_reduce(h->d_partial.p, h->d_sets.p, ns, kLmSlot, h->d_setsum.p, s);
    reduce_setsums(h, ns);
    rec.t_lm_kernel_ms += tm.ms();
    for (int i = 0; i < ns; ++i)
      for (int k = 0; k < np; ++k) costs[k] += h->h_setsum.p[(size_t)kLmSlot * i + k];
  }
  rec.multi_cost_passes++;
}

static void lm_prepare(e3d_icp* h, LmSystem& L, std::vector<PairJob>& jobs, int M) {
  // group the non-empty pairs (global counts: every rank builds the same list, a rank without correspondences of a pair
  // contributes an empty set) by full-pass mode; assign LM blocks
  L.sets.clear();
  L.mode_begin.assign(5, 0); L.mode_blocks.assign(5, 0); L.mode_block_base.assign(5, 0);
  std::vector<std::vector<PairJob*>> by_mode(4);
  for (PairJob& j : jobs) {
    if (j.gcount == 0) continue;
    const int si = j.impl_src - 1, ti = j.impl_tgt - 1;
    int mode;
    if (si < 0 || ti < 0) mode = kModeOne;
    else mode = (si < ti) ? kModeTwoCross : kModeTwo;
    by_mode[mode].push_back(&j);
  }
  int block = 0;
  const int ns_total = (int)(by_mode[1].size() + by_mode[2].size() + by_mode[3].size());
  h->h_sets.reserve(std::max(ns_total, 1));
  h->d_sets.reserve(std::max(ns_total, 1));
  std::vector<int> block_set;
  for (int m = 1; m <= 3; ++m) {
    L.mode_begin[m] = (int)L.sets.size();
    L.mode_block_base[m] = block;
    for (PairJob* j : by_mode[m]) {
      const int i = (int)L.sets.size();
      LmSet& S = h->h_sets.p[i];
      if (j->resident) {
        // resident rows: the pass walks the listed 64-row groups; local halves get the outer pose of this iteration
        const PairState& ps = *j->resident;
        const Cloud& src = (j->src == M) ? *h->fixed : *h->clouds[j->src];
        const Cloud& tgt = (j->tgt == M) ? *h->fixed : *h->clouds[j->tgt];
        S.A = ps.pA.p; S.B = ps.pB.p; S.C = ps.pC.p; S.glist = ps.glist.p; S.n = j->vrows;
        if ((size_t)j->vrows == resident_rows_cap(ps.n)) S.glist = nullptr;      // every group is listed: rows [0, n) as they lie
        S.outer = (ps.src_global ? 0 : 1) | (ps.tgt_global ? 0 : 2);
        S.Tos = to_affine(src.T); S.Tot = to_affine(tgt.T);
      } else {
        S.A = h->cA.p + j->corr_off; S.B = h->cB.p + j->corr_off; S.C = h->cC.p + j->corr_off; S.glist = nullptr; S.n = j->count;
        S.outer = 0;
        S.Tos = Affine{}; S.Tot = Affine{};
      }
      S.block_begin = block; S.nblocks = lm_blocks_for(S.n, ns_total);
      S.mode = m;
      S.side = (j->impl_src - 1 >= 0) ? 0 : 1;
      for (int b = 0; b < S.nblocks; ++b) block_set.push_back(i);
      block += S.nblocks;
      L.sets.push_back(j);
    }
    L.mode_blocks[m] = block - L.mode_block_base[m];
  }
  L.total_blocks = block;
  h->d_block_set.reserve(std::max(block, 1));
  h->d_partial.reserve((size_t)std::max(block, 1) * kLmSlot);
  h->d_setsum.reserve((size_t)std::max(ns_total, 1) * kLmSlot);
  h->h_setsum.reserve((size_t)std::max(ns_total, 1) * kLmSlot);
  if (block > 0) {
    E3D_HIP(hipMemcpyAsync(h->d_block_set.p, block_set.data(), sizeof(int) * block, hipMemcpyHostToDevice, h->stream));
    sync(h);
  }
}

// PointToPlaneICPImpl::compute  (icp_point_to_plane_impl.h:115-293)
static void lm_compute(e3d_icp* h, LmSystem& L, std::vector<SE3f>& poses, e3d_icp_iter_record& rec) {
  const int nv = L.nv;
  std::vector<double> H, b, Hn, bn;
  double cost = 0, new_cost = 0;
  double lambda = 0.1;
  lm_evaluate(h, L, poses, true, H, b, cost, rec);
  rec.initial_cost = cost;
  rec.final_cost = cost;
  struct SolveScratch { std::vector<double> Hl, x, W; std::vector<int> perm; };
  std::vector<SolveScratch> scratch(10);
  // what the quadratic model promises for a try: cost(-x) ~ cost - (2 b.x - x.H x) (H: the upper triangle, as the solver reads it)
  std::vector<double> model_gain(10, 0.0), step_norm(10, 0.0);
  auto candidate = [&](double lam, std::vector<SE3f>& out, SolveScratch& sc, int k_try) {
    sc.Hl = H;
    sc.x.resize((size_t)std::max(nv, 1));
    for (int i = 0; i < nv; ++i) sc.Hl[(size_t)i * nv + i] += lam;         // additive damping (impl.h:223)
    if (nv > 0) ldlt_solve_upper(sc.Hl.data(), nv, b.data(), sc.x.data(), sc.W, sc.perm);
    {
      double bx = 0, xHx = 0, xx = 0;
      for (int i = 0; i < nv; ++i) {
        bx += b[i] * sc.x[i]; xx += sc.x[i] * sc.x[i];
        xHx += H[(size_t)i * nv + i] * sc.x[i] * sc.x[i];
        for (int j = i + 1; j < nv; ++j) xHx += 2.0 * H[(size_t)i * nv + j] * sc.x[i] * sc.x[j];
      }
      model_gain[k_try] = 2.0 * bx - xHx; step_norm[k_try] = std::sqrt(xx);
    }
    out.resize(poses.size());
    out[0] = poses[0];
    for (size_t ci = 1; ci < poses.size(); ++ci) out[ci] = se3_apply_update(&sc.x[6 * (ci - 1)], poses[ci]);   // impl.h:235
  };
  // the tries' solves side by side on the handle's host threads when the system is large (E3D_LM_SOLVE_THREADS: 0 = on the caller's)
  static const int solve_threads = [] { const char* e = getenv("E3D_LM_SOLVE_THREADS"); return e ? atoi(e) : 4; }();
  if (nv >= 30 && solve_threads > 0 && !h->solve_pool) h->solve_pool.reset(new SolvePool(std::min(solve_threads, 9)));
  // Candidate poses are compared as f32 bit patterns: a pass at a pose that was already evaluated returns the cost it returned
  // then, bit for bit (same rows, same per-correspondence f32 code, same reduction tree), so it need not run.  At the end of an
  // outer iteration the update x is so small that exp(-x).cast<float>() * pose rounds back to the pose for most of the ten
  // damping values: those tries have new_cost == cost and are rejected as the reference rejects them (impl.h:240-283 needs
  // new_cost < cost), tries with identical poses share one evaluation.  The sequential decision rule is unchanged.
  auto same_poses = [](const std::vector<SE3f>& a, const std::vector<SE3f>& b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(SE3f)) == 0;
  };
  static_assert(sizeof(SE3f) == 7 * sizeof(float), "SE3f is compared as raw floats");
  // All ten tries of an LM step as (lambda, poses): try 0 at the current lambda, tries 1..9 with lambda doubled after every
  // rejection (impl.h:216-283).  The costs of the tries in [first, 10) whose poses are new and distinct come from one multi-pose pass
  // (one new pose: the plain cost pass); a try whose poses equal the current ones costs `cost`.
  std::vector<std::vector<SE3f>> cand(10);
  std::vector<double> lam(10), costs(10);
  // lam[first..9] from lam_first; the poses of tries [first, last); returns lambda after ten rejections
  auto tries_from = [&](int first, int last, double lam_first) {
    LmTick tick(g_lm_prof.solve_ms); g_lm_prof.solves += last - first;
    double l = lam_first;
    for (int k = first; k < 10; ++k) { lam[k] = l; l = 2.f * l; }
    if (h->solve_pool && nv >= 30 && last - first > 1) {
      const std::function<void(int)> one = [&](int i) { candidate(lam[first + i], cand[first + i], scratch[first + i], first + i); };
      h->solve_pool->parallel_for(last - first, one);
    } else {
      for (int k = first; k < last; ++k) candidate(lam[k], cand[k], scratch[0], k);
    }
    return l;
  };
  auto evaluate_tries = [&](int first) {
    std::vector<std::vector<SE3f>> distinct;
    std::vector<int> slot(10, -1);                         // -1: the pose equals the current one (cost known)
    std::vector<double> dcosts;
    for (int k = first; k < 10; ++k) {
      if (same_poses(cand[k], poses)) continue;
      for (size_t q = 0; q < distinct.size() && slot[k] < 0; ++q) if (same_poses(cand[k], distinct[q])) slot[k] = (int)q;
      if (slot[k] < 0) { slot[k] = (int)distinct.size(); distinct.push_back(cand[k]); }
    }
    if (distinct.size() == 1) {            // one new pose: the plain cost pass (HBM bound; same bits)
      std::vector<double> Hx, bx;
      dcosts.assign(1, 0.0);
      lm_evaluate(h, L, distinct[0], false, Hx, bx, dcosts[0], rec);
    } else if (!distinct.empty()) { lm_evaluate_costs(h, L, distinct, dcosts, rec); rec.multi_cost_poses += (int)distinct.size(); }
    else rec.lm_passes_skipped++;
    for (int k = first; k < 10; ++k) costs[k] = (slot[k] < 0) ? cost : dcosts[slot[k]];
    {
      static const bool lm_trace = getenv("E3D_LM_TRACE") != nullptr;
      if (lm_trace) {                                   // how many clouds keep their f32 pose in each try (diagnostics)
        fprintf(stderr, "[lm trace] tries %d..9, clouds with an unchanged pose of %zu:", first, poses.size());
        for (int k = first; k < 10; ++k) {
          int same = 0;
          for (size_t ci = 0; ci < poses.size(); ++ci) if (std::memcmp(&cand[k][ci], &poses[ci], sizeof(SE3f)) == 0) ++same;
          fprintf(stderr, " %d", same);
        }
        fprintf(stderr, "\n");
      }
    }
  };
  // Try 0 is usually accepted, so its cost is evaluated together with the next step's H and b (one fused pass) -- except in the LM
  // step an alignment that has settled ends with: there all ten tries are rejected, and the fused pass's H and b are thrown away.
  // The step at which the previous outer iteration's LM ended is the prediction: from that step on, try 0 joins tries 1..9 in the
  // multi-pose cost pass (ten poses), and only an accepted try pays a fused pass at its pose.  Same costs bit for bit, same
  // sequential decisions; a wrong prediction costs one cost-only evaluation of try 0.  (E3D_LM_SPECULATE=0: always the fused pass.)
  static const bool speculate = [] { const char* e = getenv("E3D_LM_SPECULATE"); return !(e && e[0] == '0'); }();
  // (Speculating one step LATER -- E3D_LM_SPEC_OFFSET=1 -- was measured too: once the model's gain is below the f32 noise of the
  // cost, relative 1e-7 and less (E3D_LM_TRACE prints both), a step is accepted or not like a coin is tossed and the LM ends at the
  // previous iteration's step in half of the cases; but the ten tries of such a step round to three to seven distinct poses, the
  // multi-pose pass is cheaper than a fused pass + nine poses, and speculating at the same step stays ahead: settling iterations of
  // the headline scene 99.7 -> 103.9 ms with the offset, 102.7 without speculation.)
  static const int spec_offset = [] { const char* e = getenv("E3D_LM_SPEC_OFFSET"); return e ? atoi(e) : 0; }();
  const int predicted_end = h->lm_prev_end_step;
  h->lm_prev_end_step = -1;
  for (int it = 0; it < h->max_inner; ++it) {
    rec.inner_iterations++;
    bool applied = false;
    const double trace_cost_before = cost;
    double trace_try0_cost = 0.0;
    const bool batch_all = speculate && predicted_end >= 0 && it >= predicted_end + spec_offset;
    int hit = -1;
    double lam_end = lambda;
    if (batch_all) {
      lam_end = tries_from(0, 10, lambda);
      evaluate_tries(0);
      trace_try0_cost = costs[0];
      for (int k = 0; k < 10; ++k) if (costs[k] < cost) { hit = k; break; }
    } else {
      lam_end = tries_from(0, 1, lambda);                  // only try 0 is needed yet: the other nine are solved if it is rejected
      if (same_poses(cand[0], poses)) { new_cost = cost; rec.lm_passes_skipped++; }
      else lm_evaluate(h, L, cand[0], true, Hn, bn, new_cost, rec);
      trace_try0_cost = new_cost;
      if (new_cost < cost) {
        poses = cand[0]; H.swap(Hn); b.swap(bn); cost = new_cost;
        lambda = 0.5f * lambda;
        applied = true;
      } else {
        tries_from(1, 10, lam[1]);
        evaluate_tries(1);
        for (int k = 1; k < 10; ++k) if (costs[k] < cost) { hit = k; break; }
      }
    }
    if (!applied) {
      if (hit >= 0) {
        double c2;
        lm_evaluate(h, L, cand[hit], true, Hn, bn, c2, rec);   // H, b at the accepted pose; c2 == costs[hit] bit for bit
        poses = cand[hit]; H.swap(Hn); b.swap(bn); cost = c2;
        lambda = 0.5f * lam[hit];
        applied = true;
      } else {
        lambda = lam_end;   // ten rejections: lambda doubled ten times
      }
    }
    rec.final_cost = cost;
    {
      static const bool lm_trace = getenv("E3D_LM_TRACE") != nullptr;
      if (lm_trace) fprintf(stderr, "[lm trace] inner %d batch_all %d hit %d applied %d predicted_end %d try0: model gain %.6e (of cost %.9e) step %.3e lambda %.3e cost after %.12e\n", it, (int)batch_all, hit, (int)applied, predicted_end, model_gain[0], trace_cost_before, step_norm[0], lam[0], trace_try0_cost);
    }
    if (!applied) { h->lm_prev_end_step = it; break; }
  }
}

// PointToPlaneICP::AlignMeshes  (icp_point_to_plane.cc:169-342)
static bool align_meshes(e3d_icp* h, float max_d, float thr, bool print, int iteration) {
  e3d_icp_iter_record rec{};
  rec.iteration = iteration;
  hipStream_t s = h->stream;
  EventTimer t_tr, t_nn, t_lm;
  const int M = (int)h->clouds.size();
  const bool has_fixed = (bool)h->fixed;
  int n_impl = 0, fixed_vertex = -1;

  t_tr.start(s);
  if (has_fixed) {
    Cloud& f = *h->fixed;
    fixed_vertex = n_impl++;
    f.cloud_index = fixed_vertex;
    if (!grid_usable(f, max_d)) build_grid(h, f, max_d);
    // fixed cloud lives in the global frame: G4 = L4 (identity transform is exact), bbox recomputed
    transform_cloud(h, f);
  }
  for (int i = 0; i < M; ++i) {
    Cloud& c = *h->clouds[i];
    if (!grid_usable(c, max_d)) build_grid(h, c, max_d);
    transform_cloud(h, c);
    c.cloud_index = n_impl++;
  }
  t_tr.stop(s);

  // pair list in the sequential order of the reference's ik loop (cc:208-309)
  std::vector<PairJob> jobs;
  for (int ik = 0; ik < M * M; ++ik) {
    const int i = ik / M, k = ik % M;
    if (i != k && bbox_intersects(*h->clouds[i], *h->clouds[k]))
      jobs.push_back({i, k, h->clouds[i]->cloud_index, h->clouds[k]->cloud_index});
    if (i == k && has_fixed && bbox_intersects(*h->fixed, *h->clouds[i])) {
      jobs.push_back({i, M, h->clouds[i]->cloud_index, fixed_vertex});
      jobs.push_back({M, i, fixed_vertex, h->clouds[i]->cloud_index});
    }
  }
  t_nn.start(s);
  h->corr_used = 0;
  {
    // correspondences <= queries: size the planes once (no hipMalloc/hipFree inside the iteration loop)
    // Pairs of the certificate path keep RESIDENT rows (one row per query, 48 B + 4 B, for as long as the grids live) when all of
    // them fit beside what is already allocated; otherwise every pair goes through the compacted planes as before.
    size_t qtot = 0, rows_new = 0;
    auto slice_len = [&](const Cloud& src) {
      return (size_t)((unsigned __int128)src.n * (unsigned)(h->rank + 1) / (unsigned)h->world) -
             (size_t)((unsigned __int128)src.n * (unsigned)h->rank / (unsigned)h->world);
    };
    bool resident = h->resident_rows;
    for (const PairJob& j : jobs) {
      const Cloud& src = (j.src == M) ? *h->fixed : *h->clouds[j.src];
      const Cloud& tgt = (j.tgt == M) ? *h->fixed : *h->clouds[j.tgt];
      const size_t nq = slice_len(src);
      if (!pair_uses_rows(h, tgt) || nq == 0 || tgt.n == 0) continue;
      auto it = h->pair_state.find(std::make_pair(j.src, j.tgt));
      if (it == h->pair_state.end() || it->second->pA.cap < resident_rows_cap(nq)) rows_new += resident_bytes(nq);
    }
    // rows of pairs that have left the job list (their bounding boxes no longer intersect) go back first
    for (auto& kv : h->pair_state) {
      bool listed = false;
      for (const PairJob& j : jobs) if (j.src == kv.first.first && j.tgt == kv.first.second) { listed = true; break; }
      if (!listed) { PairState& ps = *kv.second; ps.pA.release(); ps.pB.release(); ps.pC.release(); ps.rows_valid = false; }
    }
    auto release_rows = [&]() {
      for (auto& kv : h->pair_state) { PairState& ps = *kv.second; ps.pA.release(); ps.pB.release(); ps.pC.release(); ps.rows_valid = false; }
    };
    if (resident && rows_new > 0) {
      // beside the rows: the batch scratch (12 B per query of a batch, at most 64 M queries) and what the search allocates on
      // first use (sort buffers, block counts: ~40 B per query of the largest pair)
      size_t free_b = 0, total_b = 0, q_max = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      for (const PairJob& j : jobs) q_max = std::max(q_max, slice_len((j.src == M) ? *h->fixed : *h->clouds[j.src]));
      const double extra = 12.0 * (double)std::min<size_t>(q_max * kPairBatch, kBatchQueries) + 40.0 * (double)q_max;
      if ((double)rows_new + extra > 0.8 * (double)free_b) resident = false;
    }
    if (resident) {
      // The rows are allocated HERE, all of them, so that running out of memory is noticed before any pair has been searched: on
      // failure every pair takes the compacted planes this iteration (the old data flow), exactly as when the estimate says no.
      try {
        for (const PairJob& j : jobs) {
          const Cloud& src = (j.src == M) ? *h->fixed : *h->clouds[j.src];
          const Cloud& tgt = (j.tgt == M) ? *h->fixed : *h->clouds[j.tgt];
          const size_t nq = slice_len(src);
          if (!pair_uses_rows(h, tgt) || nq == 0 || tgt.n == 0) continue;
          const size_t j0 = (size_t)((unsigned __int128)src.n * (unsigned)h->rank / (unsigned)h->world);
          PairState& ps = pair_state_for(h, j.src, j.tgt, src, tgt, j0, nq);
          const size_t cap = resident_rows_cap(nq);
          if (ps.pA.cap < cap || ps.pB.cap < cap || ps.pC.cap < cap) {
            ps.rows_valid = false;
            ps.pA.reserve(cap); ps.pB.reserve(cap); ps.pC.reserve(cap);
          }
          ps.plane_match.reserve(nq); ps.glist.reserve(div_up(nq, 64));
        }
      } catch (const Error&) {
        (void)hipGetLastError();
        release_rows();
        resident = false;
      }
    }
    if (!resident) release_rows();
    h->resident_now = resident;
    for (const PairJob& j : jobs) {
      const Cloud& src = (j.src == M) ? *h->fixed : *h->clouds[j.src];
      const Cloud& tgt = (j.tgt == M) ? *h->fixed : *h->clouds[j.tgt];
      if (resident && pair_uses_rows(h, tgt)) continue;           // no compacted rows for this pair
      qtot += slice_len(src);
    }
    // two directed pairs: one slot per query (no reallocation ever); many pairs: most queries of a pair find no partner, so
    // start from last iteration's total (+ 12 %) or a quarter of the queries and let find_pair grow the planes if needed
    size_t want = qtot;
    if (jobs.size() > 2 && qtot > h->cA.cap) {
      // one slot per query as well if HBM has room for it (growing 100 GB planes means hipMalloc + copy + hipFree of that size:
      // seconds per outer iteration while the correspondence count still climbs); otherwise start small and grow
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const double need = 48.0 * (double)(qtot - h->cA.cap);
      if (need > 0.75 * (double)free_b)
        want = std::min(qtot, std::max(h->last_corr_total + h->last_corr_total / 8 + (size_t)(1 << 20), qtot / 4));
    }
    if (want > h->cA.cap) { h->cA.reserve(want); h->cB.reserve(want); h->cC.reserve(want); }
  }
  {
    // Pairs of the certificate search with resident rows run in batches of up to kPairBatch (two host round trips per batch);
    // every other pair (sparse targets, forced kernel modes, compacted rows, the sequential distance sum) one by one.
    static const bool batching = [] { const char* e = getenv("E3D_ICP_BATCH"); return !(e && e[0] == '0'); }();
    std::vector<BatchItem> batch;
    size_t batch_queries = 0;                               // (kBatchQueries bounds a batch's scratch)
    // E3D_ICP_BATCH: 2 (default) one launch per kernel and batch (find_pairs_multi), 1 one launch per kernel and pair with the host
    // round trips batched (find_pairs_batched), 0 pair by pair (find_pair)
    static const int batch_mode = [] { const char* e = getenv("E3D_ICP_BATCH"); return e ? atoi(e) : 2; }();
    auto flush = [&]() {
      if (batch.empty()) return;
      if (!(batch_mode >= 2 && find_pairs_multi(h, batch, max_d, rec))) find_pairs_batched(h, batch, max_d, rec);
      batch.clear(); batch_queries = 0;
    };
    for (size_t p = 0; p < jobs.size(); ++p) {
      PairJob& j = jobs[p];
      Cloud& src = (j.src == M) ? *h->fixed : *h->clouds[j.src];
      Cloud& tgt = (j.tgt == M) ? *h->fixed : *h->clouds[j.tgt];
      // this rank's slice of the source cloud (cell order); world == 1 => the whole cloud
      const size_t j0 = (size_t)((unsigned __int128)src.n * (unsigned)h->rank / (unsigned)h->world);
      const size_t j1 = (size_t)((unsigned __int128)src.n * (unsigned)(h->rank + 1) / (unsigned)h->world);
      const bool whole = !h->comm && h->world <= 1 && (j1 - j0) == src.n;
      if (batching && !nn_profile() && h->resident_now && pair_uses_rows(h, tgt) && j1 > j0 && tgt.n > 0 && !(h->sequential_dsum && whole)) {
        j.count = 0; j.dsum = 0.0; j.corr_off = h->corr_used;
        BatchItem it{};
        it.job = &j; it.src = &src; it.tgt = &tgt; it.j0 = j0; it.n = j1 - j0;
        it.ps = &pair_state_for(h, j.src, j.tgt, src, tgt, j0, j1 - j0);
        if (batch_queries + it.n > kBatchQueries) flush();
        batch.push_back(it);
        batch_queries += it.n;
        if (batch.size() == kPairBatch) flush();
        continue;
      }
      flush();                                              // (keeps the pairs' order on the stream)
      find_pair(h, src, tgt, max_d, j, j0, j1, rec);
      rec.queries += (long long)(j1 - j0);
      rec.correspondences += j.count;
    }
    flush();
  }
  t_nn.stop(s);
  if (nn_profile()) {
    fprintf(stderr, "[nn profile] certify + bounded search %.1f ms, keys + sort %.1f ms, match scan + compaction (+ plane growth) %.1f ms, %zu pairs\n",
            g_nn_prof[0], g_nn_prof[1], g_nn_prof[2], jobs.size());
    for (double& v : g_nn_prof) v = 0;
  }
  h->last_corr_total = h->corr_used;
  // global per-pair counts (what the reference prints); local counts stay in j.count for the LM sets
  std::vector<long long> gcount(jobs.size());
  std::vector<double> gdsum(jobs.size());
  for (size_t p = 0; p < jobs.size(); ++p) { gcount[p] = jobs[p].count; gdsum[p] = jobs[p].dsum; }
  if (sharded(h) && !jobs.empty()) {
    std::vector<double> buf(2 * jobs.size(), 0.0);
    for (size_t p = 0; p < jobs.size(); ++p) { buf[2 * p] = (double)jobs[p].count; buf[2 * p + 1] = jobs[p].dsum; }
    reduce_host(h, buf.data(), buf.size());
    for (size_t p = 0; p < jobs.size(); ++p) { gcount[p] = (long long)buf[2 * p]; gdsum[p] = buf[2 * p + 1]; }
  }
  for (size_t p = 0; p < jobs.size(); ++p) jobs[p].gcount = gcount[p];
  for (size_t p = 0; p < jobs.size(); ++p) {
    const PairJob& j = jobs[p];
    const int psrc = (j.impl_src == fixed_vertex) ? -1 : j.impl_src;
    const int ptgt = (j.impl_tgt == fixed_vertex) ? -1 : j.impl_tgt;
    h->pair_records.push_back({iteration, psrc, ptgt, (int64_t)gcount[p], gdsum[p]});
    if (print && h->rank == 0) {
      char avg[64] = "";
      if (gcount[p] > 0) {
        const float a = jobs[p].dsum_f32 ? (float)gdsum[p] / (float)(size_t)gcount[p]        // float / size_t, as the reference divides
                                         : (float)(gdsum[p] / (double)gcount[p]);
        snprintf(avg, sizeof avg, " (avg. distance: %g)", (double)a);
      }
      if (psrc >= 0 && ptgt >= 0) printf("  found correspondences from %d to %d: %lld%s\n", j.impl_src, j.impl_tgt, gcount[p], avg);
      else if (ptgt < 0) printf("  found correspondences from %d to fixed clouds: %lld%s\n", j.impl_src, gcount[p], avg);
      else printf("  found correspondences from fixed clouds to %d: %lld%s\n", j.impl_tgt, gcount[p], avg);
    }
  }

  // impl.setMaxIterations(150); impl.compute()  (cc:312-316)
  t_lm.start(s);
  std::vector<SE3f> poses((size_t)n_impl);
  LmSystem L;
  L.n_impl = n_impl;
  L.nv = 6 * (n_impl - 1);
  { LmTick tick(g_lm_prof.prepare_ms); lm_prepare(h, L, jobs, M); }
  if (n_impl >= 1) lm_compute(h, L, poses, rec);
  t_lm.stop(s);
  if (lm_profile()) {
    fprintf(stderr, "[lm profile] it %d: prepare %.2f ms, %d solves %.2f ms, %d evaluations %.2f ms (of which waiting for the sums + reduction %.2f ms: longest wait %.2f ms; "
            "all-reduce callback %.2f ms, longest %.2f ms)\n", iteration, g_lm_prof.prepare_ms, g_lm_prof.solves, g_lm_prof.solve_ms, g_lm_prof.evals, g_lm_prof.eval_ms,
            g_lm_prof.reduce_ms, g_lm_prof.wait_max, g_lm_prof.callback_ms, g_lm_prof.callback_max);
    g_lm_prof = LmProf{};
  }

  // pose write-back (cc:318-341): new = Affine3f(pose.matrix()) * global_T_cloud
  bool converged = true;
  std::vector<float> T_before((size_t)12 * (size_t)(M + 1));
  for (int i = 0; i < M; ++i) std::memcpy(&T_before[12 * (size_t)i], h->clouds[i]->T, sizeof(float) * 12);
  if (has_fixed) std::memcpy(&T_before[12 * (size_t)M], h->fixed->T, sizeof(float) * 12);
  for (int i = 0; i < M; ++i) {
    Cloud& c = *h->clouds[i];
    const SE3f& p = poses[c.cloud_index];
    float R[9];
    quat_to_matrix<float>(p.q.w, p.q.x, p.q.y, p.q.z, R);
    float Tn[12];
    for (int r = 0; r < 3; ++r) {
      for (int col = 0; col < 3; ++col)
        Tn[4 * r + col] = dot3(R[3 * r], R[3 * r + 1], R[3 * r + 2], c.T[col], c.T[4 + col], c.T[8 + col]);
      Tn[4 * r + 3] = dot3(R[3 * r], R[3 * r + 1], R[3 * r + 2], c.T[3], c.T[7], c.T[11]) + p.t[r];
    }
    const float dx = c.T[3] - Tn[3], dy = c.T[7] - Tn[7], dz = c.T[11] - Tn[11];
    const float movement = std::sqrt(dx * dx + (dy * dy + dz * dz));
    if (movement > thr) converged = false;
    if (print && h->rank == 0) printf("  %d moved by %g\n", c.cloud_index, (double)movement);
    // certificates of the NN search: how far can any point of this cloud have moved, how exact is its transform
    c.last_motion = pose_motion_bound(c, c.T, Tn);
    c.cum_motion += c.last_motion;
    c.err_max = std::max(c.err_max, std::max(pose_rounding_bound(c, c.T), pose_rounding_bound(c, Tn)));
    std::memcpy(c.T, Tn, sizeof Tn);
  }
  // the pairs' own motion accumulators (every pair that keeps a state, whether it was searched in this iteration or not)
  for (auto& kv : h->pair_state) {
    const int si = kv.first.first, ti = kv.first.second;
    if (si < 0 || ti < 0 || si > M || ti > M || (si == M && !has_fixed) || (ti == M && !has_fixed)) continue;
    const Cloud& src = (si == M) ? *h->fixed : *h->clouds[si];
    const Cloud& tgt = (ti == M) ? *h->fixed : *h->clouds[ti];
    const double glob = ((si == M) ? 0.0 : src.last_motion) + ((ti == M) ? 0.0 : tgt.last_motion);
    if (!accumulate_pair_motion(*kv.second, src, tgt, &T_before[12 * (size_t)si], src.T, &T_before[12 * (size_t)ti], tgt.T, glob)) ++h->nn_global_bound_updates;
  }
  {
    static const bool want_stats = [] { const char* e = getenv("E3D_NN_STATS"); return e && e[0] == '1'; }();
    if (want_stats && h->nn_global_bound_updates)
      fprintf(stderr, "[nn] %lld pose updates of a pair went into the clouds' global motion bound (a pose that is not near-rigid)\n", h->nn_global_bound_updates);
  }
  rec.t_transform_ms = t_tr.ms();
  rec.t_nn_ms = t_nn.ms();
  rec.t_nn_sort_ms = h->tm_sort.take(); rec.t_nn_scan_ms = h->tm_scan.take(); rec.t_nn_compact_ms = h->tm_compact.take();
  { const double tb = h->tm_bounded.take(); rec.t_nn_bounded_ms += tb; rec.t_nn_query_ms += tb; }
  { const double tc = h->tm_certify.take(); rec.t_nn_certify_ms += tc; rec.t_nn_query_ms += tc; }
  { const double ts = h->tm_search.take(); rec.t_nn_search_ms += ts; rec.t_nn_query_ms += ts; }
  rec.t_lm_ms = t_lm.ms();
  h->iter_records.push_back(rec);
  return converged;
}

}  // namespace e3d

// =================================================================================================
// C-ABI
// =================================================================================================
#define E3D_TRY try {
#define E3D_CATCH()                                                                         \
  } catch (const e3d::Error& e) { e3d::set_last_error(e.what()); return e.code; }           \
  catch (const std::exception& e) { e3d::set_last_error(e.what()); return E3D_ERR_INVALID; }

extern "C" {

int e3d_abi_version(void) { return E3D_ABI_VERSION; }

const char* e3d_last_error(void) { return e3d::last_error_cstr(); }

int e3d_init(int device) {
  E3D_TRY
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(E3D_ERR_NO_DEVICE, "no HIP device visible (libe3dhip needs an MI355X / gfx950 GPU)");
  if (device < 0 || device >= n) throw Error(E3D_ERR_INVALID, fmt("device %d out of range (%d devices)", device, n));
  E3D_HIP(hipSetDevice(device));
  g_device = device;
  return n;
  E3D_CATCH()
}

int e3d_set_nn_mode(int mode) {
  if (mode < 0 || mode > 5) { e3d::set_last_error("e3d_set_nn_mode: mode must be 0..5"); return E3D_ERR_INVALID; }
  g_nn_mode = mode;
  return 0;
}

e3d_icp_t* e3d_icp_create(void) {
  try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(E3D_ERR_NO_DEVICE, "no HIP device visible (libe3dhip needs an MI355X / gfx950 GPU)");
    E3D_HIP(hipSetDevice(g_device));
    std::unique_ptr<e3d_icp> h(new e3d_icp());
    h->device = g_device;
    h->nn_mode = g_nn_mode;
    E3D_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    return h.release();
  } catch (const std::exception& e) {
    e3d::set_last_error(e.what());
    return nullptr;
  }
}

void e3d_icp_destroy(e3d_icp_t* icp) { delete icp; }

int e3d_icp_add_cloud(e3d_icp_t* h, const float* xyz, const float* normals, size_t n, const float T[12], int fixed) {
  E3D_TRY
  if (!h || (!xyz && n) || (!normals && n) || !T) throw Error(E3D_ERR_INVALID, "e3d_icp_add_cloud: null argument");
  if (n >= (size_t)1 << 31) throw Error(E3D_ERR_INVALID, "e3d_icp_add_cloud: more than 2^31-1 points (reference uses int indices)");
  E3D_HIP(hipSetDevice(h->device));
  if (fixed) {
    // transform to the global frame once and concatenate (icp_point_to_plane.cc:112-127)
    if (!h->fixed) { h->fixed.reset(new Cloud()); h->fixed->fixed = true; }
    Cloud& f = *h->fixed;
    const size_t n0 = f.n;
    f.raw_xyz.grow_keep(3 * (n0 + n), 3 * n0, h->stream);
    f.raw_nrm.grow_keep(3 * (n0 + n), 3 * n0, h->stream);
    if (n > 0) {
      DevBuf<float> tx, tn;
      tx.reserve(3 * n); tn.reserve(3 * n);
      copy_in(tx.p, xyz, sizeof(float) * 3 * n, h->stream);
      copy_in(tn.p, normals, sizeof(float) * 3 * n, h->stream);
      ensure_bbox_scratch(h);
      launch_transform_aos(tx.p, tn.p, n, to_affine(T), f.raw_xyz.p + 3 * n0, f.raw_nrm.p + 3 * n0, h->bbox_partial.p,
                           h->bbox_out.p, h->stream);
      sync(h);
    }
    f.n = n0 + n;
    f.grid_valid = false;
    return -1;
  }
  std::unique_ptr<Cloud> c(new Cloud());
  std::memcpy(c->T, T, sizeof(float) * 12);
  upload_cloud(h, *c, xyz, normals, n);
  h->clouds.push_back(std::move(c));
  return (int)h->clouds.size() - 1;
  E3D_CATCH()
}

int e3d_icp_run(e3d_icp_t* h, float max_d, int initial_iteration, int max_num_iterations, float thr, int print) {
  E3D_TRY
  if (!h) throw Error(E3D_ERR_INVALID, "e3d_icp_run: null handle");
  if (h->clouds.empty()) throw Error(E3D_ERR_INVALID, "e3d_icp_run: no clouds to optimize (reference: CHECK(!clouds_.empty()))");
  E3D_HIP(hipSetDevice(h->device));
  for (int i = initial_iteration; i < initial_iteration + max_num_iterations; ++i) {
    if (print && h->rank == 0) printf("-- Alignment iteration %d --\n", i);
    const bool converged = align_meshes(h, max_d, thr, print != 0, i);
    if (converged) {
      if (print && h->rank == 0) { printf("Convergence is assumed as the maximum movement is less than the threshold.\n"); fflush(stdout); }
      return 1;
    }
  }
  if (print) fflush(stdout);
  return 0;
  E3D_CATCH()
}

int e3d_icp_get_pose(e3d_icp_t* h, int idx, float T[12]) {
  E3D_TRY
  if (!h || !T) throw Error(E3D_ERR_INVALID, "e3d_icp_get_pose: null argument");
  if (idx < 0 || idx >= (int)h->clouds.size()) throw Error(E3D_ERR_INDEX, fmt("cloud index %d out of range", idx));
  std::memcpy(T, h->clouds[idx]->T, sizeof(float) * 12);
  return 0;
  E3D_CATCH()
}

int e3d_icp_set_sequential_distance_sum(e3d_icp_t* h, int enable) {
  if (!h) { e3d::set_last_error("e3d_icp_set_sequential_distance_sum: null handle"); return E3D_ERR_INVALID; }
  h->sequential_dsum = enable != 0;
  return 0;
}

int e3d_icp_set_resident_rows(e3d_icp_t* h, int enable) {
  if (!h) { e3d::set_last_error("e3d_icp_set_resident_rows: null handle"); return E3D_ERR_INVALID; }
  h->resident_rows = enable != 0;
  return 0;
}

int e3d_icp_set_max_inner_iterations(e3d_icp_t* h, int n) {
  if (!h || n < 0) { e3d::set_last_error("e3d_icp_set_max_inner_iterations: bad argument"); return E3D_ERR_INVALID; }
  h->max_inner = n;
  return 0;
}

size_t e3d_icp_num_pair_records(const e3d_icp_t* h) { return h ? h->pair_records.size() : 0; }
const e3d_icp_pair_record* e3d_icp_pair_records(const e3d_icp_t* h) { return h ? h->pair_records.data() : nullptr; }
size_t e3d_icp_num_iter_records(const e3d_icp_t* h) { return h ? h->iter_records.size() : 0; }
const e3d_icp_iter_record* e3d_icp_iter_records(const e3d_icp_t* h) { return h ? h->iter_records.data() : nullptr; }
void e3d_icp_clear_records(e3d_icp_t* h) { if (h) { h->pair_records.clear(); h->iter_records.clear(); } }

int e3d_icp_set_shard(e3d_icp_t* h, int rank, int world, e3d_allreduce_fn fn, void* user) {
  if (!h || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) {
    e3d::set_last_error("e3d_icp_set_shard: bad argument");
    return E3D_ERR_INVALID;
  }
  h->rank = rank; h->world = world; h->allreduce = fn; h->allreduce_user = user;
  return 0;
}

int e3d_icp_set_comm(e3d_icp_t* h, e3d_comm_t* comm) {
  if (!h) { e3d::set_last_error("e3d_icp_set_comm: null handle"); return E3D_ERR_INVALID; }
  if (comm && comm->device != h->device) { e3d::set_last_error("e3d_icp_set_comm: communicator and handle live on different devices"); return E3D_ERR_INVALID; }
  h->comm = comm;
  h->rank = comm ? comm->rank : 0; h->world = comm ? comm->world : 1;
  h->allreduce = nullptr; h->allreduce_user = nullptr;
  return 0;
}

// ---- stand-alone entry points ------------------------------------------------------------------
int64_t e3d_find_correspondences(const float* sxyz, size_t ns, const float* txyz, size_t nt, float d,
                                 int32_t* match_index, float* sq_distance) {
  E3D_TRY
  if ((!sxyz && ns) || (!txyz && nt) || !match_index || !sq_distance) throw Error(E3D_ERR_INVALID, "e3d_find_correspondences: null argument");
  std::unique_ptr<e3d_icp> h(e3d_icp_create());
  if (!h) throw Error(E3D_ERR_NO_DEVICE, e3d::last_error_cstr());
  Cloud src, tgt;
  upload_cloud(h.get(), src, sxyz, sxyz, ns);   // normals unused here
  upload_cloud(h.get(), tgt, txyz, txyz, nt);
  build_grid(h.get(), src, d);
  build_grid(h.get(), tgt, d);
  transform_cloud(h.get(), src);
  transform_cloud(h.get(), tgt);
  hipStream_t s = h->stream;
  DevBuf<int> out_idx; DevBuf<float> out_d2;
  out_idx.reserve(ns); out_d2.reserve(ns);
  int64_t count = 0;
  if (ns > 0) {
    if (nt > 0) {
      h->match_pos.reserve(ns); h->match_d2.reserve(ns);
      const int mode = g_nn_mode;
      const bool dense = mode >= 2 || (mode == 0 && (double)tgt.n >= 4.0 * (double)std::max(tgt.n_cells, 1u));
      const unsigned* order = nullptr;
      if (dense) {
        h->keys_a.reserve(ns); h->keys_b.reserve(ns); h->vals_a.reserve(ns); h->vals_b.reserve(ns);
        const InvMap im = make_invmap(tgt);
        launch_query_keys(src.G4.p, ns, tgt.grid, im, tgt.qrange, h->keys_a.p, h->vals_a.p, s);
        sort_pairs_u64_u32(h->keys_a.p, h->keys_b.p, h->vals_a.p, h->vals_b.p, ns, tgt.key_bits, h->sort_temp, s);
        bool source_order = false;
        if (tgt.has_dense && mode != 2) {
          h->lbe_scratch.reserve(ns);
          source_order = launch_rows(mode, tgt, src.G4.p, h->vals_b.p, ns, im, radius_sq(d), make_cert_params(tgt, MotionBound{}), h->match_pos.p,
                                     h->match_d2.p, h->lbe_scratch.p, nullptr, s);
        } else {
          launch_nn_cells(src.G4.p, h->vals_b.p, ns, tgt.G4.p, tgt.table.p, nullptr, tgt.grid, im, tgt.qrange, radius_sq(d), h->match_pos.p, h->match_d2.p, s);
        }
        order = source_order ? nullptr : h->vals_b.p;
      } else {
        launch_nn_query(src.G4.p, ns, tgt.G4.p, tgt.table.p, tgt.grid, make_invmap(tgt), radius_sq(d), h->match_pos.p, h->match_d2.p, s);
      }
      launch_unpermute_matches(h->match_pos.p, h->match_d2.p, order, ns, src.G4.p, tgt.G4.p, out_idx.p, out_d2.p, s);
      copy_out(match_index, out_idx.p, sizeof(int) * ns, s);
      copy_out(sq_distance, out_d2.p, sizeof(float) * ns, s);
      const size_t nb = div_up(ns, kBlock);
      h->block_counts.reserve(nb); h->block_offsets.reserve(nb); h->block_d2.reserve(nb);
      h->d_total.reserve(3); h->d_total_d2.reserve(1); h->h_total.reserve(3);
      h->chunk_sum.reserve(div_up(nb, 256) + 1); h->chunk_d2.reserve(div_up(nb, 256) + 1);
      launch_match_scan(h->match_pos.p, h->match_d2.p, ns, h->block_counts.p, h->block_offsets.p, h->block_d2.p, h->chunk_sum.p, h->chunk_d2.p, h->d_total.p, h->d_total_d2.p, s);
      copy_out(h->h_total.p, h->d_total.p, sizeof(unsigned long long), s);
      sync(h.get());
      count = (int64_t)h->h_total.p[0];
    } else {
      E3D_HIP(hipMemsetAsync(out_idx.p, 0xFF, sizeof(int) * ns, s));
      E3D_HIP(hipMemsetAsync(out_d2.p, 0, sizeof(float) * ns, s));
      copy_out(match_index, out_idx.p, sizeof(int) * ns, s);
      copy_out(sq_distance, out_d2.p, sizeof(float) * ns, s);
      sync(h.get());
    }
  }
  return count;
  E3D_CATCH()
}

int e3d_transform_cloud(const float* xyz, const float* normals, size_t n, const float T[12], float* oxyz,
                        float* onrm, float bmin[3], float bmax[3]) {
  E3D_TRY
  if ((!xyz && n) || !T || (!oxyz && n) || !bmin || !bmax) throw Error(E3D_ERR_INVALID, "e3d_transform_cloud: null argument");
  std::unique_ptr<e3d_icp> h(e3d_icp_create());
  if (!h) throw Error(E3D_ERR_NO_DEVICE, e3d::last_error_cstr());
  hipStream_t s = h->stream;
  DevBuf<float> ix, in, ox, on;
  ix.reserve(3 * n); ox.reserve(3 * n);
  copy_in(ix.p, xyz, sizeof(float) * 3 * n, s);
  if (normals) { in.reserve(3 * n); on.reserve(3 * n); copy_in(in.p, normals, sizeof(float) * 3 * n, s); }
  ensure_bbox_scratch(h.get());
  launch_transform_aos(ix.p, normals ? in.p : nullptr, n, to_affine(T), ox.p, normals ? on.p : nullptr,
                       h->bbox_partial.p, h->bbox_out.p, s);
  copy_out(oxyz, ox.p, sizeof(float) * 3 * n, s);
  if (normals && onrm) copy_out(onrm, on.p, sizeof(float) * 3 * n, s);
  copy_out(h->h_bbox.p, h->bbox_out.p, sizeof(float) * 6, s);
  sync(h.get());
  for (int k = 0; k < 3; ++k) { bmin[k] = h->h_bbox.p[k]; bmax[k] = h->h_bbox.p[3 + k]; }
  return 0;
  E3D_CATCH()
}

int e3d_icp_pair_system(const float* sxyz, const float* snrm, const float* txyz, const float* tnrm,
                        const int32_t* iq, const int32_t* im, int64_t n, const float sq[4], const float st[3],
                        const float tq[4], const float tt[3], double H[144], double b[12], double* cost) {
  E3D_TRY
  if (!sxyz || !snrm || !txyz || !tnrm || (!iq && n) || (!im && n) || !sq || !st || !tq || !tt || !H || !b || !cost)
    throw Error(E3D_ERR_INVALID, "e3d_icp_pair_system: null argument");
  std::unique_ptr<e3d_icp> h(e3d_icp_create());
  if (!h) throw Error(E3D_ERR_NO_DEVICE, e3d::last_error_cstr());
  hipStream_t s = h->stream;
  // the caller's clouds may be host or device memory of unknown size: gather on the host side of the ABI is not
  // possible without sizes, so sizes are derived from the index lists
  int32_t ms = -1, mt = -1;
  std::vector<int32_t> hq((size_t)n), hm((size_t)n);
  E3D_HIP(hipMemcpy(hq.data(), iq, sizeof(int32_t) * (size_t)n, hipMemcpyDefault));
  E3D_HIP(hipMemcpy(hm.data(), im, sizeof(int32_t) * (size_t)n, hipMemcpyDefault));
  for (int64_t c = 0; c < n; ++c) { ms = std::max(ms, hq[c]); mt = std::max(mt, hm[c]); }
  const size_t ns = (size_t)(ms + 1), nt = (size_t)(mt + 1);
  DevBuf<float> dsx, dsn, dtx, dtn; DevBuf<int> dq, dm;
  dsx.reserve(3 * ns); dsn.reserve(3 * ns); dtx.reserve(3 * nt); dtn.reserve(3 * nt); dq.reserve(n); dm.reserve(n);
  copy_in(dsx.p, sxyz, sizeof(float) * 3 * ns, s); copy_in(dsn.p, snrm, sizeof(float) * 3 * ns, s);
  copy_in(dtx.p, txyz, sizeof(float) * 3 * nt, s); copy_in(dtn.p, tnrm, sizeof(float) * 3 * nt, s);
  copy_in(dq.p, hq.data(), sizeof(int) * (size_t)n, s); copy_in(dm.p, hm.data(), sizeof(int) * (size_t)n, s);
  h->cA.reserve(n); h->cB.reserve(n); h->cC.reserve(n);
  launch_gather_corr(dsx.p, dsn.p, dtx.p, dtn.p, dq.p, dm.p, (size_t)n, h->cA.p, h->cB.p, h->cC.p, s);
  LmSet S{};
  S.A = h->cA.p; S.B = h->cB.p; S.C = h->cC.p; S.glist = nullptr; S.outer = 0;
  S.n = n; S.block_begin = 0; S.nblocks = lm_blocks_for(n); S.mode = kModeTwoCross; S.side = 0;
  quat_to_matrix<float>(sq[0], sq[1], sq[2], sq[3], S.Rs);
  quat_to_matrix<float>(tq[0], tq[1], tq[2], tq[3], S.Rt);
  for (int k = 0; k < 3; ++k) { S.ts[k] = st[k]; S.tt[k] = tt[k]; }
  DevBuf<LmSet> dset; dset.reserve(1);
  DevBuf<int> dbs; dbs.reserve(S.nblocks);
  std::vector<int> bs(S.nblocks, 0);
  DevBuf<double> part, out; part.reserve((size_t)S.nblocks * kLmSlot); out.reserve(kLmSlot);
  copy_in(dset.p, &S, sizeof S, s); copy_in(dbs.p, bs.data(), sizeof(int) * S.nblocks, s);
  launch_lm_pass(kModeTwoCross, dset.p, dbs.p, 0, S.nblocks, part.p, s);
  launch_lm_reduce(part.p, dset.p, 1, kLmSlot, out.p, s);
  double r[kLmSlot];
  copy_out(r, out.p, sizeof r, s);
  sync(h.get());
  std::memset(H, 0, sizeof(double) * 144); std::memset(b, 0, sizeof(double) * 12);
  *cost = r[0];
  int k = 1;
  for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) { H[12 * a + c] = r[k]; H[12 * c + a] = r[k]; ++k; }
  for (int a = 0; a < 6; ++a) b[a] = r[22 + a];
  k = 28;
  for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) { H[12 * (6 + a) + 6 + c] = r[k]; H[12 * (6 + c) + 6 + a] = r[k]; ++k; }
  for (int a = 0; a < 6; ++a) b[6 + a] = r[49 + a];
  for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) { H[12 * a + 6 + c] = r[55 + 6 * a + c]; H[12 * (6 + c) + a] = r[55 + 6 * a + c]; }
  return 0;
  E3D_CATCH()
}

}  // extern "C"
