// e3d_icp_kernels.hpp -- launch interface of the ICP kernels (see e3d_icp_kernels.hip).
#pragma once

#include "e3d_common.hpp"
#include "e3d_kernels.hpp"

namespace e3d {

// LM pass modes: which Gramian blocks a directed pair (src -> tgt) contributes
// (icp_point_to_plane_impl.h:82-113).  Variables exist for every impl cloud except cloud 0.
enum { kModeCost = 0,      // cost only
       kModeOne = 1,       // exactly one side has variables: 21 + 6 (+ cost)
       kModeTwo = 2,       // both sides, off-diagonal block would land in the lower triangle => dropped [QUIRK]
       kModeTwoCross = 3   // both sides, src block before tgt block: SS, TT, ST
};
constexpr int kLmMaxPoses = 10;      // LM tries (0 or 1)..9 evaluated by one k_lm_cost_multi pass
constexpr int kLmSlot = 91;          // doubles per block partial / per set result
constexpr int kMaxBboxBlocks = 2048;
constexpr int kRowCap = 128;         // candidates staged in LDS per wave and batch (k_nn_rows)
// An entry of a certificate search's FAR list is a source position (< 2^31) and, in bit 31, "this query had no partner": the flag
// rides through the key kernels and the radix sort as part of the value, every reader of a list masks it (kListIndexMask).
constexpr unsigned kListNoPartner = 0x80000000u, kListIndexMask = 0x7FFFFFFFu;
constexpr int kRowSpan = 4;          // max x-extent (cells) of a row segment handled at once (k_nn_rows)
constexpr int kNNCap = 256;          // candidates staged in LDS per wave and batch (k_nn_cells)

// Target-grid cell range a query must fall into to have any candidate (cells of the target's points +- 2),
// used to build dense sort keys for the queries.
struct QueryRange { int lo[3]; unsigned D[3]; };

// One directed pair's correspondence rows (three float4 planes: A = {sp.xyz, sn.x}, B = {sn.yz, tp.xy}, C = {tp.z, tn.xyz}),
// with the inner poses (R = so3().matrix() in f32, row-major) of its two impl clouds.  Two row layouts:
//   * compacted (glist == nullptr): rows [0, n) hold the pair's correspondences in source order, both halves in the GLOBAL
//     frame of the outer iteration (k_compact_corr; rewritten every outer iteration);
//   * resident (glist != nullptr): one row per QUERY of the pair at its source position, zero rows (normals 0: residuals and
//     Jacobian rows are exactly 0) for queries without a partner; glist lists the 64-row groups that hold at least one
//     correspondence, n = 64 * number of listed groups is the number of virtual rows the pass walks.  A half whose cloud can move
//     is kept in the cloud's LOCAL frame and the pass applies the outer pose (Tos / Tot, PCL's operation order: the bits of G4)
//     before the inner one, so a row only changes when the query's partner does (k_corr_update).
struct LmSet {
  const float4 *A, *B, *C; // row 0 of the pair's planes
  const unsigned* glist;   // resident rows: indices of the active 64-row groups (ascending); nullptr: compacted rows
  long long n;             // (virtual) rows
  int block_begin, nblocks;
  int mode;                // full-pass mode of this set
  int side;                // kModeOne: 0 = source has the variables, 1 = target
  int outer;               // bit 0: source half is local (apply Tos), bit 1: target half is local (apply Tot)
  float Rs[9], ts[3], Rt[9], tt[3];
  Affine Tos, Tot;         // global_T_cloud of the outer iteration for local halves
};

// inner poses of one set for one candidate LM try
struct LmPose { float Rs[9], ts[3], Rt[9], tt[3]; };

int launch_transform_aos(const float* xyz, const float* nrm, size_t n, const Affine& T, float* oxyz, float* onrm,
                         float* bbox_partial, float* bbox_out, hipStream_t s);
int launch_transform_bbox(const float4* L4, size_t n, const Affine& T, float4* G4, float* bbox_partial,
                          float* bbox_out, hipStream_t s);
void launch_bbox_aos(const float* xyz, size_t n, float* bbox_partial, float* bbox_out, hipStream_t s);
void launch_cell_keys(const float* xyz, size_t n, const GridDesc& g, unsigned long long* keys, unsigned* vals,
                      hipStream_t s);
void launch_permute(const float* xyz, const float* nrm, const unsigned* order, size_t n, float4* L4, float4* LN,
                    hipStream_t s);
void launch_count_cells(const unsigned long long* keys, size_t n, unsigned* counter, hipStream_t s);
void launch_build_table(const unsigned long long* keys, size_t n, HashEntry* table, unsigned mask, hipStream_t s);
void launch_nn_query(const float4* Gsrc, size_t n_src, const float4* Gtgt, const HashEntry* table, const GridDesc& g,
                     const InvMap& im, float r2, int* match_pos, float* match_d2, hipStream_t s);
void launch_match_scan(const int* match_pos, const float* match_d2, size_t n, unsigned* block_counts,
                       unsigned* block_offsets, double* block_d2, unsigned long long* chunk_sum, double* chunk_d2,
                       unsigned long long* total, double* total_d2, hipStream_t s);
void launch_query_keys(const float4* Gsrc, size_t n, const GridDesc& g, const InvMap& im, const QueryRange& qr,
                       unsigned long long* keys, unsigned* vals, hipStream_t s);
void launch_query_keys32(const float4* Gsrc, size_t n, const GridDesc& g, const InvMap& im, const QueryRange& qr,
                         unsigned* keys, unsigned* vals, hipStream_t s);
void launch_nn_cells(const float4* Gsrc, const unsigned* order, size_t n, const float4* Gtgt, const HashEntry* table,
                     const unsigned* dense_start, const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2,
                     int* match_pos, float* match_d2, hipStream_t s);
// Motion of a query relative to its target since the pair's state was created (round 5: a bound PER QUERY).  With M_k the map from
// the source's local frame into the target's at outer iteration k and c the centre of the source's bounding box, query p moves
// |M_{k+1} p - M_k p| <= ||dM_L|| |p - c| + |dM c| in the target's frame, where the target's points rest: the accumulated bound is
// a * rho + b with a = sum ||dM_L||, b = sum |dM c| (host, f64) and rho = |p - c| -- for rigid poses the distance of the query's
// GLOBAL position to the centre's (cs), which the kernels evaluate themselves.  The bound every point of a cloud obeys (||dL|| R +
// |dc|, rounds 2 - 4) is that of its corners: 7 - 14 cm per outer iteration while two scans are degrees apart, where a query two
// metres from the centre moves 2 cm.  a = 0, b = the clouds' bounds: the old certificate (non-rigid poses, E3D_NN_PERQUERY=0).
struct MotionBound {
  float a, b;            // rounded down (the certificate's writers) or up (k_nn_certify)
  float cs[3];           // global position of the source's bounding-box centre at the current pose
  float rho_err;         // absolute error of rho = |q - cs| as a kernel evaluates it (host: the rounding of the query's global
                         // coordinates -- it grows with their magnitude -- and of cs), rounded up
};
// certificate side of k_nn_rows: lbe = min(sqrt(second smallest d2), block_dist * cell_scale - cell_sub) + motion_lo
struct CertParams {
  float cell_scale;      // sigma_min * cell size (global distance of one local cell), rounded down
  float cell_sub;        // sigma_min * slack of the global -> local mapping and the cell boundaries, rounded up
  MotionBound lo;        // accumulated motion bound of the pair at this outer iteration, rounded down
};
// k_nn_rows writes its results at the queries' SOURCE positions order[pos] (match_pos, match_d2, lbe in source order)
void launch_nn_rows(const float4* Gsrc, const unsigned* order, size_t n, const float4* Gtgt, const unsigned* dense_start,
                    const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2, const CertParams& cert, int* match_pos,
                    float* match_d2, float* lbe, int* match2, hipStream_t s);
void launch_query_keys_list(const float4* Gsrc, const unsigned* list, size_t n, const GridDesc& g, const InvMap& im,
                            const QueryRange& qr, unsigned long long* keys, unsigned* vals, hipStream_t s);
void launch_query_keys32_list(const float4* Gsrc, const unsigned* list, size_t n, const GridDesc& g, const InvMap& im,
                              const QueryRange& qr, unsigned* keys, unsigned* vals, hipStream_t s);
// settles every query whose old partner is provably still the unique nearest neighbour within the radius (lbe - cum_up > new
// distance); lists the others: todo_near (old partner within sqrt(near2)) / todo_far, lengths in counts[0..1] (see k_nn_certify)
// none_near: queries without a partner go to todo_near as well (k_nn_bounded searches them beyond the radius)
// occupancy bits of the 27-cell blocks (stride_w 32-bit words per (y, z) row of the dense directory's range; tmp: as large as occ)
void launch_block_occupancy(const unsigned* dense_start, const QueryRange& qr, unsigned stride_w, unsigned* tmp, unsigned* occ, hipStream_t s);
// keys of the listed queries (nullptr: all) whose block holds a candidate, compacted; the others are settled (count[0] = pairs kept).
// from_state (list == nullptr only): match / match2 hold the last search's result -- queries without a partner are flagged from it
void launch_query_keys_prune(bool keys32, const float4* Gsrc, const unsigned* list, size_t n, const unsigned* occ, unsigned stride_w, const GridDesc& g,
                             const InvMap& im, const QueryRange& qr, float r2, const CertParams& cert, void* keys, unsigned* vals, unsigned* count,
                             int* match, int* match2, float* match_d2, float* lbe, bool from_state, hipStream_t s);
void launch_nn_certify(const float4* Gsrc, size_t n, const float4* Gtgt, const MotionBound& cum_up, float r2, float near2, bool none_near, int* match, int* match2,
                       const float* lbe, float* match_d2, unsigned* todo_near, unsigned* todo_far, unsigned* counts, hipStream_t s);
// bounded search (k_nn_bounded) of the listed queries around their old partners
struct BoundParams {
  float margin;          // the search covers radius (distance of the old partner) + margin: room for the next certificates
  float rho_scale;       // (1 + 1e-5) / sigma_min(target pose), rounded up
  float rho_pad;         // absolute slack of the global -> local mapping (local units), rounded up
  MotionBound lo;        // accumulated motion bound of the pair at this outer iteration, rounded down
  float cell_scale, cell_sub;   // as in CertParams: covered global distance = (distance to the scanned box's faces in cells) * cell_scale - cell_sub
  float np_extra;        // a query without a partner searches radius + np_extra (its certificate: nothing nearer than that)
};
// half_prefix: the half-cell directory (8 prefix bytes per grid cell; k_nn_bounded_half, used for long lists or always) or nullptr
void launch_nn_bounded(const float4* Gsrc, const unsigned* list, size_t n_list, const float4* Gtgt, const unsigned* dense_start,
                       const unsigned long long* half_prefix, bool half_always, const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2,
                       const BoundParams& bp, int* match, int* match2, float* match_d2, float* lbe, hipStream_t s);
// ---- a batch of directed pairs per launch (round 5; the kernels and why: e3d_icp_kernels.hip "a BATCH of directed pairs") --------
constexpr int kNnBatchPairs = 32;    // pairs per batch
constexpr int kNnBatchJobs = 2 * kNnBatchPairs;   // (pair, list) jobs of the bounded search: the near list and a short far list per pair
// what the one-pair kernels take as arguments, per pair of the batch (device pointers; block-uniform reads)
struct NnPairDev {
  const float4* Gsrc;              // this rank's slice of the source cloud, global frame
  const float4* Gtgt;              // target cloud, global frame
  const unsigned* S;               // target's dense cell-start directory
  const unsigned long long* H8;    // target's half-cell directory
  int *match, *match2;             // per-query state of the pair
  float* lbe;
  float* match_d2;                 // squared distances of this search (batch scratch)
  unsigned *todo_near, *todo_far;  // lists of the queries the certificates did not settle (batch scratch)
  unsigned* counts;                // their lengths: two words of the batch's array
  unsigned n;                      // queries
  int none_near;
  MotionBound cum_up;
  float near2;
  GridDesc g; InvMap im; QueryRange qr; BoundParams bp;
  // resident rows (k_corr_update)
  const float4 *Psrc, *LNsrc, *Ptgt, *LNtgt;
  int src_global, tgt_global;
  Affine Tsrc, Ttgt;
  float4 *A, *B, *C;
  int* plane_match;
  unsigned* glist;
  // per-block results of the row update that the certificate kernel writes itself for blocks it settles whole (round 6; nullptr: off)
  unsigned* upd_counts; double* upd_d2; unsigned* upd_groups; unsigned char* upd_done;
  // far list of this search (k_query_keys_multi, k_nn_rows_multi; round 6)
  const unsigned* far_list;        // the listed queries (nullptr: all far_n queries of the pair)
  const unsigned* occ;             // target's occupancy bits of the 27-cell blocks
  unsigned far_n, occ_stride;
  int far_flags;                   // bit 0: flags "had no partner" come from the state (list == nullptr), bit 1: settle the queries of empty blocks, bit 2: seeds
  unsigned rows_off, rows_n;       // the pair's stretch of the batch's sorted (key, query) array
  // seeds (k_query_seed_multi; far_flags bit 2): far-list queries with a probe of their own half cell (or their old partner) nearer
  // than sqrt(seed2) take that point as match[j] and go to seed_list -- a job of the bounded search -- instead of sort + k_nn_rows
  unsigned* seed_list;
  float seed2;
};
constexpr unsigned kQueryKeysBlock = 2048;   // queries per block of the compacting key kernels
constexpr unsigned kQuerySeedBlock = 1024;   // ... of the seeding key kernel (k_query_seed_multi)
struct NnBatchDev {
  int n_pairs, n_jobs;
  int key_shift;                       // bits of the cell keys in the batch's sort keys; the pair's index sits above them
  unsigned far_end[kNnBatchPairs];     // exclusive ends of the pairs' block ranges in the far lists' key kernel
  unsigned rows_end[kNnBatchPairs];    // ... in k_nn_rows_multi
  unsigned cert_end[kNnBatchPairs];    // exclusive ends of the pairs' block ranges in the certificate launch
  unsigned upd_end[kNnBatchPairs];     // ... in the row update = of their entries in the per-block result arrays
  unsigned chunk_end[kNnBatchPairs];   // ... of their 256-block chunks in the totals
  unsigned job_end[kNnBatchJobs];      // ... of the list jobs' block ranges in the bounded search
  int job_pair[kNnBatchJobs];
  unsigned job_n[kNnBatchJobs];
  const unsigned* job_list[kNnBatchJobs];
  NnPairDev pair[kNnBatchPairs];
};
void launch_nn_certify_multi(const NnBatchDev* batch, unsigned n_blocks, float r2, hipStream_t s);
void launch_nn_bounded_half_multi(const NnBatchDev* batch, unsigned n_blocks, float r2, hipStream_t s);
// far lists of a batch: counts[0] = (key, query) pairs written by all pairs, counts[1 + p] = by pair p (cleared by the caller)
void launch_query_keys_multi(bool keys32, const NnBatchDev* batch, unsigned n_blocks, float r2, void* keys, unsigned* vals, unsigned* counts, hipStream_t s);
// the same with seeds for the pairs whose far_flags carry bit 2 (block ranges far_end in units of kQuerySeedBlock queries):
// counts[1 + kNnBatchPairs + p] = queries of pair p written to its seed_list
void launch_query_seed_multi(bool keys32, const NnBatchDev* batch, unsigned n_blocks, float r2, void* keys, unsigned* vals, unsigned* counts, hipStream_t s);
void launch_nn_rows_multi(const NnBatchDev* batch, unsigned n_blocks, const unsigned* order, float r2, hipStream_t s);
void launch_corr_update_multi(const NnBatchDev* batch, unsigned n_blocks, unsigned* block_counts, double* block_d2, unsigned* block_groups, hipStream_t s);
// totals[3 p ..] = correspondences, active groups, rows rewritten of pair p; total_d2[p]; the pairs' group lists (three launches)
void launch_corr_totals_multi(const NnBatchDev* batch, int n_pairs, unsigned n_chunks, const unsigned* block_counts, const double* block_d2,
                              const unsigned* block_groups, unsigned long long* chunk_sum, double* chunk_d2, unsigned* chunk_groups,
                              unsigned* chunk_rewritten, unsigned long long* totals, double* total_d2, hipStream_t s);
constexpr int kNnCertBlockQueries = 2048;   // queries per block of the certificate kernels (kCertPerWave x waves per block)
constexpr int kNnScanChunk = 256;           // blocks per chunk of the totals
void launch_half_keys(const float* xyz, size_t n, const GridDesc& g, unsigned* keys, unsigned* vals, hipStream_t s);
void launch_cell_keys_ordered(const float* xyz, const unsigned* order, size_t n, const GridDesc& g, unsigned long long* keys, hipStream_t s);
void launch_half_prefix(const unsigned long long* keys, const float4* L4, size_t n, const GridDesc& g, const QueryRange& qr,
                        const unsigned* dense_start, unsigned long long* half_prefix, hipStream_t s);
struct MfParams { float S, r2s, eta2, delta4, delta4sq; };      // filter constants of k_nn_mfma (see mfma_filter_params)
bool mfma_filter_params(double cell, double sigma_max, int row_span, float r2, MfParams* P);
void launch_nn_mfma(const float4* Gsrc, const unsigned* order, size_t n, const float4* Gtgt, const unsigned* dense_start,
                    const GridDesc& g, const InvMap& im, const QueryRange& qr, float r2, const MfParams& P, int* match_pos,
                    float* match_d2, hipStream_t s);
int nn_row_span();
void launch_dense_counts(const unsigned long long* keys, size_t n, const QueryRange& qr, unsigned* counts, hipStream_t s);
// n_reserve: the temporary storage is sized for a sort of that many elements (a later sort of the same arrays; 0: for n)
void sort_pairs_u32_u32(unsigned* keys_in, unsigned* keys_out, unsigned* vals_in, unsigned* vals_out, size_t n,
                        int end_bit, DevBuf<char>& temp, hipStream_t s, size_t n_reserve = 0);
// in-place exclusive MAX scan of n unsigned values (rocPRIM, e3d_sort.hip); one-off per grid build
void exclusive_max_scan_u32(unsigned* data, size_t n, DevBuf<char>& temp, hipStream_t s);
void launch_compact_corr(const int* match_pos, const unsigned* order, size_t n, const unsigned* block_offsets, const float4* Gsrc,
                         const float4* LNsrc, const Affine& Tsrc, const float4* Gtgt, const float4* LNtgt,
                         const Affine& Ttgt, float4* A, float4* B, float4* C, size_t out_base, hipStream_t s);
void launch_gather_corr(const float* sxyz, const float* snrm, const float* txyz, const float* tnrm, const int* iq,
                        const int* im, size_t n, float4* A, float4* B, float4* C, hipStream_t s);
void launch_unpermute_matches(const int* match_pos, const float* match_d2, const unsigned* order, size_t n, const float4* Gsrc,
                              const float4* Gtgt, int* out_idx, float* out_d2, hipStream_t s);
void launch_match_d2_by_original(const int* match_pos, const float* match_d2, const unsigned* order, size_t n, const float4* Gsrc, float* out,
                                 hipStream_t s);
void launch_lm_pass(int mode, const LmSet* sets, const int* block_set, int block_base, int nblocks, double* partial, hipStream_t s);
void launch_lm_cost_multi(const LmSet* sets, const LmPose* poses, int n_sets, int n_poses, const int* block_set, int nblocks,
                          double* partial, hipStream_t s);
// Resident rows of one directed pair (see LmSet): rewrites the rows whose partner changed since the planes were last brought up
// to date (plane_match = the partner each row encodes, -1 zero row, anything else below -1 = never written), and produces the
// per-block match counts / squared-distance sums (same arithmetic as launch_match_scan) plus the number of active 64-row groups
// per block; launch_corr_totals then yields totals[0] = correspondences, totals[1] = active groups, totals[2] = rows rewritten
// (chunk_groups: 2 x the number of 256-block chunks), total_d2 and the group list.
void launch_corr_update(const int* match, int* plane_match, const float* match_d2, size_t n, const float4* Psrc, const float4* LNsrc,
                        bool src_global, const Affine& Tsrc, const float4* Ptgt, const float4* LNtgt, bool tgt_global, const Affine& Ttgt,
                        float4* A, float4* B, float4* C, unsigned* block_counts, double* block_d2, unsigned* block_groups, hipStream_t s);
void launch_corr_totals(size_t n, const unsigned* block_counts, const double* block_d2, const unsigned* block_groups,
                        unsigned long long* chunk_sum, double* chunk_d2, unsigned* chunk_groups, unsigned long long* totals,
                        double* total_d2, unsigned* glist, hipStream_t s);
void launch_lm_reduce(const double* partial, const LmSet* sets, int n_sets, int nacc, double* out, hipStream_t s);

// radix sort of (cell key, point index) pairs -- rocPRIM device primitive (e3d_sort.hip)
void sort_pairs_u64_u32(unsigned long long* keys_in, unsigned long long* keys_out, unsigned* vals_in,
                        unsigned* vals_out, size_t n, int end_bit, DevBuf<char>& temp, hipStream_t s, size_t n_reserve = 0);

}  // namespace e3d
