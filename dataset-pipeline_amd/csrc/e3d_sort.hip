// e3d_sort.hip -- device radix sort of (cell key, point index) pairs.  The sort itself is the
// rocPRIM library primitive (its onesweep kernel with the gfx950 tuning).  Used at every grid build (one-off per cloud and search radius) AND on the
// per-iteration path: once per batch of directed pairs for the far lists of the first outer iterations (e3d_icp.hip: find_pairs_multi).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "e3d_icp_kernels.hpp"

namespace e3d {

// rocPRIM's temporary storage for a pair sort is a second copy of the arrays (8 - 12 B per element) plus histograms.  The far lists of an
// ICP run are sorted once per batch of pairs and outer iteration, and the number of queries that survive the pruning key kernel GROWS
// while two scans approach each other: a buffer sized for the sort at hand was re-allocated inside later, timed iterations (hipFree +
// hipMalloc of 2 GB beside 125 GB of resident rows: milliseconds as a rule, 2.8 s once in eight sessions).  n_reserve = the largest n a
// later sort of the same arrays can have (the lists' total length): the buffer is sized for that once.  E3D_ALLOC_TRACE=1 reports growth.
template <class K>
static void sort_pairs_impl(K* keys_in, K* keys_out, unsigned* vals_in, unsigned* vals_out, size_t n, int end_bit, DevBuf<char>& temp, hipStream_t s,
                            size_t n_reserve) {
  if (n == 0) return;
  size_t bytes = 0, need = 0;
  E3D_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s));
  need = bytes;
  if (bytes > temp.cap && n_reserve > n)
    E3D_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, n_reserve, 0u, (unsigned)end_bit, s));
  need = std::max(need, bytes);
  static const bool trace = [] { const char* e = getenv("E3D_ALLOC_TRACE"); return e && e[0] == '1'; }();
  if (trace && need > temp.cap) fprintf(stderr, "[sort] temporary storage %zu -> %zu bytes (n = %zu, this sort needs %zu)\n", temp.cap, need, n, bytes);
  temp.reserve(need);
  E3D_HIP(rocprim::radix_sort_pairs(temp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s));
}

void sort_pairs_u64_u32(unsigned long long* keys_in, unsigned long long* keys_out, unsigned* vals_in,
                        unsigned* vals_out, size_t n, int end_bit, DevBuf<char>& temp, hipStream_t s, size_t n_reserve) {
  sort_pairs_impl(keys_in, keys_out, vals_in, vals_out, n, end_bit, temp, s, n_reserve);
}

void sort_pairs_u32_u32(unsigned* keys_in, unsigned* keys_out, unsigned* vals_in, unsigned* vals_out, size_t n,
                        int end_bit, DevBuf<char>& temp, hipStream_t s, size_t n_reserve) {
  sort_pairs_impl(keys_in, keys_out, vals_in, vals_out, n, end_bit, temp, s, n_reserve);
}

void exclusive_max_scan_u32(unsigned* data, size_t n, DevBuf<char>& temp, hipStream_t s) {
  if (n == 0) return;
  size_t bytes = 0;
  E3D_HIP(rocprim::exclusive_scan(nullptr, bytes, data, data, 0u, n, rocprim::maximum<unsigned>(), s));
  temp.reserve(bytes);
  E3D_HIP(rocprim::exclusive_scan(temp.p, bytes, data, data, 0u, n, rocprim::maximum<unsigned>(), s));
}

}  // namespace e3d
