// e3d_sort.hip -- device radix sort of (cell key, point index) pairs.  The sort itself is the
// rocPRIM library primitive (its onesweep kernel with the gfx950 tuning).  Used at every grid build (one-off per cloud and search radius) AND on the
// per-iteration path: once per batch of directed pairs for the far lists of the first outer iterations (e3d_icp.hip: find_pairs_multi).
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "e3d_icp_kernels.hpp"

namespace e3d {

void sort_pairs_u64_u32(unsigned long long* keys_in, unsigned long long* keys_out, unsigned* vals_in,
                        unsigned* vals_out, size_t n, int end_bit, DevBuf<char>& temp, hipStream_t s) {
  if (n == 0) return;
  size_t bytes = 0;
  E3D_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s));
  temp.reserve(bytes);
  E3D_HIP(rocprim::radix_sort_pairs(temp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s));
}

void sort_pairs_u32_u32(unsigned* keys_in, unsigned* keys_out, unsigned* vals_in, unsigned* vals_out, size_t n,
                        int end_bit, DevBuf<char>& temp, hipStream_t s) {
  if (n == 0) return;
  size_t bytes = 0;
  E3D_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s));
  temp.reserve(bytes);
  E3D_HIP(rocprim::radix_sort_pairs(temp.p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s));
}

void exclusive_max_scan_u32(unsigned* data, size_t n, DevBuf<char>& temp, hipStream_t s) {
  if (n == 0) return;
  size_t bytes = 0;
  E3D_HIP(rocprim::exclusive_scan(nullptr, bytes, data, data, 0u, n, rocprim::maximum<unsigned>(), s));
  temp.reserve(bytes);
  E3D_HIP(rocprim::exclusive_scan(temp.p, bytes, data, data, 0u, n, rocprim::maximum<unsigned>(), s));
}

}  // namespace e3d
