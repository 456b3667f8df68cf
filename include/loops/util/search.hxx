/**
 * @file search.hxx
 * @brief The merge-path diagonal split.
 *
 * Given the tile-end sequence `a` (length a_len) and the atom sequence `b` (length b_len,
 * a counting sequence), `_binary_search(d, a, b, a_len, b_len)` returns the point
 * (x, y), x + y = d, where the merge path crosses diagonal d: x tiles and y atoms have been
 * consumed.  Semantics restated from the reference (include/loops/util/search.hxx:35-60):
 *
 *   x_min = max(int(d) - int(b_len), 0);   x_max = min(int(d), int(a_len));
 *   i*    = first i in [x_min, x_max) with NOT (a[i] <= b[d - i - 1])   (x_max if none,
 *           x_min if the interval is empty -- this is what makes past-the-end diagonals
 *           clamp to (a_len, b_len));
 *   return { unsigned(min(i*, a_len)), unsigned(d - i*) }.
 *
 * All arithmetic is `int`, results are `unsigned int`, exactly as in the reference, so the
 * coordinates are bit-identical (tests/test_schedules_gpu.py pins them against the oracle and
 * against the reference's own device code).  The search itself is a hand-written halving
 * loop (no thrust): every probe is one load from `a` -- global memory for the per-block
 * search, LDS for the per-thread search inside a merge tile.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <loops/container/coordinate.hxx>

namespace loops {
namespace search {

template <typename offset_t, typename xit_t, typename yit_t>
__host__ __device__ __forceinline__ coordinate_t<unsigned int> _binary_search(const offset_t& diagonal,
                                                                              const xit_t a,
                                                                              const yit_t b,
                                                                              const offset_t& a_len,
                                                                              const offset_t& b_len) {
  const int d = static_cast<int>(diagonal);
  int lo = d - static_cast<int>(b_len);
  if (lo < 0) lo = 0;
  int hi = d < static_cast<int>(a_len) ? d : static_cast<int>(a_len);
  // lower_bound over [lo, hi): an empty interval (lo >= hi) yields lo.
  int count = hi - lo;
  while (count > 0) {
    const int half = count >> 1;
    const int mid = lo + half;
    if (a[mid] <= b[d - mid - 1]) {
      lo = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  coordinate_t<unsigned int> c;
  c.x = static_cast<unsigned int>(lo < static_cast<int>(a_len) ? lo : static_cast<int>(a_len));
  c.y = static_cast<unsigned int>(d - lo);
  return c;
}

}  // namespace search
}  // namespace loops
