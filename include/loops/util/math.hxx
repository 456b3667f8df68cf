/**
 * @file math.hxx
 * @brief Integer helpers. `ceil_div(n, d) = n / d + (n % d != 0)`, result in the numerator's
 * type (reference: include/loops/util/math.hxx:26-31; pinned by unittests/test_util_math.cu:22-67,
 * including n = INT64_MAX where the (n + d - 1) / d form would overflow).
 */
#pragma once

#include <hip/hip_runtime.h>

namespace loops {
namespace math {

template <class numerator_t, class denominator_t>
__host__ __device__ __forceinline__ constexpr numerator_t ceil_div(numerator_t const& n, denominator_t const& d) {
  return static_cast<numerator_t>(n / d + (n % d != 0 ? 1 : 0));
}

/// Number of halving steps a lower_bound over `n` elements needs at most.
__host__ __device__ __forceinline__ constexpr int log2_ceil(unsigned long long n) {
  int r = 0;
  while ((1ull << r) < n) ++r;
  return r;
}

}  // namespace math
}  // namespace loops
