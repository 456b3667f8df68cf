/**
 * @file equal.hxx
 * @brief `util::equal(d_ptr, h_ptr, n, error_op, verbose)`: number of positions where a DEVICE array
 * and a HOST array differ under `error_op(device_value, host_value)` (default: `!=`); with `verbose`
 * every mismatch is printed at full precision.  Interface of the reference's helper
 * (util/equal.hxx:27-68).  The device array is staged through a bounded host buffer, so comparing a
 * multi-GB result does not need a second full host copy.
 */
#pragma once

#include <algorithm>
#include <cstddef>
#include <iomanip>
#include <iostream>
#include <limits>
#include <vector>

#include <loops/backend/xpu.hxx>

namespace loops {
namespace util {
namespace detail {
inline auto default_comparator = [](auto& a, auto& b) -> bool { return a != b; };
constexpr std::size_t compare_chunk = std::size_t(1) << 22;  // elements staged per copy
}  // namespace detail

template <typename type_t, typename comp_t = decltype(detail::default_comparator)>
std::size_t equal(const type_t* d_ptr, const type_t* h_ptr, const std::size_t n,
                  comp_t error_op = detail::default_comparator, const bool verbose = false) {
  std::vector<type_t> staged(std::min(n, detail::compare_chunk));
  std::size_t mismatches = 0;
  for (std::size_t base = 0; base < n; base += staged.size()) {
    const std::size_t m = std::min(staged.size(), n - base);
    (void)xpu::memcpy(staged.data(), d_ptr + base, m * sizeof(type_t), xpu::memcpy_device_to_host);
    for (std::size_t k = 0; k < m; ++k) {
      if (!error_op(staged[k], h_ptr[base + k])) continue;
      ++mismatches;
      if (verbose)
        std::cout << "Error[" << base + k << "]: " << std::setprecision(std::numeric_limits<type_t>::digits10)
                  << staged[k] << " != " << h_ptr[base + k] << std::endl;
    }
  }
  return mismatches;
}

}  // namespace util
}  // namespace loops
