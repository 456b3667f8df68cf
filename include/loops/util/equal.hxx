/** @file equal.hxx  `util::equal(d_ptr, h_ptr, n, error_op, verbose)`: count device-vs-host
 *  mismatches under a caller-supplied predicate (reference include/loops/util/equal.hxx:46-70). */
#pragma once

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
}

template <typename type_t, typename comp_t = decltype(detail::default_comparator)>
std::size_t equal(const type_t* d_ptr, const type_t* h_ptr, const std::size_t n,
                  comp_t error_op = detail::default_comparator, const bool verbose = false) {
  std::vector<type_t> d(n);
  (void)xpu::memcpy(d.data(), d_ptr, n * sizeof(type_t), xpu::memcpy_device_to_host);
  std::size_t errors = 0;
  for (std::size_t i = 0; i < n; ++i) {
    if (error_op(d[i], h_ptr[i])) {
      if (verbose)
        std::cout << "Error[" << i << "]: " << std::setprecision(std::numeric_limits<type_t>::digits10) << d[i]
                  << " != " << h_ptr[i] << std::endl;
      ++errors;
    }
  }
  return errors;
}

}  // namespace util
}  // namespace loops
