/**
 * @file core.hxx
 * @brief The vocabulary every other header of the library builds on, in one place: memory spaces and
 * pointer unwrapping (`loops::memory`), the host/device vector aliases, the small POD aggregates that
 * cross the host/device boundary (`coordinate_t`, `pair`) and the integer helpers (`loops::math`).
 *
 * The public include paths a gunrock/loops user knows (<loops/memory.hxx>, <loops/util/math.hxx>,
 * <loops/container/vector.hxx>, <loops/container/coordinate.hxx>) forward here; names, template
 * parameters and semantics are the ones the reference's callers rely on (memory.hxx:20-54,
 * util/math.hxx:15-29, container/vector.hxx:24-37, container/coordinate.hxx:15-20).
 */
#pragma once

#include <cstddef>
#include <type_traits>

#include <hip/hip_runtime.h>

#include <thrust/device_ptr.h>
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

namespace loops {

// ------------------------------------------------------------------------------ memory spaces
namespace memory {

enum memory_space_t { device, host, managed };

/// Raw address behind a thrust device pointer ...
template <typename type_t>
inline type_t* raw_pointer_cast(thrust::device_ptr<type_t> pointer) {
  return thrust::raw_pointer_cast(pointer);
}

/// ... and the identity for plain pointers, so generic code can unwrap either.
template <typename type_t>
__host__ __device__ inline type_t* raw_pointer_cast(type_t* pointer) {
  return pointer;
}

}  // namespace memory

using namespace memory;

// ------------------------------------------------------------------------------------ storage
namespace detail {
template <typename type_t, memory_space_t space>
struct vector_of {
  using type = thrust::device_vector<type_t>;  // device and managed data live in device vectors
};
template <typename type_t>
struct vector_of<type_t, memory_space_t::host> {
  using type = thrust::host_vector<type_t>;
};
}  // namespace detail

/// Owning 1-D storage in the given memory space.
template <typename type_t, memory_space_t space = memory_space_t::device>
using vector_t = typename detail::vector_of<type_t, space>::type;
template <typename type_t>
using host_vector_t = vector_t<type_t, memory_space_t::host>;
template <typename type_t>
using device_vector_t = vector_t<type_t, memory_space_t::device>;

// ----------------------------------------------------------------------------- POD aggregates
/// (x, y) position; x counts tiles and y atoms wherever a merge path is involved.
template <typename index_t>
struct coordinate_t {
  index_t x;
  index_t y;
};

/// first/second aggregate usable in device code (what `setup<work_oriented>::init()` returns).
template <typename first_t, typename second_t>
struct pair {
  first_t first;
  second_t second;
};

// ------------------------------------------------------------------------------ integer helpers
namespace math {

/// n / d rounded up, in the numerator's type.
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
