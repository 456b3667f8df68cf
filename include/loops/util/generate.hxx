/**
 * @file generate.hxx
 * @brief Deterministic test-input generators: `generate::random::uniform_distribution(begin, end,
 * lo, hi, seed)` fills a host or device range with per-element hashed-seed random numbers;
 * `generate::random::csr(rows, cols, sparsity, out)` builds a random CSR matrix.
 *
 * Element i is drawn from a minstd_rand engine (x <- 48271 x mod 2^31-1) seeded with
 * `hash(i) * seed`, ONE draw, mapped to [lo, hi] -- inclusive integer range when the bounds are
 * integral (the examples call it with (1, 10, 42u), so x is integer-valued), real range
 * otherwise.  This restates the arithmetic the reference obtains from Thrust
 * (include/loops/util/generate.hxx:33-79 + thrust::minstd_rand / uniform_int_distribution /
 * uniform_real_distribution) so the vectors are bit-identical; it is written as
 * a plain HIP kernel (one element per lane) rather than a thrust::transform over a lambda.
 */
#pragma once

#include <chrono>
#include <cstddef>
#include <iterator>
#include <type_traits>

#include <thrust/iterator/iterator_traits.h>
#include <thrust/device_ptr.h>

#include <loops/backend/xpu.hxx>
#include <loops/container/formats.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace generate {
namespace random {

__forceinline__ __host__ __device__ unsigned int hash(unsigned int a) {
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}

namespace detail {

/// One minstd_rand draw for element i, as a real in [0, 1): (u - 1) / (1 + (m - 2)).
template <typename real_t>
__host__ __device__ inline real_t unit_draw(std::size_t i, unsigned int useed) {
  constexpr unsigned long long m = 2147483647ull;
  const unsigned int seed = hash(static_cast<unsigned int>(i)) * useed;
  unsigned long long s = static_cast<unsigned long long>(seed) % m;
  if (s == 0) s = 1;
  const unsigned long long u = (48271ull * s) % m;
  return static_cast<real_t>(u - 1ull) / (static_cast<real_t>(1) + static_cast<real_t>((m - 1ull) - 1ull));
}

template <typename out_t, typename bound_t>
__host__ __device__ inline out_t draw(std::size_t i, bound_t lo, bound_t hi, unsigned int useed) {
  if constexpr (std::is_floating_point<bound_t>::value) {
    const bound_t r = unit_draw<bound_t>(i, useed);
    return static_cast<out_t>(r * (hi - lo) + lo);
  } else {  // inclusive integer range via a real draw over [lo, hi + 1)
    const double r = unit_draw<double>(i, useed);
    return static_cast<out_t>(static_cast<bound_t>(r * ((static_cast<double>(hi) + 1.0) - static_cast<double>(lo)) +
                                                    static_cast<double>(lo)));
  }
}

template <typename out_t, typename bound_t>
__global__ void fill_kernel(out_t* out, std::size_t n, bound_t lo, bound_t hi, unsigned int useed) {
  const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
  for (std::size_t i = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = draw<out_t, bound_t>(i, lo, hi, useed);
}

template <typename iterator_t>
using is_device_iterator =
    std::is_convertible<typename thrust::iterator_system<iterator_t>::type, thrust::device_system_tag>;

}  // namespace detail

template <typename iterator_t, typename type_t>
void uniform_distribution(iterator_t begin_it, iterator_t end_it, type_t begin, type_t end,
                          unsigned int useed = static_cast<unsigned int>(
                              std::chrono::system_clock::now().time_since_epoch().count())) {
  using out_t = typename thrust::iterator_value<iterator_t>::type;
  const std::size_t n = static_cast<std::size_t>(end_it - begin_it);
  if (n == 0) return;
  if constexpr (detail::is_device_iterator<iterator_t>::value) {
    out_t* ptr = thrust::raw_pointer_cast(&(*begin_it));
    const unsigned int blocks = static_cast<unsigned int>((n + 255) / 256 > 65535 ? 65535 : (n + 255) / 256);
    hipLaunchKernelGGL((detail::fill_kernel<out_t, type_t>), dim3(blocks), dim3(256), 0, 0, ptr, n, begin, end, useed);
    (void)xpu::stream_synchronize(0);
  } else {
    for (std::size_t i = 0; i < n; ++i) begin_it[i] = detail::draw<out_t, type_t>(i, begin, end, useed);
  }
}

using namespace memory;

/// Random CSR with ~sparsity * rows * cols entries (duplicates removed).
template <typename index_t, typename offset_t, typename value_t>
void csr(std::size_t rows, std::size_t cols, float sparsity, csr_t<index_t, offset_t, value_t>& matrix) {
  const std::size_t nnzs = static_cast<std::size_t>(sparsity * (rows * cols));
  coo_t<index_t, value_t, memory_space_t::host> coo(rows, cols, nnzs);
  uniform_distribution(coo.row_indices.begin(), coo.row_indices.end(), index_t(0), index_t(rows - 1));
  uniform_distribution(coo.col_indices.begin(), coo.col_indices.end(), index_t(0), index_t(cols - 1));
  uniform_distribution(coo.values.begin(), coo.values.end(), value_t(0.0), value_t(1.0));
  coo.remove_duplicates();
  matrix = csr_t<index_t, offset_t, value_t>(coo);
}

}  // namespace random
}  // namespace generate
}  // namespace loops
