/**
 * @file matrix.cuh
 * @brief Dense row-major matrix for SpMM: `matrix_t<value_t, space>` owns `rows * cols` values in
 * `m_data` and doubles as its own device view -- copying a matrix_t (which is how it is passed to a
 * kernel) copies only {rows, cols, m_data_ptr}, never the storage.  Member names and the (r, c) /
 * flat [] accessors as the reference's callers use them (container/matrix.cuh:8-55).
 */
#pragma once

#include <cstddef>

#include <loops/core.hxx>

namespace loops {
namespace detail {

/// Non-owning {extent, pointer}: what device code sees of a dense matrix.
template <typename value_t>
struct dense_view_t {
  std::size_t rows = 0;
  std::size_t cols = 0;
  value_t* m_data_ptr = nullptr;

  __host__ __device__ __forceinline__ std::size_t flat(int r, int c) const {
    return cols * static_cast<std::size_t>(r) + static_cast<std::size_t>(c);
  }
};

}  // namespace detail

template <typename value_t, memory_space_t space = memory_space_t::device>
struct matrix_t : detail::dense_view_t<value_t> {
  using view_t = detail::dense_view_t<value_t>;

  vector_t<value_t, space> m_data;  ///< the storage (empty in a copy)

  matrix_t() = default;
  matrix_t(std::size_t r, std::size_t c) : view_t{r, c, nullptr}, m_data(r * c) {
    this->m_data_ptr = memory::raw_pointer_cast(m_data.data());
  }

  /// Shallow: the copy refers to `other`'s storage.
  __host__ __device__ matrix_t(const matrix_t& other) : view_t(static_cast<const view_t&>(other)) {}

  __host__ __device__ __forceinline__ value_t operator()(int r, int c) const { return this->m_data_ptr[this->flat(r, c)]; }
  __host__ __device__ __forceinline__ value_t& operator()(int r, int c) { return this->m_data_ptr[this->flat(r, c)]; }
  __host__ __device__ __forceinline__ value_t operator[](std::size_t i) const { return this->m_data_ptr[i]; }
  __host__ __device__ __forceinline__ value_t& operator[](std::size_t i) { return this->m_data_ptr[i]; }
};

}  // namespace loops
