/**
 * @file matrix.cuh
 * @brief `matrix_t<value_t, space>`: dense row-major matrix for SpMM's B and C
 * (reference include/loops/container/matrix.cuh:9-52).  Copies are non-owning views so the
 * object can be passed by value into kernels.
 */
#pragma once

#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {

template <typename value_t, memory_space_t space = memory_space_t::device>
struct matrix_t {
  std::size_t rows;
  std::size_t cols;

  vector_t<value_t, space> m_data;
  value_t* m_data_ptr;

  matrix_t() : rows(0), cols(0), m_data(), m_data_ptr(nullptr) {}
  matrix_t(std::size_t r, std::size_t c)
      : rows(r), cols(c), m_data(r * c), m_data_ptr(memory::raw_pointer_cast(m_data.data())) {}

  /// View of `other` (shares storage; what a kernel receives).
  __host__ __device__ matrix_t(const matrix_t<value_t, space>& other)
      : rows(other.rows), cols(other.cols), m_data_ptr(other.m_data_ptr) {}

  __host__ __device__ __forceinline__ value_t operator()(int r, int c) const { return m_data_ptr[cols * r + c]; }
  __host__ __device__ __forceinline__ value_t& operator()(int r, int c) { return m_data_ptr[cols * r + c]; }
  __host__ __device__ __forceinline__ value_t operator[](std::size_t i) const { return m_data_ptr[i]; }
  __host__ __device__ __forceinline__ value_t& operator[](std::size_t i) { return m_data_ptr[i]; }
};

}  // namespace loops
