/**
 * @file ell.hxx
 * @brief `ell_t`: ELLPACK container -- every row padded to `pitch` = max nonzeros per row, row-major,
 * padding cells carry column `sentinel()` = -1 and value 0 (reference include/loops/container/ell.hxx:45-160).
 */
#pragma once

#include <algorithm>

#include <thrust/copy.h>

#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {
using namespace memory;

template <typename index_t, typename value_t, memory_space_t space = memory_space_t::device>
struct ell_t {
  std::size_t rows;
  std::size_t cols;
  std::size_t nnzs;   ///< real (unpadded) nonzeros
  std::size_t pitch;  ///< padded row length

  vector_t<index_t, space> indices;  ///< rows * pitch column ids (sentinel() in padding)
  vector_t<value_t, space> values;   ///< rows * pitch

  static __host__ __device__ index_t sentinel() { return static_cast<index_t>(-1); }

  ell_t() : rows(0), cols(0), nnzs(0), pitch(0) {}
  ell_t(std::size_t r, std::size_t c, std::size_t nnz, std::size_t p)
      : rows(r), cols(c), nnzs(nnz), pitch(p), indices(r * p, sentinel()), values(r * p, value_t(0)) {}

  template <auto rhs_space>
  ell_t(const ell_t<index_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), pitch(rhs.pitch), indices(rhs.indices), values(rhs.values) {}

  /// Pre-flight probe: the pitch a CSR matrix would need (rows * pitch cells get allocated).
  template <typename offset_t, auto rhs_space>
  static std::size_t max_nnz_per_row(const csr_t<index_t, offset_t, value_t, rhs_space>& csr) {
    vector_t<offset_t, memory_space_t::host> off(csr.offsets);
    std::size_t widest = 0;
    for (std::size_t r = 0; r < csr.rows; ++r) widest = std::max<std::size_t>(widest, off[r + 1] - off[r]);
    return widest;
  }

  template <typename offset_t, auto rhs_space>
  ell_t(const csr_t<index_t, offset_t, value_t, rhs_space>& csr) : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs) {
    csr_t<index_t, offset_t, value_t, memory_space_t::host> h(csr);
    pitch = 0;
    for (std::size_t r = 0; r < rows; ++r) pitch = std::max<std::size_t>(pitch, h.offsets[r + 1] - h.offsets[r]);
    thrust::host_vector<index_t> h_idx(rows * pitch, sentinel());
    thrust::host_vector<value_t> h_val(rows * pitch, value_t(0));
    for (std::size_t r = 0; r < rows; ++r) {
      std::size_t cell = r * pitch;
      for (auto k = h.offsets[r]; k < h.offsets[r + 1]; ++k, ++cell) {
        h_idx[cell] = h.indices[k];
        h_val[cell] = h.values[k];
      }
    }
    indices = h_idx;
    values = h_val;
  }
};

}  // namespace loops
