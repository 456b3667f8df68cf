/**
 * @file csr.hxx
 * @brief `csr_t<index_t, offset_t, value_t, space>`: compressed sparse row container
 * (reference include/loops/container/csr.hxx:36-95): offsets[rows + 1], indices[nnz], values[nnz].
 */
#pragma once

#include <utility>

#include <loops/container/formats.hxx>
#include <loops/container/detail/convert.hxx>
#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {
using namespace memory;

template <typename index_t, typename offset_t, typename value_t, memory_space_t space = memory_space_t::device>
struct csr_t {
  std::size_t rows;
  std::size_t cols;
  std::size_t nnzs;

  vector_t<offset_t, space> offsets;  ///< Ap, length rows + 1
  vector_t<index_t, space> indices;   ///< Aj, length nnzs
  vector_t<value_t, space> values;    ///< Ax, length nnzs

  csr_t() : rows(0), cols(0), nnzs(0) {}
  csr_t(std::size_t r, std::size_t c, std::size_t nnz)
      : rows(r), cols(c), nnzs(nnz), offsets(r + 1), indices(nnz), values(nnz) {}

  template <auto rhs_space>
  csr_t(const csr_t<index_t, offset_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), offsets(rhs.offsets), indices(rhs.indices),
        values(rhs.values) {}

  /// COO -> CSR: copy into this memory space, order row-major, compress the row ids.
  template <auto rhs_space>
  csr_t(const coo_t<index_t, value_t, rhs_space>& coo)
      : rows(coo.rows), cols(coo.cols), nnzs(coo.nnzs), offsets(coo.rows + 1) {
    coo_t<index_t, value_t, space> sorted(coo);
    sorted.sort_by_row();
    indices = std::move(sorted.col_indices);
    values = std::move(sorted.values);
    detail::indices_to_offsets(sorted.row_indices, offsets);
  }
};

}  // namespace loops
