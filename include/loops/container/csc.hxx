/**
 * @file csc.hxx
 * @brief `csc_t`: compressed sparse column container (reference include/loops/container/csc.hxx:34-120):
 * offsets[cols + 1], indices[nnz] = row ids, values[nnz].
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
struct csc_t {
  std::size_t rows;
  std::size_t cols;
  std::size_t nnzs;

  vector_t<offset_t, space> offsets;  ///< column offsets, length cols + 1
  vector_t<index_t, space> indices;   ///< row ids, length nnzs
  vector_t<value_t, space> values;

  csc_t() : rows(0), cols(0), nnzs(0) {}
  csc_t(std::size_t r, std::size_t c, std::size_t nnz)
      : rows(r), cols(c), nnzs(nnz), offsets(c + 1), indices(nnz), values(nnz) {}

  template <auto rhs_space>
  csc_t(const csc_t<index_t, offset_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), offsets(rhs.offsets), indices(rhs.indices),
        values(rhs.values) {}

  template <auto rhs_space>
  csc_t(const coo_t<index_t, value_t, rhs_space>& coo)
      : rows(coo.rows), cols(coo.cols), nnzs(coo.nnzs), offsets(coo.cols + 1) {
    coo_t<index_t, value_t, space> sorted(coo);
    sorted.sort_by_column();
    indices = std::move(sorted.row_indices);
    values = std::move(sorted.values);
    detail::indices_to_offsets(sorted.col_indices, offsets);
  }

  template <auto rhs_space>
  csc_t(const csr_t<index_t, offset_t, value_t, rhs_space>& csr) : csc_t(coo_t<index_t, value_t, space>(csr)) {}
};

}  // namespace loops
