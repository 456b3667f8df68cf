/**
 * @file coo.hxx
 * @brief `coo_t<index_t, value_t, space>`: coordinate-format container (row, col, value triplets).
 * Member surface follows the reference (include/loops/container/coo.hxx:38-170): public
 * rows/cols/nnzs + three vectors, cross-space copy, construction from CSR, `sort_by_row()`,
 * `sort_by_column()`, `remove_duplicates()`.
 */
#pragma once

#include <thrust/iterator/zip_iterator.h>
#include <thrust/tuple.h>
#include <thrust/unique.h>

#include <loops/container/formats.hxx>
#include <loops/container/detail/convert.hxx>
#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {
using namespace memory;

template <typename index_t, typename value_t, memory_space_t space = memory_space_t::device>
struct coo_t {
  std::size_t rows;
  std::size_t cols;
  std::size_t nnzs;

  vector_t<index_t, space> row_indices;  ///< I, length nnzs
  vector_t<index_t, space> col_indices;  ///< J, length nnzs
  vector_t<value_t, space> values;       ///< V, length nnzs

  coo_t() : rows(0), cols(0), nnzs(0) {}
  coo_t(std::size_t r, std::size_t c, std::size_t nnz)
      : rows(r), cols(c), nnzs(nnz), row_indices(nnz), col_indices(nnz), values(nnz) {}

  template <auto rhs_space>
  coo_t(const coo_t<index_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), row_indices(rhs.row_indices), col_indices(rhs.col_indices),
        values(rhs.values) {}

  /// Expand a CSR matrix (one row id per nonzero).
  template <auto rhs_space, typename offset_t>
  coo_t(const csr_t<index_t, offset_t, value_t, rhs_space>& csr)
      : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs), row_indices(csr.nnzs), col_indices(csr.indices),
        values(csr.values) {
    vector_t<offset_t, space> row_offsets = csr.offsets;
    detail::offsets_to_indices(row_offsets, row_indices);
  }

  /// Expand a CSC matrix (one column id per nonzero; column-major order).  With `csr_t(coo)` -- which sorts row-major -- this
  /// is the CSC -> CSR path of the header API (no counterpart in the reference, whose conversions stop at csc_t(csr)):
  /// `csr_t<...> csr(coo_t<...>(csc));` then any CSR schedule or `spmv_plan_t`, instead of the scatter of csc_thread_mapped.
  template <auto rhs_space, typename offset_t>
  coo_t(const csc_t<index_t, offset_t, value_t, rhs_space>& csc)
      : rows(csc.rows), cols(csc.cols), nnzs(csc.nnzs), row_indices(csc.indices), col_indices(csc.nnzs), values(csc.values) {
    vector_t<offset_t, space> col_offsets = csc.offsets;
    detail::offsets_to_indices(col_offsets, col_indices);
  }

  /// Row-major order (ties by column).
  void sort_by_row() { detail::order_by<index_t, value_t, space>(row_indices, col_indices, values); }
  /// Column-major order (ties by row).
  void sort_by_column() { detail::order_by<index_t, value_t, space>(col_indices, row_indices, values); }

  /// Sort row-major and keep the first entry of every duplicated (row, col).
  void remove_duplicates() {
    sort_by_row();
    auto begin = thrust::make_zip_iterator(thrust::make_tuple(row_indices.begin(), col_indices.begin()));
    auto end = thrust::make_zip_iterator(thrust::make_tuple(row_indices.end(), col_indices.end()));
    auto new_end = thrust::unique_by_key(begin, end, values.begin());
    nnzs = static_cast<std::size_t>(new_end.second - values.begin());
    row_indices.resize(nnzs);
    col_indices.resize(nnzs);
    values.resize(nnzs);
  }
};

}  // namespace loops
