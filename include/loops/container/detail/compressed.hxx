/**
 * @file compressed.hxx
 * @brief `detail::compressed_t`: the storage and conversions CSR and CSC have in common.  Both are
 * "compressed along one axis": `offsets[major + 1]` delimit, per major index (row for CSR, column for
 * CSC), a run of `indices` (the minor index) and `values`.  csr_t / csc_t (csr.hxx, csc.hxx) are this
 * type with the axis fixed; their public members (rows, cols, nnzs, offsets, indices, values) and
 * constructor signatures are the ones callers of the reference use (container/csr.hxx:36-95,
 * container/csc.hxx:36-107).
 */
#pragma once

#include <cstddef>
#include <utility>

#include <loops/core.hxx>
#include <loops/container/detail/convert.hxx>

namespace loops {

template <typename index_t, typename value_t, memory_space_t space>
struct coo_t;

namespace detail {

enum class major_axis { row, column };

template <major_axis AXIS, typename index_t, typename offset_t, typename value_t, memory_space_t space>
struct compressed_t {
  std::size_t rows = 0;
  std::size_t cols = 0;
  std::size_t nnzs = 0;

  vector_t<offset_t, space> offsets;  ///< length (rows or cols) + 1
  vector_t<index_t, space> indices;   ///< minor index of every nonzero, length nnzs
  vector_t<value_t, space> values;    ///< length nnzs

  compressed_t() = default;

  /// Uninitialised storage for an r x c matrix with nnz nonzeros.
  compressed_t(std::size_t r, std::size_t c, std::size_t nnz)
      : rows(r), cols(c), nnzs(nnz), offsets(major_extent(r, c) + 1), indices(nnz), values(nnz) {}

  /// Same matrix, possibly from the other memory space.
  template <memory_space_t rhs_space>
  explicit compressed_t(const compressed_t<AXIS, index_t, offset_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), offsets(rhs.offsets), indices(rhs.indices),
        values(rhs.values) {}

  /// From triplets: bring them into this memory space, order them along the major axis (stable in the
  /// minor one), keep the minor ids + values, compress the major ids into offsets.
  template <memory_space_t rhs_space>
  explicit compressed_t(const coo_t<index_t, value_t, rhs_space>& triplets)
      : rows(triplets.rows), cols(triplets.cols), nnzs(triplets.nnzs),
        offsets(major_extent(triplets.rows, triplets.cols) + 1) {
    coo_t<index_t, value_t, space> ordered(triplets);
    if constexpr (AXIS == major_axis::row) {
      ordered.sort_by_row();
      detail::indices_to_offsets(ordered.row_indices, offsets);
      indices = std::move(ordered.col_indices);
    } else {
      ordered.sort_by_column();
      detail::indices_to_offsets(ordered.col_indices, offsets);
      indices = std::move(ordered.row_indices);
    }
    values = std::move(ordered.values);
  }

  static constexpr std::size_t major_extent(std::size_t r, std::size_t c) {
    return AXIS == major_axis::row ? r : c;
  }
};

}  // namespace detail
}  // namespace loops
