/**
 * @file csc.hxx
 * @brief `csc_t<index_t, offset_t, value_t, space>`: compressed sparse COLUMN container -- offsets[cols + 1],
 * indices[nnz] (row ids), values[nnz] (reference include/loops/container/csc.hxx:36-107).  Storage and
 * the COO conversion live in detail::compressed_t; a CSR source goes through COO.
 */
#pragma once

#include <loops/container/formats.hxx>
#include <loops/container/detail/compressed.hxx>

namespace loops {

template <typename index_t, typename offset_t, typename value_t, memory_space_t space = memory_space_t::device>
struct csc_t : detail::compressed_t<detail::major_axis::column, index_t, offset_t, value_t, space> {
  using storage_t = detail::compressed_t<detail::major_axis::column, index_t, offset_t, value_t, space>;

  csc_t() = default;
  csc_t(std::size_t r, std::size_t c, std::size_t nnz) : storage_t(r, c, nnz) {}

  /// Copy across memory spaces.
  template <auto rhs_space>
  csc_t(const csc_t<index_t, offset_t, value_t, rhs_space>& rhs)
      : storage_t(static_cast<const detail::compressed_t<detail::major_axis::column, index_t, offset_t, value_t, rhs_space>&>(rhs)) {}

  /// COO -> CSC (the input is copied; it need not be sorted).
  template <auto rhs_space>
  csc_t(const coo_t<index_t, value_t, rhs_space>& coo) : storage_t(coo) {}

  /// CSR -> CSC: expand the rows to triplets, then compress by column.
  template <auto rhs_space>
  csc_t(const csr_t<index_t, offset_t, value_t, rhs_space>& csr) : csc_t(coo_t<index_t, value_t, space>(csr)) {}
};

}  // namespace loops
