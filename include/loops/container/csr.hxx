/**
 * @file csr.hxx
 * @brief `csr_t<index_t, offset_t, value_t, space>`: compressed sparse ROW container -- offsets[rows + 1],
 * indices[nnz] (column ids), values[nnz] (reference include/loops/container/csr.hxx:36-95).  Storage and
 * the COO conversion live in detail::compressed_t.
 */
#pragma once

#include <loops/container/formats.hxx>
#include <loops/container/detail/compressed.hxx>

namespace loops {

template <typename index_t, typename offset_t, typename value_t, memory_space_t space = memory_space_t::device>
struct csr_t : detail::compressed_t<detail::major_axis::row, index_t, offset_t, value_t, space> {
  using storage_t = detail::compressed_t<detail::major_axis::row, index_t, offset_t, value_t, space>;

  csr_t() = default;
  csr_t(std::size_t r, std::size_t c, std::size_t nnz) : storage_t(r, c, nnz) {}

  /// Copy across memory spaces.
  template <auto rhs_space>
  csr_t(const csr_t<index_t, offset_t, value_t, rhs_space>& rhs)
      : storage_t(static_cast<const detail::compressed_t<detail::major_axis::row, index_t, offset_t, value_t, rhs_space>&>(rhs)) {}

  /// COO -> CSR (the input is copied; it need not be sorted).
  template <auto rhs_space>
  csr_t(const coo_t<index_t, value_t, rhs_space>& coo) : storage_t(coo) {}
};

}  // namespace loops
