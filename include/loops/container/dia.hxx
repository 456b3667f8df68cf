/**
 * @file dia.hxx
 * @brief `dia_t`: diagonal-format container.  `diag_offsets[d]` = (col - row) of stored diagonal d
 * (ascending), `values` column-major: values[d * stride + r] with stride = rows
 * (reference include/loops/container/dia.hxx:69-230).
 */
#pragma once

#include <algorithm>
#include <unordered_set>
#include <vector>

#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {
using namespace memory;

template <typename index_t, typename offset_t, typename value_t, memory_space_t space = memory_space_t::device>
struct dia_t {
  std::size_t rows;
  std::size_t cols;
  std::size_t nnzs;           ///< original nonzero count
  std::size_t stride;         ///< length of one stored diagonal (= rows)
  std::size_t num_diagonals;  ///< distinct (col - row) values

  vector_t<index_t, space> diag_offsets;
  vector_t<value_t, space> values;

  dia_t() : rows(0), cols(0), nnzs(0), stride(0), num_diagonals(0) {}

  /// Pre-flight probe: how many diagonals a CSR matrix touches (num_diagonals * rows cells).
  template <auto rhs_space, typename csr_offset_t>
  static std::size_t count_diagonals(const csr_t<index_t, csr_offset_t, value_t, rhs_space>& csr) {
    csr_t<index_t, csr_offset_t, value_t, memory_space_t::host> h(csr);
    std::unordered_set<index_t> seen;
    for (std::size_t r = 0; r < h.rows; ++r)
      for (auto a = h.offsets[r]; a < h.offsets[r + 1]; ++a)
        seen.insert(static_cast<index_t>(h.indices[a]) - static_cast<index_t>(r));
    return seen.size();
  }

  template <auto rhs_space>
  dia_t(const dia_t<index_t, offset_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), stride(rhs.stride), num_diagonals(rhs.num_diagonals),
        diag_offsets(rhs.diag_offsets), values(rhs.values) {}

  template <auto rhs_space, typename csr_offset_t>
  dia_t(const csr_t<index_t, csr_offset_t, value_t, rhs_space>& csr)
      : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs), stride(csr.rows) {
    csr_t<index_t, csr_offset_t, value_t, memory_space_t::host> h(csr);
    std::vector<index_t> diags;
    {
      std::unordered_set<index_t> seen;
      for (std::size_t r = 0; r < rows; ++r)
        for (auto a = h.offsets[r]; a < h.offsets[r + 1]; ++a)
          if (seen.insert(static_cast<index_t>(h.indices[a]) - static_cast<index_t>(r)).second)
            diags.push_back(static_cast<index_t>(h.indices[a]) - static_cast<index_t>(r));
    }
    std::sort(diags.begin(), diags.end());
    num_diagonals = diags.size();
    std::vector<value_t> cells(num_diagonals * stride, value_t{0});
    for (std::size_t r = 0; r < rows; ++r) {
      for (auto a = h.offsets[r]; a < h.offsets[r + 1]; ++a) {
        const index_t off = static_cast<index_t>(h.indices[a]) - static_cast<index_t>(r);
        const std::size_t d = std::lower_bound(diags.begin(), diags.end(), off) - diags.begin();
        cells[d * stride + r] = h.values[a];
      }
    }
    diag_offsets = vector_t<index_t, memory_space_t::host>(diags.begin(), diags.end());
    values = vector_t<value_t, memory_space_t::host>(cells.begin(), cells.end());
  }
};

}  // namespace loops
