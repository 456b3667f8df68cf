/**
 * @file bcsr.hxx
 * @brief `bcsr_t<R, C, ...>`: block-CSR container with dense R x C blocks, row-major inside a block:
 * values[(b * R + i) * C + j] = A[block_row * R + i][block_col(b) * C + j].  The host builder from
 * CSR sorts each block-row's block columns ascending, zero-fills padding cells and lets a repeated
 * (row, col) entry overwrite (reference include/loops/container/bcsr.hxx:60-197).
 */
#pragma once

#include <algorithm>
#include <vector>

#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {
using namespace memory;

template <std::size_t R, std::size_t C, typename index_t, typename offset_t, typename value_t,
          memory_space_t space = memory_space_t::device>
struct bcsr_t {
  static_assert(R > 0 && C > 0, "bcsr_t: block dimensions must be positive");
  static constexpr std::size_t kBlockRows = R;
  static constexpr std::size_t kBlockCols = C;
  static constexpr std::size_t kBlockSize = R * C;

  std::size_t rows;
  std::size_t cols;
  std::size_t nnzs;            ///< nonzeros of the source matrix
  std::size_t num_block_rows;  ///< ceil(rows / R)
  std::size_t num_block_cols;  ///< ceil(cols / C)
  std::size_t num_blocks;      ///< stored (non-empty) blocks

  vector_t<offset_t, space> block_offsets;     ///< num_block_rows + 1
  vector_t<index_t, space> block_col_indices;  ///< num_blocks
  vector_t<value_t, space> values;             ///< num_blocks * R * C
  /// What algorithms::spmv::bcsr_thread_mapped<4, 4> found out about the block-row lengths on its first call on this matrix
  /// (kernels::bcsr_rows_even / bcsr_rows_skewed; 0 = not looked at yet): which of its two kernels later calls launch.  A hint
  /// only -- offsets edited in place afterwards keep the old choice, the product stays right.  Reset it to 0 to have them looked at again.
  mutable int row_length_class = 0;

  bcsr_t() : rows(0), cols(0), nnzs(0), num_block_rows(0), num_block_cols(0), num_blocks(0) {}

  template <auto rhs_space>
  bcsr_t(const bcsr_t<R, C, index_t, offset_t, value_t, rhs_space>& rhs)
      : rows(rhs.rows), cols(rhs.cols), nnzs(rhs.nnzs), num_block_rows(rhs.num_block_rows),
        num_block_cols(rhs.num_block_cols), num_blocks(rhs.num_blocks), block_offsets(rhs.block_offsets),
        block_col_indices(rhs.block_col_indices), values(rhs.values), row_length_class(rhs.row_length_class) {}

  template <auto rhs_space, typename csr_offset_t>
  bcsr_t(const csr_t<index_t, csr_offset_t, value_t, rhs_space>& csr) : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs) {
    num_block_rows = (rows + R - 1) / R;
    num_block_cols = (cols + C - 1) / C;
    csr_t<index_t, csr_offset_t, value_t, memory_space_t::host> h(csr);

    std::vector<offset_t> boff(num_block_rows + 1, offset_t{0});
    std::vector<index_t> bcol;
    std::vector<value_t> cells;
    std::vector<index_t> here;  // block columns of the current block-row
    for (std::size_t br = 0; br < num_block_rows; ++br) {
      const std::size_t r_lo = br * R, r_hi = std::min(r_lo + R, rows);
      here.clear();
      for (std::size_t r = r_lo; r < r_hi; ++r)
        for (auto a = h.offsets[r]; a < h.offsets[r + 1]; ++a)
          here.push_back(static_cast<index_t>(static_cast<std::size_t>(h.indices[a]) / C));
      std::sort(here.begin(), here.end());
      here.erase(std::unique(here.begin(), here.end()), here.end());
      const std::size_t first = bcol.size();
      bcol.insert(bcol.end(), here.begin(), here.end());
      cells.resize(cells.size() + here.size() * kBlockSize, value_t{0});
      boff[br + 1] = static_cast<offset_t>(bcol.size());
      for (std::size_t r = r_lo; r < r_hi; ++r) {
        for (auto a = h.offsets[r]; a < h.offsets[r + 1]; ++a) {
          const std::size_t col = static_cast<std::size_t>(h.indices[a]);
          const std::size_t slot = std::lower_bound(here.begin(), here.end(), static_cast<index_t>(col / C)) - here.begin();
          cells[(first + slot) * kBlockSize + (r - r_lo) * C + col % C] = h.values[a];
        }
      }
    }
    num_blocks = bcol.size();
    block_offsets = vector_t<offset_t, memory_space_t::host>(boff.begin(), boff.end());
    block_col_indices = vector_t<index_t, memory_space_t::host>(bcol.begin(), bcol.end());
    values = vector_t<value_t, memory_space_t::host>(cells.begin(), cells.end());
  }
};

}  // namespace loops
