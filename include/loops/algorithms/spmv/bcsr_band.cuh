/**
 * @file bcsr_band.cuh
 * @brief `algorithms::spmv::bcsr_band_t<index_t, offset_t>`: a 4 x 4 fp32 BCSR held in the block-band layout
 * (loops/kernels/bcsr_band.hxx) -- bands of block-rows whose row sums live in LDS as fp64 words, the band's blocks sorted by
 * block column so that the 16-byte x gathers of a wavefront share lines, block products on v_mfma_f32_4x4x1.  A held plan for
 * repeated `algorithms::spmv::bcsr_thread_mapped<4, 4>` products (reference algorithms/spmv/bcsr_thread_mapped.cuh:36-123); the
 * header-API twin of loops_bcsr_band_plan_* (include/loops_amd.h).  No reference counterpart for the layout.
 *
 *   algorithms::spmv::bcsr_band_t<int, int> A(bcsr);          // bcsr: bcsr_t<4, 4, int, int, float> on the device
 *   A.spmv(x_padded, y);                                      // y = bcsr * x  (x padded to 4 * num_block_cols, as for bcsr_thread_mapped)
 *
 * One product in flight per object (it owns the partial-vector scratch).  The band sums are fp64 LDS atomics that arrive in no
 * fixed order: see include/loops_amd.h for when that cannot matter.
 */
#pragma once

#include <loops/container/bcsr.hxx>
#include <loops/container/vector.hxx>
#include <loops/error.hxx>
#include <loops/kernels/bcsr_band.hxx>
#include <loops/util/timer.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t>
struct bcsr_band_t {
  static_assert(sizeof(index_t) == 4 && sizeof(offset_t) == 4, "bcsr_band_t: 32-bit indices and offsets");
  std::size_t rows, num_block_rows, num_block_cols, num_blocks;
  kernels::bcsr_band_storage arrays;  ///< HB, bands, steps, chunks, kernel shape and the owned device arrays

  /// @param band_block_rows 0 = automatic (kernels::bcsr_band_block_rows), or a power of two in [16, 4096]
  /// @param target_chunks 0 = automatic (kernels::rowband_target_chunks)
  explicit bcsr_band_t(bcsr_t<4, 4, index_t, offset_t, float>& b, int band_block_rows = 0, int target_chunks = 0, xpu::stream_t stream = 0)
      : rows(b.rows), num_block_rows(b.num_block_rows), num_block_cols(b.num_block_cols), num_blocks(b.num_blocks) {
    const int err = kernels::bcsr_band_create(stream, static_cast<int>(rows), static_cast<int>(num_block_rows), static_cast<int>(num_block_cols),
                                              static_cast<int>(num_blocks), reinterpret_cast<const int*>(b.block_offsets.data().get()),
                                              reinterpret_cast<const int*>(b.block_col_indices.data().get()), b.values.data().get(),
                                              band_block_rows, target_chunks, arrays);
    error::throw_if_exception(err == kernels::rowband_e_badarg,
                              "bcsr_band_t: band_block_rows must be a power of two in [16, 4096] and every block column inside [0, num_block_cols)");
    error::throw_if_exception(err == kernels::rowband_e_range, "bcsr_band_t: row code + block column do not fit one 32-bit word at this band height");
    error::throw_if_exception(err != 0, "bcsr_band_t: build failed");
  }

  kernels::bcsr_band_view view() const { return arrays.view(); }

  /// Times every compiled kernel shape and keeps the plan's own unless another one is measurably faster (kernels::bcsr_band_tune).
  void tune(int repeats = 10, xpu::stream_t stream = 0) {
    error::throw_if_exception(kernels::bcsr_band_tune(stream, arrays, repeats, nullptr) != 0, "bcsr_band_t::tune failed");
  }

  /// y = A x; asynchronous on `stream`.  `x` holds 4 * num_block_cols elements (padded, as bcsr_thread_mapped wants it).
  void spmv_async(vector_t<float>& x, vector_t<float>& y, xpu::stream_t stream = 0) {
    if (rows == 0) return;
    if (num_blocks == 0) {
      (void)hipMemsetAsync(y.data().get(), 0, sizeof(float) * rows, stream);
      return;
    }
    kernels::launch_bcsr_band(stream, view(), x.data().get(), y.data().get());
  }

  util::timer_t spmv(vector_t<float>& x, vector_t<float>& y, xpu::stream_t stream = 0) {
    util::timer_t timer(stream);
    timer.start();
    spmv_async(x, y, stream);
    (void)xpu::stream_synchronize(stream);
    timer.stop();
    return timer;
  }
};

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
