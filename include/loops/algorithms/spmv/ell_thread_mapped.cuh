/**
 * @file ell_thread_mapped.cuh
 * @brief `algorithms::spmv::ell_thread_mapped(ell, x, y, stream)`: one thread per row over
 * `layout::ell`, padding cells (column < 0) skipped (reference
 * include/loops/algorithms/spmv/ell_thread_mapped.cuh:28-84).
 */
#pragma once

#include <loops/schedule.hxx>
#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/device.hxx>
#include <loops/util/math.hxx>
#include <loops/util/timer.hxx>
#include <loops/algorithms/spmv/launch_box.hxx>
#include <loops/memory.hxx>
#include <loops/kernels/ell_spmv.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename setup_t, typename index_t, typename type_t>
__global__ void __ell_thread_mapped(setup_t config, const index_t* indices, const type_t* values, const type_t* x,
                                    type_t* y) {
  for (auto row : config.tiles()) {
    type_t sum = 0;
    for (auto atom : config.atoms(row)) {
      const index_t col = indices[atom];
      if (col >= 0) sum += values[atom] * x[col];
    }
    y[row] = sum;
  }
}

/// The drop-in entry (reference ell_thread_mapped.cuh:53: y overwritten, the call returns after the stream has drained):
/// since round 4 it launches the row-split kernel (loops/kernels/ell_spmv.hxx: G lanes per row read it as contiguous
/// 16-byte pieces and reduce across lanes; C2-shaped ELL 0.405 -> 0.094 ms).  The reference's lane-per-row kernel stays as
/// `__ell_thread_mapped` behind ell_thread_mapped_schedule_api.
template <typename index_t, typename type_t>
void ell_thread_mapped(ell_t<index_t, type_t>& ell, vector_t<type_t>& x, vector_t<type_t>& y,
                       xpu::stream_t stream = 0) {
  kernels::launch_ell_row_split(stream, ell.rows, ell.pitch, ell.indices.data().get(), ell.values.data().get(),
                                x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
}

template <typename index_t, typename type_t>
void ell_thread_mapped_schedule_api(ell_t<index_t, type_t>& ell, vector_t<type_t>& x, vector_t<type_t>& y,
                                    xpu::stream_t stream = 0) {
  using layout_t = layout::ell<index_t, index_t>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, index_t, std::size_t,
                                  std::size_t, layout_t>;
  setup_t config(layout_t(static_cast<index_t>(ell.rows), static_cast<index_t>(ell.pitch)));
  constexpr std::size_t block_size = 128;
  if (ell.rows > 0)
    launch::non_cooperative(stream, __ell_thread_mapped<setup_t, index_t, type_t>,
                            dim3(static_cast<unsigned>(math::ceil_div(ell.rows, block_size))), dim3(block_size), config,
                            ell.indices.data().get(), ell.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
}

/// The name the row-split kernel had before ell_thread_mapped was routed to it (kept for callers of rounds 1-3).
template <typename index_t, typename type_t>
void ell_row_mapped(ell_t<index_t, type_t>& ell, vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
  ell_thread_mapped(ell, x, y, stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
