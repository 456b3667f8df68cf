/**
 * @file dia_thread_mapped.cuh
 * @brief `algorithms::spmv::dia_thread_mapped(dia, x, y, stream) -> util::timer_t`: one thread per row
 * over `layout::dia`; atom a of row r is diagonal d = a - r * num_diagonals, column r + offset[d]
 * (skipped when it falls outside the matrix); values are column-major, so a wavefront reads
 * values[d * stride + r] coalesced (reference include/loops/algorithms/spmv/dia_thread_mapped.cuh:36-110).
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
#include <loops/kernels/dia_spmv.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename setup_t, typename index_t, typename type_t>
__global__ void __dia_thread_mapped(setup_t config, std::size_t cols, std::size_t stride, std::size_t num_diagonals,
                                    const index_t* diag_offsets, const type_t* values, const type_t* x, type_t* y) {
  for (auto r : config.tiles()) {
    type_t acc = type_t{0};
    for (auto a : config.atoms(r)) {
      const std::size_t d = a - r * num_diagonals;
      const long c = static_cast<long>(r) + static_cast<long>(diag_offsets[d]);
      if (c >= 0 && c < static_cast<long>(cols)) acc += values[d * stride + r] * x[c];
    }
    y[r] = acc;
  }
}

/// The drop-in entry (reference dia_thread_mapped.cuh:67: y overwritten, the timer brackets the kernel): since round 4 it
/// launches the four-rows-per-lane kernel (loops/kernels/dia_spmv.hxx: 16-byte loads, several diagonals in flight; 38 ->
/// 26 us on the 2^20-row, 11-diagonal case).  The reference's lane-per-row kernel stays as `__dia_thread_mapped` behind
/// dia_thread_mapped_schedule_api.
template <typename index_t, typename offset_t, typename type_t>
util::timer_t dia_thread_mapped(dia_t<index_t, offset_t, type_t>& dia, vector_t<type_t>& x, vector_t<type_t>& y,
                                xpu::stream_t stream = 0) {
  util::timer_t timer(stream);
  timer.start();
  kernels::launch_dia_row4(stream, static_cast<int>(dia.rows), static_cast<int>(dia.cols), dia.stride,
                           static_cast<int>(dia.num_diagonals), dia.diag_offsets.data().get(), dia.values.data().get(),
                           x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

template <typename index_t, typename offset_t, typename type_t>
util::timer_t dia_thread_mapped_schedule_api(dia_t<index_t, offset_t, type_t>& dia, vector_t<type_t>& x, vector_t<type_t>& y,
                                             xpu::stream_t stream = 0) {
  using layout_t = layout::dia<std::size_t, std::size_t>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, std::size_t, std::size_t, std::size_t,
                                  std::size_t, layout_t>;
  setup_t config(layout_t(dia.rows, dia.num_diagonals));
  constexpr std::size_t block_size = 128;
  util::timer_t timer(stream);
  timer.start();
  if (dia.rows > 0)
    launch::non_cooperative(stream, __dia_thread_mapped<setup_t, index_t, type_t>,
                            dim3(static_cast<unsigned>(math::ceil_div(dia.rows, block_size))), dim3(block_size), config,
                            dia.cols, dia.stride, dia.num_diagonals, dia.diag_offsets.data().get(),
                            dia.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

/// The name the four-rows-per-lane kernel had before dia_thread_mapped was routed to it (kept for callers of rounds 1-3).
template <typename index_t, typename offset_t, typename type_t>
util::timer_t dia_row_mapped(dia_t<index_t, offset_t, type_t>& dia, vector_t<type_t>& x, vector_t<type_t>& y,
                             xpu::stream_t stream = 0) {
  return dia_thread_mapped(dia, x, y, stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
