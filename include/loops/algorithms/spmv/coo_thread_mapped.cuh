/**
 * @file coo_thread_mapped.cuh
 * @brief `algorithms::spmv::coo_thread_mapped(coo, x, y, stream) -> util::timer_t`: one thread per
 * nonzero over `layout::coo`, atomicAdd into y (reference
 * include/loops/algorithms/spmv/coo_thread_mapped.cuh:37-100).  y must be zero-filled.
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
#include <loops/kernels/coo_spmv.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename setup_t, typename index_t, typename type_t>
__global__ void __coo_thread_mapped(setup_t config, const index_t* row_indices, const index_t* col_indices,
                                    const type_t* values, const type_t* x, type_t* y) {
  for (auto t : config.tiles())
    for (auto atom : config.atoms(t)) atomicAdd(&y[row_indices[atom]], values[atom] * x[col_indices[atom]]);
}

/// The drop-in entry (reference coo_thread_mapped.cuh:60: y zero-filled by the caller, any triplet order, the timer brackets
/// the kernel): since round 4 it launches the run kernel (loops/kernels/coo_spmv.hxx: a lane owns 8 consecutive triplets,
/// 16-byte loads, one atomicAdd per run of equal rows; C2 2.00 -> 0.113 ms).  The reference's one-atomic-per-triplet kernel
/// stays as `__coo_thread_mapped` behind coo_thread_mapped_schedule_api.
template <typename index_t, typename type_t>
util::timer_t coo_thread_mapped(coo_t<index_t, type_t>& coo, vector_t<type_t>& x, vector_t<type_t>& y,
                                xpu::stream_t stream = 0) {
  util::timer_t timer(stream);
  timer.start();
  kernels::launch_coo_runs(stream, coo.nnzs, coo.row_indices.data().get(), coo.col_indices.data().get(),
                           coo.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

template <typename index_t, typename type_t>
util::timer_t coo_thread_mapped_schedule_api(coo_t<index_t, type_t>& coo, vector_t<type_t>& x, vector_t<type_t>& y,
                                             xpu::stream_t stream = 0) {
  using layout_t = layout::coo<index_t, index_t>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, index_t, std::size_t,
                                  std::size_t, layout_t>;
  setup_t config(layout_t(static_cast<index_t>(coo.nnzs)));
  constexpr std::size_t block_size = 128;
  util::timer_t timer(stream);
  timer.start();
  if (coo.nnzs > 0)
    launch::non_cooperative(stream, __coo_thread_mapped<setup_t, index_t, type_t>,
                            dim3(static_cast<unsigned>(math::ceil_div(coo.nnzs, block_size))), dim3(block_size), config,
                            coo.row_indices.data().get(), coo.col_indices.data().get(), coo.values.data().get(),
                            x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

/// The name the run kernel had before coo_thread_mapped was routed to it (kept for callers of rounds 1-3).
template <typename index_t, typename type_t>
util::timer_t coo_run_mapped(coo_t<index_t, type_t>& coo, vector_t<type_t>& x, vector_t<type_t>& y,
                             xpu::stream_t stream = 0) {
  return coo_thread_mapped(coo, x, y, stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
