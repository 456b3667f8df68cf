/**
 * @file flat_partitioned.cuh
 * @brief `algorithms::spmv::flat_partitioned<K = 8>(csr, x, y, stream) -> util::timer_t`:
 * thread_mapped over `layout::flat_uniform_occupancy<K, layout::csr>` -- perfectly balanced
 * K-nonzero tiles; a lane reads its K atoms with 16-byte loads, recovers the original row with ONE
 * `base().tile_of` and follows the row ends along its atoms; runs of same-row products are stitched
 * across the 64 lanes and ONE atomicAdd per row and wavefront reaches y
 * (kernels::flat_partitioned_stitched_spmv; the reference-shaped per-atom kernel stays available as
 * kernels::flat_partitioned_spmv, the per-thread-run kernel of round 1 as flat_partitioned_runs_spmv)
 * (reference include/loops/algorithms/spmv/flat_partitioned.cuh:46-110).  y must be zero-filled
 * (a freshly constructed vector_t is).
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
#include <loops/kernels/launch.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <std::size_t K = 8, typename index_t, typename offset_t, typename type_t>
util::timer_t flat_partitioned(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                               xpu::stream_t stream = 0) {
  util::timer_t timer(stream);
  timer.start();
  kernels::launch_flat_partitioned<K>(stream, csr.rows, csr.nnzs, csr.offsets.data().get(), csr.indices.data().get(),
                                      csr.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
