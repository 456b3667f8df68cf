/**
 * @file csc_thread_mapped.cuh
 * @brief `algorithms::spmv::csc_thread_mapped(csc, x, y, stream) -> util::timer_t`: one thread per
 * column over `layout::csc`, x[col] read once, atomicAdd per nonzero (reference
 * include/loops/algorithms/spmv/csc_thread_mapped.cuh:37-96).  y must be zero-filled.
 */
#pragma once

#include <loops/schedule.hxx>
#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/device.hxx>
#include <loops/util/math.hxx>
#include <loops/util/timer.hxx>
#include <type_traits>
#include <loops/algorithms/spmv/launch_box.hxx>
#include <loops/memory.hxx>
#include <loops/kernels/csc_spmv.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename setup_t, typename index_t, typename type_t>
__global__ void __csc_thread_mapped(setup_t config, const index_t* row_indices, const type_t* values,
                                    const type_t* x, type_t* y) {
  for (auto col : config.tiles()) {
    const type_t x_col = x[col];
    for (auto atom : config.atoms(col)) atomicAdd(&y[row_indices[atom]], values[atom] * x_col);
  }
}

/// The drop-in entry (reference csc_thread_mapped.cuh:59: y zero-filled by the caller, the timer brackets the kernel):
/// since round 4 it launches the nonzero-split kernel (loops/kernels/csc_spmv.hxx: the nonzeros are split evenly over
/// the lanes, 16-byte loads -- no lane owns a hub column) or, from 2^20 nonzeros on, the binned product.  A caller that multiplies more than once should hold a
/// csc_plan_t (storage transposed to CSR once: 1.04 -> 0.10 ms on C2).  The reference's lane-per-column kernel stays as
/// `__csc_thread_mapped` behind csc_thread_mapped_schedule_api.
template <typename index_t, typename offset_t, typename type_t>
util::timer_t csc_thread_mapped(csc_t<index_t, offset_t, type_t>& csc, vector_t<type_t>& x, vector_t<type_t>& y,
                                xpu::stream_t stream = 0) {
  // From 2^20 nonzeros on the products TRAVEL instead of one memory-side atomic each (kernels::launch_csc_binned: products, one radix
  // pass into bins of 4 096 rows, LDS sums -- C2 0.26 against 1.03 ms, a hub row of 2^19 nonzeros 0.21 against 6.6); its scratch is
  // this call's (allocated outside the timed region).
  constexpr bool binnable = std::is_same<index_t, int>::value && (std::is_same<type_t, float>::value || std::is_same<type_t, double>::value);
  vector_t<unsigned char> scratch;
  bool binned = false;
  if constexpr (binnable) {
    if (csc.nnzs >= (std::size_t(1) << 20) && csc.rows < (std::size_t(1) << 31) && csc.nnzs < (std::size_t(1) << 31)) {
      scratch = vector_t<unsigned char>(kernels::csc_binned_scratch_bytes<int, type_t>(static_cast<int>(csc.rows), static_cast<int>(csc.nnzs)));
      binned = true;
    }
  }
  util::timer_t timer(stream);
  timer.start();
  if constexpr (binnable) {
    if (binned)
      kernels::launch_csc_binned(stream, static_cast<int>(csc.rows), static_cast<int>(csc.cols), static_cast<int>(csc.nnzs), csc.offsets.data().get(),
                                 csc.indices.data().get(), csc.values.data().get(), x.data().get(), y.data().get(), scratch.data().get());
  }
  if (!binned)
    kernels::launch_csc_nonzero_split(stream, static_cast<int>(csc.cols), static_cast<int>(csc.nnzs),
                                      csc.offsets.data().get(), csc.indices.data().get(), csc.values.data().get(),
                                      x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

template <typename index_t, typename offset_t, typename type_t>
util::timer_t csc_thread_mapped_schedule_api(csc_t<index_t, offset_t, type_t>& csc, vector_t<type_t>& x, vector_t<type_t>& y,
                                             xpu::stream_t stream = 0) {
  using layout_t = layout::csc<index_t, offset_t>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, offset_t, std::size_t,
                                  std::size_t, layout_t>;
  setup_t config(layout_t(csc.offsets.data().get(), static_cast<index_t>(csc.cols), static_cast<offset_t>(csc.nnzs)));
  constexpr std::size_t block_size = 128;
  util::timer_t timer(stream);
  timer.start();
  if (csc.cols > 0)
    launch::non_cooperative(stream, __csc_thread_mapped<setup_t, index_t, type_t>,
                            dim3(static_cast<unsigned>(math::ceil_div(csc.cols, block_size))), dim3(block_size), config,
                            csc.indices.data().get(), csc.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

/// The name the nonzero-split kernel had before csc_thread_mapped was routed to it (kept for callers of rounds 1-3).
template <typename index_t, typename offset_t, typename type_t>
util::timer_t csc_nonzero_mapped(csc_t<index_t, offset_t, type_t>& csc, vector_t<type_t>& x, vector_t<type_t>& y,
                                 xpu::stream_t stream = 0) {
  return csc_thread_mapped(csc, x, y, stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
