/**
 * @file thread_mapped.cuh
 * @brief `algorithms::spmv::thread_mapped(csr, x, y, stream)`: row per thread through
 * `schedule::setup<thread_mapped>` (reference include/loops/algorithms/spmv/thread_mapped.cuh:27-91).
 * Blocks on the stream before returning, like the reference wrapper.  The kernel keeps the schedule's row-to-thread map and
 * walks a row 16 / 4 atoms at a time (kernels::thread_mapped_batched_spmv): the same bits as the reference's loop, which
 * remains available through loops_spmv_csr_schedule_api_f32 / kernels::launch_thread_mapped(..., reference_shape = true).
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

template <typename index_t, typename offset_t, typename type_t>
void thread_mapped(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                   xpu::stream_t stream = 0) {
  kernels::launch_thread_mapped(stream, csr.rows, csr.cols, csr.nnzs, csr.offsets.data().get(),
                                csr.indices.data().get(), csr.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
