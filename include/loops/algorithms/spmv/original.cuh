/**
 * @file original.cuh
 * @brief `algorithms::spmv::original`: the hand-written grid-stride row-per-thread baseline with no
 * schedule object (reference include/loops/algorithms/spmv/original.cuh:26-75).
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
void original(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
              xpu::stream_t stream = 0) {
  kernels::launch_original(stream, csr.rows, csr.cols, csr.nnzs, csr.offsets.data().get(), csr.indices.data().get(),
                           csr.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
