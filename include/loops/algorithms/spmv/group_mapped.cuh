/**
 * @file group_mapped.cuh
 * @brief `algorithms::spmv::group_mapped(csr, x, y, stream)`: a workgroup owns 256 consecutive rows
 * and sweeps their concatenated nonzeros lane-strided (reference
 * include/loops/algorithms/spmv/group_mapped.cuh:27-105 -- CUDA-only there; this is the CDNA4
 * implementation: the merge-tile engine over the workgroup's own rows, groups of more than 24 tiles published and swept by a
 * second launch, one workgroup per 2 tiles (kernels/group_mapped_spmv.hxx) -- no floating-point atomics, no plan).  y does not
 * have to be zero-filled by the caller.
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
#include <loops/kernels/group_mapped_spmv.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
void group_mapped(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                  xpu::stream_t stream = 0) {
  constexpr int block_size = launch_t<type_t>::block_size;
  constexpr int items_per_thread = launch_t<type_t>::items_per_thread;
  // control words + carry-outs of the shared-out groups: zero-filled scratch of this call (thrust value-initialises), the words of
  // the column sample behind them (an x of 6 MB or more: plain or phased gathers decided on the device)
  const std::size_t share_bytes = (kernels::group_share_scratch_bytes<type_t>(static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), block_size,
                                                                              items_per_thread) + 255) & ~std::size_t(255);
  vector_t<unsigned char> scratch(share_bytes + kernels::scatter_scratch_words * sizeof(unsigned int));
  unsigned int* stats = reinterpret_cast<unsigned int*>(scratch.data().get() + share_bytes);
  kernels::launch_group_mapped_shared<block_size, items_per_thread, (items_per_thread % 2 == 0)>(
      stream, static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), csr.offsets.data().get(), csr.indices.data().get(),
      csr.values.data().get(), x.data().get(), y.data().get(), scratch.data().get(), nullptr, static_cast<int>(csr.cols), stats, /*share=*/true,
      /*resample=*/true, /*timed_path=*/true);  // (from 6 MB on: below, a group's partly empty last tile makes the phased order a loss)
  (void)xpu::stream_synchronize(stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
