/**
 * @file bcsr_thread_mapped.cuh
 * @brief `algorithms::spmv::bcsr_thread_mapped<R, C>(bcsr, x, y, stream) -> util::timer_t`
 * (reference include/loops/algorithms/spmv/bcsr_thread_mapped.cuh:36-123).  4 x 4 fp32 blocks
 * take the MFMA path (four chained v_mfma_f32_4x4x1_16b_f32 per block, 16 block-rows per
 * wavefront); every other shape and fp64 run the coalesced lane-group kernels of kernels/bcsr_spmv.hxx
 * (launch_bcsr_coalesced); index types other than int keep the thread-per-block-row schedule-API kernel.
 * x must be padded to num_block_cols * C entries.
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
#include <loops/kernels/bcsr_spmv.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <std::size_t R, std::size_t C, typename index_t, typename offset_t, typename type_t>
util::timer_t bcsr_thread_mapped(bcsr_t<R, C, index_t, offset_t, type_t>& bcsr, vector_t<type_t>& x,
                                 vector_t<type_t>& y, xpu::stream_t stream = 0) {
  util::timer_t timer(stream);
  timer.start();
  if constexpr (R == 4 && C == 4 && std::is_same<type_t, float>::value && std::is_same<index_t, int>::value &&
                std::is_same<offset_t, int>::value) {
    kernels::launch_bcsr4x4_mfma(stream, static_cast<int>(bcsr.rows), static_cast<int>(bcsr.num_block_rows),
                                 static_cast<int>(bcsr.num_blocks), bcsr.block_offsets.data().get(),
                                 bcsr.block_col_indices.data().get(), bcsr.values.data().get(), x.data().get(),
                                 y.data().get());
  } else if constexpr (std::is_same<index_t, int>::value && std::is_same<offset_t, int>::value &&
                       (std::is_same<type_t, float>::value || std::is_same<type_t, double>::value)) {
    // every other shape / fp64: the coalesced lane-group kernels (whole lines of consecutive blocks per slot of lanes,
    // block inner product on the VALU) -- the 2 x 2 example of the reference takes this path
    kernels::launch_bcsr_coalesced<static_cast<int>(R), static_cast<int>(C), type_t>(
        stream, static_cast<int>(bcsr.rows), static_cast<int>(bcsr.num_block_rows), static_cast<int>(bcsr.num_blocks),
        bcsr.block_offsets.data().get(), bcsr.block_col_indices.data().get(), bcsr.values.data().get(), x.data().get(),
        y.data().get());
  } else {
    using layout_t = layout::bcsr<index_t, offset_t>;
    using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, offset_t, std::size_t,
                                    std::size_t, layout_t>;
    setup_t config(layout_t(bcsr.block_offsets.data().get(), static_cast<index_t>(bcsr.num_block_rows),
                            static_cast<offset_t>(bcsr.num_blocks)));
    constexpr std::size_t block_size = 128;
    if (bcsr.num_block_rows > 0)
      launch::non_cooperative(stream, kernels::bcsr_thread_mapped_spmv<R, C, setup_t, index_t, type_t>,
                              dim3(static_cast<unsigned>(math::ceil_div(bcsr.num_block_rows, block_size))),
                              dim3(block_size), config, bcsr.rows, bcsr.block_col_indices.data().get(),
                              bcsr.values.data().get(), x.data().get(), y.data().get());
  }
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
