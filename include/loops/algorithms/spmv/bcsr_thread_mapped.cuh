/**
 * @file bcsr_thread_mapped.cuh
 * @brief `algorithms::spmv::bcsr_thread_mapped<R, C>(bcsr, x, y, stream) -> util::timer_t`
 * (reference include/loops/algorithms/spmv/bcsr_thread_mapped.cuh:36-123).  4 x 4 fp32 blocks
 * take the MFMA path (four chained v_mfma_f32_4x4x1_16b_f32 per block, 16 block-rows per
 * wavefront -- or, where the block-row lengths are skewed, the same product on merge-path tiles: kernels/bcsr_merge_path.hxx);
 * every other shape and fp64 run the coalesced lane-group kernels of kernels/bcsr_spmv.hxx
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
#include <loops/kernels/bcsr_merge_path.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <std::size_t R, std::size_t C, typename index_t, typename offset_t, typename type_t>
util::timer_t bcsr_thread_mapped(bcsr_t<R, C, index_t, offset_t, type_t>& bcsr, vector_t<type_t>& x,
                                 vector_t<type_t>& y, xpu::stream_t stream = 0) {
  constexpr bool mfma_shape = R == 4 && C == 4 && std::is_same<type_t, float>::value && std::is_same<index_t, int>::value &&
                              std::is_same<offset_t, int>::value;
  // 4 x 4 fp32.  Even block-row lengths: the thread_mapped MFMA kernel; skewed ones (a block-row is a serial chain there): the same
  // product on the merge-path schedule (kernels::bcsr_row_length_class).  Looked at once per matrix object -- a probe over the
  // offsets + one 4-byte copy on the first call, remembered in bcsr.row_length_class -- and outside the timed region, like the
  // merge-path scratch of this call.
  [[maybe_unused]] bool on_merge_path = false;
  vector_t<unsigned char> merge_scratch;
  if constexpr (mfma_shape) {
    const int nbr = static_cast<int>(bcsr.num_block_rows), nb = static_cast<int>(bcsr.num_blocks);
    const bool merge_capable = static_cast<long long>(nbr) + nb < (1ll << 31) - 4096;
    if (bcsr.row_length_class == 0 && nbr > 0 && nb > 0 && merge_capable) {
      vector_t<unsigned int> words(8, 0u);  // [0] the class, [2..5] the probe's counters (zero-filled)
      kernels::launch_bcsr_skew_probe(stream, nbr, nb, bcsr.block_offsets.data().get(),
                                      reinterpret_cast<kernels::bcsr_skew_ctl*>(words.data().get() + 2), words.data().get());
      unsigned int found = 0;
      if (hipMemcpyAsync(&found, words.data().get(), sizeof(found), hipMemcpyDeviceToHost, stream) == hipSuccess &&
          hipStreamSynchronize(stream) == hipSuccess)
        bcsr.row_length_class = static_cast<int>(found);
    }
    if (bcsr.row_length_class == kernels::bcsr_rows_skewed && merge_capable) {
      merge_scratch = vector_t<unsigned char>(kernels::bcsr_merge_scratch_bytes(nbr, nb));
      on_merge_path = true;
    }
  }
  util::timer_t timer(stream);
  timer.start();
  if constexpr (mfma_shape) {
    const int nbr = static_cast<int>(bcsr.num_block_rows), nb = static_cast<int>(bcsr.num_blocks);
    if (on_merge_path)
      kernels::launch_bcsr4x4_merge_path(stream, static_cast<int>(bcsr.rows), nbr, nb, bcsr.block_offsets.data().get(),
                                         bcsr.block_col_indices.data().get(), bcsr.values.data().get(), x.data().get(), y.data().get(),
                                         merge_scratch.data().get());
    else
      kernels::launch_bcsr4x4_mfma(stream, static_cast<int>(bcsr.rows), nbr, nb, bcsr.block_offsets.data().get(),
                                   bcsr.block_col_indices.data().get(), bcsr.values.data().get(), x.data().get(), y.data().get());
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
