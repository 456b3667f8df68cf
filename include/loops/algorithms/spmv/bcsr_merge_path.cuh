/**
 * @file bcsr_merge_path.cuh
 * @brief `algorithms::spmv::bcsr_merge_path(bcsr, x, y, stream) -> util::timer_t`: the product of
 * `bcsr_thread_mapped<4, 4>` (reference algorithms/spmv/bcsr_thread_mapped.cuh:36-123) on the merge-path schedule -- equal tiles
 * of (block-row ends, blocks), so that a block-row of any length is shared among workgroups (kernels/bcsr_merge_path.hxx).  For
 * BCSR matrices whose block-row lengths are skewed (64 block-rows of 16 384 blocks among 2^17 short ones: 39 us against 1.8 ms
 * from the thread_mapped kernels); where the lengths are uniform `bcsr_thread_mapped` is ~20 % faster.  4 x 4 fp32 blocks, int
 * indices; x padded to 4 * num_block_cols; y needs no zero-fill.  No reference counterpart.
 */
#pragma once

#include <loops/container/bcsr.hxx>
#include <loops/container/vector.hxx>
#include <loops/kernels/bcsr_merge_path.hxx>
#include <loops/util/timer.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

inline util::timer_t bcsr_merge_path(bcsr_t<4, 4, int, int, float>& bcsr, vector_t<float>& x, vector_t<float>& y, xpu::stream_t stream = 0) {
  // coordinates + carry-outs of this call (the reference's wrappers allocate their plan per call too: merge_path_flat.cuh:111-114)
  vector_t<unsigned char> scratch(kernels::bcsr_merge_scratch_bytes(static_cast<int>(bcsr.num_block_rows), static_cast<int>(bcsr.num_blocks)));
  util::timer_t timer(stream);
  timer.start();
  kernels::launch_bcsr4x4_merge_path(stream, static_cast<int>(bcsr.rows), static_cast<int>(bcsr.num_block_rows), static_cast<int>(bcsr.num_blocks),
                                     bcsr.block_offsets.data().get(), bcsr.block_col_indices.data().get(), bcsr.values.data().get(), x.data().get(),
                                     y.data().get(), scratch.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
