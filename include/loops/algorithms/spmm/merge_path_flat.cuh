/**
 * @file merge_path_flat.cuh
 * @brief `algorithms::spmm::merge_path_flat(csr, B, C, stream) -> util::timer_t`: C = A * B with A
 * in CSR and dense row-major B, C on the merge-path schedule (loops/kernels/merge_path_spmm.hxx).
 * The reference ships SpMM only as the per-thread loop `spmm::thread_mapped`
 * (include/loops/algorithms/spmm/thread_mapped.cuh:68-94); this is the load-balanced, coalesced
 * counterpart with the calling convention of `spmv::merge_path_flat` (plan built first, not timed;
 * the returned timer brackets the SpMM kernels).  C needs no zero-fill.
 */
#pragma once

#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/container/matrix.cuh>

namespace loops {
namespace algorithms {
namespace spmm {

/// Plan of the SpMM kernel: merge tiles of spmv::launch_t (256 x 8 / 256 x 4) -- the SpMM stages the tile's
/// row ends and {B-row offset, value} pairs in LDS, so it keeps the smaller tile.
template <typename index_t, typename offset_t, typename type_t>
using merge_path_plan_t =
    schedule::merge_path::preprocess_t<spmv::launch_t<type_t>::block_size, spmv::launch_t<type_t>::items_per_thread,
                                       index_t, offset_t, std::size_t, std::size_t>;

/// SpMM with a prebuilt plan and caller-held carry-out scratch (plan.merge_tiles() * B.cols values);
/// asynchronous on `stream`.
template <typename index_t, typename offset_t, typename type_t>
void merge_path_flat_async(const merge_path_plan_t<index_t, offset_t, type_t>& plan,
                           csr_t<index_t, offset_t, type_t>& csr, matrix_t<type_t>& B, matrix_t<type_t>& C,
                           vector_t<type_t>& carry, xpu::stream_t stream = 0) {
  constexpr int block_size = spmv::launch_t<type_t>::block_size;
  constexpr int items_per_thread = spmv::launch_t<type_t>::items_per_thread;
  kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                static_cast<int>(plan.merge_tiles())};
  kernels::launch_merge_path_spmm<block_size, items_per_thread>(
      stream, view, carry.data().get(), static_cast<int>(csr.rows), static_cast<int>(csr.cols),
      static_cast<int>(csr.nnzs), csr.offsets.data().get(), csr.indices.data().get(), csr.values.data().get(),
      static_cast<const type_t*>(B.m_data_ptr), static_cast<int>(B.cols), B.cols, C.m_data_ptr, C.cols);
}

template <typename index_t, typename offset_t, typename type_t>
util::timer_t merge_path_flat(csr_t<index_t, offset_t, type_t>& csr, matrix_t<type_t>& B, matrix_t<type_t>& C,
                              xpu::stream_t stream = 0) {
  using plan_t = merge_path_plan_t<index_t, offset_t, type_t>;
  plan_t plan(typename plan_t::layout_t(csr.offsets.data().get(), static_cast<index_t>(csr.rows),
                                        static_cast<offset_t>(csr.nnzs)),
              stream, plan_t::prepass_always);
  vector_t<type_t> carry(plan.merge_tiles() * B.cols + 1);
  util::timer_t timer(stream);
  timer.start();
  merge_path_flat_async(plan, csr, B, C, carry, stream);
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

}  // namespace spmm
}  // namespace algorithms
}  // namespace loops
