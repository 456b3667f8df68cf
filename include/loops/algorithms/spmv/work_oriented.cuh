/**
 * @file work_oriented.cuh
 * @brief `algorithms::spmv::work_oriented(csr, x, y, stream)`: even share of (rows + nonzeros)
 * per thread of an occupancy-sized grid (reference include/loops/algorithms/spmv/work_oriented.cuh:33-121).
 * Runs the fused persistent kernel (loops/kernels/merge_path_spmv.hxx: work_oriented_spmv_fused): each
 * workgroup walks an even contiguous share of 1-4 merge tiles with the open row carried in registers
 * (one 512 x 8 tile per workgroup with phased x gathers -- merge_path_flat's launch -- where the columns look scattered over a large x).  No atomics;
 * y does not have to be zero-filled.
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
#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/error.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
using work_oriented_plan_t = schedule::merge_path::preprocess_t<launch_t<type_t>::block_size, launch_t<type_t>::items_per_thread,
                                                                index_t, offset_t, std::size_t, std::size_t>;

/// work_oriented over a caller-held plan (built once per sparsity structure with `prepass_always`): the persistent kernel
/// + its fix-up, no coordinate pre-pass and no synchronisation per call.
template <typename index_t, typename offset_t, typename type_t>
void work_oriented_async(const work_oriented_plan_t<index_t, offset_t, type_t>& plan, csr_t<index_t, offset_t, type_t>& csr,
                         vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
  error::throw_if_exception(static_cast<unsigned long long>(csr.rows) + static_cast<unsigned long long>(csr.nnzs) >= (1ull << 31) - 4096,
                            "work_oriented: rows + nnz must stay below 2^31");
  constexpr int block_size = launch_t<type_t>::block_size;
  constexpr int items_per_thread = launch_t<type_t>::items_per_thread;
  kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                static_cast<int>(plan.merge_tiles())};
  kernels::launch_work_oriented_fused<block_size, items_per_thread, (items_per_thread % 2 == 0)>(
      stream, view, static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), csr.offsets.data().get(),
      csr.indices.data().get(), csr.values.data().get(), x.data().get(), y.data().get());
}

template <typename index_t, typename offset_t, typename type_t>
void work_oriented(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                   xpu::stream_t stream = 0) {
  error::throw_if_exception(static_cast<unsigned long long>(csr.rows) + static_cast<unsigned long long>(csr.nnzs) >= (1ull << 31) - 4096,
                            "work_oriented: rows + nnz must stay below 2^31 (the merge-path search arithmetic is int, as in util/search.hxx:46-47)");
  constexpr int block_size = launch_t<type_t>::block_size;
  constexpr int items_per_thread = launch_t<type_t>::items_per_thread;
  using plan_t = schedule::merge_path::preprocess_t<block_size, items_per_thread, index_t, offset_t, std::size_t,
                                                    std::size_t>;
  // Columns scattered over an x of 3 MB or more (kernels::columns_look_scattered, the guess `merge_path_flat` goes by): the share is
  // one 512 x 8 tile per workgroup and the launch is merge_path_flat's phased kernel (a persistent form with phased gathers was
  // measured 10 % behind it: 74 VGPRs, three resident workgroups; profiles/r05_work_oriented_shares_experiment.txt).
  if (kernels::columns_worth_sampling(static_cast<long long>(csr.nnzs), static_cast<long long>(csr.cols), static_cast<int>(sizeof(type_t)))) {
    vector_t<unsigned int> scratch(kernels::scatter_scratch_words);
    if (kernels::columns_look_scattered(stream, csr.indices.data().get(), static_cast<long long>(csr.nnzs), static_cast<long long>(csr.cols),
                                        static_cast<int>(sizeof(type_t)), scratch.data().get())) {
      if (kernels::phased_config_for(static_cast<long long>(csr.cols), static_cast<int>(sizeof(type_t))).parts >= 16) {  // (as merge_path_flat)
        using tall_t = merge_path_plan_of_t<256, 16, index_t, offset_t>;
        tall_t tall(typename tall_t::layout_t(csr.offsets.data().get(), static_cast<index_t>(csr.rows), static_cast<offset_t>(csr.nnzs)),
                    stream, tall_t::prepass_always);
        tall.classify(stream);
        merge_path_flat_phased_async_with<256, 16>(tall, csr, x, y, stream);
        (void)xpu::stream_synchronize(stream);
        return;
      }
      using wide_t = merge_path_plan_of_t<512, 8, index_t, offset_t>;
      wide_t wide(typename wide_t::layout_t(csr.offsets.data().get(), static_cast<index_t>(csr.rows), static_cast<offset_t>(csr.nnzs)),
                  stream, wide_t::prepass_always);
      wide.classify(stream);
      merge_path_flat_phased_async_with<512, 8>(wide, csr, x, y, stream);
      (void)xpu::stream_synchronize(stream);
      return;
    }
  }
  plan_t plan(typename plan_t::layout_t(csr.offsets.data().get(), static_cast<index_t>(csr.rows),
                                        static_cast<offset_t>(csr.nnzs)),
              stream, plan_t::prepass_always);
  kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                static_cast<int>(plan.merge_tiles())};
  kernels::launch_work_oriented_fused<block_size, items_per_thread, (items_per_thread % 2 == 0)>(
      stream, view, static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), csr.offsets.data().get(),
      csr.indices.data().get(), csr.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
