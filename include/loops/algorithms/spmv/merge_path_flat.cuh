/**
 * @file merge_path_flat.cuh
 * @brief `algorithms::spmv::merge_path_flat(csr, x, y, stream) -> util::timer_t`: the headline
 * load-balanced CSR SpMV.  Same signature and timing convention as the reference wrapper
 * (include/loops/algorithms/spmv/merge_path_flat.cuh:97-139: the coordinate pre-pass is built
 * first and NOT timed; the returned timer brackets the SpMV kernels only), but the work is done
 * by the fused CDNA4 kernel (loops/kernels/merge_path_spmv.hxx) -- no per-nonzero atomics, y
 * needs no zero-fill, deterministic summation order.
 *
 * `merge_path_flat(plan, csr, x, y, stream)` reuses a caller-held `preprocess_t` (the plan only
 * depends on csr.offsets), which is how an iterative solver should call it.
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
#include <loops/error.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
using merge_path_plan_t =
    schedule::merge_path::preprocess_t<merge_path_launch_t<type_t>::block_size,
                                       merge_path_launch_t<type_t>::items_per_thread, index_t,
                                       offset_t, std::size_t, std::size_t>;

/// The plan of an explicit tile shape, and the small-tile plan of the launch box (`launch_t`: 256 x 8 / 256 x 4): the shape
/// matrices whose rows are all short run best with -- one kernel, no carry-outs (`preprocess_t::classify`).
template <std::size_t TPB, std::size_t IPT, typename index_t, typename offset_t>
using merge_path_plan_of_t = schedule::merge_path::preprocess_t<TPB, IPT, index_t, offset_t, std::size_t, std::size_t>;
template <typename index_t, typename offset_t, typename type_t>
using merge_path_small_plan_t = merge_path_plan_of_t<launch_t<type_t>::block_size, launch_t<type_t>::items_per_thread, index_t, offset_t>;

/// SpMV with a prebuilt plan of tile shape TPB x IPT; asynchronous on `stream`.
template <std::size_t TPB, std::size_t IPT, typename index_t, typename offset_t, typename type_t>
void merge_path_flat_async_with(const merge_path_plan_of_t<TPB, IPT, index_t, offset_t>& plan,
                                csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                                xpu::stream_t stream = 0, bool planned = false) {
  error::throw_if_exception(static_cast<unsigned long long>(csr.rows) + static_cast<unsigned long long>(csr.nnzs) >= (1ull << 31) - 4096,
                            "merge_path_flat: rows + nnz must stay below 2^31 (the merge-path search arithmetic is int, as in util/search.hxx:46-47)");
  kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                static_cast<int>(plan.merge_tiles()), plan.self_complete(), plan.head_starts()};
  kernels::launch_merge_path_fused<static_cast<int>(TPB), static_cast<int>(IPT), (IPT % 2 == 0), false>(
      stream, view, static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), csr.offsets.data().get(),
      csr.indices.data().get(), csr.values.data().get(), x.data().get(), y.data().get(), 3, planned);
}

/// The same product through the PHASED-gather kernel (kernels::merge_path_spmv_fused_phased: a tile's x gathers in 8 passes by
/// column range, clock-aligned across workgroups -- for columns scattered over an x of about one XCD's L2; same bits as the
/// plain kernel; shapes 512 x 8 and 256 x 16 only).  Pick it by measurement (spmv_plan_t does).  No reference counterpart.
template <std::size_t TPB, std::size_t IPT, typename index_t, typename offset_t, typename type_t>
void merge_path_flat_phased_async_with(const merge_path_plan_of_t<TPB, IPT, index_t, offset_t>& plan,
                                       csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                                       xpu::stream_t stream = 0, bool planned = false) {
  static_assert((TPB == 512 && IPT == 8) || (TPB == 256 && IPT == 16), "phased gathers: 512 x 8 or 256 x 16 merge tiles");
  error::throw_if_exception(static_cast<unsigned long long>(csr.rows) + static_cast<unsigned long long>(csr.nnzs) >= (1ull << 31) - 4096,
                            "merge_path_flat: rows + nnz must stay below 2^31 (the merge-path search arithmetic is int, as in util/search.hxx:46-47)");
  kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                static_cast<int>(plan.merge_tiles()), plan.self_complete(), plan.head_starts()};
  kernels::launch_merge_path_fused_phased<static_cast<int>(TPB), static_cast<int>(IPT)>(
      stream, view, static_cast<int>(csr.rows), static_cast<int>(csr.cols), static_cast<int>(csr.nnzs), csr.offsets.data().get(),
      csr.indices.data().get(), csr.values.data().get(), x.data().get(), y.data().get(), 3, planned);
}

/// SpMV with a prebuilt plan (the merge_path launch box: 512 x 8 / 512 x 4); asynchronous on `stream`.
template <typename index_t, typename offset_t, typename type_t>
void merge_path_flat_async(const merge_path_plan_t<index_t, offset_t, type_t>& plan,
                           csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                           xpu::stream_t stream = 0) {
  merge_path_flat_async_with<merge_path_launch_t<type_t>::block_size, merge_path_launch_t<type_t>::items_per_thread>(plan, csr, x, y,
                                                                                                                      stream);
}

/// The same product for one rank of a row-range sharded multi-GPU SpMV: every finished row of y is ALSO stored to the same
/// element of `peers.count` peer-mapped vectors (loops/kernels/merge_path_spmv.hxx `fanout_store`; `peers.base[p]` = where
/// this shard's y[0] lives in peer p's full-length vector) -- the allgatherv(y) issued from the kernels' epilogue
/// (no reference counterpart).  Asynchronous on `stream`; one barrier among the ranks remains per step.
template <typename index_t, typename offset_t, typename type_t>
void merge_path_flat_fanout_async(const merge_path_plan_t<index_t, offset_t, type_t>& plan,
                                  csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                                  const kernels::peer_fanout<type_t>& peers, xpu::stream_t stream = 0) {
  error::throw_if_exception(static_cast<unsigned long long>(csr.rows) + static_cast<unsigned long long>(csr.nnzs) >= (1ull << 31) - 4096,
                            "merge_path_flat_fanout: rows + nnz must stay below 2^31");
  error::throw_if_exception(peers.count < 0 || peers.count > kernels::max_peers, "merge_path_flat_fanout: at most 7 peers");
  constexpr int block_size = merge_path_launch_t<type_t>::block_size;
  constexpr int items_per_thread = merge_path_launch_t<type_t>::items_per_thread;
  kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                static_cast<int>(plan.merge_tiles())};
  kernels::launch_merge_path_fused_fanout<block_size, items_per_thread>(
      stream, view, static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), csr.offsets.data().get(), csr.indices.data().get(),
      csr.values.data().get(), x.data().get(), y.data().get(), peers);
}

template <typename index_t, typename offset_t, typename type_t>
util::timer_t merge_path_flat(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y,
                              xpu::stream_t stream = 0) {
  error::throw_if_exception(static_cast<unsigned long long>(csr.rows) + static_cast<unsigned long long>(csr.nnzs) >= (1ull << 31) - 4096,
                            "merge_path_flat: rows + nnz must stay below 2^31 (the merge-path search arithmetic is int, as in util/search.hxx:46-47)");
  using plan_t = merge_path_plan_t<index_t, offset_t, type_t>;
  using small_t = merge_path_small_plan_t<index_t, offset_t, type_t>;
  // The tile shape follows the structure (part of the untimed plan set-up, like the reference's pre-pass): if no merge tile of
  // the small shape starts more than a workgroup's worth of nonzeros inside a row, the product runs as ONE kernel over
  // 256 x 8 tiles -- band / FEM / short-row matrices (25 us against 29-35 with larger tiles on the 2^20 x 16 band matrix);
  // otherwise (rows longer than a tile) 512 x 8 tiles + the carry-out fix-up (C2: 94.4 against 95.9 us).
  const typename small_t::layout_t lay(csr.offsets.data().get(), static_cast<index_t>(csr.rows), static_cast<offset_t>(csr.nnzs));
  util::timer_t timer(stream);
  // Columns scattered over an x of 3 MB or more (a structural guess from 16 384 sampled pairs of nonzeros, part of
  // the untimed set-up: kernels::columns_look_scattered): 512 x 8 tiles with PHASED x gathers (C2 94.8 -> 82.9
  // us, same bits; fp64 188 -> 141) -- self-completing or not.
  if (kernels::columns_worth_sampling(static_cast<long long>(csr.nnzs), static_cast<long long>(csr.cols), static_cast<int>(sizeof(type_t)))) {
    vector_t<unsigned int> scratch(kernels::scatter_scratch_words);
    if (kernels::columns_look_scattered(stream, csr.indices.data().get(), static_cast<long long>(csr.nnzs), static_cast<long long>(csr.cols),
                                        static_cast<int>(sizeof(type_t)), scratch.data().get())) {
      // (16 or 32 parts -- x beyond 6 MB, 12 MB for 8-byte values: 256 x 16 tiles, twice the gathers per lane and pass; 3-11 % faster)
      if (kernels::phased_config_for(static_cast<long long>(csr.cols), static_cast<int>(sizeof(type_t))).parts >= 16) {
        using tall_t = merge_path_plan_of_t<256, 16, index_t, offset_t>;
        tall_t tall(lay, stream, tall_t::prepass_always);
        tall.classify(stream);
        timer.start();
        merge_path_flat_phased_async_with<256, 16>(tall, csr, x, y, stream);
        (void)xpu::stream_synchronize(stream);
        timer.stop();
        return timer;
      }
      using wide_t = merge_path_plan_of_t<512, 8, index_t, offset_t>;
      wide_t wide(lay, stream, wide_t::prepass_always);
      wide.classify(stream);
      timer.start();
      merge_path_flat_phased_async_with<512, 8>(wide, csr, x, y, stream);
      (void)xpu::stream_synchronize(stream);
      timer.stop();
      return timer;
    }
  }
  {
    small_t small(lay, stream, small_t::prepass_always);
    if (small.classify(stream) || small.merge_tiles() <= 1) {
      timer.start();
      merge_path_flat_async_with<launch_t<type_t>::block_size, launch_t<type_t>::items_per_thread>(small, csr, x, y, stream);
      (void)xpu::stream_synchronize(stream);
      timer.stop();
      return timer;
    }
  }
  // Coordinates for every merge tile (the fused kernel always consumes the table).
  plan_t plan(lay, stream, plan_t::prepass_always);
  plan.classify(stream);
  timer.start();
  merge_path_flat_async(plan, csr, x, y, stream);
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
