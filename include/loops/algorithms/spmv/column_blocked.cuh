/**
 * @file column_blocked.cuh
 * @brief `algorithms::spmv::column_blocked_t<index_t, offset_t, type_t>`: a CSR held column-blocked
 * ("stacked", loops/kernels/column_blocked.hxx) for matrices / multi-GPU shards whose x does not fit
 * the 4 MB per-XCD L2, and `spmv(x, y)` over it -- the fused merge_path_flat kernel on the stacked CSR
 * plus a K-way row reduce.  No reference counterpart (the reference leaves the x gather to the cache).
 *
 *   algorithms::spmv::column_blocked_t<int, int, float> A(csr);          // K picked from csr.cols
 *   A.spmv(x, y);                                                       // y = csr * x
 */
#pragma once

#include <vector>

#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/kernels/column_blocked.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
struct column_blocked_t {
  using plan_t = merge_path_plan_t<index_t, offset_t, type_t>;

  std::size_t rows, cols, nnzs;
  int num_blocks;
  std::vector<int> bounds;            ///< num_blocks + 1 column boundaries
  vector_t<offset_t> offsets;         ///< num_blocks * rows + 1 stacked offsets
  vector_t<index_t> indices;          ///< stacked column indices (global ids)
  vector_t<type_t> values;            ///< stacked values
  vector_t<int> perm;                 ///< stacked position -> original position
  vector_t<type_t> partial;           ///< num_blocks * rows partial results
  plan_t plan;

  /// Blocks of x of about 2 MB, at most half the mean row length (each block adds `rows` row-end items
  /// to the merge path) and at most 64.
  static int automatic_blocks(std::size_t cols, std::size_t rows, std::size_t nnzs) {
    int k = 1;
    while (k < 64 && cols * sizeof(type_t) / k > (std::size_t(2) << 20)) k *= 2;
    const std::size_t mean = rows ? nnzs / rows : 0;
    int cap = 2;
    while (cap < 64 && std::size_t(cap) * 2 <= mean / 2) cap *= 2;
    return k < cap ? k : cap;
  }

  /// @param blocks 0 = automatic; @param block_bounds optional explicit boundaries (blocks + 1 values).
  explicit column_blocked_t(csr_t<index_t, offset_t, type_t>& csr, int blocks = 0, const int* block_bounds = nullptr,
                            xpu::stream_t stream = 0)
      : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs),
        num_blocks(blocks > 0 ? blocks : automatic_blocks(csr.cols, csr.rows, csr.nnzs)),
        bounds(num_blocks + 1), offsets(std::size_t(num_blocks) * csr.rows + 1), indices(csr.nnzs), values(csr.nnzs),
        perm(csr.nnzs), partial(std::size_t(num_blocks) * csr.rows),
        plan(build(checked(csr), block_bounds, stream), stream, plan_t::prepass_always) {
    plan.classify(stream);  // short stacked rows only -> one SpMV kernel, no carry-outs
  }

  /// y = A x; asynchronous on `stream`.
  void spmv_async(vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    constexpr int block_size = merge_path_launch_t<type_t>::block_size;
    constexpr int items_per_thread = merge_path_launch_t<type_t>::items_per_thread;
    kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                  static_cast<int>(plan.merge_tiles()), plan.self_complete(), plan.head_starts()};
    kernels::launch_merge_path_fused<block_size, items_per_thread, (items_per_thread % 2 == 0), false>(
        stream, view, static_cast<int>(num_blocks * rows), static_cast<int>(nnzs), offsets.data().get(),
        indices.data().get(), values.data().get(), x.data().get(), partial.data().get(), 3, true);
    kernels::launch_reduce_blocks<type_t>(stream, partial.data().get(), static_cast<int>(rows), num_blocks,
                                          y.data().get());
  }

  /// y = A x for one rank of a row-range sharded multi-GPU SpMV: the block reduce also stores the finished y to the
  /// peer-mapped vectors `peers` (kernels::reduce_blocks_x4_fanout; see merge_path_flat_fanout_async).  Asynchronous.
  void spmv_fanout_async(vector_t<type_t>& x, vector_t<type_t>& y, const kernels::peer_fanout<type_t>& peers,
                         xpu::stream_t stream = 0) {
    error::throw_if_exception(peers.count < 0 || peers.count > kernels::max_peers,
                              "column_blocked_t::spmv_fanout_async: peers.count must be 0 .. 7");
    constexpr int block_size = merge_path_launch_t<type_t>::block_size;
    constexpr int items_per_thread = merge_path_launch_t<type_t>::items_per_thread;
    kernels::merge_plan_view view{plan.data(), plan.carry_rows(), plan.template carry_values<type_t>(),
                                  static_cast<int>(plan.merge_tiles()), plan.self_complete(), plan.head_starts()};
    kernels::launch_merge_path_fused<block_size, items_per_thread, (items_per_thread % 2 == 0), false>(
        stream, view, static_cast<int>(num_blocks * rows), static_cast<int>(nnzs), offsets.data().get(),
        indices.data().get(), values.data().get(), x.data().get(), partial.data().get(), 3, true);
    kernels::launch_reduce_blocks_fanout<type_t>(stream, partial.data().get(), static_cast<int>(rows), num_blocks,
                                                 y.data().get(), peers);
  }

  util::timer_t spmv(vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    util::timer_t timer(stream);
    timer.start();
    spmv_async(x, y, stream);
    (void)xpu::stream_synchronize(stream);
    timer.stop();
    return timer;
  }

 private:
  /// The stacked CSR has num_blocks * rows rows: ITS rows + nnz must fit the int merge-path arithmetic.
  csr_t<index_t, offset_t, type_t>& checked(csr_t<index_t, offset_t, type_t>& csr) {
    error::throw_if_exception(static_cast<unsigned long long>(num_blocks) * csr.rows + csr.nnzs >= (1ull << 31) - 4096,
                              "column_blocked_t: num_blocks * rows + nnz must stay below 2^31");
    return csr;
  }
  typename plan_t::layout_t build(csr_t<index_t, offset_t, type_t>& csr, const int* block_bounds, xpu::stream_t stream) {
    for (int k = 0; k <= num_blocks; ++k)
      bounds[k] = block_bounds ? block_bounds[k] : static_cast<int>(static_cast<long long>(cols) * k / num_blocks);
    vector_t<int> bounds_dev(bounds.begin(), bounds.end());
    const std::size_t temp_bytes = kernels::column_blocked_temp_bytes(static_cast<int>(nnzs), static_cast<int>(num_blocks * rows));
    vector_t<char> temp(temp_bytes);
    kernels::column_blocked_view<index_t, offset_t, type_t> view{static_cast<int>(rows), static_cast<int>(cols),
                                                                 static_cast<int>(nnzs), num_blocks, offsets.data().get(),
                                                                 indices.data().get(), values.data().get(), perm.data().get()};
    kernels::build_column_blocked(stream, csr.offsets.data().get(), csr.indices.data().get(), csr.values.data().get(),
                                  bounds_dev.data().get(), view, temp.data().get(), temp_bytes);
    (void)xpu::stream_synchronize(stream);  // temporaries die with this scope
    return typename plan_t::layout_t(offsets.data().get(), static_cast<index_t>(num_blocks * rows),
                                     static_cast<offset_t>(nnzs));
  }
};

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
