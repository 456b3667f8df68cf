/**
 * @file ell_merge_path.cuh
 * @brief `algorithms::spmv::ell_merge_path(ell, x, y, stream) -> util::timer_t`: the merge-path
 * schedule driven by a NON-CSR layout (`layout::ell`: tile ends are (row + 1) * pitch, produced
 * by a functor) -- the layout-generic proof of the schedule
 * (reference include/loops/algorithms/spmv/ell_merge_path.cuh:32-125).
 *
 * The wrapper runs the FUSED merge-tile engine (loops/kernels/merge_path_spmv.hxx) over the ELL cells: the
 * schedule's own coordinate pre-pass over layout::ell, the row ends from `kernels::ell_row_end` instead of an
 * offsets array, padding cells (negative column) contributing 0 -- no atomics, y need not be zero-filled,
 * deterministic order.  `__ell_merge_path` below is the reference-shaped kernel written against the public
 * schedule API (one atomicAdd per cell, y zero-filled by the caller); `ell_merge_path_atomic` launches it.
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
#include <loops/error.hxx>
#include <loops/kernels/launch.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <std::size_t threads_per_block, std::size_t items_per_thread, typename meta_t, typename setup_t,
          typename layout_t, typename index_t, typename type_t>
__global__ void __launch_bounds__(int(threads_per_block))
__ell_merge_path(meta_t meta, layout_t lay, const index_t* indices, const type_t* values, const type_t* x, type_t* y) {
  __shared__ typename setup_t::storage_t temporary_storage;
  setup_t config(meta, temporary_storage, lay);
  auto map = config.init();
  if (!config.is_valid_accessor(map)) return;
#pragma unroll
  for (auto item : config.virtual_idx()) {
    auto nz = config.atom_idx(item, map);
    auto row = config.tile_idx(map);
    if (config.atoms_counting_it[map.y] < temporary_storage.tile_end_offset[map.x]) {
      const index_t col = indices[nz];
      if (col >= 0) atomicAdd(&(y[row]), values[nz] * x[col]);
      map.y++;
    } else {
      map.x++;
    }
  }
}

/// Reference-shaped: the schedule-API kernel above, one atomicAdd per cell; y must be zero-filled.
template <typename index_t, typename type_t>
util::timer_t ell_merge_path_atomic(ell_t<index_t, type_t>& ell, vector_t<type_t>& x, vector_t<type_t>& y,
                                    xpu::stream_t stream = 0) {
  using layout_t = layout::ell<index_t, index_t>;
  constexpr std::size_t block_size = launch_t<type_t>::block_size;
  constexpr std::size_t items_per_thread = launch_t<type_t>::items_per_thread;
  using meta_t = schedule::merge_path::preprocess_t<block_size, items_per_thread, index_t, index_t, std::size_t,
                                                    std::size_t, layout_t>;
  using setup_t = schedule::setup<schedule::algorithms_t::merge_path_flat, block_size, items_per_thread, index_t,
                                  index_t, std::size_t, std::size_t, layout_t>;
  layout_t lay(static_cast<index_t>(ell.rows), static_cast<index_t>(ell.pitch));
  meta_t meta(lay, stream);
  util::timer_t timer(stream);
  timer.start();
  if (meta.merge_tiles() > 0)
    launch::non_cooperative(stream,
                            __ell_merge_path<block_size, items_per_thread, meta_t, setup_t, layout_t, index_t, type_t>,
                            dim3(static_cast<unsigned>(meta.merge_tiles())), dim3(block_size), meta, lay,
                            ell.indices.data().get(), ell.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

template <typename index_t, typename type_t>
util::timer_t ell_merge_path(ell_t<index_t, type_t>& ell, vector_t<type_t>& x, vector_t<type_t>& y,
                             xpu::stream_t stream = 0) {
  using layout_t = layout::ell<index_t, index_t>;
  constexpr int block_size = merge_path_launch_t<type_t>::block_size;
  constexpr int items_per_thread = merge_path_launch_t<type_t>::items_per_thread;
  using meta_t = schedule::merge_path::preprocess_t<block_size, items_per_thread, index_t, index_t, std::size_t,
                                                    std::size_t, layout_t>;
  error::throw_if_exception(ell.rows * ell.pitch + ell.rows >= (std::size_t(1) << 31) - 4096,
                            "ell_merge_path: rows * pitch + rows must stay below 2^31 (int merge-path arithmetic)");
  layout_t lay(static_cast<index_t>(ell.rows), static_cast<index_t>(ell.pitch));
  meta_t meta(lay, stream, meta_t::prepass_always);  // coordinates of every merge tile (untimed, like the reference's pre-pass)
  util::timer_t timer(stream);
  timer.start();
  kernels::merge_plan_view view{meta.data(), meta.carry_rows(), meta.template carry_values<type_t>(),
                                static_cast<int>(meta.merge_tiles())};
  kernels::launch_ell_merge_path_fused<block_size, items_per_thread>(stream, view, static_cast<int>(ell.rows),
                                                                     static_cast<int>(ell.pitch), ell.indices.data().get(),
                                                                     ell.values.data().get(), x.data().get(), y.data().get());
  (void)xpu::stream_synchronize(stream);
  timer.stop();
  return timer;
}

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
