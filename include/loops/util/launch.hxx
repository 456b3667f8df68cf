/**
 * @file launch.hxx
 * @brief Kernel launch helpers: `launch::non_cooperative(stream, kernel, grid, block, args...)`
 * and `launch::cooperative(...)` (reference: include/loops/util/launch.hxx:33-75).  No dynamic
 * LDS is used anywhere in the library: schedules declare their LDS statically via
 * `setup::storage_t`.
 */
#pragma once

#include <cstddef>
#include <utility>

#include <loops/backend/xpu.hxx>

namespace loops {
namespace launch {

template <typename func_t, typename... args_t>
void non_cooperative(xpu::stream_t stream, const func_t& kernel, dim3 number_of_blocks, dim3 threads_per_block,
                     args_t&&... args) {
  hipLaunchKernelGGL(kernel, number_of_blocks, threads_per_block, 0, stream, std::forward<args_t>(args)...);
}

/// Grid-synchronising launch; the grid must be fully resident (see launch_box::occupancy_grid).
template <typename func_t, typename... args_t>
void cooperative(xpu::stream_t stream, const func_t& kernel, std::size_t number_of_blocks,
                 std::size_t threads_per_block, args_t&&... args) {
  void* argument_ptrs[sizeof...(args_t) == 0 ? 1 : sizeof...(args_t)] = {
      const_cast<void*>(static_cast<const void*>(&args))...};
  (void)xpu::launch_cooperative_kernel<func_t>(&kernel, dim3(number_of_blocks), dim3(threads_per_block), argument_ptrs, 0,
                                         stream);
}

}  // namespace launch
}  // namespace loops
