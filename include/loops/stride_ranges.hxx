/**
 * @file stride_ranges.hxx
 * @brief Device-side strided ranges used inside user kernels and the schedules:
 * `grid_stride_range`, `block_stride_range`, `custom_stride_range`
 * (reference: include/loops/stride_ranges.hxx:16-62).
 */
#pragma once

#include <loops/range.hxx>

namespace loops {

template <typename T>
using step_range_t = typename range_proxy<T>::step_range_proxy;

/// begin + global thread rank, advancing by the total number of launched threads.
template <typename T>
__device__ __forceinline__ step_range_t<T> grid_stride_range(T begin, T end) {
  return step_range_t<T>(begin + T(blockDim.x * blockIdx.x + threadIdx.x), end, T(gridDim.x * blockDim.x));
}

/// [begin, end) advancing by the workgroup size (the caller adds its own lane offset).
template <typename T>
__device__ __forceinline__ step_range_t<T> block_stride_range(T begin, T end) {
  return step_range_t<T>(begin, end, T(blockDim.x));
}

template <typename T>
__device__ __forceinline__ step_range_t<T> custom_stride_range(T begin, T end, T stride) {
  return step_range_t<T>(begin, end, stride);
}

}  // namespace loops
