/**
 * @file wave.hxx
 * @brief 64-lane wavefront primitives for CDNA (gfx950): lane id, inclusive / exclusive
 * prefix sums and the segmented (head-flagged) prefix sum the merge-path SpMV uses to
 * combine per-thread partial rows without atomics.  All of them are log2(64) = 6 cross-lane
 * steps on registers; none touches LDS or needs a barrier.  Every lane of the wavefront must
 * execute the call (cross-lane reads of inactive lanes are undefined).
 */
#pragma once

#include <hip/hip_runtime.h>

namespace loops {
namespace wave {

constexpr int size = 64;

__device__ __forceinline__ int lane() { return static_cast<int>(__lane_id()); }

/// Inclusive prefix sum over sub-groups of `width` lanes (power of two <= 64).
template <int width = size, typename T>
__device__ __forceinline__ T inclusive_sum(T v) {
  const int l = lane() & (width - 1);
#pragma unroll
  for (int d = 1; d < width; d <<= 1) {
    T up = __shfl_up(v, d, width);
    if (l >= d) v += up;
  }
  return v;
}

template <int width = size, typename T>
__device__ __forceinline__ T exclusive_sum(T v) {
  return inclusive_sum<width>(v) - v;
}

/// Sum over the whole sub-group, returned in every lane.
template <int width = size, typename T>
__device__ __forceinline__ T reduce_sum(T v) {
#pragma unroll
  for (int d = width >> 1; d > 0; d >>= 1) v += __shfl_xor(v, d, width);
  return v;
}

/**
 * Segmented inclusive prefix sum across the 64 lanes.  `head` marks lanes that START a new
 * segment (their own value is the first element of it).  On return `v` holds the running sum
 * of the lane's segment up to and including the lane, and `head` holds "a segment head exists
 * at or before this lane" -- i.e. whether the lane's running sum is closed off from whatever
 * came before lane 0 (the caller's carry-in from the previous wavefront).
 */
template <typename T>
__device__ __forceinline__ void segmented_inclusive_sum(T& v, bool& head) {
  const int l = lane();
  int h = head ? 1 : 0;
#pragma unroll
  for (int d = 1; d < size; d <<= 1) {
    T up_v = __shfl_up(v, d);
    int up_h = __shfl_up(h, d);
    if (l >= d) {
      if (!h) v += up_v;
      h |= up_h;
    }
  }
  head = h != 0;
}

}  // namespace wave
}  // namespace loops
