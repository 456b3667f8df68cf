/**
 * @file wave.hxx
 * @brief 64-lane wavefront primitives for CDNA (gfx950): lane id, inclusive / exclusive
 * prefix sums and the segmented (head-flagged) prefix sum the merge-path SpMV uses to
 * combine per-thread partial rows without atomics.
 *
 * The full-wavefront scans run on the VALU with DPP (data-parallel primitives) modifiers --
 * `row_shr:1/2/4/8` inside each 16-lane row, then `row_bcast:15` and `row_bcast:31` to carry
 * across rows -- six combine steps, no LDS traffic (`__shfl_up` lowers to `ds_bpermute_b32`,
 * an LDS-pipeline instruction with ~100 clk latency each; twelve of them back to back used to be
 * the longest dependent chain of the merge-tile engine).  Sub-wave groups use the shuffle form.
 * Every lane of the wavefront must execute the calls (cross-lane reads of inactive lanes are
 * undefined).
 */
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace loops {
namespace wave {

constexpr int size = 64;

__device__ __forceinline__ int lane() { return static_cast<int>(__lane_id()); }

namespace dpp {
// DPP control words (gfx9): row_shr:n = 0x110 + n, wave_shr:1 = 0x138, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143.
constexpr int row_shr1 = 0x111, row_shr2 = 0x112, row_shr4 = 0x114, row_shr8 = 0x118;
constexpr int wave_shr1 = 0x138, wave_shl1 = 0x130, row_bcast15 = 0x142, row_bcast31 = 0x143;

/// Lane i receives `v` of the lane selected by CTRL; lanes without a source (or masked out by
/// ROW_MASK) receive `fill`.
template <int CTRL, int ROW_MASK = 0xF, typename T>
__device__ __forceinline__ T move(T v, T fill) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "dpp::move: 32- or 64-bit values");
  if constexpr (sizeof(T) == 4) {
    const int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), CTRL,
                                              ROW_MASK, 0xF, false);
    return __builtin_bit_cast(T, r);
  } else {
    const std::uint64_t b = __builtin_bit_cast(std::uint64_t, v), f = __builtin_bit_cast(std::uint64_t, fill);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(
        static_cast<int>(f & 0xFFFFFFFFu), static_cast<int>(b & 0xFFFFFFFFu), CTRL, ROW_MASK, 0xF, false));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(
        static_cast<int>(f >> 32), static_cast<int>(b >> 32), CTRL, ROW_MASK, 0xF, false));
    return __builtin_bit_cast(T, (static_cast<std::uint64_t>(hi) << 32) | lo);
  }
}
}  // namespace dpp

/// Inclusive prefix sum over sub-groups of `width` lanes (power of two <= 64).
template <int width = size, typename T>
__device__ __forceinline__ T inclusive_sum(T v) {
  if constexpr (width == size) {
    v += dpp::move<dpp::row_shr1>(v, T(0));
    v += dpp::move<dpp::row_shr2>(v, T(0));
    v += dpp::move<dpp::row_shr4>(v, T(0));
    v += dpp::move<dpp::row_shr8>(v, T(0));
    v += dpp::move<dpp::row_bcast15, 0xA>(v, T(0));
    v += dpp::move<dpp::row_bcast31, 0xC>(v, T(0));
    return v;
  } else {
    const int l = lane() & (width - 1);
#pragma unroll
    for (int d = 1; d < width; d <<= 1) {
      T up = __shfl_up(v, d, width);
      if (l >= d) v += up;
    }
    return v;
  }
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

/// Value of the previous lane (lane 0 receives `fill`).
template <typename T>
__device__ __forceinline__ T shift_up1(T v, T fill) {
  return dpp::move<dpp::wave_shr1>(v, fill);
}

/// Value of the next lane (lane 63 receives `fill`).
template <typename T>
__device__ __forceinline__ T shift_down1(T v, T fill) {
  return dpp::move<dpp::wave_shl1>(v, fill);
}

/**
 * Segmented inclusive prefix sum across the 64 lanes.  `head` marks lanes that START a new
 * segment (their own value is the first element of it).  On return `v` holds the running sum
 * of the lane's segment up to and including the lane, and `head` holds "a segment head exists
 * at or before this lane" -- i.e. whether the lane's running sum is closed off from whatever
 * came before lane 0 (the caller's carry-in from the previous wavefront).
 * The combine (v, h) <- (h ? v : v + v', h | h') is associative, so the row_shr / row_bcast
 * ladder of the plain scan applies unchanged.
 */
template <typename T>
__device__ __forceinline__ void segmented_inclusive_sum(T& v, bool& head) {
  int h = head ? 1 : 0;
#define LOOPS_SEG_STEP(CTRL, MASK)                       \
  {                                                      \
    const T up_v = dpp::move<CTRL, MASK>(v, T(0));       \
    const int up_h = dpp::move<CTRL, MASK>(h, 0);        \
    v = h ? v : v + up_v;                                \
    h |= up_h;                                           \
  }
  LOOPS_SEG_STEP(dpp::row_shr1, 0xF)
  LOOPS_SEG_STEP(dpp::row_shr2, 0xF)
  LOOPS_SEG_STEP(dpp::row_shr4, 0xF)
  LOOPS_SEG_STEP(dpp::row_shr8, 0xF)
  LOOPS_SEG_STEP(dpp::row_bcast15, 0xA)
  LOOPS_SEG_STEP(dpp::row_bcast31, 0xC)
#undef LOOPS_SEG_STEP
  head = h != 0;
}

}  // namespace wave
}  // namespace loops
