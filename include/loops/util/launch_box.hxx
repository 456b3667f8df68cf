/**
 * @file launch_box.hxx
 * @brief Compile-time, architecture-keyed launch parameters
 * (`launch_params_t<flags, block, items_per_thread, lds_bytes>` selected by `launch_box_t<...>`)
 * and `occupancy_grid()` for persistent-style launches.
 *
 * Same surface as the reference (include/loops/util/launch_box.hxx:48-239) so user launch boxes
 * keep compiling: the full flag enumeration is kept (NVIDIA `sm_*` bits never match here), the
 * first entry whose flags contain the target (or `fallback`) wins.  The target is gfx950 unless
 * `LOOPS_TARGET_GFX` says otherwise.
 */
#pragma once

#include <cstddef>
#include <type_traits>

#include <loops/backend/xpu.hxx>
#include <loops/util/device.hxx>

#ifndef LOOPS_TARGET_GFX
#define LOOPS_TARGET_GFX 0x950
#endif

namespace loops {
namespace launch_box {

enum sm_flag_t : unsigned int {
  fallback = 1u << 0,
  sm_70 = 1u << 1, sm_72 = 1u << 2, sm_75 = 1u << 3, sm_80 = 1u << 4,
  sm_86 = 1u << 5, sm_89 = 1u << 6, sm_90 = 1u << 7, sm_100 = 1u << 8,
  gfx906 = 1u << 16, gfx908 = 1u << 17, gfx90a = 1u << 18, gfx942 = 1u << 19, gfx950 = 1u << 20,
  gfx1030 = 1u << 21, gfx1100 = 1u << 22, gfx1200 = 1u << 23, gfx1201 = 1u << 24,
};

constexpr sm_flag_t operator|(sm_flag_t a, sm_flag_t b) {
  return static_cast<sm_flag_t>(static_cast<unsigned int>(a) | static_cast<unsigned int>(b));
}
constexpr sm_flag_t operator&(sm_flag_t a, sm_flag_t b) {
  return static_cast<sm_flag_t>(static_cast<unsigned int>(a) & static_cast<unsigned int>(b));
}

constexpr sm_flag_t flag_of_gfx(int gfx) {
  return gfx == 0x950 ? gfx950 : gfx == 0x942 ? gfx942 : gfx == 0x90a ? gfx90a : gfx == 0x908 ? gfx908
       : gfx == 0x906 ? gfx906 : gfx == 0x1030 ? gfx1030 : gfx == 0x1100 ? gfx1100 : gfx == 0x1200 ? gfx1200
       : gfx == 0x1201 ? gfx1201 : static_cast<sm_flag_t>(0u);
}
constexpr sm_flag_t flag_of(int /*cuda compute capability: never a target here*/) { return static_cast<sm_flag_t>(0u); }

constexpr sm_flag_t target_flag = flag_of_gfx(LOOPS_TARGET_GFX);

template <sm_flag_t sm_flags_, std::size_t block_size_, std::size_t items_per_thread_ = 1,
          std::size_t shared_memory_bytes_ = 0>
struct launch_params_t {
  static constexpr sm_flag_t sm_flags = sm_flags_;
  static constexpr std::size_t block_size = block_size_;
  static constexpr std::size_t items_per_thread = items_per_thread_;
  static constexpr std::size_t shared_memory_bytes = shared_memory_bytes_;
};

namespace detail {
template <sm_flag_t>
struct dependent_false : std::false_type {};

template <sm_flag_t target, typename... params_t>
struct select {
  static_assert(dependent_false<target>::value,
                "launch_box_t: no launch_params_t matches LOOPS_TARGET_GFX and no fallback was given.");
};
template <sm_flag_t target, typename head_t, typename... tail_t>
struct select<target, head_t, tail_t...> {
  static constexpr bool matched = static_cast<unsigned int>(head_t::sm_flags & target) != 0u ||
                                  static_cast<unsigned int>(head_t::sm_flags & fallback) != 0u;
  using type = typename std::conditional_t<matched, std::common_type<head_t>, select<target, tail_t...>>::type;
};
}  // namespace detail

template <typename... params_t>
struct launch_box_t : detail::select<target_flag, params_t...>::type {};

/// Blocks that are co-resident for `kernel` at `block_size`: occupancy per CU x number of CUs.
template <typename kernel_t>
inline std::size_t occupancy_grid(const kernel_t& kernel, int block_size, std::size_t dynamic_lds_bytes = 0) {
  int blocks_per_cu = 0;
  (void)xpu::occupancy_max_active_blocks_per_multiprocessor(&blocks_per_cu, kernel, block_size, dynamic_lds_bytes);
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  return static_cast<std::size_t>(blocks_per_cu) * static_cast<std::size_t>(device::multi_processor_count());
}

}  // namespace launch_box
}  // namespace loops
