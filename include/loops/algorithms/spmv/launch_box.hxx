/**
 * @file launch_box.hxx
 * @brief `launch_t<type_t>`: (workgroup size, merge items per thread) for the SpMV kernels, keyed
 * by the compile target (LOOPS_TARGET_GFX) through `launch_box_t` (util/launch_box.hxx).  The
 * reference keeps one "analytical" pair per NVIDIA SM generation and per CDNA generation
 * (algorithms/spmv/launch_box.hxx:56-90); the pairs below are the ones MEASURED on MI355X
 * (loops_autotune_merge_path_f32, 128x7 .. 1024x8).  `launch_t`: 256 x 8 merge items for 4-byte values
 * (256 x 4 for 8-byte ones) -- work_oriented, group_mapped, SpMM, where the tile also sizes LDS row-end
 * and B-row staging.  `merge_path_launch_t`: 512 x 8 (512 x 4) for the SpMV merge_path_flat kernel, whose
 * bit-mask engine keeps only the products in LDS: best or within 3 % of best on every structure tried
 * (C2 98.2 vs 100.5 us for 256 x 8; power-law rows with 8192-wide column bands 44.6 vs 48.8 us).
 */
#pragma once

#include <cstddef>

#include <loops/util/launch_box.hxx>

namespace loops {
namespace algorithms {
namespace spmv {
namespace detail {

/// Merge items per thread: halve the 4-byte figure for 8-byte values (same LDS footprint per tile).
template <typename type_t>
constexpr std::size_t items_for(std::size_t four_byte_items) {
  return sizeof(type_t) > 4 ? 4 : four_byte_items;
}

template <typename type_t>
using cdna3_cdna4_t = launch_box::launch_params_t<launch_box::gfx942 | launch_box::gfx950, 256, items_for<type_t>(8)>;
template <typename type_t>
using cdna3_cdna4_merge_path_t =
    launch_box::launch_params_t<launch_box::gfx942 | launch_box::gfx950, 512, items_for<type_t>(8)>;
template <typename type_t>
using earlier_cdna_t =
    launch_box::launch_params_t<launch_box::gfx906 | launch_box::gfx908 | launch_box::gfx90a, 256, items_for<type_t>(7)>;
template <typename type_t>
using anything_else_t = launch_box::launch_params_t<launch_box::fallback, 256, items_for<type_t>(8)>;

}  // namespace detail

template <typename type_t>
using launch_t = launch_box::launch_box_t<detail::cdna3_cdna4_t<type_t>, detail::earlier_cdna_t<type_t>,
                                          detail::anything_else_t<type_t>>;

/// Launch box of the SpMV merge_path_flat kernel (and of the plans it consumes).
template <typename type_t>
using merge_path_launch_t = launch_box::launch_box_t<detail::cdna3_cdna4_merge_path_t<type_t>,
                                                     detail::earlier_cdna_t<type_t>, detail::anything_else_t<type_t>>;

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
