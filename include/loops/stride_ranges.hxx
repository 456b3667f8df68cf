/** @file stride_ranges.hxx  `grid_stride_range`, `block_stride_range`, `custom_stride_range`, `step_range_t`:
 * defined with the range types in range.hxx. */
#pragma once
#include <loops/range.hxx>
