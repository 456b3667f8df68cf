/** @file coordinate.hxx  2-D merge-path coordinate {x = tiles consumed, y = atoms consumed}
 *  (reference: include/loops/container/coordinate.hxx). */
#pragma once
namespace loops {
template <typename index_t>
struct coordinate_t {
  index_t x;
  index_t y;
};
}  // namespace loops
