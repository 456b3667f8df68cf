/**
 * @file thread_mapped.hxx
 * @brief `setup<thread_mapped, 1, 1, ...>`: thread g visits tiles g, g + G, ... (G = threads in
 * the grid) and walks each tile's atoms sequentially.  Host-constructible POD passed by value
 * into the kernel.  Member surface (tiles(), atoms(t), atoms(t, fn), layout()) as in the reference,
 * include/loops/schedule/thread_mapped.hxx:40-129.
 */
#pragma once

#include <loops/schedule/setup.hxx>
#include <loops/range.hxx>

namespace loops {
namespace schedule {

template <typename tiles_type, typename atoms_type, typename tile_size_type, typename atom_size_type,
          typename layout_type>
class setup<algorithms_t::thread_mapped, 1, 1, tiles_type, atoms_type, tile_size_type, atom_size_type, layout_type>
    : public detail::layout_bound<tiles_type, atoms_type, tile_size_type, atom_size_type, layout_type> {
  using bound_t = detail::layout_bound<tiles_type, atoms_type, tile_size_type, atom_size_type, layout_type>;

 public:
  using typename bound_t::tile_size_t;
  using bound_t::bound_t;  // (), (tile_ends, num_tiles, num_atoms), (layout)

  /// Tiles owned by the calling thread: its grid rank, then one grid further, ...
  __device__ step_range_t<tile_size_t> tiles() const {
    return grid_stride_range(tile_size_t(0), static_cast<tile_size_t>(this->layout_.num_tiles()));
  }

  /// Atoms of `tile`, first to last.
  __device__ auto atoms(const tile_size_t& tile) const { return span(this->layout_.tile_begin(tile), tile); }

  /// Atoms of `tile` from a caller-supplied first atom `first_atom(tile)` on (resuming mid-tile).
  template <typename fn_t>
  __device__ auto atoms(const tile_size_t& tile, fn_t first_atom) const {
    return span(first_atom(tile), tile);
  }

 private:
  template <typename first_t>
  __device__ auto span(first_t first, const tile_size_t& tile) const {
    return loops::range(first, this->layout_.tile_end(tile));
  }
};

}  // namespace schedule
}  // namespace loops
