/**
 * @file thread_mapped.hxx
 * @brief `setup<thread_mapped, 1, 1, ...>`: thread g visits tiles g, g + G, ... (G = threads in
 * the grid) and walks each tile's atoms sequentially.  Host-constructible POD passed by value
 * into the kernel.  Member surface follows include/loops/schedule/thread_mapped.hxx:40-129 of
 * the reference (tiles(), atoms(t), atoms(t, fn), layout()).
 */
#pragma once

#include <loops/schedule.hxx>
#include <loops/stride_ranges.hxx>
#include <loops/container/layout.hxx>

namespace loops {
namespace schedule {

template <typename tiles_type, typename atoms_type, typename tile_size_type, typename atom_size_type,
          typename layout_type>
class setup<algorithms_t::thread_mapped, 1, 1, tiles_type, atoms_type, tile_size_type, atom_size_type, layout_type> {
 public:
  using tiles_t = tiles_type;
  using atoms_t = atoms_type;
  using tiles_iterator_t = tiles_t*;
  using atoms_iterator_t = atoms_t*;
  using tile_size_t = tile_size_type;
  using atom_size_t = atom_size_type;
  using layout_t = layout_type;

  __host__ __device__ setup() : layout_() {}
  __host__ __device__ setup(tiles_t* tiles, tile_size_t num_tiles, atom_size_t num_atoms)
      : layout_(tiles, num_tiles, num_atoms) {}
  __host__ __device__ explicit setup(layout_t layout) : layout_(layout) {}

  /// Tiles owned by the calling thread.
  __device__ step_range_t<tile_size_t> tiles() const {
    return grid_stride_range(tile_size_t(0), static_cast<tile_size_t>(layout_.num_tiles()));
  }

  /// Atoms of `tile`, first to last.
  __device__ auto atoms(const tile_size_t& tile) const {
    return loops::range(layout_.tile_begin(tile), layout_.tile_end(tile));
  }

  /// Atoms of `tile` starting from a caller-supplied first atom `first_atom(tile)`.
  template <typename fn_t>
  __device__ auto atoms(const tile_size_t& tile, fn_t first_atom) const {
    return loops::range(first_atom(tile), layout_.tile_end(tile));
  }

  __host__ __device__ const layout_t& layout() const { return layout_; }

 private:
  layout_t layout_;
};

}  // namespace schedule
}  // namespace loops
