/**
 * @file partitioning.hxx
 * @brief `flat_uniform_occupancy<K, base>`: a layout ADAPTOR that re-bins the atoms of any
 * base layout into uniform tiles of K atoms (last tile clipped), so a tile-per-thread
 * schedule becomes perfectly balanced; `base().tile_of(a)` recovers the original tile (row)
 * of an atom.  Reference: include/loops/container/partitioning.hxx:72-141; known answers in
 * unittests/test_layout_flat_partitioner.cu:24-112 (K=2 over 7 atoms -> 4 tiles of sizes
 * 2,2,2,1; tile_of(a) = a / K).
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#include <loops/iterator.hxx>

namespace loops {
namespace layout {

template <std::size_t K, typename base_layout_type>
struct flat_uniform_occupancy {
  static_assert(K > 0, "flat_uniform_occupancy: K must be positive.");

  using base_layout_t = base_layout_type;
  using tile_id_t = typename base_layout_t::tile_id_t;
  using atom_id_t = typename base_layout_t::atom_id_t;
  using tile_end_iterator_t = iterator::uniform_tile_end<tile_id_t, atom_id_t>;

  static constexpr atom_id_t kAtomsPerTile = static_cast<atom_id_t>(K);

  base_layout_t base_;

  __host__ __device__ flat_uniform_occupancy() : base_() {}
  __host__ __device__ explicit flat_uniform_occupancy(base_layout_t base) : base_(base) {}

  __host__ __device__ const base_layout_t& base() const { return base_; }

  __host__ __device__ tile_id_t num_tiles() const {
    return static_cast<tile_id_t>((base_.num_atoms() + kAtomsPerTile - 1) / kAtomsPerTile);
  }
  __host__ __device__ atom_id_t num_atoms() const { return base_.num_atoms(); }
  __host__ __device__ atom_id_t tile_begin(tile_id_t t) const { return static_cast<atom_id_t>(t) * kAtomsPerTile; }
  __host__ __device__ atom_id_t tile_end(tile_id_t t) const {
    const atom_id_t e = static_cast<atom_id_t>(t + 1) * kAtomsPerTile;
    const atom_id_t total = base_.num_atoms();
    return e < total ? e : total;
  }
  __host__ __device__ atom_id_t tile_size(tile_id_t t) const { return tile_end(t) - tile_begin(t); }
  __host__ __device__ tile_end_iterator_t tile_end_iter() const {
    return tile_end_iterator_t{kAtomsPerTile, base_.num_atoms()};
  }
  __host__ __device__ tile_id_t tile_of(atom_id_t a) const { return static_cast<tile_id_t>(a / kAtomsPerTile); }
};

}  // namespace layout
}  // namespace loops
