/**
 * @file partitioning.hxx
 * @brief `layout::flat_uniform_occupancy<K, base_layout_t>`: re-tiles any layout into tiles of exactly K
 * consecutive atoms (the last one may be short), ignoring the base layout's own tile boundaries -- a
 * uniform-occupancy work list on top of an irregular one.  It satisfies the layout contract
 * (num_tiles / num_atoms / tile_begin / tile_end / tile_size / tile_end_iter / tile_of) and keeps the
 * base view reachable through base(); `flat_partitioned<K>` SpMV asks that for the row of an atom.
 * Reference: include/loops/container/partitioning.hxx:72-141.
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#include <loops/iterator.hxx>

namespace loops {
namespace layout {
namespace detail {

/// Arithmetic of "tiles of K atoms over [0, total)".
template <typename tile_id_t, typename atom_id_t, atom_id_t K>
struct fixed_tiling {
  __host__ __device__ static constexpr tile_id_t count(atom_id_t total) {
    return static_cast<tile_id_t>(total / K + (total % K != 0 ? 1 : 0));
  }
  __host__ __device__ static constexpr atom_id_t first(tile_id_t t) { return static_cast<atom_id_t>(t) * K; }
  __host__ __device__ static constexpr atom_id_t last(tile_id_t t, atom_id_t total) {
    const atom_id_t unclipped = first(t) + K;
    return unclipped < total ? unclipped : total;
  }
  __host__ __device__ static constexpr tile_id_t owner(atom_id_t a) { return static_cast<tile_id_t>(a / K); }
};

}  // namespace detail

template <std::size_t K, typename base_layout_type>
struct flat_uniform_occupancy {
  static_assert(K > 0, "flat_uniform_occupancy: K must be positive.");

  using base_layout_t = base_layout_type;
  using tile_id_t = typename base_layout_t::tile_id_t;
  using atom_id_t = typename base_layout_t::atom_id_t;
  using tile_end_iterator_t = iterator::uniform_tile_end<tile_id_t, atom_id_t>;

  static constexpr atom_id_t kAtomsPerTile = static_cast<atom_id_t>(K);
  using tiling_t = detail::fixed_tiling<tile_id_t, atom_id_t, kAtomsPerTile>;

  base_layout_t base_;

  __host__ __device__ flat_uniform_occupancy() : base_() {}
  __host__ __device__ explicit flat_uniform_occupancy(base_layout_t base) : base_(base) {}

  __host__ __device__ const base_layout_t& base() const { return base_; }

  __host__ __device__ atom_id_t num_atoms() const { return base_.num_atoms(); }
  __host__ __device__ tile_id_t num_tiles() const { return tiling_t::count(num_atoms()); }
  __host__ __device__ atom_id_t tile_begin(tile_id_t t) const { return tiling_t::first(t); }
  __host__ __device__ atom_id_t tile_end(tile_id_t t) const { return tiling_t::last(t, num_atoms()); }
  __host__ __device__ atom_id_t tile_size(tile_id_t t) const { return tile_end(t) - tile_begin(t); }
  __host__ __device__ tile_id_t tile_of(atom_id_t a) const { return tiling_t::owner(a); }
  __host__ __device__ tile_end_iterator_t tile_end_iter() const {
    return tile_end_iterator_t{kAtomsPerTile, num_atoms()};
  }
};

}  // namespace layout
}  // namespace loops
