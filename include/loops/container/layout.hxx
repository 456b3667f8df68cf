/**
 * @file layout.hxx
 * @brief Layout views: the tile/atom contract every schedule consumes.
 *
 * A layout view is a small, non-owning, trivially-copyable object passed BY VALUE into
 * kernels.  It answers "how many tiles / atoms" and "which atoms belong to tile t":
 *
 *   num_tiles()  num_atoms()
 *   tile_begin(t)  tile_end(t)  tile_size(t)
 *   tile_end_iter()      random-access object `it` with it[k] == tile_end(k)
 *   tile_of(a)           tile that owns atom a
 *
 * Invariants (restated from the reference contract, include/loops/container/layout.hxx:16-55,
 * and pinned by its unittests/test_layout_contract.hxx:30-88): tile_begin(0) == 0,
 * tile_end(T-1) == num_atoms(), tile ends are monotone, tile_size == end - begin,
 * tile_begin(tile_of(a)) <= a < tile_end(tile_of(a)).
 *
 * Six views (csr, csc, bcsr, coo, ell, dia) + the `flat_uniform_occupancy` partitioner
 * (container/partitioning.hxx).  The three compressed formats share one implementation
 * (they differ only in what a "tile" means: row / column / block-row); the two
 * fixed-pitch formats share another.  Tile-end sequences are the register-only functors
 * of loops/iterator.hxx -- raw `const atom*` for compressed formats, so merge-path code can
 * stage them into LDS with plain coalesced loads.
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#include <loops/iterator.hxx>

namespace loops {
namespace layout {
namespace detail {

/// offsets[0..T] compressed view: tile t owns atoms [offsets[t], offsets[t+1]).
template <typename tile_id_type, typename atom_id_type>
struct compressed {
  using tile_id_t = tile_id_type;
  using atom_id_t = atom_id_type;
  using tile_end_iterator_t = atom_id_t const*;

  atom_id_t const* offsets_;  ///< length num_tiles + 1, non-decreasing
  tile_id_t n_tiles_;
  atom_id_t n_atoms_;

  __host__ __device__ compressed() : offsets_(nullptr), n_tiles_(0), n_atoms_(0) {}
  __host__ __device__ compressed(atom_id_t const* offsets, tile_id_t num_tiles, atom_id_t num_atoms)
      : offsets_(offsets), n_tiles_(num_tiles), n_atoms_(num_atoms) {}

  __host__ __device__ tile_id_t num_tiles() const { return n_tiles_; }
  __host__ __device__ atom_id_t num_atoms() const { return n_atoms_; }
  __host__ __device__ atom_id_t tile_begin(tile_id_t t) const { return offsets_[t]; }
  __host__ __device__ atom_id_t tile_end(tile_id_t t) const { return offsets_[t + 1]; }
  __host__ __device__ atom_id_t tile_size(tile_id_t t) const { return offsets_[t + 1] - offsets_[t]; }
  __host__ __device__ tile_end_iterator_t tile_end_iter() const { return offsets_ + 1; }

  /// Smallest t with tile_end(t) > a (skips empty tiles): O(log T) halving search.
  __host__ __device__ tile_id_t tile_of(atom_id_t a) const {
    tile_id_t first = 0;
    tile_id_t count = n_tiles_;
    while (count > 0) {
      tile_id_t half = count >> 1;
      if (offsets_[first + half + 1] <= a) {
        first += half + 1;
        count -= half + 1;
      } else {
        count = half;
      }
    }
    return first;
  }
};

/// Every tile owns exactly `pitch` atoms: tile t = [t * pitch, (t + 1) * pitch).
template <typename tile_id_type, typename atom_id_type>
struct fixed_pitch {
  using tile_id_t = tile_id_type;
  using atom_id_t = atom_id_type;
  using tile_end_iterator_t = iterator::uniform_tile_end<tile_id_t, atom_id_t>;

  tile_id_t n_tiles_;
  atom_id_t pitch_;

  __host__ __device__ fixed_pitch() : n_tiles_(0), pitch_(0) {}
  __host__ __device__ fixed_pitch(tile_id_t num_tiles, atom_id_t pitch) : n_tiles_(num_tiles), pitch_(pitch) {}

  __host__ __device__ tile_id_t num_tiles() const { return n_tiles_; }
  __host__ __device__ atom_id_t num_atoms() const { return static_cast<atom_id_t>(n_tiles_) * pitch_; }
  __host__ __device__ atom_id_t tile_begin(tile_id_t t) const { return static_cast<atom_id_t>(t) * pitch_; }
  __host__ __device__ atom_id_t tile_end(tile_id_t t) const { return static_cast<atom_id_t>(t + 1) * pitch_; }
  __host__ __device__ atom_id_t tile_size(tile_id_t) const { return pitch_; }
  __host__ __device__ tile_end_iterator_t tile_end_iter() const { return tile_end_iterator_t{pitch_, num_atoms()}; }
  __host__ __device__ tile_id_t tile_of(atom_id_t a) const { return static_cast<tile_id_t>(a / pitch_); }
};

}  // namespace detail

/// CSR: tile = row, atom = nonzero (reference layout.hxx:88-149).
template <typename tile_id_type, typename atom_id_type>
struct csr : detail::compressed<tile_id_type, atom_id_type> {
  using base_t = detail::compressed<tile_id_type, atom_id_type>;
  __host__ __device__ csr() : base_t() {}
  __host__ __device__ csr(atom_id_type const* offsets, tile_id_type num_tiles, atom_id_type num_atoms)
      : base_t(offsets, num_tiles, num_atoms) {}
};

/// CSC: tile = column, atom = nonzero (reference layout.hxx:313-359).
template <typename tile_id_type, typename atom_id_type>
struct csc : detail::compressed<tile_id_type, atom_id_type> {
  using base_t = detail::compressed<tile_id_type, atom_id_type>;
  __host__ __device__ csc() : base_t() {}
  __host__ __device__ csc(atom_id_type const* offsets, tile_id_type num_tiles, atom_id_type num_atoms)
      : base_t(offsets, num_tiles, num_atoms) {}
};

/// BCSR: tile = block-row, atom = dense R x C block (reference layout.hxx:240-285).
template <typename tile_id_type, typename atom_id_type>
struct bcsr : detail::compressed<tile_id_type, atom_id_type> {
  using base_t = detail::compressed<tile_id_type, atom_id_type>;
  __host__ __device__ bcsr() : base_t() {}
  __host__ __device__ bcsr(atom_id_type const* offsets, tile_id_type num_tiles, atom_id_type num_atoms)
      : base_t(offsets, num_tiles, num_atoms) {}
};

/// ELL: tile = row, `pitch` = padded row length (reference layout.hxx:444-496).
template <typename tile_id_type, typename atom_id_type>
struct ell : detail::fixed_pitch<tile_id_type, atom_id_type> {
  using base_t = detail::fixed_pitch<tile_id_type, atom_id_type>;
  __host__ __device__ ell() : base_t() {}
  __host__ __device__ ell(tile_id_type num_tiles, atom_id_type pitch) : base_t(num_tiles, pitch) {}
};

/// DIA: tile = row, `pitch` = number of stored diagonals (reference layout.hxx:167-217).
template <typename tile_id_type, typename atom_id_type>
struct dia : detail::fixed_pitch<tile_id_type, atom_id_type> {
  using base_t = detail::fixed_pitch<tile_id_type, atom_id_type>;
  __host__ __device__ dia() : base_t() {}
  __host__ __device__ dia(tile_id_type num_rows, atom_id_type num_diags) : base_t(num_rows, num_diags) {}
};

/// COO: every nonzero is its own tile (reference layout.hxx:386-421).
template <typename tile_id_type, typename atom_id_type>
struct coo {
  using tile_id_t = tile_id_type;
  using atom_id_t = atom_id_type;
  using tile_end_iterator_t = iterator::counting<atom_id_t>;

  atom_id_t n_nzs_;

  __host__ __device__ coo() : n_nzs_(0) {}
  __host__ __device__ explicit coo(atom_id_t nnz) : n_nzs_(nnz) {}

  __host__ __device__ tile_id_t num_tiles() const { return static_cast<tile_id_t>(n_nzs_); }
  __host__ __device__ atom_id_t num_atoms() const { return n_nzs_; }
  __host__ __device__ atom_id_t tile_begin(tile_id_t t) const { return static_cast<atom_id_t>(t); }
  __host__ __device__ atom_id_t tile_end(tile_id_t t) const { return static_cast<atom_id_t>(t) + 1; }
  __host__ __device__ atom_id_t tile_size(tile_id_t) const { return atom_id_t{1}; }
  __host__ __device__ tile_end_iterator_t tile_end_iter() const { return tile_end_iterator_t(atom_id_t{1}); }
  __host__ __device__ tile_id_t tile_of(atom_id_t a) const { return static_cast<tile_id_t>(a); }
};

}  // namespace layout
}  // namespace loops

#include <loops/container/partitioning.hxx>
