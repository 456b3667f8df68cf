/**
 * @file iterator.hxx
 * @brief Two tiny POD random-access "iterators" the layout views hand to the schedules:
 * a counting sequence and an affine/clamped tile-end functor sequence.  They replace the
 * thrust counting/transform iterators of the reference (layout.hxx:184-187,398,456-459)
 * with register-only objects that cost nothing inside a wavefront.
 */
#pragma once

#include <hip/hip_runtime.h>

namespace loops {
namespace iterator {

/// it[k] == base + k.
template <typename T>
struct counting {
  T base;
  __host__ __device__ constexpr explicit counting(T b = T(0)) : base(b) {}
  template <typename I>
  __host__ __device__ constexpr T operator[](I k) const { return base + static_cast<T>(k); }
  __host__ __device__ constexpr T operator*() const { return base; }
  __host__ __device__ constexpr counting operator+(T k) const { return counting(base + k); }
};

/// it[k] == min((k + 1) * pitch, total): uniform tiles of `pitch` atoms, last one clipped.
template <typename tile_t, typename atom_t>
struct uniform_tile_end {
  atom_t pitch;
  atom_t total;
  template <typename I>
  __host__ __device__ constexpr atom_t operator[](I k) const {
    atom_t e = static_cast<atom_t>(static_cast<tile_t>(k) + 1) * pitch;
    return e < total ? e : total;
  }
};

}  // namespace iterator
}  // namespace loops
