/**
 * @file setup.hxx
 * @brief What every schedule shares: the `algorithms_t` tags, the primary `schedule::setup<...>`
 * template the kernels are written against, and `schedule::detail::layout_bound` -- the part of a
 * schedule object that just carries the layout view (typedef surface, the three constructors, layout()).
 *
 * The tag names and the template parameter list ARE the library's public interface
 * (`schedule::setup<scheme, TPB, TPT, tiles_t, atoms_t, tile_size_t, atom_size_t, layout_t>`, reference
 * include/loops/schedule.hxx:27-66): kernels written for gunrock/loops name them verbatim.
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#include <loops/backend/xpu.hxx>
#include <loops/container/layout.hxx>

namespace loops {
namespace schedule {

enum algorithms_t {
  merge_path_flat,  ///< even share per workgroup + per thread (merge path)
  work_oriented,    ///< even share per thread (merge path)
  thread_mapped,    ///< tile per thread
  group_mapped,     ///< tiles per wavefront/workgroup, atoms lane-strided
  bucketing,        ///< declared by the reference (schedule.hxx:31), never implemented there either
};

template <algorithms_t scheme,
          std::size_t threads_per_block,
          std::size_t threads_per_tile,
          typename tiles_t,
          typename atoms_t,
          typename tile_size_t = std::size_t,
          typename atom_size_t = std::size_t,
          typename layout_type = layout::csr<tiles_t, atoms_t>>
class setup;

namespace detail {

/// The layout-carrying part of a schedule: the public typedefs kernels use, construction from either a
/// CSR-shaped (tile-end array, #tiles, #atoms) triple or a ready layout view, and read access to it.
/// Host-constructible, trivially copyable into a kernel.
template <typename tiles_type, typename atoms_type, typename tile_size_type, typename atom_size_type,
          typename layout_type>
class layout_bound {
 public:
  using tiles_t = tiles_type;
  using atoms_t = atoms_type;
  using tiles_iterator_t = tiles_t*;
  using atoms_iterator_t = atoms_t*;
  using tile_size_t = tile_size_type;
  using atom_size_t = atom_size_type;
  using layout_t = layout_type;

  __host__ __device__ layout_bound() : layout_() {}
  __host__ __device__ layout_bound(tiles_t* tile_ends, tile_size_t num_tiles, atom_size_t num_atoms)
      : layout_(tile_ends, num_tiles, num_atoms) {}
  __host__ __device__ explicit layout_bound(layout_t view) : layout_(view) {}

  __host__ __device__ const layout_t& layout() const { return layout_; }

 protected:
  layout_t layout_;
};

}  // namespace detail
}  // namespace schedule
}  // namespace loops
