/**
 * @file work_oriented.hxx
 * @brief `setup<work_oriented, TPB, IPT, ...>`: every THREAD of the grid gets an even share of
 * the merged work list (tiles + atoms).
 *
 *   num_threads = gridDim.x * TPB;   w = ceil_div(tiles + atoms, num_threads)
 *   thread g:  lo = min(w * g, total), hi = min(lo + w, total)
 *              st = split(lo), en = split(hi)            (merge-path diagonal splits)
 *   complete tiles  t in [st.tile, en.tile): atoms [cursor, tile_end(t)), cursor starts at st.atom
 *   remainder       tile en.tile, atoms [cursor, en.atom)
 *
 * Assignment semantics restated from include/loops/schedule/work_oriented.hxx:79-143,156-179
 * of the reference and pinned bit-exact against the oracle (and the reference's device code).
 * MI355X notes: the two diagonal splits are walked in ONE interleaved halving loop, so their
 * dependent loads (L2 / Infinity-Cache latency each) overlap instead of running back to back;
 * tile ends are read through `tile_end_iter()`, so any layout works.
 */
#pragma once

#include <loops/schedule/setup.hxx>
#include <loops/memory.hxx>
#include <loops/stride_ranges.hxx>
#include <loops/util/math.hxx>
#include <loops/container/layout.hxx>

namespace loops {
namespace schedule {

template <std::size_t THREADS_PER_BLOCK, std::size_t ITEMS_PER_THREAD, typename tiles_type, typename atoms_type,
          typename tile_size_type, typename atom_size_type, typename layout_type>
class setup<algorithms_t::work_oriented, THREADS_PER_BLOCK, ITEMS_PER_THREAD, tiles_type, atoms_type,
            tile_size_type, atom_size_type, layout_type> {
 public:
  using tiles_t = tiles_type;
  using atoms_t = atoms_type;
  using tiles_iterator_t = tiles_t*;
  using atoms_iterator_t = atoms_t*;
  using tile_size_t = tile_size_type;
  using atom_size_t = atom_size_type;
  using layout_t = layout_type;

  /// (tile, atom) pair; `init()` returns pair<split_t, split_t> = {start, end}.
  using split_t = pair<atom_size_t, atom_size_t>;
  using map_t = pair<split_t, split_t>;

  enum : unsigned int {
    threads_per_block = THREADS_PER_BLOCK,
    items_per_thread = ITEMS_PER_THREAD,
    items_per_tile = threads_per_block * items_per_thread,
  };

  __device__ __forceinline__ setup(tiles_iterator_t _tiles, tile_size_t _num_tiles, atom_size_t _num_atoms)
      : setup(layout_t(_tiles, _num_tiles, _num_atoms)) {}

  __device__ __forceinline__ explicit setup(layout_t _layout)
      : layout_(_layout),
        total_work(_layout.num_tiles() + _layout.num_atoms()),
        num_threads(gridDim.x * threads_per_block),
        work_per_thread(math::ceil_div(total_work, num_threads)) {}

  /// The calling thread's {start, end} splits of the merged work list.
  __device__ __forceinline__ map_t init() const {
    const std::size_t tid = threadIdx.x + blockIdx.x * blockDim.x;
    const std::size_t begin = work_per_thread * tid < total_work ? work_per_thread * tid : total_work;
    const std::size_t end = begin + work_per_thread < total_work ? begin + work_per_thread : total_work;
    return split2(static_cast<atom_size_t>(begin), static_cast<atom_size_t>(end));
  }

  /// Tiles that END inside the thread's share.
  __device__ __forceinline__ step_range_t<tiles_t> tiles(map_t& m) const {
    return custom_stride_range(tiles_t(m.first.first), tiles_t(m.second.first), tiles_t(1));
  }

  /// Atoms of complete tile `t` that belong to the thread (advances the thread's atom cursor).
  __device__ __forceinline__ step_range_t<atoms_t> atoms(tiles_t t, map_t& m) const {
    const atoms_t next = layout_.tile_end(t);
    const atoms_t first = static_cast<atoms_t>(m.first.second);
    m.first.second += static_cast<atom_size_t>(next - first);
    return custom_stride_range(first, next, atoms_t(1));
  }

  /// The (at most one) tile whose head belongs to this thread but which ends in a later share.
  __device__ __forceinline__ step_range_t<tiles_t> remainder_tiles(map_t& m) const {
    return custom_stride_range(tiles_t(m.second.first), tiles_t(m.second.first + 1), tiles_t(1));
  }

  __device__ __forceinline__ step_range_t<atoms_t> remainder_atoms(map_t& m) const {
    return custom_stride_range(atoms_t(m.first.second), atoms_t(m.second.second), atoms_t(1));
  }

  __host__ __device__ const layout_t& layout() const { return layout_; }

 private:
  /// Both diagonal splits in one interleaved halving loop (same result as two
  /// search::_binary_search calls; int arithmetic as in the reference's private `search`).
  __device__ __forceinline__ map_t split2(atom_size_t d0, atom_size_t d1) const {
    const auto a = layout_.tile_end_iter();
    const int a_len = static_cast<int>(layout_.num_tiles());
    const int b_len = static_cast<int>(layout_.num_atoms());
    const int e0 = static_cast<int>(d0), e1 = static_cast<int>(d1);
    int lo0 = e0 - b_len > 0 ? e0 - b_len : 0, lo1 = e1 - b_len > 0 ? e1 - b_len : 0;
    int n0 = (e0 < a_len ? e0 : a_len) - lo0, n1 = (e1 < a_len ? e1 : a_len) - lo1;
    while (n0 > 0 || n1 > 0) {
      const int h0 = n0 >> 1, h1 = n1 >> 1;
      const int m0 = lo0 + h0, m1 = lo1 + h1;
      // Issue both probes before either compare so the two loads are in flight together.
      const long long p0 = n0 > 0 ? static_cast<long long>(a[m0]) : 0;
      const long long p1 = n1 > 0 ? static_cast<long long>(a[m1]) : 0;
      if (n0 > 0) {
        if (p0 <= static_cast<long long>(e0 - m0 - 1)) { lo0 = m0 + 1; n0 -= h0 + 1; } else { n0 = h0; }
      }
      if (n1 > 0) {
        if (p1 <= static_cast<long long>(e1 - m1 - 1)) { lo1 = m1 + 1; n1 -= h1 + 1; } else { n1 = h1; }
      }
    }
    map_t m;
    m.first.first = static_cast<atom_size_t>(lo0 < a_len ? lo0 : a_len);
    m.first.second = static_cast<atom_size_t>(e0 - lo0);
    m.second.first = static_cast<atom_size_t>(lo1 < a_len ? lo1 : a_len);
    m.second.second = static_cast<atom_size_t>(e1 - lo1);
    return m;
  }

  layout_t layout_;
  std::size_t total_work;
  std::size_t num_threads;
  std::size_t work_per_thread;
};

}  // namespace schedule
}  // namespace loops
