/**
 * @file group_mapped.hxx
 * @brief `setup<group_mapped, TPB, TPT, ...>`: a GROUP of TPT threads (a sub-wave, one
 * 64-lane wavefront, or the whole workgroup) owns TPT consecutive tiles and sweeps the
 * concatenation of their atoms lane-strided, so one long tile is shared by the whole group.
 *
 *   thread with grid rank g owns tile g (or none);  n = tile_size(g)
 *   p_st   = exclusive prefix sum of n inside the group (kept in LDS), aggregate = sum(n)
 *   rank r visits virtual atoms v = r, r + TPT, ... < aggregate
 *   v_tile = upper_bound(p_st[0..length), v) - 1,   length = tiles really owned by the group
 *   tile   = group_base + v_tile,   atom = tile_begin(tile) + v - p_st[v_tile]
 *
 * Semantics restated from include/loops/schedule/group_mapped.hxx:104-192 of the reference.
 * The reference implementation is CUDA-only (cooperative_groups block_tile_memory +
 * cg::exclusive_scan) and is EXCLUDED from its HIP build (schedule.hxx:69-74), so there is no
 * reference HIP behaviour: this file is a fresh CDNA4 design.  Groups are described by
 * `group_t` (thread_rank / meta_group_rank / size / sync); the prefix sum is a 6-step
 * cross-lane scan per 64-lane wavefront (loops/util/wave.hxx), stitched through LDS only when
 * the group spans several wavefronts.  `warp_mapped` means one CDNA wavefront = 64 lanes.
 */
#pragma once

#include <loops/schedule/setup.hxx>
#include <loops/stride_ranges.hxx>
#include <loops/util/wave.hxx>
#include <loops/container/layout.hxx>

namespace loops {
namespace schedule {

/// A tile of `SIZE` consecutive threads of the workgroup (SIZE: power of two <= 64, or TPB).
template <unsigned int SIZE>
struct group_t {
  unsigned int rank_;        ///< rank inside the group
  unsigned int meta_rank_;   ///< index of the group inside the workgroup
  unsigned long long grid_;  ///< rank of the thread inside the grid
  __device__ __forceinline__ unsigned int thread_rank() const { return rank_; }
  __device__ __forceinline__ unsigned int meta_group_rank() const { return meta_rank_; }
  __device__ __forceinline__ static constexpr unsigned int size() { return SIZE; }
  __device__ __forceinline__ unsigned long long grid_rank() const { return grid_; }
  /// Group-wide barrier: free inside one wavefront, a workgroup barrier otherwise.
  __device__ __forceinline__ void sync() const {
    if constexpr (SIZE > wave::size) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
  }
};

template <std::size_t THREADS_PER_BLOCK, std::size_t THREADS_PER_TILE, typename tiles_type, typename atoms_type,
          typename tile_size_type, typename atom_size_type, typename layout_type>
class setup<algorithms_t::group_mapped, THREADS_PER_BLOCK, THREADS_PER_TILE, tiles_type, atoms_type,
            tile_size_type, atom_size_type, layout_type> {
  static_assert(THREADS_PER_TILE <= wave::size ? (wave::size % THREADS_PER_TILE == 0)
                                               : (THREADS_PER_TILE == THREADS_PER_BLOCK),
                "group_mapped: a group is a power-of-two slice of a wavefront, or the whole workgroup.");
  static_assert(THREADS_PER_BLOCK % THREADS_PER_TILE == 0, "group_mapped: TPB must be a multiple of TPT.");

 public:
  using tiles_t = tiles_type;
  using atoms_t = atoms_type;
  using tiles_iterator_t = tiles_t*;
  using atoms_iterator_t = atoms_t*;
  using tile_size_t = tile_size_type;
  using atom_size_t = atom_size_type;
  using layout_t = layout_type;

  enum : unsigned int {
    threads_per_block = THREADS_PER_BLOCK,
    threads_per_tile = THREADS_PER_TILE,
    tiles_per_block = THREADS_PER_BLOCK / THREADS_PER_TILE,
    waves_per_block = (THREADS_PER_BLOCK + wave::size - 1) / wave::size,
  };
  using partition_t = group_t<threads_per_tile>;

  /// LDS scratch of the schedule (declare one `__shared__ storage_t` per workgroup).
  struct alignas(16) storage_t {
    atoms_t tile_aggregates[tiles_per_block];
    atoms_t wave_totals[waves_per_block];
    atoms_t atoms_offsets[threads_per_block];
    tiles_t tiles_indices[threads_per_block];
  };

  storage_t& buffer;

  __device__ __forceinline__ setup(storage_t& _buffer, tiles_iterator_t _tiles, tile_size_t _num_tiles,
                                   atom_size_t _num_atoms)
      : buffer(_buffer), layout_(_tiles, _num_tiles, _num_atoms) {}
  __device__ __forceinline__ setup(storage_t& _buffer, layout_t _layout) : buffer(_buffer), layout_(_layout) {}

  /// Form the groups and publish which tile each thread owns (-1: none).
  __device__ __forceinline__ partition_t partition() {
    partition_t p;
    p.rank_ = threadIdx.x % threads_per_tile;
    p.meta_rank_ = threadIdx.x / threads_per_tile;
    p.grid_ = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    buffer.tiles_indices[threadIdx.x] =
        p.grid_ < static_cast<unsigned long long>(layout_.num_tiles()) ? static_cast<tiles_t>(p.grid_) : tiles_t(-1);
    return p;
  }

  /// Prefix-sum the group's tile sizes and return the calling thread's virtual atoms.
  __device__ step_range_t<atoms_t> atom_accessor(partition_t& p) {
    atoms_t* p_st = buffer.atoms_offsets + p.meta_group_rank() * threads_per_tile;
    atoms_t n = 0;
    if (p.grid_rank() < static_cast<unsigned long long>(layout_.num_tiles()))
      n = layout_.tile_size(static_cast<tiles_t>(p.grid_rank()));

    atoms_t aggregate;
    if constexpr (threads_per_tile <= wave::size) {
      const atoms_t incl = wave::inclusive_sum<threads_per_tile>(n);
      p_st[p.thread_rank()] = incl - n;
      aggregate = __shfl(incl, threads_per_tile - 1, threads_per_tile);
    } else {
      const unsigned int w = threadIdx.x / wave::size;
      const atoms_t incl = wave::inclusive_sum(n);
      if (wave::lane() == wave::size - 1) buffer.wave_totals[w] = incl;
      __syncthreads();
      atoms_t before = 0, total = 0;
#pragma unroll
      for (unsigned int i = 0; i < waves_per_block; ++i) {
        const atoms_t t = buffer.wave_totals[i];
        before += i < w ? t : 0;
        total += t;
      }
      p_st[p.thread_rank()] = before + incl - n;
      aggregate = total;
    }
    if (p.thread_rank() == 0) buffer.tile_aggregates[p.meta_group_rank()] = aggregate;
    p.sync();
    return custom_stride_range(atoms_t(p.thread_rank()), aggregate, atoms_t(p.size()));
  }

  /// Number of tiles the group really owns (the last group of the grid may be short).
  __device__ __forceinline__ int get_length(partition_t& p) const {
    const long long base = static_cast<long long>(p.grid_rank()) - p.thread_rank();
    long long length = base + p.size();
    if (static_cast<long long>(layout_.num_tiles()) < length) length = static_cast<long long>(layout_.num_tiles());
    return static_cast<int>(length - base);
  }

  /// Group-local index of the tile that owns virtual atom `virtual_atom`.
  __device__ __forceinline__ tiles_t tile_accessor(atoms_t& virtual_atom, partition_t& p) const {
    const atoms_t* p_st = buffer.atoms_offsets + p.meta_group_rank() * threads_per_tile;
    int first = 0;
    int count = get_length(p);
    if (count < 0) count = 0;
    while (count > 0) {  // upper_bound: first index with p_st[i] > virtual_atom
      const int half = count >> 1;
      if (p_st[first + half] <= virtual_atom) {
        first += half + 1;
        count -= half + 1;
      } else {
        count = half;
      }
    }
    return static_cast<tiles_t>(first - 1);
  }

  __device__ __forceinline__ bool is_valid_accessor(tiles_t& tile_id, partition_t& p) const {
    return tile_id < get_length(p);
  }

  /// Global tile id of group-local tile `v_tile_id`.
  __device__ __forceinline__ tiles_t tile_id(tiles_t& v_tile_id, partition_t& p) const {
    return buffer.tiles_indices[v_tile_id + p.meta_group_rank() * p.size()];
  }

  /// Global atom id of virtual atom `v_atom`, which lives in tile `tile_id` (group-local `v_tile_id`).
  __device__ __forceinline__ atoms_t atom_id(atoms_t& v_atom, tiles_t& tile_id, tiles_t& v_tile_id,
                                             partition_t& p) const {
    const atoms_t* p_st = buffer.atoms_offsets + p.meta_group_rank() * threads_per_tile;
    return layout_.tile_begin(tile_id) + v_atom - p_st[v_tile_id];
  }

  __host__ __device__ const layout_t& layout() const { return layout_; }

 private:
  layout_t layout_;
};

/// One 64-lane wavefront per group (the reference's `warp_mapped` names a 32-thread warp,
/// group_mapped.hxx:201-203; on CDNA the hardware group is the wavefront).
template <std::size_t threads_per_block, typename tiles_t, typename atoms_t>
using warp_mapped = setup<algorithms_t::group_mapped, threads_per_block, wave::size, tiles_t, atoms_t>;

/// The whole workgroup is one group.
template <std::size_t threads_per_block, typename tiles_t, typename atoms_t>
using block_mapped = setup<algorithms_t::group_mapped, threads_per_block, threads_per_block, tiles_t, atoms_t>;

}  // namespace schedule
}  // namespace loops
