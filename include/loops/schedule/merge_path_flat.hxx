/**
 * @file merge_path_flat.hxx
 * @brief `setup<merge_path_flat, TPB, IPT, ...>`: the merged work list (tiles + atoms) is cut
 * into merge tiles of TPB * IPT items, one per WORKGROUP; inside the workgroup every thread
 * gets exactly IPT consecutive items.
 *
 *   M = ceil_div(tiles + atoms, TPB * IPT)                       merge tiles = workgroups
 *   workgroup b:  start = split(b * TPB * IPT), end = split((b + 1) * TPB * IPT)   (global)
 *                 tile ends [start.x, end.x + IPT) staged in LDS (clamped to the last tile)
 *   thread t:     local start = split(t * IPT) over the LDS tile ends / atoms from start.y
 *   then IPT steps:  atom if (start.y + y) < lds_tile_end[x] (visit, ++y) else ++x
 *
 * Semantics restated from the reference (include/loops/schedule/merge_path_flat.hxx:45-76
 * pre-pass kernel, :99-172 preprocess_t, :193-390 setup; consumption loop
 * include/loops/algorithms/spmv/merge_path_flat.cuh:71-82); coordinates are bit-identical
 * (int search arithmetic, unsigned coordinates) and pinned against the oracle and against the
 * reference's own device code.
 *
 * Differences by design (MI355X):
 *  - `preprocess_t` is a plain handle {coords*, sizes}: the host object owns one device
 *    allocation (coordinates + the carry-out scratch of the fused SpMV), kernel-side copies
 *    are non-owning views.  (In the reference the by-value kernel copy loses the coordinate
 *    pointer -- SURVEY Q1 -- so its pre-pass result is never consumed; here it is.)
 *  - the global per-workgroup search runs on two lanes of wavefront 0 only when no
 *    precomputed coordinate table is present.
 */
#pragma once

#include <cstddef>
#include <limits>

#include <loops/error.hxx>
#include <loops/schedule/setup.hxx>
#include <loops/stride_ranges.hxx>
#include <loops/iterator.hxx>
#include <loops/util/math.hxx>
#include <loops/util/search.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/device.hxx>
#include <loops/container/coordinate.hxx>
#include <loops/container/layout.hxx>

namespace loops {
namespace schedule {

using coord_t = coordinate_t<unsigned int>;

namespace merge_path {

/// coord[i] = split(i * TPB * IPT) for i in [0, M]: one thread per merge-tile boundary.
template <std::size_t THREADS_PER_BLOCK, std::size_t ITEMS_PER_THREAD, typename layout_t, typename tile_size_t,
          typename atom_size_t>
__global__ void generate_search_coordinates(layout_t layout, tile_size_t num_tiles, atom_size_t num_atoms,
                                            std::size_t num_merge_tiles, coord_t* d_tile_coordinates) {
  using atoms_t = typename layout_t::atom_id_t;
  constexpr std::size_t items_per_tile = THREADS_PER_BLOCK * ITEMS_PER_THREAD;
  const std::size_t i = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i <= num_merge_tiles) {
    const atom_size_t diagonal = static_cast<atom_size_t>(i * items_per_tile);
    d_tile_coordinates[i] = search::_binary_search(diagonal, layout.tile_end_iter(), iterator::counting<atoms_t>(0),
                                                   static_cast<atom_size_t>(num_tiles),
                                                   static_cast<atom_size_t>(num_atoms));
  }
}

/// For every merge tile b: head_start[b] = first atom of the tile (row) the merge tile starts in, and
/// *flag |= 1 if that is more than `limit` atoms before the merge tile (see preprocess_t::classify).
template <typename layout_t>
__global__ void classify_tile_heads(layout_t layout, const coord_t* coords, std::size_t num_merge_tiles,
                                    unsigned int limit, int* flag, int* head_start) {
  const std::size_t b = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= num_merge_tiles) return;
  const coord_t c = coords[b];
  unsigned int start = c.y;
  if (c.x < static_cast<unsigned int>(layout.num_tiles())) {
    start = static_cast<unsigned int>(layout.tile_begin(c.x));
    if (c.y - start > limit) atomicOr(flag, 1);
  }
  head_start[b] = static_cast<int>(start);
}

/**
 * Host-side plan for a (layout, TPB, IPT) triple: the per-workgroup start coordinates, plus
 * the scratch the fused SpMV needs for rows that straddle workgroups (one {row, partial}
 * carry-out per merge tile).  Copies are non-owning views (safe to pass by value to kernels).
 */
template <std::size_t THREADS_PER_BLOCK, std::size_t ITEMS_PER_THREAD, typename tiles_type, typename atoms_type,
          typename tile_size_type, typename atom_size_type, typename layout_type = layout::csr<tiles_type, atoms_type>>
class preprocess_t {
 public:
  using tiles_t = tiles_type;
  using atoms_t = atoms_type;
  using tiles_iterator_t = tiles_t*;
  using atoms_iterator_t = atoms_t*;
  using tile_size_t = tile_size_type;
  using atom_size_t = atom_size_type;
  using layout_t = layout_type;

  static constexpr std::size_t items_per_tile = THREADS_PER_BLOCK * ITEMS_PER_THREAD;
  /// Below this many merge tiles the per-workgroup search is done in-kernel (two lanes).
  static constexpr std::size_t min_tiles_for_prepass = 256;
  /// Third constructor argument: when to run the coordinate pre-pass kernel.
  enum : int { prepass_auto = -1, prepass_never = 0, prepass_always = 1 };

  preprocess_t(tiles_iterator_t _tiles, tile_size_t _num_tiles, atom_size_t _num_atoms, xpu::stream_t stream = 0)
      : preprocess_t(layout_t(_tiles, _num_tiles, _num_atoms), stream) {}

  explicit preprocess_t(layout_t _layout, xpu::stream_t stream = 0, int prepass = prepass_auto)
      : total_work(static_cast<std::size_t>(_layout.num_tiles()) + static_cast<std::size_t>(_layout.num_atoms())),
        num_merge_tiles(math::ceil_div(total_work, items_per_tile)),
        d_tile_coordinates(nullptr),
        d_scratch(nullptr),
        owner(true),
        self_complete_(false),
        layout_copy(_layout) {
    // one allocation: [coords (M+1) x 8 B | carry values (M+2) x 8 B | carry rows (M+2) x 4 B |
    //                  head flag 4 B | head starts (M+1) x 4 B]
    const std::size_t coord_bytes = (num_merge_tiles + 1) * sizeof(coord_t);
    const std::size_t bytes = coord_bytes + (num_merge_tiles + 2) * (sizeof(double) + sizeof(int)) +
                              (num_merge_tiles + 2) * sizeof(int);
    error::throw_if_exception(xpu::malloc(&d_scratch, bytes), "merge_path::preprocess_t: allocation failed.");
    if (prepass == prepass_always || (prepass == prepass_auto && num_merge_tiles >= min_tiles_for_prepass)) {
      d_tile_coordinates = static_cast<coord_t*>(d_scratch);
      constexpr std::size_t block = 256;
      launch::non_cooperative(
          stream, generate_search_coordinates<THREADS_PER_BLOCK, ITEMS_PER_THREAD, layout_t, tile_size_t, atom_size_t>,
          dim3(static_cast<unsigned int>(math::ceil_div(num_merge_tiles + 1, block))), dim3(block), _layout,
          static_cast<tile_size_t>(_layout.num_tiles()), static_cast<atom_size_t>(_layout.num_atoms()),
          num_merge_tiles, d_tile_coordinates);
    }
  }

  /// Non-owning view (this is what travels into kernels by value).
  __host__ __device__ preprocess_t(preprocess_t const& rhs)
      : total_work(rhs.total_work),
        num_merge_tiles(rhs.num_merge_tiles),
        d_tile_coordinates(rhs.d_tile_coordinates),
        d_scratch(rhs.d_scratch),
        owner(false),
        self_complete_(rhs.self_complete_),
        layout_copy(rhs.layout_copy) {}
  preprocess_t& operator=(preprocess_t const&) = delete;

  __host__ __device__ ~preprocess_t() {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (owner && d_scratch) (void)xpu::free(d_scratch);
#endif
  }

  /// Per-workgroup start coordinates (M + 1 entries) or nullptr (search in-kernel).
  __host__ __device__ coord_t* data() const { return d_tile_coordinates; }
  __host__ __device__ std::size_t merge_tiles() const { return num_merge_tiles; }

  /// Carry-out scratch of the fused SpMV: row ids then partial sums, one per merge tile.
  template <typename type_t>
  __host__ __device__ type_t* carry_values() const {
    static_assert(sizeof(type_t) <= sizeof(double), "carry scratch is sized for <= 8-byte values");
    return reinterpret_cast<type_t*>(static_cast<char*>(d_scratch) + (num_merge_tiles + 1) * sizeof(coord_t));
  }
  __host__ __device__ int* carry_rows() const {
    return reinterpret_cast<int*>(static_cast<char*>(d_scratch) + (num_merge_tiles + 1) * sizeof(coord_t) +
                                  (num_merge_tiles + 2) * sizeof(double));
  }

  /// Decides (one stream synchronisation) whether the fused SpMV over this plan can run as ONE kernel: true
  /// when no merge tile starts more than THREADS_PER_BLOCK atoms inside a tile of the layout (row) -- every
  /// merge tile can then re-read the short head of its first row itself and nothing is carried between
  /// tiles (kernels::merge_path_spmv_fused_self).  Needs the coordinate table (prepass run).
  bool classify(xpu::stream_t stream = 0) {
    self_complete_ = false;
    if (!d_tile_coordinates || num_merge_tiles <= 1) return false;
    int* flag = head_flag();
    if (hipMemsetAsync(flag, 0, sizeof(int), stream) != hipSuccess) return false;
    constexpr std::size_t block = 256;
    launch::non_cooperative(stream, classify_tile_heads<layout_t>,
                            dim3(static_cast<unsigned int>(math::ceil_div(num_merge_tiles, block))), dim3(block),
                            layout_copy, d_tile_coordinates, num_merge_tiles,
                            static_cast<unsigned int>(THREADS_PER_BLOCK), flag, head_starts());
    int host_flag = 1;
    if (hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
    if (xpu::stream_synchronize(stream) != 0) return false;
    self_complete_ = host_flag == 0;
    return self_complete_;
  }
  __host__ __device__ bool self_complete() const { return self_complete_; }
  __host__ __device__ int* head_starts() const { return head_flag() + 1; }

 private:
  std::size_t total_work;
  std::size_t num_merge_tiles;
  __host__ __device__ int* head_flag() const { return carry_rows() + (num_merge_tiles + 2); }

  coord_t* d_tile_coordinates;
  void* d_scratch;
  bool owner;
  bool self_complete_;
  layout_t layout_copy;
};

}  // namespace merge_path

template <std::size_t THREADS_PER_BLOCK, std::size_t ITEMS_PER_THREAD, typename tiles_type, typename atoms_type,
          typename tile_size_type, typename atom_size_type, typename layout_type>
class setup<algorithms_t::merge_path_flat, THREADS_PER_BLOCK, ITEMS_PER_THREAD, tiles_type, atoms_type,
            tile_size_type, atom_size_type, layout_type> {
 public:
  using tiles_t = tiles_type;
  using atoms_t = atoms_type;
  using tiles_iterator_t = tiles_t*;
  using atoms_iterator_t = atoms_t*;
  using tile_size_t = tile_size_type;
  using atom_size_t = atom_size_type;
  using layout_t = layout_type;
  using meta_t = merge_path::preprocess_t<THREADS_PER_BLOCK, ITEMS_PER_THREAD, tiles_type, atoms_type,
                                          tile_size_type, atom_size_type, layout_type>;

  enum : unsigned int {
    threads_per_block = THREADS_PER_BLOCK,
    items_per_thread = ITEMS_PER_THREAD,
    items_per_tile = threads_per_block * items_per_thread,
  };

  /// LDS scratch: the workgroup's two global coordinates + the staged tile ends.
  struct storage_t {
    coord_t tile_coords[2];
    tiles_t tile_end_offset[items_per_thread + items_per_tile + 1];
  };

  storage_t& buffer;
  meta_t& meta;

  iterator::counting<atoms_t> atoms_counting_it;  ///< it[y] = first atom of the merge tile + y
  iterator::counting<tiles_t> tiles_counting_it;  ///< it[x] = first tile of the merge tile + x

  __device__ __forceinline__ setup(meta_t& _meta, storage_t& _buffer, tiles_iterator_t _tiles,
                                   tile_size_t _num_tiles, atom_size_t _num_atoms)
      : setup(_meta, _buffer, layout_t(_tiles, _num_tiles, _num_atoms)) {}

  __device__ __forceinline__ setup(meta_t& _meta, storage_t& _buffer, layout_t _layout)
      : buffer(_buffer),
        meta(_meta),
        layout_(_layout),
        total_work(static_cast<std::size_t>(_layout.num_tiles()) + static_cast<std::size_t>(_layout.num_atoms())),
        num_merge_tiles(math::ceil_div(total_work, std::size_t(items_per_tile))),
        tile_num_tiles(0),
        tile_num_atoms(0) {}

  /// Collective (whole workgroup): returns the calling thread's LOCAL start coordinate, or the
  /// invalid coordinate when the workgroup has no merge tile.
  __device__ __forceinline__ coord_t init() {
    const std::size_t tid = static_cast<std::size_t>(blockIdx.x) * gridDim.y + blockIdx.y;
    if (tid >= num_merge_tiles)
      return coord_t{std::numeric_limits<unsigned int>::max(), std::numeric_limits<unsigned int>::max()};

    if (threadIdx.x < 2) {
      if (meta.data() == nullptr) {
        const atom_size_t diagonal = static_cast<atom_size_t>((tid + threadIdx.x) * items_per_tile);
        buffer.tile_coords[threadIdx.x] = search::_binary_search(
            diagonal, layout_.tile_end_iter(), iterator::counting<atoms_t>(0),
            static_cast<atom_size_t>(layout_.num_tiles()), static_cast<atom_size_t>(layout_.num_atoms()));
      } else {
        buffer.tile_coords[threadIdx.x] = meta.data()[tid + threadIdx.x];
      }
    }
    __syncthreads();

    const coord_t tile_start = buffer.tile_coords[0];
    const coord_t tile_end = buffer.tile_coords[1];
    tile_num_tiles = static_cast<tile_size_t>(tile_end.x - tile_start.x);
    tile_num_atoms = static_cast<atom_size_t>(tile_end.y - tile_start.y);

    // Stage the tile ends of this merge tile (+ IPT of slack so the IPT-step walk never reads
    // past the staged window) into LDS; indices are clamped to the last real tile.
    const auto end_offsets = layout_.tile_end_iter();
    const int last_tile = static_cast<int>(layout_.num_tiles()) - 1;
    for (int item = threadIdx.x; item < static_cast<int>(tile_num_tiles) + static_cast<int>(items_per_thread);
         item += threads_per_block) {
      int t = static_cast<int>(tile_start.x) + item;
      t = t < last_tile ? t : last_tile;
      buffer.tile_end_offset[item] = end_offsets[t];
    }

    tiles_counting_it = iterator::counting<tiles_t>(static_cast<tiles_t>(tile_start.x));
    atoms_counting_it = iterator::counting<atoms_t>(static_cast<atoms_t>(tile_start.y));
    __syncthreads();

    return search::_binary_search(atom_size_t(threadIdx.x * items_per_thread), buffer.tile_end_offset,
                                  iterator::counting<atoms_t>(static_cast<atoms_t>(tile_start.y)),
                                  static_cast<atom_size_t>(tile_num_tiles), tile_num_atoms);
  }

  __device__ __forceinline__ bool is_valid_accessor(coord_t& coord) const {
    return coord.x != std::numeric_limits<unsigned int>::max() && coord.y != std::numeric_limits<unsigned int>::max();
  }

  /// The thread's IPT merge steps: 0 .. IPT-1.
  __device__ __forceinline__ step_range_t<int> virtual_idx() const {
    return custom_stride_range(int(0), int(items_per_thread), int(1));
  }

  /// Global atom id at the thread's current position (clamped to the last atom; not advanced).
  __device__ __forceinline__ atoms_t atom_idx(int /*vid*/, coord_t& coord) const {
    const atoms_t a = atoms_counting_it[coord.y];
    const atoms_t last = static_cast<atoms_t>(layout_.num_atoms()) - 1;
    return a < last ? a : last;
  }

  /// Global tile id at the thread's current position (not advanced).
  __device__ __forceinline__ tiles_t tile_idx(coord_t& coord) const { return tiles_counting_it[coord.x]; }

  __device__ __forceinline__ tile_size_t num_tiles() const { return tile_num_tiles; }
  __device__ __forceinline__ atom_size_t num_atoms() const { return tile_num_atoms; }

  __host__ __device__ const layout_t& layout() const { return layout_; }

 private:
  layout_t layout_;
  std::size_t total_work;
  std::size_t num_merge_tiles;
  tile_size_t tile_num_tiles;
  atom_size_t tile_num_atoms;
};

}  // namespace schedule
}  // namespace loops
