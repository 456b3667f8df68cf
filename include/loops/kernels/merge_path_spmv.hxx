/**
 * @file merge_path_spmv.hxx
 * @brief The headline kernel: fused merge-path CSR SpMV for CDNA4 (gfx950, 64-lane wavefronts).
 *
 * Same decomposition as `schedule::setup<merge_path_flat, TPB, IPT>` -- workgroup b owns merge
 * tile b = diagonals [b*TPB*IPT, (b+1)*TPB*IPT) of the (row-ends, nonzeros) merge path and
 * thread t owns IPT consecutive merge items of it, i.e. the SAME (tile, atom) -> (workgroup,
 * thread) assignment the reference's schedule hands out (schedule/merge_path_flat.hxx:267-335,
 * algorithms/spmv/merge_path_flat.cuh:71-82) -- but the per-nonzero global atomicAdd of the
 * reference kernel is gone:
 *
 *  1. STREAM   the tile's col_idx / values are read from HBM once, coalesced, 16 bytes per lane
 *              (global_load_dwordx4 from a 16-byte aligned base), x is gathered through L2 /
 *              Infinity Cache, and the products land in LDS (2048 x 4 B per workgroup).
 *              Row-end offsets of the tile are staged in LDS next to them.
 *  2. SPLIT    each thread finds its start on the merge path: row ends mark their merged position in
 *              an LDS bit mask and the start is a prefix popcount (MASK engines, merge_path_flat), or
 *              a halving search over the LDS-resident row ends (<= 11 LDS probes).
 *  3. WALK     IPT merge steps per thread out of LDS: accumulate in a register, store y[row]
 *              directly for every row that both starts and ends inside the thread.
 *  4. STITCH   partial rows crossing thread boundaries are combined with a 6-step 64-lane
 *              segmented prefix sum on the VALU (DPP row_shr / row_bcast, wave::segmented_inclusive_sum),
 *              wavefronts are stitched
 *              through 4 LDS words, and the one partial row leaving the workgroup goes to a
 *              {row, value} carry-out slot.
 *  5. FIX-UP   a tiny second kernel adds the carry-outs to y (rows longer than a merge tile
 *              span several workgroups: max degree 2^14 vs 2048-item tiles).  Plans in which no
 *              tile starts more than TPB nonzeros inside a row skip 5 altogether: the tile is
 *              extended backwards to the start of its first row (merge_path_spmv_fused_self).
 *
 * Workgroup -> tile mapping is XCD-contiguous (detail::xcd_contiguous): the hardware deals
 * workgroups round-robin to the 8 XCDs, the kernels renumber them so that each XCD walks ONE
 * contiguous run of tiles and its private L2 sees one neighbourhood of x.
 *
 * Steps 1-4 are the `merge_tile_engine`; the three tuned CSR kernels differ only in how they
 * cut the merge path into tiles: merge_path_flat (one plan tile per workgroup), work_oriented
 * (an even contiguous share of 1-4 plan tiles per workgroup, eight shares per resident workgroup,
 * the open row carried in a register) and group_mapped (the tiles of the workgroup's own 256 rows,
 * found in LDS -- no plan; heavy groups shared out: group_mapped_spmv.hxx).
 *
 * y needs NO zero-fill: every row is stored exactly once by the thread that consumes its
 * row-end item; the summation order is deterministic (no floating-point atomics).
 *
 * HBM traffic per merge tile: natoms * (4 + sizeof(T)) streamed + (nrows + IPT) * 4 row ends +
 * nrows * sizeof(T) stores (+ 8 B coordinates, + 12 B carry-out) -- the algorithmic minimum of
 * SURVEY 8(d) plus ~1 %; x is served by L2 / MALL.
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include <loops/container/coordinate.hxx>
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

using coord_t = coordinate_t<unsigned int>;

namespace detail {

/// Cache policy of a tile's memory accesses -- the `NT` template parameter of the engine and its kernels:
///   0 (false)                 plain global loads
///   1 (true)                  non-temporal global loads of col_idx / values
///   policy::make(i, v, x)     buffer loads carrying explicit gfx950 cache-policy bits (the `aux` operand of
///                             raw.buffer.load: SC0 = 1, NT = 2, SC1 = 16) for the col_idx stream, the values stream
///                             and the x gather; x_aux = 0 keeps the gather a plain global load (no 4 GB limit on x)
namespace policy {
constexpr int sc0 = 1, nt = 2, sc1 = 16;
constexpr int make(int idx_aux, int val_aux, int x_aux = 0) { return 0x10000 | idx_aux | (val_aux << 5) | (x_aux << 10); }
constexpr bool buffered(int p) { return (p & 0x10000) != 0; }
constexpr int idx_aux(int p) { return p & 31; }
constexpr int val_aux(int p) { return (p >> 5) & 31; }
constexpr int x_aux(int p) { return (p >> 10) & 31; }
constexpr unsigned int rsrc_flags = 0x00020000u;  // gfx9 raw buffer: DATA_FORMAT = 32 bit, no swizzle, no TID
/// policy::phased(M): plain loads, the x gathers of a tile issued in M passes by column range (merge_tile_engine::
/// stream_vectors; M a power of two).  The part a tile starts with is (clock / period) mod M; shift and period are run-time
/// values (`phase_args`, chosen by the launcher from the matrix's column count: kernels::phased_config_for).
constexpr int phased_flag = 0x20000;
constexpr int phased(int m) { return phased_flag | m; }
constexpr int phases(int p) { return (p & phased_flag) != 0 ? (p & 0xff) : 0; }
/// policy::phased_auto(M): both forms in one kernel, chosen at RUN time by `phase_args::enabled` (workgroup-uniform): what the
/// plan-less entry points launch behind column_scatter_sample -- the device decides, the host never waits for the answer.
constexpr int phased_auto_flag = 0x40000;
constexpr int phased_auto(int m) { return phased_flag | phased_auto_flag | m; }
constexpr bool phased_is_auto(int p) { return (p & phased_auto_flag) != 0; }
/// policy::windowed(L) -- MEASUREMENT ONLY (libloops_probes.so; profiles/r05_c3_lds_window_experiment.txt): every interior tile
/// copies 2^L consecutive elements of x, centred on the column of the tile's middle row (`phase_args::window_cols` = the column
/// count), into LDS with coalesced 16-byte loads; gathers whose column falls inside are served from LDS, the others from
/// memory.  For matrices whose nonzeros sit near the diagonal at the scale of a tile's rows.  Same products, same bits.
constexpr int windowed_flag = 0x80000;
constexpr int windowed(int log2_w) { return windowed_flag | log2_w; }
constexpr int window(int p) { return (p & windowed_flag) != 0 ? 1 << (p & 0xff) : 0; }
}  // namespace policy

/// Run-time side of the phased gathers: part of a column = min(col >> shift, M - 1); the clock's current part =
/// umulhi(low 32 bits of s_memrealtime, inv_ticks) mod M with inv_ticks = 2^32 / (period in 10 ns ticks).
struct phase_args {
  unsigned int shift = 0;
  unsigned int inv_ticks = 0;
  unsigned int enabled = 1;  ///< policy::phased_auto kernels only: 0 = gather as the plain kernel does
  unsigned int window_cols = 0;  ///< policy::windowed kernels only: columns of the matrix (>= the window)
  unsigned int window_lo = 0;    ///< (set per tile by the engine: first column of the tile's window)
};

/// LDS window of x of a policy::windowed engine (nothing otherwise: empty base, the storage keeps its size).
template <typename T, int W>
struct window_store {
  alignas(16) T xwin[W];
};
template <typename T>
struct window_store<T, 0> {};

/// The rule of kernels::columns_look_scattered on the four counts of column_scatter_sample (far pairs in one part of x / seen, adjacent
/// pairs in one 128-byte line / seen): the far pairs share a part LESS than twice as often as uniformly scattered columns would
/// (1 / parts) -- columns that use the parts unevenly (hub columns at neighbouring ids: R-MAT graphs in generator order, 3.2 / parts
/// at scale 23) leave most passes nearly empty and already hit the L2 -- and fewer than a quarter of the adjacent pairs share a line.
__host__ __device__ inline bool scatter_counts_say_phase(unsigned int far_same, unsigned int far_seen, unsigned int near_same,
                                                         unsigned int near_seen, unsigned int parts) {
  return far_seen >= 1024u && static_cast<unsigned long long>(far_same) * parts < 2ull * far_seen && 4ull * near_same < near_seen;
}

/// 4 consecutive elements at byte offset `byte_off` of buffer `r` with cache-policy bits AUX (see policy).
template <typename T, int AUX>
__device__ __forceinline__ void buffer_load4(const __amdgpu_buffer_rsrc_t r, const int byte_off, T (&out)[4]) {
  using w4 = unsigned int __attribute__((ext_vector_type(4)));
  if constexpr (sizeof(T) == 4) {
    using v4 = T __attribute__((ext_vector_type(4)));
    const w4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
    const v4 v = __builtin_bit_cast(v4, raw);
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
  } else {
    static_assert(sizeof(T) == 8, "buffer_load4: 4- or 8-byte elements");
    using v2 = T __attribute__((ext_vector_type(2)));
    const w4 ra = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
    const w4 rb = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 16, 0, AUX);
    const v2 a = __builtin_bit_cast(v2, ra), b = __builtin_bit_cast(v2, rb);
    out[0] = a.x;
    out[1] = a.y;
    out[2] = b.x;
    out[3] = b.y;
  }
}

/// One element at byte offset `byte_off` of buffer `r` with cache-policy bits AUX.
template <typename T, int AUX>
__device__ __forceinline__ T buffer_load1(const __amdgpu_buffer_rsrc_t r, const unsigned int byte_off) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, static_cast<int>(byte_off), 0, AUX));
  } else {
    using w2 = unsigned int __attribute__((ext_vector_type(2)));
    const w2 raw = __builtin_amdgcn_raw_buffer_load_b64(r, static_cast<int>(byte_off), 0, AUX);
    return __builtin_bit_cast(T, raw);
  }
}

/// 4 consecutive elements with the widest loads the element size allows: one 16-byte
/// global_load_dwordx4 for 32-bit elements, two for 64-bit ones.  (Written against the native
/// ext-vector types: element-wise bit_cast of a __vector_size__ vector miscompiles on ROCm 7.2
/// -- every lane of the result aliases element 0.)
template <typename V, bool NT>
__device__ __forceinline__ V load_vec(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(static_cast<const V*>(p));
  else return *static_cast<const V*>(p);
}

template <typename T, bool NT>
__device__ __forceinline__ void load4(const T* __restrict__ p, T (&out)[4]) {
  if constexpr (sizeof(T) == 4) {
    using v4 = T __attribute__((ext_vector_type(4)));
    const v4 v = load_vec<v4, NT>(p);
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
  } else {
    static_assert(sizeof(T) == 8, "load4: 4- or 8-byte elements");
    using v2 = T __attribute__((ext_vector_type(2)));
    const v2 a = load_vec<v2, NT>(p);
    const v2 b = load_vec<v2, NT>(p + 2);
    out[0] = a.x;
    out[1] = a.y;
    out[2] = b.x;
    out[3] = b.y;
  }
}

/// Tile owned by workgroup `i` of `m` when XCD (i mod 8) is to work on one contiguous run of
/// tiles.  Workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2 each): a contiguous run
/// per XCD keeps the x / B rows that neighbouring tiles share (banded, clustered, web-graph
/// matrices) in ONE L2 instead of all eight.
__device__ __forceinline__ int xcd_contiguous(int i, int m) {
  constexpr int XCDS = 8;
  const int k = i % XCDS, j = i / XCDS;
  const int base = m / XCDS, rem = m % XCDS;
  return k * base + (k < rem ? k : rem) + j;
}

/// LDS index of product slot i.  With PAD one word of padding every 32 slots turns the
/// stride-IPT (IPT = 8) per-thread walk into a conflict-free ds_read_b32 pattern.
template <bool PAD>
__device__ __forceinline__ int slot(int i) {
  if constexpr (PAD) return i + (i >> 5);
  else return i;
}

}  // namespace detail

/// Where a finished row of y goes.  `plain_store`: y[r] = v.  `fanout_store`: y[r] = v AND the same element of up to
/// seven peer vectors -- the allgatherv(y) of a multi-GPU SpMV fused into the epilogue: `base[p]` is a
/// peer-mapped pointer (xGMI, hipIpcOpenMemHandle / peer access) to where THIS shard's y[0] lives in peer p's full
/// vector, written with system-scope (write-through) stores so the data is in the peer's memory when the kernel ends.
template <typename type_t>
struct plain_store {
  type_t* __restrict__ y;
  __device__ __forceinline__ void operator()(const int r, const type_t v) const { y[r] = v; }
  __device__ __forceinline__ type_t load(const int r) const { return y[r]; }
};
constexpr int max_peers = 7;  // 8 GPUs per node
template <typename type_t>
struct peer_fanout {
  int count;
  type_t* base[max_peers];
};
template <typename type_t>
struct fanout_store {
  type_t* __restrict__ y;
  peer_fanout<type_t> peers;
  __device__ __forceinline__ void operator()(const int r, const type_t v) const {
    y[r] = v;
    for (int p = 0; p < peers.count; ++p) __hip_atomic_store(peers.base[p] + r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __device__ __forceinline__ type_t load(const int r) const { return y[r]; }
};

/// Row ends of a CSR: row r ends at offsets[r + 1].
template <typename offset_t>
struct csr_row_end {
  const offset_t* __restrict__ offsets;
  __device__ __forceinline__ offset_t operator()(const int r) const { return offsets[r + 1]; }
};
/// Row ends of an ELL matrix (rows x pitch cells, row-major): row r ends at (r + 1) * pitch -- no array at all
/// (layout::ell::tile_end_iter of the layout contract, container/layout.hxx).
struct ell_row_end {
  int pitch;
  __device__ __forceinline__ int operator()(const int r) const { return (r + 1) * pitch; }
};

/// coord[i] = merge-path split at diagonal i * tile_items, i in [0, M]  (one lane each).  `row_end(r)` = end of
/// row r on the nonzero axis (csr_row_end / ell_row_end above, or any callable).
template <typename row_end_t>
__global__ void merge_path_coordinates_of(const row_end_t row_end, int rows, int nnz, int tile_items, int num_merge_tiles,
                                          coord_t* __restrict__ coords) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > num_merge_tiles) return;
  const long long dl = static_cast<long long>(i) * tile_items;
  const int d = static_cast<int>(dl);  // int, like the reference (search.hxx:46-47)
  int lo = d - nnz > 0 ? d - nnz : 0;
  int count = (d < rows ? d : rows) - lo;
  while (count > 0) {
    const int half = count >> 1;
    const int mid = lo + half;
    if (static_cast<int>(row_end(mid)) <= d - mid - 1) {
      lo = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  coords[i] = coord_t{static_cast<unsigned int>(lo < rows ? lo : rows), static_cast<unsigned int>(d - lo)};
}

/// Largest "head" over all merge tiles: head(b) = nonzeros of the row a tile STARTS in that lie before the
/// tile (coords[b].y - offsets[coords[b].x]).  If no head exceeds `limit`, every row crosses at most one tile
/// boundary and the closing tile can re-read the <= limit nonzeros it is missing itself: the plan is
/// "self-completing" -- no carry-outs, no fix-up kernel (merge_path_spmv_fused<..., SELF = true>).
/// Also records head_start[b] = the first nonzero of that row (what the self-completing kernel starts from).
template <typename offset_t>
__global__ void merge_path_head_check(const coord_t* __restrict__ coords, const int num_merge_tiles, const int rows,
                                      const offset_t* __restrict__ offsets, const int limit, int* __restrict__ flag,
                                      int* __restrict__ head_start) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= num_merge_tiles) return;
  const coord_t c = coords[b];
  int start = static_cast<int>(c.y);
  if (static_cast<int>(c.x) < rows) {
    start = static_cast<int>(offsets[c.x]);
    if (static_cast<int>(c.y) - start > limit) atomicOr(flag, 1);
  }
  head_start[b] = start;
}

/**
 * The merge-tile engine shared by the three tuned CSR SpMV kernels.  One call processes ONE
 * merge tile -- `nrows` row ends and `natoms` nonzeros starting at (row0, nz0) on the merge path,
 * nrows + natoms <= TPB * IPT -- with the whole workgroup:
 *
 *   STREAM  col_idx / values of [nz0, nz0 + natoms) -> products in LDS (16 B per lane loads)
 *   SPLIT   per-thread start on the merge path (halving search over the LDS row ends `re`)
 *   WALK    IPT merge steps per thread out of LDS; rows that start and end inside a thread are
 *           stored straight to y
 *   STITCH  partial rows across threads (64-lane segmented prefix sum) and wavefronts (LDS)
 *
 * `re[i]` must hold the end offset of row (row0 + i) for i < nrows + IPT (clamped past the
 * last row), in LDS, visible to the whole workgroup once the engine's first barrier is passed.
 * `carry_in` is the partial sum of row `row0` accumulated by earlier tiles of the SAME workgroup
 * (0 for the first); the return value is the partial sum of the row still open when the tile
 * ends (row0 + nrows), uniform across the workgroup.  Collective: every thread of the workgroup
 * must call it; contains 3 workgroup barriers.
 */
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t,
          bool MASK = false, bool PADCOLS = false>
struct merge_tile_engine {
  /// PADCOLS: a negative column index marks a padding cell (ELL, container/ell.hxx:31-55): it contributes 0 and its
  /// x is never read (the gather goes to x[0] and the product is discarded).
  static __device__ __forceinline__ unsigned int gather_index(const index_t col) {
    if constexpr (PADCOLS) return col < 0 ? 0u : static_cast<unsigned int>(col);
    else return static_cast<unsigned int>(col);  // column ids are non-negative: zero-extend
  }
  static __device__ __forceinline__ type_t product(const index_t col, const type_t val, const type_t xv) {
    if constexpr (PADCOLS) return col < 0 ? type_t(0) : val * xv;
    else return val * xv;
  }

  static constexpr int TILE = TPB * IPT;
  static constexpr int WAVES = TPB / wave::size;
  static constexpr int NPROD = TILE + 4;                            // + alignment slack
  static constexpr int KV = (NPROD + 4 * TPB - 1) / (4 * TPB);      // vector-load rounds per thread

  static constexpr int WINDOW = detail::policy::window(NT);
  static_assert(WINDOW % (4 * TPB) == 0, "x window: whole 16-byte vectors per thread");

  struct storage_t : detail::window_store<type_t, WINDOW> {
    // + a 4-slot dump group for the surplus lanes of the last STREAM round; 16-byte aligned so that unpadded engines
    // store a lane's four products with one ds_write_b128 (4 x ds_write_b32 at a 16-byte lane stride conflict 4 ways)
    alignas(16) type_t prod[PAD ? (NPROD + 4) + ((NPROD + 4) >> 5) + 1 : NPROD + 4];
    type_t wave_val[WAVES];
    int wave_head[WAVES];
    type_t carry;
    // MASK split: bit p of `mask` = 1 iff merged position p of the tile is a row end
    unsigned int mask[MASK ? TILE / 32 + 2 : 1];
    int wave_ends[MASK ? WAVES : 1];
    // MASK engines: the first YT rows a tile completes are staged here and leave as coalesced stores
    type_t ytile[MASK ? TILE / 4 : 1];
  };
  static constexpr int YT = MASK ? TILE / 4 : 0;

  /// MASK engines: zero the row-end marks (whole workgroup; follow with a barrier before marking).
  static __device__ __forceinline__ void clear_marks(storage_t& s) {
    for (int i = threadIdx.x; i < TILE / 32 + 2; i += TPB) s.mask[i] = 0u;
  }
  /// MASK engines: row (row0 + i) of the tile ends at nonzero offset `end` (tile starts at nonzero nz0).
  static __device__ __forceinline__ void mark_row_end(storage_t& s, int i, int end, int nz0) {
    const int p = (end - nz0) + i;
    atomicOr(&s.mask[p >> 5], 1u << (p & 31));
  }

  /// STREAM of a 16-byte aligned tile in three straight-line phases, so that every load of a phase is in flight
  /// before the first one is waited for: (1) col_idx / values vectors, (2) the x gathers, (3) products -> LDS.
  /// `mark` runs between (1) and (2).  INTERIOR: no load of the tile can pass the end of the arrays.
  template <bool INTERIOR, typename mark_t>
  static __device__ __forceinline__ void stream_vectors(storage_t& s, const int abase, const int nz1, const int nnz,
                                                        const index_t* __restrict__ indices,
                                                        const type_t* __restrict__ values,
                                                        const type_t* __restrict__ x, mark_t& mark,
                                                        const detail::phase_args phase = {}) {
    const int tid = threadIdx.x;
    index_t col[KV][4];
    type_t val[KV][4];
    type_t xv[KV][4];
    if constexpr (INTERIOR) {
      // (windowed engines: the window's loads leave first -- their address needs nothing but the tile's coordinates)
      constexpr int WV = WINDOW > 0 ? WINDOW / (4 * TPB) : 1;
      type_t win[WV][4];
      if constexpr (WINDOW > 0) {
#pragma unroll
        for (int v = 0; v < WV; ++v) detail::load4<type_t, false>(x + phase.window_lo + (v * TPB + tid) * 4, win[v]);
      }
      // Branch-free: a lane whose vector lies behind the tile's last nonzero loads the tile's LAST vector instead
      // (same line for all of them) and its products land in slots the walk never reads.
      int emax = (nz1 - 1) & ~3;
      emax = emax > abase ? emax : abase;
      if constexpr (detail::policy::buffered(NT)) {
        // explicit cache-policy bits: the tile's streams as buffer loads relative to the tile's (scalar) base
        namespace pol = detail::policy;
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<index_t*>(indices + abase), 0, KV * TPB * 4 * static_cast<int>(sizeof(index_t)), pol::rsrc_flags);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<type_t*>(values + abase), 0, KV * TPB * 4 * static_cast<int>(sizeof(type_t)), pol::rsrc_flags);
#pragma unroll
        for (int k = 0; k < KV; ++k) {
          int e = (k * TPB + tid) * 4;
          e = e < emax - abase ? e : emax - abase;
          detail::buffer_load4<index_t, pol::idx_aux(NT)>(ri, e * static_cast<int>(sizeof(index_t)), col[k]);
          detail::buffer_load4<type_t, pol::val_aux(NT)>(rv, e * static_cast<int>(sizeof(type_t)), val[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
          int e = abase + (k * TPB + tid) * 4;
          e = e < emax ? e : emax;
          detail::load4<index_t, NT == 1>(indices + static_cast<unsigned int>(e), col[k]);
          detail::load4<type_t, NT == 1>(values + static_cast<unsigned int>(e), val[k]);
        }
      }
      mark();
      if constexpr (detail::policy::buffered(NT) && detail::policy::x_aux(NT) != 0) {
        // (measurement shapes only: 32-bit byte offsets limit x to 4 GB)
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<type_t*>(x), 0, -1, detail::policy::rsrc_flags);
#pragma unroll
        for (int k = 0; k < KV; ++k) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            xv[k][j] = detail::buffer_load1<type_t, detail::policy::x_aux(NT)>(
                rx, gather_index(col[k][j]) * static_cast<unsigned int>(sizeof(type_t)));
        }
      } else if constexpr (WINDOW > 0) {
        // window -> LDS; gathers outside it leave first (execution-masked), then every lane reads LDS (index clamped:
        // no branch), then the select
#pragma unroll
        for (int v = 0; v < WV; ++v) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s.xwin[(v * TPB + tid) * 4 + j] = win[v][j];
        }
        __syncthreads();
        type_t xl[KV][4];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned int c = gather_index(col[k][j]);
            xv[k][j] = type_t(0);
            if (c - phase.window_lo >= static_cast<unsigned int>(WINDOW)) xv[k][j] = x[c];
          }
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned int d = gather_index(col[k][j]) - phase.window_lo;
            d = d < static_cast<unsigned int>(WINDOW) ? d : static_cast<unsigned int>(WINDOW - 1);
            xl[k][j] = s.xwin[d];
          }
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned int d = gather_index(col[k][j]) - phase.window_lo;
            xv[k][j] = d < static_cast<unsigned int>(WINDOW) ? xl[k][j] : xv[k][j];
          }
        }
      } else {
        bool phased_now = detail::policy::phases(NT) > 1;
        if constexpr (detail::policy::phased_is_auto(NT)) phased_now = phase.enabled != 0;  // (workgroup-uniform)
        if constexpr (detail::policy::phases(NT) > 1) {
         if (phased_now) {
          // PHASED gathers (merge_path_spmv_fused_phased): the tile's gathers leave in M passes, pass p taking
          // the items whose column lies in part p of x (M parts of 2^shift columns), execution-masked, each pass
          // drained (vmcnt(0)) and closed by a workgroup barrier before the next one starts.  Which part comes first is read
          // off a clock every wavefront of the chip shares (s_memrealtime, 100 MHz; the period ~ the time a pass takes), so the
          // workgroups of an XCD gather from the SAME part of x at the same time: the working set of the XCD's L2 is
          // |x| / M + the streams instead of |x| + the streams.  Results are unchanged (the same loads, another order).
          constexpr unsigned int M = detail::policy::phases(NT);
          static_assert((M & (M - 1)) == 0, "phased gathers: a power-of-two number of parts");
          const unsigned int now = static_cast<unsigned int>(__builtin_amdgcn_s_memrealtime());
          const unsigned int first = __umulhi(now, phase.inv_ticks) & (M - 1);
#pragma unroll
          for (int k = 0; k < KV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[k][j] = type_t(0);
          // 8-byte values: the addresses are formed ONCE, outside the passes -- formed inside, the compiler builds each 64-bit address
          // in the destination registers of its load and puts an s_waitcnt vmcnt(0) in front of every masked load from the second
          // pass on (196 waits in the 16-part kernel against 26 for 4-byte values, where the destination is one register)
          constexpr bool WIDE = sizeof(type_t) == 8;
          const type_t* from[WIDE ? KV : 1][4];
          if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < KV; ++k)
#pragma unroll
              for (int j = 0; j < 4; ++j) from[k][j] = x + gather_index(col[k][j]);
          }
#pragma unroll
          for (unsigned int pp = 0; pp < M; ++pp) {
            const unsigned int p = (pp + first) & (M - 1);
#pragma unroll
            for (int k = 0; k < KV; ++k) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const unsigned int c = gather_index(col[k][j]);
                unsigned int part = c >> phase.shift;
                part = part < M - 1 ? part : M - 1;
                if constexpr (WIDE) {
                  if (part == p) xv[k][j] = *from[k][j];
                } else {
                  if (part == p) xv[k][j] = x[c];
                }
              }
            }
            // (partial waits -- vmcnt(4 / 8 / 14) -- and no wait at all measured 2-3 % slower than the full drain, the drain
            // without the barrier 4 % slower: profiles/r04_phased_gather_experiments.txt)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
          }
         }
        }
        if (!phased_now) {
#pragma unroll
          for (int k = 0; k < KV; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              xv[k][j] = x[gather_index(col[k][j])];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        // the round that overhangs the array: a surplus lane's four products go to the dump group behind the array
        // (NPROD is a multiple of 4, so a lane's vector is inside or outside as a whole; no branch: a predicated store
        // would pull this round's gathers behind the other rounds' wait).  Unpadded engines: one ds_write_b128 per round.
        int i = (k * TPB + tid) * 4;
        if ((k + 1) * TPB * 4 > NPROD) i = i < NPROD ? i : NPROD;
#pragma unroll
        for (int j = 0; j < 4; ++j) s.prod[detail::slot<PAD>(i + j)] = product(col[k][j], val[k][j], xv[k][j]);
      }
      return;
    }
    bool live[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int e = abase + (k * TPB + tid) * 4;
      live[k] = e < nz1;
      if (live[k]) {
        if (e + 3 < nnz) {
          detail::load4<index_t, NT == 1>(indices + static_cast<unsigned int>(e), col[k]);
          detail::load4<type_t, NT == 1>(values + static_cast<unsigned int>(e), val[k]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = e + j < nnz;
            col[k][j] = ok ? indices[e + j] : index_t(0);
            val[k][j] = ok ? values[e + j] : type_t(0);
          }
        }
      }
    }
    mark();
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if (live[k]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[k][j] = x[gather_index(col[k][j])];
      }
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if (live[k]) {
        const int i = (k * TPB + tid) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) s.prod[detail::slot<PAD>(i + j)] = product(col[k][j], val[k][j], xv[k][j]);
      }
    }
  }

  struct no_marks {
    __device__ __forceinline__ void operator()() const {}
  };

  // ---- MEASUREMENT ONLY (libloops_probes.so, profiles/r05_c2_persistent_ordering_experiment.txt): a persistent workgroup that
  // issues a tile's x gathers BEFORE the stream loads of its next tile.  The col_idx / values vectors of a tile live in a
  // `tile_regs` the caller owns; `tile_pipe` hands the engine the loaded set of this tile and the set to fill for the next one.
  struct tile_regs {
    index_t col[KV][4];
    type_t val[KV][4];
  };
  struct no_pipe {
    static constexpr bool active = false;
  };
  /// LATE = false: the next tile's streams leave right behind this tile's gathers; true: once the gathers have RETURNED and
  /// the products are in LDS (they are in flight during the walk either way).
  template <bool LATE>
  struct tile_pipe {
    static constexpr bool active = true;
    static constexpr bool late = LATE;
    tile_regs& cur;    ///< this tile's vectors, issued by the previous tile (or the prologue) with issue_streams(abase)
    tile_regs& next;   ///< filled here, behind this tile's gathers
    int next_abase;    ///< 16-byte aligned element base of the next tile (this tile's again when there is none)
    int next_nz1;      ///< end of the next tile on the nonzero axis
  };
  /// Branch-free stream loads of the tile [abase, nz1) (abase a multiple of 4, nnz % 4 == 0): as in stream_vectors, a lane whose
  /// vector lies behind the tile's last nonzero loads the tile's LAST vector -- its products land in slots the walk never reads.
  static __device__ __forceinline__ void issue_streams(tile_regs& r, const int abase, const int nz1,
                                                       const index_t* __restrict__ indices, const type_t* __restrict__ values) {
    const int tid = threadIdx.x;
    int last = (nz1 - 1) & ~3;
    last = last > abase ? last : abase;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      int e = abase + (k * TPB + tid) * 4;
      e = e < last ? e : last;
      detail::load4<index_t, false>(indices + static_cast<unsigned int>(e), r.col[k]);
      detail::load4<type_t, false>(values + static_cast<unsigned int>(e), r.val[k]);
    }
  }
  template <typename mark_t, typename pipe_t>
  static __device__ __forceinline__ void stream_vectors_pipelined(storage_t& s, const int nnz, const index_t* __restrict__ indices,
                                                                  const type_t* __restrict__ values, const type_t* __restrict__ x,
                                                                  mark_t& mark, const pipe_t& pipe) {
    const int tid = threadIdx.x;
    type_t xv[KV][4];
    mark();
#pragma unroll
    for (int k = 0; k < KV; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[k][j] = x[gather_index(pipe.cur.col[k][j])];
    }
    __builtin_amdgcn_sched_barrier(0);  // the gathers leave before the next tile's streams
    if constexpr (!pipe_t::late) issue_streams(pipe.next, pipe.next_abase, pipe.next_nz1, indices, values);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      int i = (k * TPB + tid) * 4;
      if ((k + 1) * TPB * 4 > NPROD) i = i < NPROD ? i : NPROD;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s.prod[detail::slot<PAD>(i + j)] = product(pipe.cur.col[k][j], pipe.cur.val[k][j], xv[k][j]);
    }
    if constexpr (pipe_t::late) {
      __builtin_amdgcn_sched_barrier(0);
      issue_streams(pipe.next, pipe.next_abase, pipe.next_nz1, indices, values);
    }
  }

  /// `mark` (MASK engines): called once, after the tile's col_idx / values loads have been ISSUED and before the
  /// x gathers wait for them -- the caller's row-offset loads + mark_row_end() go there, so that their latency
  /// overlaps the stream's instead of preceding it.  The mask must have been cleared (clear_marks + barrier).
  template <typename mark_t = no_marks>
  static __device__ __forceinline__ type_t run(storage_t& s, const offset_t* re, const int row0, const int nz0,
                                               const int nrows, const int natoms, const int nnz,
                                               const index_t* __restrict__ indices,
                                               const type_t* __restrict__ values, const type_t* __restrict__ x,
                                               type_t* __restrict__ y, const type_t carry_in,
                                               mark_t mark = mark_t{}) {
    return run_to(s, re, row0, nz0, nrows, natoms, nnz, indices, values, x, plain_store<type_t>{y}, carry_in, mark);
  }

  /// The same with the destination of finished rows abstracted: `out(r, v)` stores row r of y (plain_store,
  /// fanout_store).
  template <typename store_t, typename mark_t = no_marks, typename pipe_t = no_pipe>
  static __device__ __forceinline__ type_t run_to(storage_t& s, const offset_t* re, const int row0, const int nz0,
                                                  const int nrows, const int natoms, const int nnz,
                                                  const index_t* __restrict__ indices,
                                                  const type_t* __restrict__ values, const type_t* __restrict__ x,
                                                  const store_t out, const type_t carry_in,
                                                  mark_t mark = mark_t{}, const detail::phase_args phase = {},
                                                  const pipe_t pipe = pipe_t{}) {
    const int tid = threadIdx.x;
    const int nz1 = nz0 + natoms;
    // ---- 1. STREAM ------------------------------------------------------------------------
    const int abase = VEC ? (nz0 & ~3) : nz0;  // 16-byte aligned element base of the tile
    const int shift = nz0 - abase;             // 0..3 leading elements owned by the previous tile
    if constexpr (VEC) {
      // every vector load of the tile in-bounds?  (uniform; false only for the last tile(s) of the matrix)
      if constexpr (pipe_t::active) {
        // (measurement kernels; nnz % 4 == 0 required: every vector the clamp of issue_streams leaves alone is whole)
        stream_vectors_pipelined(s, nnz, indices, values, x, mark, pipe);
      } else if (abase + KV * 4 * TPB <= nnz) {
        if constexpr (WINDOW > 0) {
          // the window: WINDOW columns around the column of the tile's middle row, 16-byte aligned, inside [0, cols)
          detail::phase_args ph = phase;
          int lo = (row0 + (nrows >> 1) - WINDOW / 2) & ~3;
          const int hi = (static_cast<int>(phase.window_cols) - WINDOW) & ~3;
          lo = lo < hi ? lo : hi;
          ph.window_lo = static_cast<unsigned int>(lo > 0 ? lo : 0);
          stream_vectors<true>(s, abase, nz1, nnz, indices, values, x, mark, ph);
        } else {
          stream_vectors<true>(s, abase, nz1, nnz, indices, values, x, mark, phase);
        }
      }
      else stream_vectors<false>(s, abase, nz1, nnz, indices, values, x, mark);
    } else {
      // Unaligned arrays: coalesced 4-byte loads, one element per lane per round.
      mark();
#pragma unroll
      for (int k = 0; k < IPT; ++k) {
        const int i = k * TPB + tid;
        if (i < natoms) {
          const int e = nz0 + i;
          s.prod[detail::slot<PAD>(i)] = product(indices[e], values[e], x[gather_index(indices[e])]);
        }
      }
    }
    __syncthreads();

    // ---- 2. SPLIT: this thread's start on the merge path -------------------------------------
    const int total = nrows + natoms;  // == TILE except in a last / short tile
    const int diag = tid * IPT;
    int tx, ty;
    type_t sum = type_t(0);
    type_t first_sum = type_t(0);
    int first_row = 0;
    bool closed = false;
    if constexpr (MASK) {
      // Row end i sits at merged position (end offset - nz0) + i.  The CALLER has cleared the mask
      // (clear_marks + barrier) and marked every row end of the tile (mark_row_end) before calling; the marks
      // are visible after the STREAM barrier above.  A thread's start is the number of marks before its first
      // item (prefix popcount: one wavefront scan + WAVES LDS words) instead of a dependent halving search, and
      // the walk reads its IPT bits instead of comparing against row ends (`re` is not used).  Same split as
      // the search (search.hxx semantics: a nonzero precedes the row end it belongs to).
      const unsigned long long window =
          (static_cast<unsigned long long>(s.mask[(diag >> 5) + 1]) << 32) | s.mask[diag >> 5];
      const unsigned int bits = static_cast<unsigned int>(window >> (diag & 31)) & ((1u << IPT) - 1u);
      const int ends = __popc(bits);
      const int before = wave::exclusive_sum(ends);
      if (wave::lane() == wave::size - 1) s.wave_ends[tid / wave::size] = before + ends;
      __syncthreads();
      tx = before;
#pragma unroll
      for (int w = 0; w < WAVES; ++w)
        if (w < tid / wave::size) tx += s.wave_ends[w];
      ty = diag - tx;
      // ---- 3. WALK: IPT merge steps, row end or nonzero by the mask bit ----------------------
      // select-based: the only control flow left is the predicated store of a completed row
      const unsigned int live = (diag + IPT <= total) ? ((1u << IPT) - 1u)
                                                      : (diag < total ? ((1u << (total - diag)) - 1u) : 0u);
      const unsigned int end_bits = bits & live;    // bit j: merge step j completes row (row0 + tx)
      const unsigned int atom_bits = ~bits & live;  // bit j: merge step j consumes a nonzero
      // the product under every step, all IPT LDS reads in flight together (their addresses depend on the mask
      // bits only): the walk itself then runs out of registers
      type_t p[IPT];
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        p[j] = s.prod[detail::slot<PAD>(ty + shift)];  // read even on a row end (in bounds, unused)
        ty += (atom_bits >> j) & 1u;
      }
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const bool end = (end_bits >> j) & 1u;
        const bool atom = (atom_bits >> j) & 1u;
        if (end && closed) {  // a row completed inside this thread: stage it (coalesced copy-out below)
          if (tx < YT) s.ytile[tx] = sum;
          else out(row0 + tx, sum);
        }
        first_sum = (end && !closed) ? sum : first_sum;
        first_row = (end && !closed) ? tx : first_row;
        closed = closed || end;
        sum = end ? type_t(0) : (atom ? sum + p[j] : sum);
        tx += end ? 1 : 0;
      }
    } else {
      {
        int lo = diag - natoms > 0 ? diag - natoms : 0;
        int count = (diag < nrows ? diag : nrows) - lo;
        while (count > 0) {
          const int half = count >> 1;
          const int mid = lo + half;
          if (re[mid] <= nz0 + (diag - mid - 1)) {
            lo = mid + 1;
            count -= half + 1;
          } else {
            count = half;
          }
        }
        tx = lo < nrows ? lo : nrows;
        ty = diag - lo;
      }

      // ---- 3. WALK: IPT merge steps out of LDS ------------------------------------------------
      int row_end = re[tx];
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        if (diag + j < total) {
          if (nz0 + ty < row_end) {  // merge step consumes a nonzero
            sum += s.prod[detail::slot<PAD>(ty + shift)];
            ++ty;
          } else {  // merge step consumes a row end: row (row0 + tx) is complete
            if (!closed) {
              first_sum = sum;
              first_row = tx;
              closed = true;
            } else {
              out(row0 + tx, sum);
            }
            sum = type_t(0);
            ++tx;
            row_end = re[tx];
          }
        }
      }
    }

    // ---- 4. STITCH: partial rows across threads / wavefronts ----------------------------------
    const int lane = wave::lane();
    const int w = tid / wave::size;
    type_t run_sum = sum;  // tail partial (row row0 + tx, still open)
    bool head = closed;    // a thread that closed a row starts a new segment with its tail
    wave::segmented_inclusive_sum(run_sum, head);
    const type_t prev_run = wave::shift_up1(run_sum, type_t(0));  // lane 0: nothing before it
    const int prev_head = wave::shift_up1(static_cast<int>(head), 0);
    if (lane == wave::size - 1) {
      s.wave_val[w] = run_sum;
      s.wave_head[w] = head ? 1 : 0;
    }
    __syncthreads();
    // open partial entering this wavefront: earlier wavefronts back to the last one that closed a
    // row, and -- if none of them did -- the workgroup's carry-in.
    type_t wave_in = type_t(0);
    bool reaches_tile_start = true;
    for (int i = w - 1; i >= 0; --i) {
      wave_in += s.wave_val[i];
      if (s.wave_head[i]) {
        reaches_tile_start = false;
        break;
      }
    }
    if (reaches_tile_start) wave_in += carry_in;
    if (closed) {
      const type_t v = first_sum + prev_run + (prev_head ? type_t(0) : wave_in);
      if (MASK && first_row < YT) s.ytile[first_row] = v;
      else out(row0 + first_row, v);
    }
    if (tid == TPB - 1) s.carry = run_sum + (head ? type_t(0) : wave_in);
    __syncthreads();
    if constexpr (MASK) {
      // rows row0 .. row0 + min(nrows, YT) - 1 were all completed in this tile: one coalesced store each
      const int staged = nrows < YT ? nrows : YT;
      for (int i = tid; i < staged; i += TPB) out(row0 + i, s.ytile[i]);
    }
    return s.carry;
  }
};

/**
 * merge_path_flat: one merge tile per workgroup, tile b = coords[b] .. coords[b + 1].
 * @tparam TPB threads per workgroup (multiple of 64), IPT merge items per thread.
 * @tparam PAD  pad the LDS product array (conflict-free walk for even IPT).
 * @tparam NT   stream col_idx / values with non-temporal loads.
 * @tparam VEC  `indices` and `values` are 16-byte aligned (checked by the host launcher).
 */
template <int TPB, int IPT, bool PAD, int NT, bool VEC, bool SELF, bool MASK, bool PADCOLS = false, typename row_end_t,
          typename index_t, typename type_t, typename store_t>
__device__ __forceinline__ void
merge_path_spmv_tile_to(const coord_t* __restrict__ coords, const int rows, const int nnz,
                        const row_end_t row_end, const index_t* __restrict__ indices,
                        const type_t* __restrict__ values, const type_t* __restrict__ x, const store_t out,
                        int* __restrict__ carry_row, type_t* __restrict__ carry_val,
                        const int* __restrict__ head_start = nullptr, const detail::phase_args phase = {}) {
  using offset_t = int;
  // SELF (self-completing plans, merge_path_head_check): the tile is EXTENDED backwards to the start of the row
  // it begins in -- at most TPB extra nonzeros -- so that row is summed completely here and nothing has to be
  // carried in from the previous tile (which sums the same nonzeros into an open tail it then discards).
  // The extended tile has up to TPB * IPT + TPB merge items: one more item per thread.
  constexpr int ITEMS = SELF ? IPT + 1 : IPT;
  constexpr bool PADDED = SELF ? (ITEMS % 2 == 0) : PAD;
  using engine_t = merge_tile_engine<TPB, ITEMS, PADDED, NT, VEC, index_t, offset_t, type_t, MASK, PADCOLS>;
  __shared__ typename engine_t::storage_t s_engine;
  __shared__ offset_t s_re[MASK ? 1 : TPB * ITEMS + ITEMS + 1];

  const int tid = threadIdx.x;
  const int b = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  // tile coordinates (wave-uniform: scalar loads).  A matrix that fits ONE merge tile needs no
  // coordinate table: its tile is the whole merge path (and no fix-up: nothing leaves the tile).
  // All three scalar loads are issued UNCONDITIONALLY and together (one wait instead of three dependent ones at the
  // head of every workgroup): `coords` holds gridDim.x + 1 entries also for a single-tile launch, where their
  // contents are ignored.
  const bool single = gridDim.x == 1;
  const coord_t m0 = coords[b];
  const coord_t m1 = coords[b + 1];
  int head = 0;
  if constexpr (SELF) head = head_start[b];  // start of row `row0` (recorded by merge_path_head_check)
  const int row0 = single ? 0 : static_cast<int>(m0.x);
  int nz0 = single ? 0 : static_cast<int>(m0.y);
  const int nrows = (single ? rows : static_cast<int>(m1.x)) - row0;
  if constexpr (SELF) nz0 = head;
  const int natoms = (single ? nnz : static_cast<int>(m1.y)) - nz0;

  if constexpr (MASK) {
    // row ends of the tile -> marks in the engine's bit mask, straight from the registers that loaded them
    engine_t::clear_marks(s_engine);
    __syncthreads();
  } else {
    // row ends of the tile -> LDS (visible after the engine's first barrier)
    for (int i = tid; i < nrows + ITEMS; i += TPB) {
      int r = row0 + i;
      r = r < rows - 1 ? r : rows - 1;
      s_re[i] = static_cast<offset_t>(row_end(r));
    }
  }
  const type_t carry = engine_t::run_to(s_engine, s_re, row0, nz0, nrows, natoms, nnz, indices, values, x, out, type_t(0), [&]() {
    if constexpr (MASK) {
      for (int i = tid; i < nrows; i += TPB)
        engine_t::mark_row_end(s_engine, i, static_cast<int>(row_end(row0 + i)), nz0);
    }
  }, phase);
  if constexpr (!SELF) {
    if (tid == 0) {
      carry_row[b] = row0 + nrows;  // == c1.x: the row still open when the tile ends
      carry_val[b] = carry;
    }
  }
}

template <int TPB, int IPT, bool PAD, int NT, bool VEC, bool SELF, bool MASK, bool PADCOLS = false, typename row_end_t,
          typename index_t, typename type_t>
__device__ __forceinline__ void
merge_path_spmv_tile(const coord_t* __restrict__ coords, const int rows, const int nnz,
                     const row_end_t row_end, const index_t* __restrict__ indices,
                     const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                     int* __restrict__ carry_row, type_t* __restrict__ carry_val,
                     const int* __restrict__ head_start = nullptr) {
  merge_path_spmv_tile_to<TPB, IPT, PAD, NT, VEC, SELF, MASK, PADCOLS>(coords, rows, nnz, row_end, indices, values, x,
                                                                        plain_store<type_t>{y}, carry_row, carry_val, head_start);
}

template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t,
          bool MASK = false>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused(const coord_t* __restrict__ coords, const int rows, const int nnz,
                      const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                      const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                      int* __restrict__ carry_row, type_t* __restrict__ carry_val) {
  merge_path_spmv_tile<TPB, IPT, PAD, NT, VEC, false, MASK>(coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values,
                                                            x, y, carry_row, carry_val);
}

/// merge_path_flat over an ELL matrix (rows x pitch cells, negative column = padding): the SAME tile function with
/// the row ends produced by a functor instead of an offsets array -- the layout-generic use of the schedule the
/// reference demonstrates with one atomicAdd per cell (algorithms/spmv/ell_merge_path.cuh:32-69), here without
/// atomics and without a zero-filled y.  `coords` from the schedule's own pre-pass over layout::ell.
template <int TPB, int IPT, bool PAD, bool VEC, typename index_t, typename type_t>
__global__ void __launch_bounds__(TPB)
ell_merge_path_spmv_fused(const coord_t* __restrict__ coords, const int rows, const int pitch,
                          const index_t* __restrict__ indices, const type_t* __restrict__ values,
                          const type_t* __restrict__ x, type_t* __restrict__ y, int* __restrict__ carry_row,
                          type_t* __restrict__ carry_val) {
  merge_path_spmv_tile<TPB, IPT, PAD, 0, VEC, false, true, true>(coords, rows, rows * pitch, ell_row_end{pitch}, indices, values,
                                                                 x, y, carry_row, carry_val);
}

/// merge_path_flat with the allgatherv(y) of a multi-GPU run fused into the epilogue: every finished row
/// goes to y AND to the same element of `peers.count` peer-mapped vectors (fanout_store).  Rows that span merge tiles
/// are completed -- on every destination -- by merge_path_spmv_fixup_to.
template <int TPB, int IPT, bool PAD, bool VEC, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_fanout(const coord_t* __restrict__ coords, const int rows, const int nnz,
                             const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                             const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                             const peer_fanout<type_t> peers, int* __restrict__ carry_row, type_t* __restrict__ carry_val) {
  merge_path_spmv_tile_to<TPB, IPT, PAD, 0, VEC, false, true>(coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values, x,
                                                              fanout_store<type_t>{y, peers}, carry_row, carry_val);
}

/// Self-completing variant (plans whose heads are all <= TPB, merge_path_head_check): every tile finishes the
/// rows it closes by itself, nothing is carried between tiles and no fix-up kernel follows.
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t,
          bool MASK = false>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_self(const coord_t* __restrict__ coords, const int* __restrict__ head_start, const int rows,
                           const int nnz, const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                           const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  merge_path_spmv_tile<TPB, IPT, PAD, NT, VEC, true, MASK>(coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values, x, y,
                                                           nullptr, static_cast<type_t*>(nullptr), head_start);
}

/// The same kernel under a second symbol for products issued through an SpMV plan handle (loops_spmv_plan_*, algorithms::spmv::spmv_plan_t:
/// the candidates it times at creation and the products it runs afterwards), so that the plain symbol's per-kernel statistics
/// in a profile are those of the direct entry points only (bench.py: the headline kernel on the headline matrix).
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t,
          bool MASK = false>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_planned(const coord_t* __restrict__ coords, const int rows, const int nnz,
                              const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                              const type_t* __restrict__ values, const type_t* __restrict__ x,
                              type_t* __restrict__ y, int* __restrict__ carry_row, type_t* __restrict__ carry_val) {
  merge_path_spmv_tile<TPB, IPT, PAD, NT, VEC, false, MASK>(coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values,
                                                            x, y, carry_row, carry_val);
}

/// merge_path_flat with PHASED x gathers (round 4): the same tile kernel, but a tile's gathers leave in PHASES passes by
/// column range -- part = min(col >> phase.shift, PHASES - 1) -- and the pass a workgroup starts with is read off the
/// chip-wide clock, so the workgroups of an XCD work on the same part of x at the same time (merge_tile_engine::
/// stream_vectors).  For matrices whose columns are scattered over an x of about the size of one XCD's L2 (C2: 4 MB = the L2):
/// the matrix stream evicts a fifth of x there, and every evicted line costs an Infinity-Cache round trip; with 8 parts the
/// L2 hit rate of the kernel goes from 80.8 % to 89 % (x misses 2.35 M -> 0.9 M per product) and the kernel from 94.9 to
/// ~84 us.  Same bits as the plain kernel.  Never a default: the autotuner / the SpMV plan adopt it by measurement (it costs
/// 2-10 % where the gathers hit anyway).
template <int TPB, int IPT, int PHASES, bool VEC, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_phased(const coord_t* __restrict__ coords, const int rows, const int nnz,
                             const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                             const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                             int* __restrict__ carry_row, type_t* __restrict__ carry_val, const detail::phase_args phase) {
  merge_path_spmv_tile_to<TPB, IPT, true, detail::policy::phased(PHASES), VEC, false, true>(
      coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values, x, plain_store<type_t>{y}, carry_row, carry_val, nullptr,
      phase);
}

/// The kernel of the PLAN-LESS entry points on matrices large enough to ask (kernels::columns_worth_sampling): plain or phased
/// gathers by what column_scatter_sample -- launched in front of it on the same stream -- wrote to `stats` (the rule of
/// columns_look_scattered, evaluated here by every workgroup: four scalar loads).  No host round trip: the call stays
/// asynchronous.
template <int TPB, int IPT, int PHASES, bool VEC, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_auto(const coord_t* __restrict__ coords, const int rows, const int nnz,
                           const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                           const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                           int* __restrict__ carry_row, type_t* __restrict__ carry_val, const unsigned int* __restrict__ stats,
                           detail::phase_args phase) {
  const unsigned int far_same = stats[0], far_seen = stats[1], near_same = stats[2], near_seen = stats[3];
  phase.enabled = detail::scatter_counts_say_phase(far_same, far_seen, near_same, near_seen, PHASES) ? 1u : 0u;
  merge_path_spmv_tile_to<TPB, IPT, true, detail::policy::phased_auto(PHASES), VEC, false, true>(
      coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values, x, plain_store<type_t>{y}, carry_row, carry_val, nullptr,
      phase);
}

/// The SELF-COMPLETING form (plans over short rows: one kernel, no carry-outs, merge_path_spmv_fused_self) with phased gathers:
/// short rows with scattered columns -- random graphs, "8 M rows x 2 nonzeros" -- gain as C2 does.
template <int TPB, int IPT, int PHASES, bool VEC, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_self_phased(const coord_t* __restrict__ coords, const int* __restrict__ head_start, const int rows,
                                  const int nnz, const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                                  const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                                  const detail::phase_args phase) {
  merge_path_spmv_tile_to<TPB, IPT, true, detail::policy::phased(PHASES), VEC, true, true>(
      coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values, x, plain_store<type_t>{y}, nullptr,
      static_cast<type_t*>(nullptr), head_start, phase);
}

/// Are the columns SCATTERED at the scale the phased gathers work on -- and not already cheap to gather?  `samples` positions
/// i = k * stride of the nonzero stream are looked at twice:
///   FAR   the pair (i, i + far + a per-sample offset below 4096), one to two merge tiles apart: do the two columns fall into the same part of x
///         (min(col >> shift, parts - 1))?  Uniformly scattered columns: ~1 / parts of the pairs; bands, host blocks, dense hub
///         rows: most of them (a tile's gathers then stay inside one or two parts whatever the order); hub COLUMNS at neighbouring
///         ids (R-MAT in generator order): 2-3 / parts -- the threshold is 2 / parts (detail::scatter_counts_say_phase);
///   NEAR  the pair (i, i + 1): do the two columns share a 128-byte line of x (col >> line_shift)?  Runs of consecutive
///         columns do -- their gathers coalesce and hit L1, phasing them only adds passes.
/// (Adjacent nonzeros say nothing about scatter: columns are sorted inside a row, so a row of 16 uniformly random columns
/// has its neighbours half a part apart.)  out[0] = far pairs in one part, out[1] = far pairs seen, out[2] = near pairs in one
/// line, out[3] = near pairs seen.
/// Launch geometry: scatter_blocks workgroups of 256 threads, one sample per thread (spread over the chip: one CU alone keeps
/// ~94 reads in flight, and these are 49 152 scattered ones); a workgroup's counts are reduced in LDS and leave as four PLAIN
/// stores into its own slot of `partials`, column_scatter_decide (one wavefront) adds the slots up into `out[0..3]`: no zero-fill
/// in front, no global atomics.  (Tried first: an atomic per wavefront on four neighbouring words -- 4 096 atomics at ~88 per
/// microsecond = a 46 us kernel; a memset + 64 atomics: 25 us per plan-less call; everything in one workgroup: 60 us.)
constexpr int scatter_blocks = 64;
constexpr int scatter_samples = scatter_blocks * 256;
constexpr int scatter_scratch_words = 4 + 4 * scatter_blocks;  ///< out[4] followed by partials[scatter_blocks][4]
template <typename index_t>
__global__ void __launch_bounds__(256)
column_scatter_sample(const index_t* __restrict__ indices, const long long nnz, const long long stride, const long long far,
                      const unsigned int shift, const unsigned int parts, const unsigned int line_shift,
                      unsigned int* __restrict__ partials) {
  __shared__ unsigned int counts[4];
  if (threadIdx.x < 4) counts[threadIdx.x] = 0u;
  __syncthreads();
  const int k = blockIdx.x * 256 + static_cast<int>(threadIdx.x);
  const long long i = static_cast<long long>(k) * stride;
  // (the distance varies from sample to sample: with equal row lengths a fixed one would always meet the same rank inside the
  // other row, and the k-th smallest of d random columns sits in much the same place in every row)
  const long long j = i + far + static_cast<long long>((static_cast<unsigned int>(k) * 2654435761u) >> 20);
  const bool near_ok = i + 1 < nnz, far_ok = near_ok && j < nnz;
  const unsigned int c0 = near_ok ? static_cast<unsigned int>(indices[i]) : 0u;
  const unsigned int c1 = near_ok ? static_cast<unsigned int>(indices[i + 1]) : 0u;
  const unsigned int c2 = far_ok ? static_cast<unsigned int>(indices[j]) : 0u;
  unsigned int a = c0 >> shift, b2 = c2 >> shift;
  a = a < parts - 1 ? a : parts - 1;
  b2 = b2 < parts - 1 ? b2 : parts - 1;
  const unsigned long long m0 = __builtin_amdgcn_ballot_w64(far_ok && a == b2), m1 = __builtin_amdgcn_ballot_w64(far_ok),
                           m2 = __builtin_amdgcn_ballot_w64(near_ok && (c0 >> line_shift) == (c1 >> line_shift)),
                           m3 = __builtin_amdgcn_ballot_w64(near_ok);
  if (wave::lane() == 0) {
    atomicAdd(&counts[0], static_cast<unsigned int>(__builtin_popcountll(m0)));
    atomicAdd(&counts[1], static_cast<unsigned int>(__builtin_popcountll(m1)));
    atomicAdd(&counts[2], static_cast<unsigned int>(__builtin_popcountll(m2)));
    atomicAdd(&counts[3], static_cast<unsigned int>(__builtin_popcountll(m3)));
  }
  __syncthreads();
  if (threadIdx.x < 4) partials[blockIdx.x * 4 + threadIdx.x] = counts[threadIdx.x];
}

/// out[c] = sum over the scatter_blocks slots of partials[slot][c] (one wavefront; scatter_blocks == 64 lanes).
__global__ void __launch_bounds__(64) column_scatter_decide(const unsigned int* __restrict__ partials, unsigned int* __restrict__ out) {
  static_assert(scatter_blocks == 64, "one lane per slot");
  const int l = threadIdx.x;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    unsigned int v = partials[l * 4 + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (l == 0) out[c] = v;
  }
}

/// (the same under the symbol of SpMV-plan handles: profile attribution only, as merge_path_spmv_fused_planned)
template <int TPB, int IPT, int PHASES, bool VEC, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
merge_path_spmv_fused_phased_planned(const coord_t* __restrict__ coords, const int rows, const int nnz,
                                     const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                                     const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                                     int* __restrict__ carry_row, type_t* __restrict__ carry_val, const detail::phase_args phase) {
  merge_path_spmv_tile_to<TPB, IPT, true, detail::policy::phased(PHASES), VEC, false, true>(
      coords, rows, nnz, csr_row_end<offset_t>{offsets}, indices, values, x, plain_store<type_t>{y}, carry_row, carry_val, nullptr,
      phase);
}

/**
 * work_oriented: every workgroup owns an even, contiguous share of the merge tiles (hence of rows +
 * nonzeros; the launcher sizes the shares at 1-4 tiles -- eight shares per resident workgroup --
 * instead of one long share per workgroup of an occupancy-sized grid: launch.hxx) and walks it tile
 * after tile, carrying the open row's partial sum in a register -- one carry-out per share instead
 * of one per tile.
 * (Even-share semantics of schedule::setup<work_oriented>, reference work_oriented.hxx:79-91,
 * at merge-tile granularity.)
 */
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t,
          bool MASK = false>
__global__ void __launch_bounds__(TPB)
work_oriented_spmv_fused(const coord_t* __restrict__ coords, const int num_merge_tiles, const int tiles_per_group,
                         const int rows, const int nnz, const offset_t* __restrict__ offsets,
                         const index_t* __restrict__ indices, const type_t* __restrict__ values,
                         const type_t* __restrict__ x, type_t* __restrict__ y, int* __restrict__ carry_row,
                         type_t* __restrict__ carry_val) {
  using engine_t = merge_tile_engine<TPB, IPT, PAD, NT, VEC, index_t, offset_t, type_t, MASK>;
  __shared__ typename engine_t::storage_t s_engine;
  __shared__ offset_t s_re[MASK ? 1 : TPB * IPT + IPT + 1];

  const int tid = threadIdx.x;
  const int g = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int t_begin = g * tiles_per_group;
  int t_end = t_begin + tiles_per_group;
  t_end = t_end < num_merge_tiles ? t_end : num_merge_tiles;
  type_t carry = type_t(0);
  int open_row = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const coord_t c0 = coords[t];
    const coord_t c1 = coords[t + 1];
    const int row0 = static_cast<int>(c0.x);
    const int nz0 = static_cast<int>(c0.y);
    const int nrows = static_cast<int>(c1.x) - row0;
    const int natoms = static_cast<int>(c1.y) - nz0;
    if constexpr (MASK) {  // row ends -> marks of the engine's bit mask (the barrier also fences the previous tile)
      engine_t::clear_marks(s_engine);
      __syncthreads();
    } else {
      for (int i = tid; i < nrows + IPT; i += TPB) {
        int r = row0 + i;
        r = r < rows - 1 ? r : rows - 1;
        s_re[i] = offsets[r + 1];
      }
    }
    // carry-in of the share's FIRST tile belongs to an earlier workgroup: it goes through the fix-up
    carry = engine_t::run(s_engine, s_re, row0, nz0, nrows, natoms, nnz, indices, values, x, y, carry, [&]() {
      if constexpr (MASK) {
        for (int i = tid; i < nrows; i += TPB)
          engine_t::mark_row_end(s_engine, i, static_cast<int>(offsets[row0 + i + 1]), nz0);
      }
    });
    open_row = row0 + nrows;
  }
  if (tid == 0 && t_begin < t_end) {
    carry_row[g] = open_row;
    carry_val[g] = carry;
  }
}

/**
 * group_mapped: workgroup b owns the TPB consecutive rows [b * TPB, (b + 1) * TPB) and sweeps the
 * concatenation of their nonzeros cooperatively (semantics of schedule::setup<group_mapped, TPB, TPB>,
 * reference group_mapped.hxx:104-192), here as a sequence of merge tiles over the workgroup's OWN
 * row offsets, which sit in LDS once: no global search, no plan, no cross-workgroup carry (a
 * workgroup owns whole rows), no atomics, no zero-fill of y.
 */
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t,
          bool MASK = false>
__global__ void __launch_bounds__(TPB)
group_mapped_spmv_fused(const int rows, const int nnz, const offset_t* __restrict__ offsets,
                        const index_t* __restrict__ indices, const type_t* __restrict__ values,
                        const type_t* __restrict__ x, type_t* __restrict__ y) {
  using engine_t = merge_tile_engine<TPB, IPT, PAD, NT, VEC, index_t, offset_t, type_t, MASK>;
  constexpr int TILE = TPB * IPT;
  __shared__ typename engine_t::storage_t s_engine;
  __shared__ offset_t s_off[TPB + 1 + IPT];  // offsets of the group's rows (+ clamped slack)

  const int tid = threadIdx.x;
  const int group_row0 = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x)) * TPB;
  int group_rows = rows - group_row0;
  group_rows = group_rows < TPB ? group_rows : TPB;
  for (int i = tid; i < TPB + 1 + IPT; i += TPB) {
    int r = group_row0 + i;
    r = r < rows ? r : rows;
    s_off[i] = offsets[r];
  }
  __syncthreads();
  const int nz_begin = s_off[0];
  const int group_atoms = s_off[group_rows] - nz_begin;
  const int total = group_rows + group_atoms;
  const offset_t* row_end = s_off + 1;  // row_end[i] = end of row group_row0 + i

  type_t carry = type_t(0);
  int tx0 = 0, ty0 = 0;  // merge-path position (rows, atoms consumed) at the start of the tile
  for (int d0 = 0; d0 < total; d0 += TILE) {
    // end of this tile on the group's merge path: split at min(d0 + TILE, total), every lane
    // does the same <= log2(TPB) LDS probes (broadcast reads)
    const int d1 = d0 + TILE < total ? d0 + TILE : total;
    int lo = d1 - group_atoms > 0 ? d1 - group_atoms : 0;
    int count = (d1 < group_rows ? d1 : group_rows) - lo;
    while (count > 0) {
      const int half = count >> 1;
      const int mid = lo + half;
      if (row_end[mid] <= nz_begin + (d1 - mid - 1)) {
        lo = mid + 1;
        count -= half + 1;
      } else {
        count = half;
      }
    }
    const int tx1 = lo < group_rows ? lo : group_rows;
    const int ty1 = d1 - lo;
    if constexpr (MASK) {  // row ends of the tile (already in LDS) -> marks of the engine's bit mask
      engine_t::clear_marks(s_engine);
      __syncthreads();
      for (int i = tid; i < tx1 - tx0; i += TPB)
        engine_t::mark_row_end(s_engine, i, static_cast<int>(row_end[tx0 + i]), nz_begin + ty0);
    }
    carry = engine_t::run(s_engine, row_end + tx0, group_row0 + tx0, nz_begin + ty0, tx1 - tx0, ty1 - ty0, nnz,
                          indices, values, x, y, carry);
    tx0 = tx1;
    ty0 = ty1;
  }
}

/// y[row] += sum of the carry-outs of the run of merge tiles that ended inside `row`.
/// Latency-bound (a few thousand lanes, each a chain of dependent loads): the neighbours' rows, the two first
/// values of the run and y[row] are requested together, so the common case (runs of one or two tiles) costs two
/// dependent memory round trips instead of four or five.
template <typename type_t, typename store_t>
__device__ __forceinline__ void merge_path_spmv_fixup_body(const int* __restrict__ carry_row, const type_t* __restrict__ carry_val,
                                                           const int num_merge_tiles, const int rows, const store_t out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_merge_tiles) return;
  const bool has_next = i + 1 < num_merge_tiles;
  const int r = carry_row[i];
  const int before = i > 0 ? carry_row[i - 1] : -1;
  const int after = has_next ? carry_row[i + 1] : -1;
  type_t s = carry_val[i];
  const type_t second = has_next ? carry_val[i + 1] : type_t(0);
  if (r >= rows || before == r) return;  // nothing open / not the first tile of the run
  const type_t old = out.load(r);
  if (after == r) {
    s += second;
    for (int j = i + 2; j < num_merge_tiles && carry_row[j] == r; ++j) s += carry_val[j];
  }
  out(r, old + s);
}

template <typename type_t>
__global__ void merge_path_spmv_fixup(const int* __restrict__ carry_row, const type_t* __restrict__ carry_val,
                                      int num_merge_tiles, int rows, type_t* __restrict__ y) {
  merge_path_spmv_fixup_body(carry_row, carry_val, num_merge_tiles, rows, plain_store<type_t>{y});
}

/// The fix-up of merge_path_spmv_fused_fanout: the completed rows are (re)written to every destination.
template <typename type_t>
__global__ void merge_path_spmv_fixup_fanout(const int* __restrict__ carry_row, const type_t* __restrict__ carry_val,
                                             int num_merge_tiles, int rows, type_t* __restrict__ y,
                                             const peer_fanout<type_t> peers) {
  merge_path_spmv_fixup_body(carry_row, carry_val, num_merge_tiles, rows, fanout_store<type_t>{y, peers});
}

}  // namespace kernels
}  // namespace loops
