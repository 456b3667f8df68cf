/**
 * @file panel_binned.hxx
 * @brief Panel-binned SpMV: the layout for matrices whose x is (much) larger than the 4 MB per-XCD L2.
 *
 * Why.  CSR SpMV issues one scattered 4-byte gather of x per nonzero, and on MI355X that gather -- not HBM -- sets the
 * time: a CU keeps ~95 reads in flight, each gather holds a slot for its whole round trip (L2 hit ~220 clks, Infinity
 * Cache ~950), so the chip serves 265 G gathers/s at best from L2 and ~55 G/s beyond it.  This layout removes
 * the gathers from the memory system altogether: x is read in PANELS of W consecutive columns that fit the 160 KB LDS of a CU, and a
 * nonzero's x value comes out of LDS (ds_read: ~9 random reads per clock and CU, twenty times the L2 path).
 *
 * Layout (a re-ordered COPY of the matrix, built once on the device, O(nnz)): nonzeros sorted by (panel p = col / W,
 * sub-band s = row / Hw, then their CSR order) -- the "A order" -- with `val`, `col16 = col - p W` (6 bytes per nonzero) and,
 * per group of 4 items, `dst4` = where the group's products go in the "B order": the same segments sorted by (sub-band,
 * panel), holding `row16 = row - s Hw` (2 bytes per nonzero).  Every (p, s) segment starts at a multiple of 4 items in both
 * orders (padding: val 0, row16 0xFFFF), so a group of 4 never straddles segments.
 *
 * y = A x in two streaming kernels:
 *   A  panel_products   one workgroup per chunk of ONE panel: x[p W .. (p + 1) W) -> LDS (coalesced), then
 *                       prod[dst4[i / 4] ..] = val[i ..] * xs[col16[i ..]], 16-byte loads and stores: the products leave in
 *                       16-byte groups, consecutive groups of a segment to consecutive addresses (7 B read + 4 B written per
 *                       nonzero, no gather leaves the CU);
 *   B  panel_reduce     one WORKGROUP per sub-band of Hw rows: the sub-band's products are ONE contiguous run of the B order,
 *                       cut at plan time into WINDOWS of at most 64 lanes x 4 items (make_windows: a piece of one segment,
 *                       or -- "packed" -- consecutive segments / segment tails of <= 64 items each; 128 for 8-byte values);
 *                       each of the 4 wavefronts takes a quarter of the windows, 8 in flight (16 + 8 bytes per lane).  Inside
 *                       a one-segment window rows are sorted: runs of equal rows are summed with the wave64 segmented prefix
 *                       sum and the run ends update the wavefront's OWN LDS accumulators with plain read-modify-writes.
 *                       Packed windows use LDS atomics (ds_add_f32 retires 0.33 lanes per clock and CU on gfx950, ds_add_f64
 *                       3-8), or -- where small segments are the rule (4-byte values) -- the run-combining path with
 *                       compare-and-swap final updates and one wavefront per sub-band.  The partial vectors are added in
 *                       wavefront order and the Hw rows of y are stored coalesced (4 B + 2 B read per nonzero).
 *                       Reproducible: no order depends on timing.
 * HBM traffic 17 B per nonzero instead of 8 B + a gather; y needs no zero-fill; no global atomics.
 *
 * When it pays: x beyond the L2 and rows spread over many panels (C3- / C5-like inputs).  Sub-bands are the unit of
 * parallelism of kernel B, so a handful of rows holding most nonzeros ("extreme skew") serialises it -- the SpMV plan
 * (loops_spmv_plan_*) adopts this layout only when it measures faster.
 * No reference counterpart (the reference leaves the gather to the cache).
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <loops/kernels/merge_path_spmv.hxx>

#ifndef LOOPS_PANEL_COMPACT_U
#define LOOPS_PANEL_COMPACT_U 4  // vectors of 4 items per lane and step of the compact kernel A (tuning knob of variant builds)
#endif
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

/// Device arrays of a panel-binned matrix (all owned by the caller / the plan).
template <typename type_t>
struct panel_binned_view {
  int rows, cols, nnz;
  int W, Hw, P, S;            ///< columns per panel, rows per sub-band, number of panels / sub-bands
  int padded;                 ///< A order: items incl. padding (multiple of 4)
  int padded_b;               ///< B order: slots incl. padding (multiple of 4); == padded unless `compact`
  int compact;                ///< 1: kernel A pre-sums runs of equal (row, panel) and the B order holds one slot per RUN
  int awin;                   ///< compact: items of one wavefront step of kernel A (64 lanes x 4): no run crosses a multiple of it
  type_t* val;                ///< [padded] A order
  unsigned short* col16;      ///< [padded] A order: column inside the panel (compact: | 0x8000 on the item that ENDS a run)
  int* dst4;                  ///< [padded / 4] A order: B-order slot of the group's first output (compact: bit 31 = the group holds padding)
  unsigned short* row16;      ///< [padded_b] B order: row inside the sub-band; 0xFFFF = padding
  int* perm;                  ///< [padded] A order: CSR position of the item (-1 = padding): value refresh
  int* segb;                  ///< [S * P + 1] B order: segment (s, p) = items [segb[s * P + p], segb[s * P + p + 1])
  int* bstart;                ///< [S + 1] B order: sub-band s owns items [bstart[s], bstart[s + 1]) (= segb[s * P])
  int* chunks;                ///< [3 * num_chunks] {panel, begin, end} work list of kernel A (A-order positions)
  int num_chunks;
  type_t* prod;               ///< [padded_b] B order: products / run sums scratch (kernel A -> kernel B)
  int* wins;                  ///< [2 * panel_window_capacity] kernel B's windows: {first item, items | packed << 16} (B order)
  int* wstart;                ///< [S + 1] sub-band s owns windows [wstart[s], wstart[s + 1])
};

/// Upper bound of the number of kernel-B windows: full windows of 256 items plus at most one remainder per segment.
inline std::size_t panel_window_capacity(int padded, long long segments) {
  return static_cast<std::size_t>(padded / 256) + static_cast<std::size_t>(segments) + 1;
}

/// Segments of at most this many items share windows ("packed"): 64 for 4-byte values, whose packed windows cost an LDS
/// float atomic (0.33 lanes per clock and CU) or a compare-and-swap per item; 128 for 8-byte values (ds_add_f64: 3-8 lanes
/// per clock and CU; C5 shard in f64: kernel B 378 -> 211 us; 256 loses on C2, whose hub rows then meet in one word).
template <typename type_t>
constexpr int panel_pack_items() { return sizeof(type_t) == 4 ? 64 : 128; }

/// Panel widths kernel A is compiled for: 64 KB of x per workgroup (512 threads, two workgroups per CU) and 128 KB (1024
/// threads, one workgroup per CU).  `value` = the narrow one.
template <typename type_t>
struct panel_width {
  static constexpr int value = 65536 / static_cast<int>(sizeof(type_t));
  static constexpr int wide = 2 * value;
};

/// Panel width for (rows, cols, nnz): 128 KB of x per panel whenever the matrix has at least four narrow panels -- half as
/// many panels means (panel, sub-band) segments twice as long (fuller windows in kernel B, longer store runs in kernel A) and
/// half as many of them; measured better or equal on every input with x >= 4 MB (C2 86 -> 73 us, C5 shard 345 -> 272,
/// C3 stand-ins 0.83 -> 0.73 / 1.21 -> 0.89 / 0.73 -> 0.62 ms, profiles/r03_panel_binned.json).
template <typename type_t>
inline int panel_columns(int /*rows*/, int cols, int /*nnz*/) {
  const int w = panel_width<type_t>::value;
  const long long P = cols > 0 ? (static_cast<long long>(cols) + w - 1) / w : 1;
  return P >= 4 ? panel_width<type_t>::wide : w;
}

namespace panel {

constexpr unsigned short pad_row = 0xFFFFu;

/// key[i] = (segment of nonzero i) << 32 | i and rc[i] = (row inside the sub-band) << 16 | (column inside the panel): everything
/// the later passes need of a nonzero besides its value, so that they gather ONE word per item.  Lane per IPT consecutive
/// nonzeros: one search for the row of the first, then a walk along the offsets .  No counting here:
/// the per-segment counts come from the SORTED keys (segment_starts) -- one global atomic per item was 0.9 of this kernel's
/// 0.92 ms on C2 (scattered atomics into L2 retire at ~18 G/s).
template <int IPT, typename index_t, typename offset_t>
__global__ void __launch_bounds__(256)
make_keys(const offset_t* __restrict__ offsets, const index_t* __restrict__ indices, const int rows, const int nnz,
          const int W, const int Hw, const int S, const int cols, unsigned long long* __restrict__ keys, unsigned int* __restrict__ rc,
          int* __restrict__ bad) {
  const long long base_ll = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * IPT;
  if (base_ll >= nnz) return;
  const int base = static_cast<int>(base_ll);
  int row = 0, count = rows;
  while (count > 0) {
    const int half = count >> 1;
    const int mid = row + half;
    if (offsets[mid + 1] <= base) {
      row = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  offset_t row_end = offsets[row + 1];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int i = base + j;
    if (i >= nnz) break;
    while (i >= row_end) row_end = offsets[++row + 1];  // skip empty rows
    unsigned int col = static_cast<unsigned int>(indices[i]);
    if (col >= static_cast<unsigned int>(cols)) {  // (also a negative index) flagged, then clamped so that nothing is written out of bounds
      *bad = 1;
      col = 0;
    }
    const unsigned int p = col / static_cast<unsigned int>(W), sb = static_cast<unsigned int>(row) / static_cast<unsigned int>(Hw);
    const unsigned int seg = p * static_cast<unsigned int>(S) + sb;
    keys[i] = (static_cast<unsigned long long>(seg) << 32) | static_cast<unsigned int>(i);
    rc[i] = ((static_cast<unsigned int>(row) - sb * static_cast<unsigned int>(Hw)) << 16) | (col - p * static_cast<unsigned int>(W));
  }
}

/// seg_start[g] = the first sorted position whose segment is >= g (g <= n: seg_start[n] = nnz): the segments' extents read off
/// the sorted keys by n + 1 independent binary searches.
__global__ void __launch_bounds__(256)
segment_starts(const unsigned long long* __restrict__ sorted, const int nnz, const int n, int* __restrict__ seg_start) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n) return;
  const unsigned long long want = static_cast<unsigned long long>(g) << 32;
  int lo = 0, count = nnz;
  while (count > 0) {
    const int half = count >> 1;
    if (sorted[lo + half] < want) {
      lo += half + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  seg_start[g] = lo;
}

/// padded[g] = the items of segment g (seg_start[g + 1] - seg_start[g]) rounded up to a multiple of 4 (g < n), padded[n] = 0.
__global__ void __launch_bounds__(256) pad_segment_sizes(const int* __restrict__ seg_start, const int n, int* __restrict__ padded) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= n) padded[g] = g < n ? (seg_start[g + 1] - seg_start[g] + 3) & ~3 : 0;
}

/// paddedT[s * P + p] = padded[p * S + s]; paddedT[S * P] = 0 (the segments in B order).
__global__ void __launch_bounds__(256)
transpose_counts(const int* __restrict__ padded, const int P, const int S, int* __restrict__ paddedT) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long n = static_cast<long long>(P) * S;
  if (t > n) return;
  if (t == n) { paddedT[t] = 0; return; }
  const int s = static_cast<int>(t / P), p = static_cast<int>(t - static_cast<long long>(s) * P);
  paddedT[t] = padded[static_cast<long long>(p) * S + s];
}

/// Compact layout: items of one wavefront step of kernel A (64 lanes x 4 consecutive items): a run of equal rows never
/// crosses a multiple of a_window, so the pre-summing needs no carry between wavefronts.
constexpr int a_window = wave::size * 4;
constexpr unsigned short run_end_bit = 0x8000u;  // of col16: "this item ENDS a run" (panels of at most 32768 columns)
/// ends[j] = 1 when sorted item j is the LAST of its run -- the next item lies in another segment or another row, or j is the
/// last item of its window of kernel A (a_window items; A positions are counted from the panel's start: chunks begin at
/// multiples of 4096 items of it) -- else 0; ends[nnz] = 0.  The exclusive scan of `ends` numbers the runs.
__global__ void __launch_bounds__(256)
mark_run_ends(const unsigned long long* __restrict__ sorted, const int* __restrict__ seg_start, const int* __restrict__ seg_dest,
              const unsigned int* __restrict__ rc, const int nnz, const int S, int* __restrict__ ends) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > nnz) return;
  if (j == nnz) { ends[j] = 0; return; }
  const unsigned long long key = sorted[j];
  const int g = static_cast<int>(key >> 32), i = static_cast<int>(key & 0xffffffffull);
  bool last = j + 1 == nnz;
  if (!last) {
    const unsigned long long nk = sorted[j + 1];
    // (inside one segment the row inside the sub-band identifies the row)
    last = static_cast<int>(nk >> 32) != g || (rc[static_cast<int>(nk & 0xffffffffull)] >> 16) != (rc[i] >> 16);
  }
  if (!last) {
    const int p = g / S;
    const int a = seg_dest[g] + (j - seg_start[g]) - seg_dest[static_cast<long long>(p) * S];
    last = (a & (a_window - 1)) == a_window - 1;
  }
  ends[j] = last ? 1 : 0;
}

/// padded_b[g] = the runs of segment g rounded up to a multiple of 4 (g < n), padded_b[n] = 0.
__global__ void __launch_bounds__(256)
pad_run_counts(const int* __restrict__ seg_start, const int* __restrict__ run_index, const int n, int* __restrict__ padded_b) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= n) padded_b[g] = g < n ? (run_index[seg_start[g + 1]] - run_index[seg_start[g]] + 3) & ~3 : 0;
}

/// Sorted position j -> its position in both orders; fills val / col16 / perm / dst4 (A order) and row16 (B order).
/// COMPACT: the B order holds one slot per run (`ends` / `run_index` of mark_run_ends): col16 carries the run-end flag in
/// bit 15, dst4 the slot of the group's first run end and, in bit 31, "this group holds padding" (the items behind the
/// group's last run end are padding then: a segment's last real item always ends a run).
template <bool COMPACT, typename type_t>
__global__ void __launch_bounds__(256)
place(const unsigned long long* __restrict__ sorted, const int* __restrict__ seg_start, const int* __restrict__ seg_dest,
      const int* __restrict__ seg_dest_b, const unsigned int* __restrict__ rc, const int* __restrict__ ends,
      const int* __restrict__ run_index, const type_t* __restrict__ values, const int nnz, const int P, const int S,
      type_t* __restrict__ val, unsigned short* __restrict__ col16, int* __restrict__ dst4, unsigned short* __restrict__ row16,
      int* __restrict__ perm) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nnz) return;
  const unsigned long long key = sorted[j];
  const int g = static_cast<int>(key >> 32), i = static_cast<int>(key & 0xffffffffull);
  const int first = seg_start[g];
  const int within = j - first;
  const int p = g / S, s = g - p * S;
  const int a = seg_dest[g] + within;
  const int bseg = seg_dest_b[static_cast<long long>(s) * P + p];
  val[a] = values[i];
  perm[a] = i;
  const unsigned int packed = rc[i];
  const unsigned int col = packed & 0xFFFFu;
  const unsigned short row = static_cast<unsigned short>(packed >> 16);
  if constexpr (COMPACT) {
    const int end = ends[j];
    const int slot = bseg + (run_index[j] - run_index[first]);  // of the run item j belongs to
    col16[a] = static_cast<unsigned short>(col | (end ? run_end_bit : 0u));
    if (end) row16[slot] = row;
    if ((within & 3) == 0) {
      const bool has_padding = within + 4 > seg_start[g + 1] - first;
      dst4[a >> 2] = slot | (has_padding ? static_cast<int>(0x80000000u) : 0);
    }
  } else {
    col16[a] = static_cast<unsigned short>(col);
    row16[bseg + within] = row;
    if ((within & 3) == 0) dst4[a >> 2] = bseg + within;  // (the real items of a segment are a prefix of it: every group starts with one)
  }
}

/// bstart[s] = seg_dest_b[s * P] (s <= S: bstart[S] = total); panel_start[p] = seg_dest[p * S] (p <= P).
__global__ void __launch_bounds__(256)
extract_starts(const int* __restrict__ seg_dest, const int* __restrict__ seg_dest_b, const int P, const int S,
               int* __restrict__ bstart, int* __restrict__ panel_start) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= S) bstart[t] = seg_dest_b[static_cast<long long>(t) * P];
  if (t <= P) panel_start[t] = seg_dest[static_cast<long long>(t) * S];
}

/// Kernel B's work list, built once: the windows of sub-band s (one thread per sub-band walks its P segments).  A window is
/// at most 256 consecutive items of the B order: either a piece of ONE segment (rows sorted: run-combining path) or
/// "packed" -- consecutive segments / segment tails of at most `pack` items each (sorted only piecewise).  FILL = false
/// counts (wstart[s] = number of windows), FILL = true writes them at wstart[s] (after the exclusive scan).
/// Why a table: the walk needs one dependent wave-uniform load per segment, and a sub-band of a matrix with column locality
/// has hundreds of segments of a few items -- done inside kernel B it was most of that kernel's time on such matrices.
template <bool FILL>
__global__ void __launch_bounds__(256)
make_windows(const int* __restrict__ segb, const int P, const int S, const int pack, int* __restrict__ wstart, int* __restrict__ wins) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int* seg = segb + static_cast<long long>(s) * P;
  int* out = FILL ? wins + 2ll * wstart[s] : nullptr;
  int count = 0, ob = 0, oe = 0;                        // [ob, oe): the open packed window
  auto emit = [&](int b, int e, int packed) {
    if constexpr (FILL) { out[2 * count] = b; out[2 * count + 1] = (e - b) | (packed << 16); }
    ++count;
  };
  int e = seg[0];
  for (int p = 0; p < P; ++p) {
    int b = e;
    e = seg[p + 1];
    if (e - b > pack) {                                 // a large segment: sorted windows, its short tail opens a packed one
      if (oe > ob) emit(ob, oe, 1);
      ob = oe = 0;
      while (e - b > 256) { emit(b, b + 256, 0); b += 256; }
      if (e - b > pack) { emit(b, e, 0); continue; }
    }
    if (e == b) continue;
    if (oe > ob && e - ob <= 256) { oe = e; continue; }  // joins the open window (B order is contiguous across segments)
    if (oe > ob) emit(ob, oe, 1);
    ob = b;
    oe = e;
  }
  if (oe > ob) emit(ob, oe, 1);
  if constexpr (!FILL) wstart[s] = count;
}

template <typename type_t>
__global__ void __launch_bounds__(256)
refresh_values(const int* __restrict__ perm, const type_t* __restrict__ values, const int padded, type_t* __restrict__ val) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < padded) {
    const int i = perm[j];
    val[j] = i >= 0 ? values[i] : type_t(0);
  }
}

/// x[base .. base + n) -> xs (16-byte loads, every thread of the workgroup); the caller synchronises.
template <int TPB, typename type_t>
__device__ __forceinline__ void load_x_panel(type_t* __restrict__ xs, const type_t* __restrict__ x, const long long base, const int n,
                                             const int tid) {
  constexpr int VW = 16 / static_cast<int>(sizeof(type_t));  // elements per 16-byte vector
  using vec_t = type_t __attribute__((ext_vector_type(VW)));
  using vec_ld_t = type_t __attribute__((ext_vector_type(VW), aligned(sizeof(type_t))));
  for (int j = tid * VW; j < n; j += TPB * VW) {
    if (j + VW <= n) {
      const vec_t v = *reinterpret_cast<const vec_ld_t*>(x + base + j);
#pragma unroll
      for (int e = 0; e < VW; ++e) xs[j + e] = v[e];
    } else {
      for (int e = 0; j + e < n; ++e) xs[j + e] = x[base + j + e];
    }
  }
}

/// 16 bytes of a group's products to `to`.
/// (`to` is aligned to the element only -- the compact B order starts a group's slots wherever its runs begin -- so the vector
/// types are declared element-aligned: one global_store_dwordx4 either way, without the undefined behaviour of a misaligned
/// naturally-aligned vector store.)
template <typename type_t>
__device__ __forceinline__ void store4(type_t* __restrict__ to, const type_t a, const type_t b, const type_t c, const type_t d) {
  if constexpr (sizeof(type_t) == 4) {
    using o4 = type_t __attribute__((ext_vector_type(4)));
    using o4_st = type_t __attribute__((ext_vector_type(4), aligned(sizeof(type_t))));
    *reinterpret_cast<o4_st*>(to) = o4{a, b, c, d};
  } else {
    using o2 = type_t __attribute__((ext_vector_type(2)));
    using o2_st = type_t __attribute__((ext_vector_type(2), aligned(sizeof(type_t))));
    *reinterpret_cast<o2_st*>(to) = o2{a, b};
    *reinterpret_cast<o2_st*>(to + 2) = o2{c, d};
  }
}

/// Kernel A: products of the chunks [g C / G, (g + 1) C / G) of workgroup g of G -- one chunk per workgroup by default (G = C;
/// see panel_products_grid) -- x panel in LDS, (re)loaded only when the panel changes:
///   prod[dst4[i / 4] ..] = val[i ..] * xs[col16[i ..]], one 16-byte store per group of 4 items.
/// 4 items per lane and vector: 16 B of values (f32; 2 x 16 B for f64), 8 B of columns, 4 B of destination; U vectors per
/// step.  Software-pipelined: the loads of step k + 1 are issued BEFORE the stores of step k (gfx9 counts loads and
/// stores in one in-order counter, vmcnt).  Loads are branch-free (a lane behind the chunk's end re-reads the chunk's first
/// vector and stores nothing).  Measured at the chip's copy rate (C5 shard: 870 MB in 0.2 ms).
template <int TPB, int W, int U, bool NT, typename type_t>
__global__ void __launch_bounds__(TPB)
panel_products(const int* __restrict__ chunks, const int num_chunks, const type_t* __restrict__ val,
               const unsigned short* __restrict__ col16, const int* __restrict__ dst4, const type_t* __restrict__ x, const int cols,
               type_t* __restrict__ prod) {
  __shared__ type_t xs[W];
  using u16x4 = unsigned short __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const int g = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int c_lo = static_cast<int>(static_cast<long long>(num_chunks) * g / static_cast<int>(gridDim.x));
  const int c_hi = static_cast<int>(static_cast<long long>(num_chunks) * (g + 1) / static_cast<int>(gridDim.x));
  constexpr int STEP = TPB * 4 * U;
  struct batch_t {
    type_t v[U][4];
    u16x4 c[U];
    int dst[U];
  };
  int begin = 0, end = 0;  // the current chunk (workgroup-uniform)
  auto load = [&](batch_t& t, const int i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = i0 + (u * TPB + tid) * 4;
      i = i < end ? i : begin;
      detail::load4<type_t, NT>(val + i, t.v[u]);
      if constexpr (NT) {
        t.c[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(col16 + i));
        t.dst[u] = __builtin_nontemporal_load(dst4 + (i >> 2));
      } else {
        t.c[u] = *reinterpret_cast<const u16x4*>(col16 + i);
        t.dst[u] = dst4[i >> 2];
      }
    }
  };
  auto consume = [&](const batch_t& t, const int i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      type_t out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = t.v[u][e] * xs[t.c[u][e]];
      if (i0 + (u * TPB + tid) * 4 < end) store4(prod + t.dst[u], out[0], out[1], out[2], out[3]);
    }
  };
  int panel_in_lds = -1;
  batch_t a, b;
  for (int c = c_lo; c < c_hi; ++c) {
    const int p = chunks[3 * c];
    begin = chunks[3 * c + 1];
    end = chunks[3 * c + 2];
    if (begin >= end) continue;  // (workgroup-uniform)
    int i0 = begin;  // (wave-uniform loop control)
    load(a, i0);     // the chunk's first stream loads are in flight while the x panel is fetched
    if (p != panel_in_lds) {
      const long long base = static_cast<long long>(p) * W;
      if (panel_in_lds >= 0) __syncthreads();  // every wavefront is done with the previous panel
      load_x_panel<TPB>(xs, x, base, cols - base < W ? static_cast<int>(cols - base) : W, tid);
      __syncthreads();
      panel_in_lds = p;
    }
    for (;;) {
      if (i0 + STEP < end) load(b, i0 + STEP);
      consume(a, i0);
      i0 += STEP;
      if (i0 >= end) break;
      if (i0 + STEP < end) load(a, i0 + STEP);
      consume(b, i0);
      i0 += STEP;
      if (i0 >= end) break;
    }
  }
}

/// Kernel A of the COMPACT layout: the same stream, but runs of equal (row, panel) -- cut at plan time so that none crosses
/// a wavefront step (a_window = 64 lanes x 4 items) -- are summed before they leave the CU: inside the lane, then across
/// lanes with the wave64 segmented prefix sum, and the item that ENDS a run (bit 15 of its col16) stores the run's sum to
/// the run's slot of the B order (dst4 of the lane's group + the run ends before it in the group).  On a matrix with column
/// locality (a row's nonzeros in one or two panels: web graphs, FEM bands) a row's 20-30 products shrink to 1-2 slots: 4 B
/// written and 6 B read back per RUN instead of per nonzero.  Items behind the last run end of a group flagged "holds
/// padding" (bit 31 of dst4) are padding and contribute nothing, whatever x holds.
/// Summation order: fixed by the layout (lane-sequential, then the scan's tree) -- reproducible.
/// Where the time goes (band C3 stand-in, 1.41 GB read + 0.08 GB written; diagnostic builds): the streams alone 262 us
/// (5.4 TB/s), + the LDS gathers and sums 323, + the stores 351-362.  Three formulations of the sums and stores were measured
/// within 4 % of each other (4 items per lane with per-lane `if`s around the stores: this one, the fastest; 8 items per
/// lane, run-end flags in the dst4 word, buffer stores that drop lanes by an out-of-range offset, mask arithmetic instead
/// of selects: 362-365): 26 vector instructions per item either way, the SIMDs 26 % busy -- what is left is that a
/// wavefront's loads are not in flight while it sums (16 wavefronts per CU: the 128 KB x panel).
template <int TPB, int W, int U, bool NT, typename type_t>
__global__ void __launch_bounds__(TPB)
panel_products_compact(const int* __restrict__ chunks, const int num_chunks, const type_t* __restrict__ val,
                       const unsigned short* __restrict__ col16, const int* __restrict__ dst4, const type_t* __restrict__ x,
                       const int cols, type_t* __restrict__ prod) {
  __shared__ type_t xs[W];
  static_assert(W <= 32768, "panel_products_compact: bit 15 of col16 is the run-end flag");
  using u16x4 = unsigned short __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const int g = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int c_lo = static_cast<int>(static_cast<long long>(num_chunks) * g / static_cast<int>(gridDim.x));
  const int c_hi = static_cast<int>(static_cast<long long>(num_chunks) * (g + 1) / static_cast<int>(gridDim.x));
  constexpr int STEP = TPB * 4 * U;
  struct batch_t {
    type_t v[U][4];
    u16x4 c[U];
    int dst[U];
  };
  int begin = 0, end = 0;  // the current chunk (workgroup-uniform)
  auto load = [&](batch_t& t, const int i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = i0 + (u * TPB + tid) * 4;
      i = i < end ? i : begin;
      detail::load4<type_t, NT>(val + i, t.v[u]);
      if constexpr (NT) {
        t.c[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(col16 + i));
        t.dst[u] = __builtin_nontemporal_load(dst4 + (i >> 2));
      } else {
        t.c[u] = *reinterpret_cast<const u16x4*>(col16 + i);
        t.dst[u] = dst4[i >> 2];
      }
    }
  };
  auto consume = [&](const batch_t& t, const int i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = i0 + (u * TPB + tid) * 4 < end;
      bool f[4];  // (a dead lane re-read the chunk's first group: all its items "end" and store nothing)
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = !live || (t.c[u][e] & run_end_bit) != 0;
      const bool has_padding = live && t.dst[u] < 0;
      type_t pr[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bool ends_behind = f[e];                             // a run end at or behind item e: e is a real item
#pragma unroll
        for (int k = e + 1; k < 4; ++k) ends_behind = ends_behind || f[k];
        const type_t xv = xs[t.c[u][e] & (run_end_bit - 1)];
        pr[e] = has_padding && !ends_behind ? type_t(0) : t.v[u][e] * xv;
      }
      type_t run[4];
      run[0] = pr[0];
#pragma unroll
      for (int e = 1; e < 4; ++e) run[e] = f[e - 1] ? pr[e] : run[e - 1] + pr[e];
      // across lanes: the lane's first run continues the previous lane's last unless that one ended on its item 3
      // (lane 0: the window's first item starts a run by construction)
      const int prev_end = wave::shift_up1(f[3] ? 1 : 0, 1);
      const bool continues = prev_end == 0;
      type_t tail = run[3];
      bool head = f[0] || f[1] || f[2] || !continues;
      wave::segmented_inclusive_sum(tail, head);
      const type_t prev_tail = wave::shift_up1(tail, type_t(0));  // (cross-lane read first, select afterwards)
      const type_t carry_in = continues ? prev_tail : type_t(0);
      if (live) {
        type_t* to = prod + (t.dst[u] & 0x7FFFFFFF);
        if (f[0] && f[1] && f[2] && f[3] && !has_padding) {  // four runs of one item (no locality here): one 16-byte store
          store4(to, run[0] + carry_in, run[1], run[2], run[3]);
        } else {
          int slot = 0;
          bool open = true;                                  // no run end in the lane before item e: the carry belongs to its run
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (f[e]) {
              to[slot] = run[e] + (open ? carry_in : type_t(0));
              ++slot;
              open = false;
            }
          }
        }
      }
    }
  };
  int panel_in_lds = -1;
  batch_t a, b;
  for (int c = c_lo; c < c_hi; ++c) {
    const int p = chunks[3 * c];
    begin = chunks[3 * c + 1];
    end = chunks[3 * c + 2];
    if (begin >= end) continue;  // (workgroup-uniform)
    int i0 = begin;  // (wave-uniform loop control)
    load(a, i0);     // the chunk's first stream loads are in flight while the x panel is fetched
    if (p != panel_in_lds) {
      const long long base = static_cast<long long>(p) * W;
      if (panel_in_lds >= 0) __syncthreads();  // every wavefront is done with the previous panel
      load_x_panel<TPB>(xs, x, base, cols - base < W ? static_cast<int>(cols - base) : W, tid);
      __syncthreads();
      panel_in_lds = p;
    }
    for (;;) {
      if (i0 + STEP < end) load(b, i0 + STEP);
      consume(a, i0);
      i0 += STEP;
      if (i0 >= end) break;
      if (i0 + STEP < end) load(a, i0 + STEP);
      consume(b, i0);
      i0 += STEP;
      if (i0 >= end) break;
    }
  }
}

/// One window of kernel B: 64 lanes x 4 consecutive items of ONE segment (rows non-decreasing; `pad_row` marks padding and
/// lanes outside the segment).  Runs of equal rows are summed -- inside a lane, then across lanes with the segmented prefix
/// sum -- and the lane-slot that ends a run adds the run's sum to the row's accumulator with a plain LDS read-modify-write:
/// inside a window every row ends exactly once, so no two lanes touch the same address (LDS float atomics, the obvious
/// alternative, retire ~0.4 lanes per clock and CU on gfx950: 5 x the time of the whole product stream).
template <typename type_t>
__device__ __forceinline__ void panel_window_add(type_t* __restrict__ acc, const int dump, const type_t (&v)[4],
                                                 const unsigned int (&r)[4]) {
  type_t run[4];
  run[0] = v[0];
#pragma unroll
  for (int e = 1; e < 4; ++e) run[e] = r[e] == r[e - 1] ? run[e - 1] + v[e] : v[e];
  const bool closed = r[3] != r[0];                                  // a run ends inside this lane
  const unsigned int prev_last = wave::shift_up1(r[3], 0xFFFFFFFEu);  // lane 0: never equal
  const unsigned int next_first = wave::shift_down1(r[0], 0xFFFFFFFDu);
  const bool continues = r[0] == prev_last;                           // my first run continues the previous lane's last
  type_t tail = run[3];
  bool head = closed || !continues;
  wave::segmented_inclusive_sum(tail, head);
  const type_t prev_tail = wave::shift_up1(tail, type_t(0));  // (cross-lane read: executed by every lane, selected afterwards)
  const type_t carry_in = continues ? prev_tail : type_t(0);
  // Branch-free read-modify-write: a slot that ends no run targets the lane's private dump word (index `dump`) and adds 0,
  // so all four reads and then all four writes are issued by every lane without touching the execution mask.
  int at[4];
  type_t add[4], old[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned int next = e < 3 ? r[e + 1] : next_first;
    const bool ends = r[e] != next && r[e] != pad_row;
    at[e] = ends ? static_cast<int>(r[e]) : dump;
    add[e] = ends ? run[e] + (r[e] == r[0] ? carry_in : type_t(0)) : type_t(0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) old[e] = acc[at[e]];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[at[e]] = old[e] + add[e];
}

/// Kernel B: one workgroup of WAVES wavefronts per sub-band.  The sub-band's products are one contiguous run of the B order
/// (`segb`, S * P + 1 ints, says where every segment starts); wavefront w takes the w-th share of the run's ITEMS, walks it
/// in windows of 64 lanes x 4 items, U windows in flight, and adds them into its OWN Hw accumulators in (dynamic) LDS; the
/// WAVES partial vectors are then added in wavefront order.  Everything a wavefront does to its accumulators is in program
/// order: the result is reproducible.
/// NT: the product / row streams are larger than the Infinity Cache (non-temporal loads).
/// Packed windows (small segments sharing a window: sorted only piecewise) add item by item with LDS atomics.
/// Since round 4 this is the kernel of 8-BYTE values only (ds_add_f64 is fast, and private accumulators keep an f64 result
/// reproducible to the last bit); 4-byte values go through panel_reduce_wide below.
template <bool NT, int WAVES, typename type_t, typename store_t>
__global__ void __launch_bounds__(WAVES * wave::size)
panel_reduce(const int* __restrict__ wstart, const int* __restrict__ wins, const int Hw, const type_t* __restrict__ prod,
             const unsigned short* __restrict__ row16, const int rows, const store_t out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char panel_lds[];
  constexpr int U = 8;      // windows in flight per wavefront
  using u16x4 = unsigned short __attribute__((ext_vector_type(4)));
  const int lane = wave::lane();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / wave::size);
  type_t* all = reinterpret_cast<type_t*>(panel_lds);
  const int stride = Hw + wave::size;                       // Hw accumulators + one private dump word per lane
  type_t* acc = all + static_cast<std::size_t>(w) * stride;
  const int dump = Hw + lane;
  const int s = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  for (int j = lane; j < stride; j += wave::size) acc[j] = type_t(0);
  // This wavefront's share: a contiguous WAVES-th of the sub-band's windows (make_windows built them: at most 256 items each,
  // pieces of one segment or packed small segments).  A run of equal rows cut between two wavefronts ends up in two
  // accumulators: fine.  Window descriptors are wave-uniform loads, U independent ones per step.
  const int w_lo = wstart[s], w_n = wstart[s + 1] - w_lo;
  const int my0 = w_lo + static_cast<int>(static_cast<long long>(w_n) * w / WAVES);
  const int my1 = w_lo + static_cast<int>(static_cast<long long>(w_n) * (w + 1) / WAVES);
  const int first = my0 < my1 ? wins[2 * my0] : 0;          // an in-bounds item for lanes that have nothing to load
  for (int base = my0; base < my1; base += U) {
    int wb[U], we[U];
    bool packed[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = base + u < my1 ? base + u : my0;          // (the surplus slots of the last step re-read a descriptor
      const int b = wins[2 * k], d = wins[2 * k + 1];         //  and are given no items)
      wb[u] = b;
      we[u] = base + u < my1 ? b + (d & 0xFFFF) : b;
      packed[u] = (d >> 16) != 0;
    }
    // branch-free loads: every vector of the U windows is requested before the first is waited for (a lane outside its
    // window re-reads the window's first vector -- in bounds -- and contributes padding)
    type_t v[U][4];
    u16x4 r16v[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = wb[u] + lane * 4;
      live[u] = i < we[u];
      const int at = live[u] ? i : (wb[u] < we[u] ? wb[u] : first);
      detail::load4<type_t, NT>(prod + at, v[u]);
      if constexpr (NT) r16v[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(row16 + at));
      else r16v[u] = *reinterpret_cast<const u16x4*>(row16 + at);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (wb[u] < we[u]) {  // (wave-uniform)
        unsigned int r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = live[u] ? static_cast<unsigned int>(r16v[u][e]) : static_cast<unsigned int>(pad_row);
        if (packed[u]) {   // small segments / segment tails sharing the window: sorted only piecewise
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r[e] != pad_row) atomicAdd(&acc[r[e]], v[u][e]);
        } else {
          panel_window_add<type_t>(acc, dump, v[u], r);
        }
      }
    }
  }
  __syncthreads();
  const long long row0 = static_cast<long long>(s) * Hw;
  for (int j = threadIdx.x; j < Hw && row0 + j < rows; j += WAVES * wave::size) {
    type_t sum = all[j];
#pragma unroll
    for (int k = 1; k < WAVES; ++k) sum += all[static_cast<std::size_t>(k) * stride + j];
    out(static_cast<int>(row0 + j), sum);
  }
}

/// Kernel B, second generation ("wide"): one workgroup of WAVES wavefronts per sub-band, ONE set of Hw accumulators per
/// workgroup held as fp64 words in LDS.  The sub-band's products are one contiguous run of the B order; it is walked in
/// windows of 64 lanes x 4 consecutive items straight from `bstart` -- no window table, no distinction between large and
/// small segments: wavefront w takes windows w, w + WAVES, ..., U of them in flight.  A lane's 4 items belong to one segment
/// (segments are padded to multiples of 4), so they are row-sorted; runs of equal ADJACENT rows are summed inside the window
/// -- in the lane, then across lanes with the wave64 segmented prefix sum, in the value type: a 256-slot partial carries at
/// most 3 + 6 roundings -- and every run end adds its sum to the row's accumulator with ds_add_f64 (3-8 lanes per clock and CU
/// on gfx950, where ds_add_f32 retires 0.33): a row that ends more than once in a window (segment boundaries) or in several
/// wavefronts at a time is the atomic unit's business.  A window without two adjacent equal rows (the rule when kernel A has
/// pre-summed the runs) skips sums and scan altogether.  (The fp64 scan of the first version was 42 of the window's 130
/// vector instructions and the kernel 78 % issue-bound on a C5 shard: profiles/r04_panel_pmc_c5_shard.json.)
/// Accuracy: products are fp32 (one rounding each), a window's partial is within 5.4e-7 of its L1 mass at worst (measured:
/// 1-2e-7), everything across windows, panels and wavefronts is fp64, y is rounded once at the store -- the north star's
/// 1e-6 holds on rows of any length.  Sums of fp32 numbers in fp64 are EXACT while a row's partials span
/// fewer than 53 - 24 - log2(n) binary orders of magnitude, so the result does not depend on the order the wavefronts'
/// atomics arrive in (nor on W / Hw) for such rows; beyond that two runs may differ in the last bit of an fp32 y.
/// LDS: Hw * 8 bytes per workgroup instead of WAVES * Hw * sizeof(type_t): sub-bands (and with them kernel A's store runs)
/// can be 2-4 x taller at the same occupancy.
template <bool NT, int WAVES, int U, typename type_t, typename store_t>
__global__ void __launch_bounds__(WAVES * wave::size)
panel_reduce_wide(const int* __restrict__ bstart, const int Hw, const type_t* __restrict__ prod,
                  const unsigned short* __restrict__ row16, const int rows, const store_t out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char panel_lds[];
  using u16x4 = unsigned short __attribute__((ext_vector_type(4)));
  double* acc = reinterpret_cast<double*>(panel_lds);
  const int lane = wave::lane();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / wave::size);
  const int s = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int b0 = bstart[s], b1 = bstart[s + 1];
  for (int j = threadIdx.x; j < Hw; j += WAVES * wave::size) acc[j] = 0.0;
  __syncthreads();
  constexpr int WIN = wave::size * 4;
  const int nw = (b1 - b0 + WIN - 1) / WIN;
  for (int k = w; k < nw; k += WAVES * U) {
    type_t v[U][4];
    u16x4 r16v[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {   // branch-free: every vector of the U windows is requested before the first wait
      const long long i = static_cast<long long>(b0) + static_cast<long long>(k + u * WAVES) * WIN + lane * 4;
      live[u] = i < b1;
      const int at = live[u] ? static_cast<int>(i) : b0;
      detail::load4<type_t, NT>(prod + at, v[u]);
      if constexpr (NT) r16v[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(row16 + at));
      else r16v[u] = *reinterpret_cast<const u16x4*>(row16 + at);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + u * WAVES < nw) {  // (wave-uniform)
        unsigned int r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = live[u] ? static_cast<unsigned int>(r16v[u][e]) : static_cast<unsigned int>(pad_row);
        const unsigned int next_first = wave::shift_down1(r[0], 0xFFFFFFFDu);
        // No two ADJACENT slots of the window share a row (the rule in a compact B order, where kernel A has already summed
        // the runs): every slot is its own run -- no sums, no scan, four atomics.  (wave-uniform)
        const bool joins = (r[0] == r[1] && r[1] != pad_row) || (r[1] == r[2] && r[2] != pad_row) || (r[2] == r[3] && r[3] != pad_row) ||
                           (r[3] == next_first && r[3] != pad_row);  // (padding ends a segment: equal pad marks join nothing)
        if (__builtin_amdgcn_ballot_w64(joins) == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r[e] != pad_row) atomicAdd(&acc[r[e]], static_cast<double>(v[u][e]));
          continue;
        }
        // Runs of equal adjacent rows: summed in the VALUE type inside the window (lane-sequential, then the scan's tree: at most
        // 3 + 6 roundings of a 256-slot partial, 5.4e-7 of its L1 mass at worst with 4-byte values), widened to fp64 at the atomic.
        type_t run[4];
        run[0] = v[u][0];
#pragma unroll
        for (int e = 1; e < 4; ++e) run[e] = r[e] == r[e - 1] ? run[e - 1] + v[u][e] : v[u][e];
        const bool closed = r[3] != r[0];
        const unsigned int prev_last = wave::shift_up1(r[3], 0xFFFFFFFEu);
        const bool continues = r[0] == prev_last;
        type_t tail = run[3];
        bool head = closed || !continues;
        wave::segmented_inclusive_sum(tail, head);
        const type_t prev_tail = wave::shift_up1(tail, type_t(0));  // (cross-lane read first, select afterwards)
        const type_t carry_in = continues ? prev_tail : type_t(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int next = e < 3 ? r[e + 1] : next_first;
          if (r[e] != next && r[e] != pad_row) atomicAdd(&acc[r[e]], static_cast<double>(run[e] + (r[e] == r[0] ? carry_in : type_t(0))));
        }
      }
    }
  }
  __syncthreads();
  const long long row0 = static_cast<long long>(s) * Hw;
  for (int j = threadIdx.x; j < Hw && row0 + j < rows; j += WAVES * wave::size) out(static_cast<int>(row0 + j), static_cast<type_t>(acc[j]));
}

}  // namespace panel

/// Accumulator rows a workgroup of kernel B may own: 4-byte values -> panel_reduce_wide, Hw fp64 words per WORKGROUP (32 KB at
/// 4096 rows: four 8-wavefront workgroups per CU); 8-byte values -> panel_reduce, Hw words per WAVEFRONT (4 x 16 KB at 2048).
template <typename type_t>
constexpr int panel_subband_rows_max() { return sizeof(type_t) == 4 ? 4096 : 2048; }

/// Sub-band height (rows per workgroup of kernel B): what matters is the size of a (panel, sub-band) segment, nnz Hw / (rows P)
/// items -- kernel A stores a segment's products as one run: Hw = the power of two that brings a segment to ~192 items, at
/// least 256 rows, at most panel_subband_rows_max (8-byte values: half of it -- 4 instead of 2 workgroups per CU -- unless
/// segments would then hold fewer than 96 items); halved while fewer than 512 sub-bands would be left.
/// Measured, 4-byte values, round 4 (tests/perf/exp_panel_reduce.py, 128 KB panels, kernel A + B): C2 52.7 / 55.3 / 57.4 us for
/// Hw = 512 / 1024 / 2048; C5 shard 231.6 / 224.4 / 247.9 us for 2048 / 4096 / 8192.
template <typename type_t>
inline int panel_subband_rows(int rows, int nnz, int P) {
  const int hw_max = panel_subband_rows_max<type_t>();
  const double per_row = rows > 0 && P > 0 ? static_cast<double>(nnz) / (static_cast<double>(rows) * static_cast<double>(P)) : 0.0;
  const double want = per_row > 0 ? 192.0 / per_row : 256.0;
  int hw = 256;
  while (hw < hw_max && hw < want) hw *= 2;
  if (sizeof(type_t) == 8 && hw == hw_max && per_row * (hw_max / 2) >= 96.0) hw = hw_max / 2;
  while (hw > 256 && static_cast<long long>(rows) / hw < 512) hw /= 2;
  return hw;
}

/// Kernel A's work list from the panel starts (A order, P + 1 entries): {panel, begin, end} triples.  A panel is cut into
/// round(items / CH) chunks of EQUAL size (a multiple of 4096 items = one load per lane of the widest workgroup), so no
/// workgroup loads a 64 / 128 KB x panel for the few thousand items left over by a fixed chunk length (8 M rows x 2
/// nonzeros: 68 K items per panel were 2 chunks, 65 536 + 2 700, and the launch two rounds of workgroups instead of one).
/// CH = 65 536 items, 131 072 once that still leaves two chunks per compute unit (a chunk starts with the fetch of its x
/// panel and ends with a drained pipeline: fewer, longer chunks where there are plenty).  Measured CH = 32 768 / 65 536 /
/// 131 072 / 262 144: C2 45 / 39 / 55 / 86 us (round 3), C5 shard 214 / 181 / 176 / 177, band C3 stand-in (compact) - / 352 /
/// 333 / 332, host-blocked - / 392 / 391 / 423.  LOOPS_PANEL_CHUNK overrides (tuning).
inline std::vector<int> panel_chunk_list(const std::vector<int>& panel_start, int P) {
  static const int forced = [] { const char* e = std::getenv("LOOPS_PANEL_CHUNK"); return e ? std::atoi(e) : 0; }();
  const long long total = P > 0 ? static_cast<long long>(panel_start[P]) - panel_start[0] : 0;
  const int CH = forced > 0 ? forced : (total >= 512ll * 131072 ? 131072 : 65536);
  std::vector<int> list;
  for (int k = 0; k < P; ++k) {
    const int b0 = panel_start[k], n = panel_start[k + 1] - b0;
    if (n <= 0) continue;
    const int pieces = n / CH + (n % CH >= CH / 2 || n < CH ? 1 : 0);
    const int size = ((n + pieces - 1) / pieces + 4095) & ~4095;
    for (int b = b0; b < b0 + n; b += size) {
      list.push_back(k);
      list.push_back(b);
      list.push_back(b + size < b0 + n ? b + size : b0 + n);
    }
  }
  return list;
}

/// EXPERIMENT switch: LOOPS_PANEL_REDUCE=1 sends 8-byte values through the wide kernel B too (0 / unset: the windowed one).
inline int panel_reduce_variant() {
  static const int v = [] { const char* e = std::getenv("LOOPS_PANEL_REDUCE"); return e ? std::atoi(e) : 0; }();
  return v;
}

/// Return codes of panel_binned_create beyond hipError_t values (the C ABI's LOOPS_E_BADARG / LOOPS_E_RANGE, include/loops_amd.h).
constexpr int panel_e_badarg = -1, panel_e_range = -2;

/// The device arrays of one panel-binned matrix, OWNED (hipMalloc / hipFree); what loops_panel_plan_* and
/// algorithms::spmv::panel_binned_t hold.  Type-erased over the value type (`vbytes`).
struct panel_binned_storage {
  int rows = 0, cols = 0, nnz = 0, vbytes = 0;
  int W = 0, Hw = 0, P = 0, S = 0, padded = 0, padded_b = 0, num_chunks = 0, compact = 0, awin = 0;
  long long runs = 0;          ///< runs of equal (row, panel) inside kernel A's windows: what a compact B order holds
  void *val = nullptr, *prod = nullptr;
  unsigned short *col16 = nullptr, *row16 = nullptr;
  int *perm = nullptr, *dst4 = nullptr, *segb = nullptr, *bstart = nullptr, *chunks = nullptr, *wins = nullptr, *wstart = nullptr;

  panel_binned_storage() = default;
  panel_binned_storage(const panel_binned_storage&) = delete;
  panel_binned_storage& operator=(const panel_binned_storage&) = delete;
  ~panel_binned_storage() { release(); }
  void release() {
    (void)hipFree(val); (void)hipFree(prod); (void)hipFree(col16); (void)hipFree(row16); (void)hipFree(perm); (void)hipFree(dst4);
    (void)hipFree(segb); (void)hipFree(bstart); (void)hipFree(chunks); (void)hipFree(wins); (void)hipFree(wstart);
    val = prod = nullptr; col16 = row16 = nullptr; perm = dst4 = segb = bstart = chunks = wins = wstart = nullptr;
  }
  template <typename type_t>
  panel_binned_view<type_t> view() const {
    return panel_binned_view<type_t>{rows, cols, nnz, W, Hw, P, S, padded, padded_b, compact, awin, static_cast<type_t*>(val), col16, dst4, row16,
                                     perm, segb, bstart, chunks, num_chunks, static_cast<type_t*>(prod), wins, wstart};
  }
};

/// When the compact layout is adopted without being asked for: the B order shrinks to at most this fraction of the nonzeros.
/// Each run costs kernel A a 4-byte store where a group of 4 single-item runs costs one 16-byte store, and the scan.  Measured
/// (tests/perf/exp_panel_reduce.py, compact against plain, kernel A + B): C2 (runs / nnz 0.50: hub rows fill their panels) 45.0
/// against 53.1 us, C3 stand-ins host-blocked (0.18) 434 against 742, band (0.10) 378 against 667, uniform (0.68) 671 against
/// 747, C5 shard (0.75) 238 against 234, 8 M rows x 2 (1.0) 80 against 69.
constexpr double panel_compact_threshold = 0.70;

/// Builds the panel-binned copy of a CSR on the device (O(nnz): one radix sort of 8-byte keys, n + 1 binary searches for the
/// segments' extents, scans, one placement pass that gathers two words per nonzero;
/// three host synchronisations: the A-order size and the run count, the B-order size, the panel starts).
/// subband_rows / panel_cols: 0 = automatic.  compact: -1 = automatic (panel_compact_threshold), 0 = never, 1 = always.
/// Returns 0, a hipError_t, panel_e_badarg (also: a column index outside [0, cols)) or panel_e_range.
template <typename index_t, typename offset_t, typename type_t>
int panel_binned_create(hipStream_t stream, int rows, int cols, int nnz, const offset_t* offsets, const index_t* indices,
                        const type_t* values, int subband_rows, int panel_cols, int compact, panel_binned_storage& out) {
  static_assert(sizeof(index_t) == 4 && sizeof(offset_t) == 4, "panel_binned_create: 32-bit indices and offsets");
  if (!offsets || rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && (!indices || !values)) || compact < -1 || compact > 1) return panel_e_badarg;
  out.release();
  out.rows = rows; out.cols = cols; out.nnz = nnz; out.vbytes = static_cast<int>(sizeof(type_t));
  out.W = panel_columns<type_t>(rows, cols, nnz);
  if (panel_cols != 0) {  // explicit: one of the two compiled widths
    if (panel_cols != panel_width<type_t>::value && panel_cols != panel_width<type_t>::wide) return panel_e_badarg;
    out.W = panel_cols;
  }
  out.P = cols > 0 ? static_cast<int>((static_cast<long long>(cols) + out.W - 1) / out.W) : 1;
  out.Hw = panel_subband_rows<type_t>(rows, nnz, out.P);
  if (subband_rows != 0) {  // explicit: a power of two, 64 .. panel_subband_rows_max
    const int cap = panel_reduce_variant() ? 4096 : panel_subband_rows_max<type_t>();
    if (subband_rows < 64 || subband_rows > cap || (subband_rows & (subband_rows - 1))) return panel_e_badarg;
    out.Hw = subband_rows;
  }
  out.S = rows > 0 ? static_cast<int>((static_cast<long long>(rows) + out.Hw - 1) / out.Hw) : 1;
  out.padded = out.padded_b = out.num_chunks = out.compact = 0;
  out.runs = 0;
  const int P = out.P, S = out.S;
  const long long segments = static_cast<long long>(P) * S;
  // every segment may carry up to 3 padding items
  if (segments > (1ll << 26) || static_cast<long long>(nnz) + 3 * segments >= (1ll << 31) - 4096) return panel_e_range;
  if (rows == 0) return 0;
  const int nseg = static_cast<int>(segments);

  // temporary storage: sort keys in / out (the `in` half is reused for the run-end flags and their scan once the sort is
  // done), the row of every nonzero, seven per-segment tables, hipcub scratch
  auto up = [](std::size_t v) { return (v + 255) & ~std::size_t(255); };
  std::size_t sort_bytes = 0, scan_bytes = 0;
  {
    unsigned long long* k = nullptr;
    int* ci = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, k, k, nnz, 32, 64);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, ci, ci, (nnz > nseg ? nnz : nseg) + 1);
  }
  const std::size_t cub_bytes_total = up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
  const std::size_t key_bytes = up((static_cast<std::size_t>(nnz) + 1) * 8), row_bytes = up(static_cast<std::size_t>(nnz) * 4);
  const std::size_t seg_bytes = up((static_cast<std::size_t>(nseg) + 1) * 4);
  const std::size_t temp_bytes = 2 * key_bytes + row_bytes + 7 * seg_bytes + 256 + cub_bytes_total;
  char* base = nullptr;
  int* panel_start = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), temp_bytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&panel_start), sizeof(int) * (static_cast<std::size_t>(P) + 1));
  struct guard_t {
    char*& a; int*& b;
    ~guard_t() { (void)hipFree(a); (void)hipFree(b); }
  } guard{base, panel_start};
  if (e != hipSuccess) return static_cast<int>(e);
  auto* keys_in = reinterpret_cast<unsigned long long*>(base);
  auto* keys_out = reinterpret_cast<unsigned long long*>(base + key_bytes);
  int* ends = reinterpret_cast<int*>(base);                                   // (over keys_in, after the sort) [nnz + 1]
  int* run_index = ends + (nnz + 1);                                          // [nnz + 1]: 2 x 4 x (nnz + 1) <= key_bytes
  auto* rc = reinterpret_cast<unsigned int*>(base + 2 * key_bytes);   // (row inside the sub-band) << 16 | column inside the panel
  int* counts = reinterpret_cast<int*>(base + 2 * key_bytes + row_bytes);  // (first of the seven per-segment tables; unused slot 0)
  const std::size_t seg_ints = seg_bytes / 4;
  int* seg_start = counts + seg_ints;        // unpadded start of segment g in the sorted keys
  int* padded_a = counts + 2 * seg_ints;     // padded size of segment g = p * S + s
  int* seg_dest = counts + 3 * seg_ints;     // A-order start of segment g
  int* padded_b = counts + 4 * seg_ints;     // B-order size of segment g (its padded item or run count)
  int* padded_t = counts + 5 * seg_ints;     // the same transposed to s * P + p
  int* seg_dest_b = counts + 6 * seg_ints;   // B-order start of segment (s, p)
  int* bad = counts + 7 * seg_ints;
  void* cub_temp = base + 2 * key_bytes + row_bytes + 7 * seg_bytes + 256;
  std::size_t cub_bytes = cub_bytes_total;
  auto scan = [&](const int* in, int* outp, int n) {
    cub_bytes = cub_bytes_total;
    return hipcub::DeviceScan::ExclusiveSum(cub_temp, cub_bytes, in, outp, n, stream);
  };
  const dim3 seg_grid(math::ceil_div(nseg + 1, 256));

  // ---- stage 1: segments of the A order, run ends
  e = hipMemsetAsync(bad, 0, sizeof(int), stream);
  if (e != hipSuccess) return static_cast<int>(e);
  constexpr int KEYS_PER_LANE = 8;
  if (nnz > 0) {
    hipLaunchKernelGGL((panel::make_keys<KEYS_PER_LANE, index_t, offset_t>), dim3(math::ceil_div(nnz, 256 * KEYS_PER_LANE)), dim3(256), 0,
                       stream, offsets, indices, rows, nnz, out.W, out.Hw, S, cols, keys_in, rc, bad);
    int end_bit = 33;
    while (end_bit < 64 && (static_cast<unsigned long long>(segments) >> (end_bit - 32)) != 0) ++end_bit;
    cub_bytes = cub_bytes_total;
    e = hipcub::DeviceRadixSort::SortKeys(cub_temp, cub_bytes, keys_in, keys_out, nnz, 32, end_bit, stream);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  // the segments' extents from the sorted keys (nnz == 0: keys_out is not read, every start is 0)
  hipLaunchKernelGGL(panel::segment_starts, seg_grid, dim3(256), 0, stream, keys_out, nnz, nseg, seg_start);
  hipLaunchKernelGGL(panel::pad_segment_sizes, seg_grid, dim3(256), 0, stream, seg_start, nseg, padded_a);
  e = scan(padded_a, seg_dest, nseg + 1);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(panel::mark_run_ends, dim3(math::ceil_div(nnz + 1, 256)), dim3(256), 0, stream, keys_out, seg_start, seg_dest, rc, nnz, S, ends);
  e = scan(ends, run_index, nnz + 1);
  if (e != hipSuccess) return static_cast<int>(e);
  int h_sizes[3] = {0, 0, 0};  // padded A items, runs, bad index seen
  e = hipMemcpyAsync(&h_sizes[0], seg_dest + nseg, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h_sizes[1], run_index + nnz, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h_sizes[2], bad, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return static_cast<int>(e);
  if (h_sizes[2] != 0) return panel_e_badarg;
  out.padded = h_sizes[0];
  out.runs = h_sizes[1];
  // (the run-end flag lives in bit 15 of col16: panels of at most 32768 columns -- both compiled widths of both value types)
  out.compact = compact == 1 || (compact == -1 && static_cast<double>(out.runs) <= panel_compact_threshold * static_cast<double>(nnz)) ? 1 : 0;
  if (out.W > 32768) out.compact = 0;
  out.awin = panel::a_window;

  // ---- stage 2: segments of the B order
  if (out.compact) hipLaunchKernelGGL(panel::pad_run_counts, seg_grid, dim3(256), 0, stream, seg_start, run_index, nseg, padded_b);
  const int* sizes_b = out.compact ? padded_b : padded_a;
  hipLaunchKernelGGL(panel::transpose_counts, seg_grid, dim3(256), 0, stream, sizes_b, P, S, padded_t);
  e = scan(padded_t, seg_dest_b, nseg + 1);
  if (e == hipSuccess) e = hipMemcpyAsync(&out.padded_b, seg_dest_b + nseg, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return static_cast<int>(e);

  // ---- stage 3: allocation and placement
  const std::size_t na = static_cast<std::size_t>(out.padded > 0 ? out.padded : 4), nb = static_cast<std::size_t>(out.padded_b > 0 ? out.padded_b : 4);
  auto alloc = [&](auto** ptr, std::size_t bytes) { if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(ptr), bytes); };
  alloc(&out.val, sizeof(type_t) * na);
  alloc(&out.prod, sizeof(type_t) * nb);
  alloc(&out.col16, sizeof(unsigned short) * na);
  alloc(&out.row16, sizeof(unsigned short) * nb);
  alloc(&out.perm, sizeof(int) * na);
  alloc(&out.dst4, sizeof(int) * (na / 4 + 1));
  alloc(&out.segb, sizeof(int) * (static_cast<std::size_t>(nseg) + 1));
  alloc(&out.bstart, sizeof(int) * (static_cast<std::size_t>(S) + 1));
  alloc(&out.wstart, sizeof(int) * (static_cast<std::size_t>(S) + 1));
  alloc(&out.wins, sizeof(int) * 2 * panel_window_capacity(out.padded_b, segments));
  if (e == hipSuccess) e = hipMemsetAsync(out.val, 0, sizeof(type_t) * na, stream);
  if (e == hipSuccess) e = hipMemsetAsync(out.col16, 0, sizeof(unsigned short) * na, stream);
  if (e == hipSuccess) e = hipMemsetAsync(out.row16, 0xFF, sizeof(unsigned short) * nb, stream);
  if (e == hipSuccess) e = hipMemsetAsync(out.perm, 0xFF, sizeof(int) * na, stream);
  if (e == hipSuccess) e = hipMemsetAsync(out.dst4, 0, sizeof(int) * (na / 4 + 1), stream);
  if (e == hipSuccess) e = hipMemsetAsync(out.prod, 0, sizeof(type_t) * nb, stream);  // (compact: kernel A never writes the padding slots)
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  if (nnz > 0) {
    auto* v = static_cast<type_t*>(out.val);
    const dim3 grid(math::ceil_div(nnz, 256));
    if (out.compact)
      hipLaunchKernelGGL((panel::place<true, type_t>), grid, dim3(256), 0, stream, keys_out, seg_start, seg_dest, seg_dest_b, rc,
                         ends, run_index, values, nnz, P, S, v, out.col16, out.dst4, out.row16, out.perm);
    else
      hipLaunchKernelGGL((panel::place<false, type_t>), grid, dim3(256), 0, stream, keys_out, seg_start, seg_dest, seg_dest_b, rc,
                         ends, run_index, values, nnz, P, S, v, out.col16, out.dst4, out.row16, out.perm);
  }
  e = hipMemcpyAsync(out.segb, seg_dest_b, sizeof(int) * (static_cast<std::size_t>(nseg) + 1), hipMemcpyDeviceToDevice, stream);
  // the windowed kernel B's work list: count per sub-band, scan, fill
  if (e == hipSuccess) e = hipMemsetAsync(out.wstart, 0, sizeof(int) * (static_cast<std::size_t>(S) + 1), stream);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  const dim3 wgrid(math::ceil_div(S, 256));
  hipLaunchKernelGGL(panel::make_windows<false>, wgrid, dim3(256), 0, stream, seg_dest_b, P, S, panel_pack_items<type_t>(), out.wstart,
                     static_cast<int*>(nullptr));
  e = scan(out.wstart, out.wstart, S + 1);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  hipLaunchKernelGGL(panel::make_windows<true>, wgrid, dim3(256), 0, stream, seg_dest_b, P, S, panel_pack_items<type_t>(), out.wstart, out.wins);
  const int m = (S > P ? S : P) + 1;
  hipLaunchKernelGGL(panel::extract_starts, dim3(math::ceil_div(m, 256)), dim3(256), 0, stream, seg_dest, seg_dest_b, P, S, out.bstart, panel_start);
  std::vector<int> ps(static_cast<std::size_t>(P) + 1, 0);
  e = hipMemcpyAsync(ps.data(), panel_start, sizeof(int) * ps.size(), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) {
    const std::vector<int> list = panel_chunk_list(ps, P);  // kernel A's work list
    out.num_chunks = static_cast<int>(list.size() / 3);
    e = hipMalloc(reinterpret_cast<void**>(&out.chunks), sizeof(int) * (list.empty() ? 3 : list.size()));
    if (e == hipSuccess && !list.empty()) e = hipMemcpy(out.chunks, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  return 0;
}

/// Workgroups of kernel A: 0 = one per chunk (the default); LOOPS_PANEL_GRID=<n> makes n workgroups walk contiguous shares
/// of the chunk list, keeping the x panel in LDS while the panel stays the same.  Measured (round 4, 256 / 512 / 1024
/// workgroups against one per chunk): host-blocked C3 stand-in 414 / 426 / 423 against 392 us, uniform 621 / 612 / 598 against
/// 567, C5 shard 189 / 181 / 186 against 186, C2 equal -- the panel fetch per chunk is not what kernel A waits for, and static
/// shares lose the balancing of the hardware dispatcher.
inline int panel_products_grid() {
  static const int v = [] { const char* e = std::getenv("LOOPS_PANEL_GRID"); return e ? std::atoi(e) : 0; }();
  return v;
}

/// y = A x over a panel-binned matrix: kernel A then kernel B.  stages: bit 0 = products, bit 1 = reduce.
/// Streams are read non-temporally unless the product's whole working set fits the Infinity Cache (see `nt` below).
template <typename type_t, typename store_t>
int launch_panel_binned_to(hipStream_t stream, const panel_binned_view<type_t>& m, const type_t* x, const store_t out, int stages = 3) {
  if (m.rows == 0) return 0;
  constexpr int W = panel_width<type_t>::value, W2 = panel_width<type_t>::wide;
  if (m.W != W && m.W != W2) return static_cast<int>(hipErrorInvalidValue);
  // Non-temporal streams unless the WHOLE working set of a product -- values, columns, destinations (7 B per item with 4-byte
  // values), products and rows (6 B per slot), plus x and y -- fits the 256 MB Infinity Cache and so survives from one product
  // to the next (C2, 219 MB: plain loads 64.7 -> 58.2 us per product, kernel A 37.4 -> 31.6); beyond it plain loads only
  // evict each other's streams (C5 shard, kernel B plain: 256 -> 291 us).
  const bool nt = static_cast<double>(m.padded) * (sizeof(type_t) + 3.0) + static_cast<double>(m.padded_b) * (sizeof(type_t) + 2.0) +
                      (static_cast<double>(m.rows) + m.cols) * sizeof(type_t) > 240e6;
  if ((stages & 1) && m.num_chunks > 0) {
    const int want = panel_products_grid();
    const int grid = want > 0 && want < m.num_chunks ? want : m.num_chunks;  // one workgroup per chunk unless asked otherwise
    auto go = [&](auto kernel, int threads) {
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, stream, m.chunks, m.num_chunks, m.val, m.col16, m.dst4, x, m.cols, m.prod);
    };
    if (m.compact) {
      if (m.awin != panel::a_window) return static_cast<int>(hipErrorInvalidValue);
      constexpr int UC = LOOPS_PANEL_COMPACT_U;
      if (m.W == W) {
        if (nt) go(panel::panel_products_compact<512, W, UC, true, type_t>, 512);
        else go(panel::panel_products_compact<512, W, UC, false, type_t>, 512);
      } else {
        if (nt) go(panel::panel_products_compact<1024, W2, UC, true, type_t>, 1024);
        else go(panel::panel_products_compact<1024, W2, UC, false, type_t>, 1024);
      }
    } else if (m.W == W) {
      if (nt) go(panel::panel_products<512, W, 4, true, type_t>, 512);
      else go(panel::panel_products<512, W, 4, false, type_t>, 512);
    } else {
      if (nt) go(panel::panel_products<1024, W2, 4, true, type_t>, 1024);
      else go(panel::panel_products<1024, W2, 4, false, type_t>, 1024);
    }
  }
  if (stages & 2) {
    const int variant = panel_reduce_variant();
    if (sizeof(type_t) == 4 || variant != 0) {  // one set of fp64 accumulators per workgroup, 8 wavefronts, 2 windows in flight each
      const std::size_t lds = static_cast<std::size_t>(m.Hw) * sizeof(double);
      auto go = [&](auto kernel, int waves) {
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kernel, dim3(m.S), dim3(waves * wave::size), lds, stream, m.bstart, m.Hw, m.prod, m.row16, m.rows, out);
      };
      // measured WAVES x U = 4x4 / 8x4 / 8x2 / 16x4 (tests/perf/exp_panel_reduce.py): C2 (Hw 512) 24.8 / 21.4 / 21.4 / 22.2 us,
      // C5 shard (Hw 4096) 99.2 / 83.1 / 79.4 / 79.0 us
      if (nt) go(panel::panel_reduce_wide<true, 8, 2, type_t, store_t>, 8);
      else go(panel::panel_reduce_wide<false, 8, 2, type_t, store_t>, 8);
      return static_cast<int>(hipGetLastError());
    }
    constexpr int waves = 4;  // (measured 4 / 2 / 1 wavefronts, round 3: C2 34 / 40 / 64 us, C5 shard 126 / 175 / 299 us)
    const std::size_t lds = static_cast<std::size_t>(waves) * (m.Hw + wave::size) * sizeof(type_t);
    auto go = [&](auto kernel) {
      // (66.5 KB at Hw = 16 KB / sizeof(T): above the 64 KB a kernel may use without asking; the attribute belongs to the
      // device's copy of THIS kernel, so it is set at every such launch -- it is cheap -- rather than remembered per lambda)
      if (lds > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (16384 + 64 * 8));
      hipLaunchKernelGGL(kernel, dim3(m.S), dim3(waves * wave::size), lds, stream, m.wstart, m.wins, m.Hw, m.prod, m.row16, m.rows, out);
    };
    if (nt) go(panel::panel_reduce<true, 4, type_t, store_t>); else go(panel::panel_reduce<false, 4, type_t, store_t>);
  }
  return static_cast<int>(hipGetLastError());
}

template <typename type_t>
int launch_panel_binned(hipStream_t stream, const panel_binned_view<type_t>& m, const type_t* x, type_t* y, int stages = 3) {
  return launch_panel_binned_to(stream, m, x, plain_store<type_t>{y}, stages);
}

/// The same product with the finished rows of y also stored to `peers.count` peer-mapped vectors (multi-GPU epilogue
/// fan-out; see fanout_store).
template <typename type_t>
int launch_panel_binned_fanout(hipStream_t stream, const panel_binned_view<type_t>& m, const type_t* x, type_t* y,
                               const peer_fanout<type_t>& peers) {
  if (peers.count < 0 || peers.count > max_peers) return static_cast<int>(hipErrorInvalidValue);
  return launch_panel_binned_to(stream, m, x, fanout_store<type_t>{y, peers}, 3);
}

}  // namespace kernels
}  // namespace loops
