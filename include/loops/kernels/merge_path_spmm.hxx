/**
 * @file merge_path_spmm.hxx
 * @brief CSR SpMM  C[m x n] = A[m x k] * B[k x n]  (dense row-major B, C) on the merge-path
 * schedule, for CDNA4.  SURVEY 8(f) row 3: the step the reference has only as a per-thread loop
 * (algorithms/spmm/thread_mapped.cuh:28-52: thread per row, columns of B in the OUTER loop, so the
 * matrix is re-read n times and every B access is a 4-byte gather).
 *
 * Decomposition: the plan's merge tiles (same coordinates as the SpMV: rows + nonzeros split
 * evenly), one workgroup per (tile, slab of G * V <= 256 columns of B).  Inside a tile
 *
 *   STAGE  col_idx / values / row ends of the tile -> LDS, coalesced, read from HBM once;
 *   SPLIT  the tile is cut into TPB/G equal merge ranges, one per sub-group of G lanes
 *          (halving search over the LDS row ends, search.hxx semantics);
 *   WALK   lane l of a sub-group owns V consecutive columns of B and C (16-byte accesses when
 *          the row pitch allows): for every nonzero of the range the sub-group reads one ROW of
 *          the slab of B coalesced (G * V * 4 bytes, U rows in flight),
 *          accumulates in a register and stores a row of C whenever a row end is consumed --
 *          rank-1 updates, no atomics, no reshaping into dense blocks;
 *   STITCH the first row a sub-group closes and its open tail go through LDS and are chained
 *          across sub-groups; the row still open at the tile end leaves as an n-wide carry-out
 *          that the fix-up kernel adds (rows longer than a tile).
 *
 * C needs no zero-fill: every row of C is stored exactly once per column (empty rows included);
 * the summation order is deterministic.
 *
 * Traffic per tile and slab: natoms * (8 + G * V * 4) bytes read (B rows through L2 / MALL),
 * nrows * G * V * 4 written.  Bound: B-row gather bandwidth (L2 / Infinity Cache / HBM depending on
 * k * n * 4 bytes vs the 4 MB L2 and 256 MB MALL).
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include <loops/kernels/merge_path_spmv.hxx>

namespace loops {
namespace kernels {

namespace detail {

/// V consecutive elements from a (V * sizeof(T))-byte aligned address with the widest loads
/// available (up to 16 bytes per instruction); native ext-vector types, see load4.
template <int V, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, T (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = *p;
  } else if constexpr (V == 2) {
    using v2 = T __attribute__((ext_vector_type(2)));
    const v2 v = *reinterpret_cast<const v2*>(p);
    out[0] = v.x;
    out[1] = v.y;
  } else {
    static_assert(V == 4, "V: 1, 2 or 4");
    load4<T, false>(p, out);
  }
}

template <int V, typename T>
__device__ __forceinline__ void store_row(T* __restrict__ p, const T (&in)[V]) {
  if constexpr (V == 1) {
    *p = in[0];
  } else if constexpr (V == 2) {
    using v2 = T __attribute__((ext_vector_type(2)));
    v2 v;
    v.x = in[0];
    v.y = in[1];
    *reinterpret_cast<v2*>(p) = v;
  } else if constexpr (sizeof(T) == 4) {
    using v4 = T __attribute__((ext_vector_type(4)));
    v4 v;
    v.x = in[0];
    v.y = in[1];
    v.z = in[2];
    v.w = in[3];
    *reinterpret_cast<v4*>(p) = v;
  } else {
    using v2 = T __attribute__((ext_vector_type(2)));
    v2 a, b;
    a.x = in[0];
    a.y = in[1];
    b.x = in[2];
    b.y = in[3];
    *reinterpret_cast<v2*>(p) = a;
    *reinterpret_cast<v2*>(p + 2) = b;
  }
}

}  // namespace detail

/**
 * @tparam TPB threads per workgroup, IPT merge items per thread (the plan's tile = TPB * IPT).
 * @tparam G   lanes per sub-group (power of two <= 64).
 * @tparam V   consecutive columns of B / C per lane (1, 2 or 4: 4-, 8- or 16-byte accesses; the
 *             launcher picks the widest the row pitch and base alignment allow).  A workgroup
 *             covers a slab of G * V columns; the per-nonzero bookkeeping is amortised over V.
 * @tparam U   rows of B in flight per sub-group.
 * grid = (merge tiles, ceil(n / (G * V))).
 */
template <int TPB, int IPT, int G, int V, int U, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
merge_path_spmm(const coord_t* __restrict__ coords, const int rows, const int nnz,
                const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                const type_t* __restrict__ values, const type_t* __restrict__ B, const int n, const std::size_t ldb,
                type_t* __restrict__ C, const std::size_t ldc, int* __restrict__ carry_row,
                type_t* __restrict__ carry_mat) {
  constexpr int TILE = TPB * IPT;
  constexpr int S = TPB / G;       // sub-groups per workgroup
  constexpr int ITEMS = TILE / S;  // merge items per sub-group
  static_assert(TPB % G == 0 && (G & (G - 1)) == 0 && G <= wave::size, "G: power of two <= 64 dividing TPB");

  struct entry_t {
    unsigned int brow;  // row offset into B in units of V elements: col_idx * (ldb / V)
    type_t val;
  };
  __shared__ offset_t s_re[TILE + 2];
  __shared__ entry_t s_ent[TILE];
  __shared__ type_t s_head[V][TPB];
  __shared__ type_t s_tail[V][TPB];
  __shared__ int s_closed[S];

  const int tid = threadIdx.x;
  // one contiguous run of merge tiles per XCD: rows of B shared by neighbouring tiles live in ONE L2
  const int b = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const coord_t c0 = coords[b];
  const coord_t c1 = coords[b + 1];
  const int row0 = static_cast<int>(c0.x);
  const int nz0 = static_cast<int>(c0.y);
  const int nrows = static_cast<int>(c1.x) - row0;
  const int natoms = static_cast<int>(c1.y) - nz0;
  const int total = nrows + natoms;

  // ---- STAGE ------------------------------------------------------------------------------
  for (int i = tid; i <= nrows; i += TPB) {
    int r = row0 + i;
    r = r < rows - 1 ? r : rows - 1;
    s_re[i] = offsets[r + 1];
  }
  const unsigned int pitch = static_cast<unsigned int>(ldb / V);  // launcher: ldb % V == 0, cols * pitch < 2^32
  for (int i = tid; i < natoms; i += TPB)
    s_ent[i] = entry_t{static_cast<unsigned int>(indices[nz0 + i]) * pitch, values[nz0 + i]};
  __syncthreads();

  // ---- SPLIT ------------------------------------------------------------------------------
  const int sg = tid / G;
  const int l = tid % G;
  const int col = (blockIdx.y * G + l) * V;
  const bool active = col < n;  // n is a multiple of V: a lane's V columns are all in or all out
  auto split = [&](int diag) {
    int lo = diag - natoms > 0 ? diag - natoms : 0;
    int count = (diag < nrows ? diag : nrows) - lo;
    while (count > 0) {
      const int half = count >> 1;
      const int mid = lo + half;
      if (s_re[mid] <= nz0 + (diag - mid - 1)) {
        lo = mid + 1;
        count -= half + 1;
      } else {
        count = half;
      }
    }
    return lo < nrows ? lo : nrows;
  };
  const int d0 = sg * ITEMS < total ? sg * ITEMS : total;
  const int d1 = d0 + ITEMS < total ? d0 + ITEMS : total;
  int row = split(d0);
  const int ty0 = d0 - row;
  const int row1 = split(d1);
  const int ty1 = d1 - row1;

  // ---- WALK -------------------------------------------------------------------------------
  // lanes past the last column read column 0 (valid memory) and never store: no predication in
  // the hot loop
  const type_t* __restrict__ Bc = B + (active ? col : 0);
  type_t acc[V], head_vec[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = head_vec[k] = type_t(0);
  int head_row = 0;
  bool closed = false;
  int row_end = s_re[row];
  auto flush = [&]() {
    if (!closed) {
#pragma unroll
      for (int k = 0; k < V; ++k) head_vec[k] = acc[k];
      head_row = row;
      closed = true;
    } else if (active) {
      detail::store_row<V>(C + static_cast<std::size_t>(row0 + row) * ldc + col, acc);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = type_t(0);
    ++row;
    row_end = s_re[row];
  };
  auto fetch = [&](const entry_t& e, type_t (&out)[V]) {
    detail::load_row<V>(Bc + static_cast<std::size_t>(e.brow) * V, out);
  };
  int a = ty0;
  for (; a + U <= ty1; a += U) {  // full batches: U rows of B in flight, no bounds checks
    entry_t e[U];
    type_t bv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) e[u] = s_ent[a + u];
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(e[u], bv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      while (nz0 + a + u >= row_end) flush();
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += e[u].val * bv[u][k];
    }
  }
  for (; a < ty1; ++a) {  // < U left over
    const entry_t e = s_ent[a];
    type_t bv[V];
    fetch(e, bv);
    while (nz0 + a >= row_end) flush();
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] += e.val * bv[k];
  }
  while (row < row1) flush();

  // ---- STITCH -----------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < V; ++k) {
    s_head[k][tid] = head_vec[k];
    s_tail[k][tid] = acc[k];
  }
  if (l == 0) s_closed[sg] = closed ? 1 : 0;
  __syncthreads();
  // partial of the open row entering this sub-group: earlier sub-groups back to the last one that
  // closed a row (what precedes the tile is added by the fix-up kernel)
  type_t in[V];
#pragma unroll
  for (int k = 0; k < V; ++k) in[k] = type_t(0);
  for (int i = sg - 1; i >= 0; --i) {
#pragma unroll
    for (int k = 0; k < V; ++k) in[k] += s_tail[k][i * G + l];
    if (s_closed[i]) break;
  }
  if (closed && active) {
#pragma unroll
    for (int k = 0; k < V; ++k) head_vec[k] += in[k];
    detail::store_row<V>(C + static_cast<std::size_t>(row0 + head_row) * ldc + col, head_vec);
  }
  if (sg == S - 1) {
    if (active) {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += closed ? type_t(0) : in[k];
      detail::store_row<V>(carry_mat + static_cast<std::size_t>(b) * n + col, acc);
    }
    if (l == 0 && blockIdx.y == 0) carry_row[b] = row0 + nrows;
  }
}

/// C[row, :] += sum of the carry-outs of the run of merge tiles that ended inside `row`.
template <typename type_t>
__global__ void merge_path_spmm_fixup(const int* __restrict__ carry_row, const type_t* __restrict__ carry_mat,
                                      const int num_merge_tiles, const int rows, const int n,
                                      type_t* __restrict__ C, const std::size_t ldc) {
  const std::size_t idx = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const std::size_t i = idx / n;
  const int c = static_cast<int>(idx % n);
  if (i >= static_cast<std::size_t>(num_merge_tiles)) return;
  const int r = carry_row[i];
  if (r >= rows) return;
  if (i > 0 && carry_row[i - 1] == r) return;  // not the first tile of the run
  type_t s = carry_mat[i * n + c];
  for (std::size_t j = i + 1; j < static_cast<std::size_t>(num_merge_tiles) && carry_row[j] == r; ++j)
    s += carry_mat[j * n + c];
  C[static_cast<std::size_t>(r) * ldc + c] += s;
}

/// The reference-shaped SpMM (thread per row, columns outer) on raw pointers, for the C ABI's
/// THREAD_MAPPED schedule and as the "before" of the comparison.
template <typename index_t, typename offset_t, typename type_t>
__global__ void thread_mapped_spmm(const int rows, const offset_t* __restrict__ offsets,
                                   const index_t* __restrict__ indices, const type_t* __restrict__ values,
                                   const type_t* __restrict__ B, const int n, const std::size_t ldb,
                                   type_t* __restrict__ C, const std::size_t ldc) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const offset_t begin = offsets[row], end = offsets[row + 1];
  for (int c = 0; c < n; ++c) {
    type_t sum = type_t(0);
    for (offset_t nz = begin; nz < end; ++nz) sum += values[nz] * B[static_cast<std::size_t>(indices[nz]) * ldb + c];
    C[static_cast<std::size_t>(row) * ldc + c] = sum;
  }
}

}  // namespace kernels
}  // namespace loops
