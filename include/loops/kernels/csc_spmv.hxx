/**
 * @file csc_spmv.hxx
 * @brief CSC SpMV kernels: y[row_indices[k]] += values[k] * x[column of k].
 *
 *  - `csc_column_spmv`: one lane per column walking its nonzeros (the shape of the reference kernel,
 *    algorithms/spmv/csc_thread_mapped.cuh:36-58): adjacent lanes read from unrelated places and a
 *    long column serialises one lane.
 *  - `csc_nonzero_split_spmv`: the nonzeros are split evenly -- a lane owns IPT consecutive nonzeros
 *    whatever column they are in (16-byte loads of row_indices / values), finds the column of its first
 *    one with a search over the column offsets and walks the offsets from there; x[col] is a broadcast
 *    read.  One atomicAdd per nonzero remains (the rows of a column are scattered by construction).
 * y must be zero-filled (same precondition as the reference).
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/util/math.hxx>

namespace loops {
namespace kernels {

template <typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(256)
csc_column_spmv(const std::size_t cols, const offset_t* __restrict__ offsets, const index_t* __restrict__ row_indices,
                const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  const std::size_t col = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  const type_t xc = x[col];
  for (offset_t k = offsets[col]; k < offsets[col + 1]; ++k) atomicAdd(&y[row_indices[k]], values[k] * xc);
}

template <int IPT, bool VEC, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(256)
csc_nonzero_split_spmv(const int cols, const int nnz, const offset_t* __restrict__ offsets,
                       const index_t* __restrict__ row_indices, const type_t* __restrict__ values,
                       const type_t* __restrict__ x, type_t* __restrict__ y) {
  static_assert(IPT % 4 == 0, "IPT: multiple of 4");
  const long long base_ll = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * IPT;
  if (base_ll >= nnz) return;
  const int base = static_cast<int>(base_ll);
  // column of nonzero `base`: last c with offsets[c] <= base
  int col = 0, count = cols;
  while (count > 0) {
    const int half = count >> 1;
    const int mid = col + half;
    if (offsets[mid + 1] <= base) {
      col = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  index_t r[IPT];
  type_t v[IPT];
  const bool full = base + IPT <= nnz;
  if (VEC && full) {
#pragma unroll
    for (int k = 0; k < IPT; k += 4) {
      index_t r4[4];
      type_t v4[4];
      detail::load4<index_t, false>(row_indices + base + k, r4);
      detail::load4<type_t, false>(values + base + k, v4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[k + j] = r4[j];
        v[k + j] = v4[j];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const bool ok = base + k < nnz;
      r[k] = ok ? row_indices[base + k] : index_t(0);
      v[k] = ok ? values[base + k] : type_t(0);
    }
  }
  offset_t col_end = offsets[col + 1];
  type_t xc = x[col];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    if (base + k < nnz) {
      while (base + k >= col_end) {  // next non-empty column
        ++col;
        col_end = offsets[col + 1];
        xc = x[col];
      }
      atomicAdd(&y[r[k]], v[k] * xc);
    }
  }
}

template <typename index_t, typename offset_t, typename type_t>
int launch_csc_column(hipStream_t stream, std::size_t cols, const offset_t* offsets, const index_t* row_indices,
                      const type_t* values, const type_t* x, type_t* y) {
  if (cols == 0) return 0;
  hipLaunchKernelGGL((csc_column_spmv<index_t, offset_t, type_t>),
                     dim3(static_cast<unsigned>(math::ceil_div(cols, std::size_t(256)))), dim3(256), 0, stream, cols,
                     offsets, row_indices, values, x, y);
  return static_cast<int>(hipGetLastError());
}

template <typename index_t, typename offset_t, typename type_t>
int launch_csc_nonzero_split(hipStream_t stream, int cols, int nnz, const offset_t* offsets, const index_t* row_indices,
                             const type_t* values, const type_t* x, type_t* y) {
  if (nnz == 0 || cols == 0) return 0;
  constexpr int IPT = 8;
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(row_indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  const dim3 grid(static_cast<unsigned>(math::ceil_div(static_cast<long long>(nnz), 256ll * IPT))), block(256);
  if (aligned)
    hipLaunchKernelGGL((csc_nonzero_split_spmv<IPT, true, index_t, offset_t, type_t>), grid, block, 0, stream, cols, nnz,
                       offsets, row_indices, values, x, y);
  else
    hipLaunchKernelGGL((csc_nonzero_split_spmv<IPT, false, index_t, offset_t, type_t>), grid, block, 0, stream, cols, nnz,
                       offsets, row_indices, values, x, y);
  return static_cast<int>(hipGetLastError());
}

// ---------------------------------------------------------------------------------------- CSC -> CSR on the device
// What a held CSC plan does once (loops_csc_plan_*): the scatter y[row] += ... of a CSC product is one global atomic per
// nonzero on this chip (~16 G/s: 1.04 ms on C2 whatever the kernel), the same matrix as CSR runs in 0.1 ms -- so a caller
// that multiplies more than once should transpose the storage once.  Keys (row << 32 | column) + the nonzero's position are
// radix-sorted; rows are counted on the way.

/// key[k] = row << 32 | column of nonzero k (column by a search over the offsets), pos[k] = k, counts[row + 1] += 1.
/// A row index outside [0, rows) sets *bad and is counted nowhere (the caller refuses the input after the pass).
template <typename index_t, typename offset_t>
__global__ void __launch_bounds__(256)
csc_transpose_keys(const int rows, const int cols, const int nnz, const offset_t* __restrict__ col_offsets,
                   const index_t* __restrict__ row_indices, unsigned long long* __restrict__ keys, int* __restrict__ pos,
                   int* __restrict__ counts, int* __restrict__ bad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  int col = 0, count = cols;   // last c with col_offsets[c] <= k
  while (count > 1) {
    const int half = count >> 1;
    if (col_offsets[col + half] <= k) { col += half; count -= half; }
    else count = half;
  }
  unsigned int r = static_cast<unsigned int>(row_indices[k]);
  const bool ok = r < static_cast<unsigned int>(rows);
  if (!ok) { *bad = 1; r = 0; }
  keys[k] = (static_cast<unsigned long long>(r) << 32) | static_cast<unsigned int>(col);
  pos[k] = k;
  if (ok) atomicAdd(counts + r + 1, 1);
}

/// The same keys from COO triplets (any order): key[k] = row << 32 | column; a row or column index out of range sets *bad.
template <typename index_t>
__global__ void __launch_bounds__(256)
coo_transpose_keys(const int rows, const int cols, const int nnz, const index_t* __restrict__ row_indices,
                   const index_t* __restrict__ col_indices, unsigned long long* __restrict__ keys, int* __restrict__ pos,
                   int* __restrict__ counts, int* __restrict__ bad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  unsigned int r = static_cast<unsigned int>(row_indices[k]), c = static_cast<unsigned int>(col_indices[k]);
  const bool ok = r < static_cast<unsigned int>(rows) && c < static_cast<unsigned int>(cols);
  if (!ok) { *bad = 1; r = 0; c = 0; }
  keys[k] = (static_cast<unsigned long long>(r) << 32) | c;
  pos[k] = k;
  if (ok) atomicAdd(counts + r + 1, 1);
}

/// indices[i] = column of the i-th nonzero in (row, column) order, values[i] = its value.
template <typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
csc_transpose_finish(const int nnz, const unsigned long long* __restrict__ keys_sorted, const int* __restrict__ perm,
                     const type_t* __restrict__ csc_values, index_t* __restrict__ indices, type_t* __restrict__ values) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  indices[i] = static_cast<index_t>(keys_sorted[i] & 0xFFFFFFFFull);
  values[i] = csc_values[perm[i]];
}

template <typename type_t>
__global__ void __launch_bounds__(256)
gather_values(const int n, const int* __restrict__ perm, const type_t* __restrict__ from, type_t* __restrict__ to) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) to[i] = from[perm[i]];
}

// ------------------------------------------------------------------------------------------------------------------
// BINNED one-shot CSC product.  csc_nonzero_split_spmv pays one memory-side atomic per nonzero (the rows of a column are
// scattered; C2: 1.1 ms = 15 G atomics/s, the reference's shape 1.15 ms).  Here the products travel instead:
//   1. csc_products: p[k] = values[k] * x[column of k] (x is a broadcast read: no gather at all);
//   2. ONE radix pass per 8 bits of (row / 4096) -- hipcub SortPairs over the row indices as they are, bits 12 and up -- moves
//      (row, product) into bins of 4 096 consecutive rows, stably;
//   3. csc_bin_bounds + csc_reduce_bins: a workgroup adds a bin's products up in LDS (fp64 words, one ds_add_f64 per product) and stores
//      adds them to the bin's 4 096 rows of y (zero-filled by the caller, like the atomic kernels' y).  A bin of more than `csc_bin_chunk`
//      products is shared by up to 32 workgroups, which add their sums with one atomic per row and workgroup.
// Sums: unordered fp64 in LDS, rounded once (several workgroups: once each) -- exact on exactly summable inputs, within an ulp of the
// fp64 sum otherwise; the reference's kernel adds with fp32 atomics in an order that varies from run to run.
constexpr int csc_bin_rows = 4096, csc_bin_shift = 12, csc_bin_chunk = 1 << 17, csc_bin_shares = 32;

template <int IPT, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(256)
csc_products(const int cols, const int nnz, const offset_t* __restrict__ offsets, const type_t* __restrict__ values, const type_t* __restrict__ x,
             type_t* __restrict__ products) {
  const long long base_ll = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * IPT;
  if (base_ll >= nnz) return;
  const int base = static_cast<int>(base_ll);
  int col = 0, count = cols;  // column of nonzero `base`: last c with offsets[c] <= base
  while (count > 0) {
    const int half = count >> 1;
    const int mid = col + half;
    if (offsets[mid + 1] <= base) {
      col = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  int next = static_cast<int>(offsets[col + 1]);
  type_t xc = x[col];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = base + i;
    if (k >= nnz) break;
    while (k >= next) {  // (empty columns are stepped over)
      ++col;
      next = static_cast<int>(offsets[col + 1]);
      xc = x[col];
    }
    products[k] = values[k] * xc;
  }
}

/// bounds[b] = first position of the sorted rows that lies in bin b or later (b in [0, bins]): a lane per bound, a halving search.
template <typename index_t>
__global__ void __launch_bounds__(256)
csc_bin_bounds(const int nnz, const int bins, const index_t* __restrict__ sorted_rows, int* __restrict__ bounds) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b > bins) return;
  const long long first_row = static_cast<long long>(b) << csc_bin_shift;
  int lo = 0, count = nnz;
  while (count > 0) {
    const int half = count >> 1;
    if (static_cast<long long>(sorted_rows[lo + half]) < first_row) {
      lo += half + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  bounds[b] = lo;
}

/// Workgroup (bin, share): a bin of at most csc_bin_chunk products is added up by share 0 alone, which adds its sums to the bin's rows
/// of y with plain read-modify-writes; a larger bin is cut into chunks of csc_bin_chunk products dealt to the gridDim.y shares in
/// turn, each of which adds its LDS sums to y with one atomic per row it touched.  Shares without work return at once.
template <typename index_t, typename type_t>
__global__ void __launch_bounds__(512)
csc_reduce_bins(const int rows, const int* __restrict__ bounds, const index_t* __restrict__ sorted_rows, const type_t* __restrict__ sorted_products,
                type_t* __restrict__ y) {
  __shared__ double s_acc[csc_bin_rows];
  const int bin = blockIdx.x;
  const long long bin_begin = bounds[bin], bin_end = bounds[bin + 1];
  const bool shared = bin_end - bin_begin > csc_bin_chunk;                            // (workgroup-uniform)
  if (!shared && blockIdx.y > 0) return;
  if (shared && bin_begin + static_cast<long long>(blockIdx.y) * csc_bin_chunk >= bin_end) return;
  for (int i = threadIdx.x; i < csc_bin_rows; i += 512) s_acc[i] = 0.0;
  __syncthreads();
  for (long long begin = bin_begin + static_cast<long long>(blockIdx.y) * csc_bin_chunk; begin < bin_end;
       begin += static_cast<long long>(gridDim.y) * csc_bin_chunk) {
    const long long end = begin + csc_bin_chunk < bin_end ? begin + csc_bin_chunk : bin_end;
    for (long long k = begin + threadIdx.x; k < end; k += 512)
      atomicAdd(&s_acc[static_cast<int>(sorted_rows[k]) & (csc_bin_rows - 1)], static_cast<double>(sorted_products[k]));
  }
  __syncthreads();
  const long long row0 = static_cast<long long>(bin) << csc_bin_shift;
  for (int i = threadIdx.x; i < csc_bin_rows; i += 512) {
    if (row0 + i >= rows) break;
    if (s_acc[i] == 0.0) continue;
    if (shared) atomicAdd(&y[row0 + i], static_cast<type_t>(s_acc[i]));
    else y[row0 + i] = y[row0 + i] + static_cast<type_t>(s_acc[i]);  // (this workgroup alone touches the bin's rows)
  }
}

// ---- the binning as kernels of its own (matrices of up to csc_own_max_bins bins = 2^21 rows): no products array, no histogram pre-pass over
// the keys, no library pass -- count (rows of a tile -> LDS histogram -> counts[bin][tile]), scan (a workgroup per bin over the tiles; one
// more over the bins' totals = the bins' bounds), scatter (products computed here; a tile's items ordered by bin in LDS, then written in
// runs), reduce as above.  C2: 0.376 (library pass) -> 0.26 ms (profiles/r06_csc_binned.txt).
constexpr int csc_own_max_bins = 512, csc_tile_threads = 512, csc_tile_ipt = 16, csc_tile = csc_tile_threads * csc_tile_ipt;

template <typename index_t>
__global__ void __launch_bounds__(csc_tile_threads)
csc_bin_count(const int nnz, const int bins, const int tiles, const index_t* __restrict__ row_indices, int* __restrict__ counts) {
  __shared__ int s_hist[csc_own_max_bins];
  for (int i = threadIdx.x; i < bins; i += csc_tile_threads) s_hist[i] = 0;
  __syncthreads();
  const long long base = static_cast<long long>(blockIdx.x) * csc_tile;
#pragma unroll
  for (int u = 0; u < csc_tile_ipt; ++u) {
    const long long k = base + threadIdx.x + static_cast<long long>(u) * csc_tile_threads;
    if (k < nnz) atomicAdd(&s_hist[static_cast<unsigned int>(row_indices[k]) >> csc_bin_shift], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += csc_tile_threads) counts[static_cast<std::size_t>(i) * tiles + blockIdx.x] = s_hist[i];
}

/// Workgroup b: counts[b][0 .. tiles) -> its exclusive prefix sums in place; totals[b] = the bin's products.
__global__ void __launch_bounds__(256)
csc_bin_scan_tiles(const int tiles, int* __restrict__ counts, int* __restrict__ totals) {
  using scan_t = hipcub::BlockScan<int, 256>;
  __shared__ typename scan_t::TempStorage temp;
  __shared__ int s_carry;
  int* row = counts + static_cast<std::size_t>(blockIdx.x) * tiles;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < tiles; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < tiles ? row[i] : 0;
    int ex, sum;
    scan_t(temp).ExclusiveSum(v, ex, sum);
    const int carry = s_carry;
    if (i < tiles) row[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + sum;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}
/// bounds[b] = sum of totals[0 .. b), b in [0, bins]  (one workgroup; bins <= csc_own_max_bins).
__global__ void __launch_bounds__(csc_own_max_bins)
csc_bin_scan_bins(const int bins, const int* __restrict__ totals, int* __restrict__ bounds) {
  using scan_t = hipcub::BlockScan<int, csc_own_max_bins>;
  __shared__ typename scan_t::TempStorage temp;
  const int v = static_cast<int>(threadIdx.x) < bins ? totals[threadIdx.x] : 0;
  int ex, sum;
  scan_t(temp).ExclusiveSum(v, ex, sum);
  if (static_cast<int>(threadIdx.x) < bins) bounds[threadIdx.x] = ex;
  if (threadIdx.x == 0) bounds[bins] = sum;
}

/// A tile of nonzeros -> (row, product) pairs in their bins.  A lane takes 16 consecutive nonzeros (column of the first by a search over the
/// offsets, then a walk: x[col] is a broadcast read), ranks them inside the tile's bins with LDS atomics, the tile's items are laid out by
/// bin in LDS and leave in runs: item i of that order goes to bounds[bin] + counts[bin][tile] + (i - first item of the bin in the tile).
template <typename offset_t, typename type_t>
__global__ void __launch_bounds__(csc_tile_threads)
csc_bin_scatter(const int cols, const int nnz, const int bins, const int tiles, const offset_t* __restrict__ offsets, const int* __restrict__ row_indices,
                const type_t* __restrict__ values, const type_t* __restrict__ x, const int* __restrict__ counts, const int* __restrict__ bounds,
                unsigned int* __restrict__ binned_rows, type_t* __restrict__ binned_products) {
  using scan_t = hipcub::BlockScan<int, csc_tile_threads>;
  __shared__ typename scan_t::TempStorage temp;
  __shared__ int s_hist[csc_own_max_bins], s_first[csc_own_max_bins];
  __shared__ unsigned int s_row[csc_tile];
  __shared__ type_t s_prod[csc_tile];
  for (int i = threadIdx.x; i < csc_own_max_bins; i += csc_tile_threads) s_hist[i] = 0;
  __syncthreads();
  const long long tile_base = static_cast<long long>(blockIdx.x) * csc_tile;
  const long long base_ll = tile_base + static_cast<long long>(threadIdx.x) * csc_tile_ipt;
  unsigned int r[csc_tile_ipt];
  type_t p[csc_tile_ipt];
  int rank[csc_tile_ipt];
  int mine = 0;
  if (base_ll < nnz) {
    const int base = static_cast<int>(base_ll);
    int col = 0, count = cols;  // column of nonzero `base`: last c with offsets[c] <= base
    while (count > 0) {
      const int half = count >> 1;
      const int mid = col + half;
      if (offsets[mid + 1] <= base) {
        col = mid + 1;
        count -= half + 1;
      } else {
        count = half;
      }
    }
    int next = static_cast<int>(offsets[col + 1]);
    type_t xc = x[col];
#pragma unroll
    for (int i = 0; i < csc_tile_ipt; ++i) {
      const int k = base + i;
      if (k >= nnz) break;
      while (k >= next) {  // (empty columns are stepped over)
        ++col;
        next = static_cast<int>(offsets[col + 1]);
        xc = x[col];
      }
      r[i] = static_cast<unsigned int>(row_indices[k]);
      p[i] = values[k] * xc;
      rank[i] = atomicAdd(&s_hist[r[i] >> csc_bin_shift], 1);
      mine = i + 1;
    }
  }
  __syncthreads();
  {  // s_first[b] = first position of bin b in the tile's by-bin order (bins <= 512 = one entry per thread)
    const int v = static_cast<int>(threadIdx.x) < bins ? s_hist[threadIdx.x] : 0;
    int ex;
    scan_t(temp).ExclusiveSum(v, ex);
    s_first[threadIdx.x] = ex;
  }
  __syncthreads();
  for (int i = 0; i < mine; ++i) {
    const int at = s_first[r[i] >> csc_bin_shift] + rank[i];
    s_row[at] = r[i];
    s_prod[at] = p[i];
  }
  __syncthreads();
  const long long left = nnz - tile_base;
  const int items = left < csc_tile ? static_cast<int>(left) : csc_tile;
  for (int i = threadIdx.x; i < items; i += csc_tile_threads) {
    const unsigned int row = s_row[i];
    const int b = static_cast<int>(row >> csc_bin_shift);
    const std::size_t to = static_cast<std::size_t>(bounds[b]) + static_cast<std::size_t>(counts[static_cast<std::size_t>(b) * tiles + blockIdx.x]) +
                           static_cast<std::size_t>(i - s_first[b]);
    binned_rows[to] = row;
    binned_products[to] = s_prod[i];
  }
}

/// Scratch of launch_csc_binned: products + sorted rows + sorted products + bin bounds + the sort's own.
template <typename index_t, typename type_t>
inline std::size_t csc_binned_scratch_bytes(int rows, int nnz) {
  const std::size_t n = static_cast<std::size_t>(nnz > 0 ? nnz : 1), bins = static_cast<std::size_t>(rows) / csc_bin_rows + 2;
  std::size_t sort_bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, static_cast<const unsigned int*>(nullptr), static_cast<unsigned int*>(nullptr),
                                            static_cast<const type_t*>(nullptr), static_cast<type_t*>(nullptr), nnz, csc_bin_shift, 32);
  auto up = [](std::size_t b) { return (b + 255) & ~std::size_t(255); };
  const std::size_t tiles = (n + csc_tile - 1) / csc_tile;
  const std::size_t own = bins <= static_cast<std::size_t>(csc_own_max_bins) ? up(sizeof(int) * bins * tiles) + up(sizeof(int) * (bins + 1)) : 0;
  const std::size_t lib = up(sizeof(type_t) * n) + up(sort_bytes);  // (the library pass: the products array and the sort's own)
  return up(sizeof(type_t) * n) + up(sizeof(index_t) * n) + up(sizeof(int) * (bins + 1)) + (own > lib ? own : lib) + 256;
}

/// y += A x for a CSC matrix by binned products (file section above): y zero-filled by the caller, as for the atomic kernels (a row's
/// sum is ADDED to what y holds).  int row indices.
template <typename offset_t, typename type_t>
int launch_csc_binned(hipStream_t stream, int rows, int cols, int nnz, const offset_t* offsets, const int* row_indices, const type_t* values,
                      const type_t* x, type_t* y, void* scratch) {
  if (rows <= 0) return 0;
  if (nnz <= 0) return 0;
  auto up = [](std::size_t b) { return (b + 255) & ~std::size_t(255); };
  const std::size_t n = static_cast<std::size_t>(nnz);
  const int bins = (rows + csc_bin_rows - 1) / csc_bin_rows;
  char* p = static_cast<char*>(scratch);
  type_t* sorted_products = reinterpret_cast<type_t*>(p); p += up(sizeof(type_t) * n);
  unsigned int* sorted_rows = reinterpret_cast<unsigned int*>(p); p += up(sizeof(int) * n);
  int* bounds = reinterpret_cast<int*>(p); p += up(sizeof(int) * (static_cast<std::size_t>(rows) / csc_bin_rows + 3));
  if (bins <= csc_own_max_bins) {  // the binning as kernels of its own
    const int tiles = static_cast<int>((n + csc_tile - 1) / csc_tile);
    int* counts = reinterpret_cast<int*>(p); p += up(sizeof(int) * static_cast<std::size_t>(bins) * tiles);
    int* totals = reinterpret_cast<int*>(p);
    hipLaunchKernelGGL((csc_bin_count<int>), dim3(tiles), dim3(csc_tile_threads), 0, stream, nnz, bins, tiles, row_indices, counts);
    hipLaunchKernelGGL(csc_bin_scan_tiles, dim3(bins), dim3(256), 0, stream, tiles, counts, totals);
    hipLaunchKernelGGL(csc_bin_scan_bins, dim3(1), dim3(csc_own_max_bins), 0, stream, bins, totals, bounds);
    hipLaunchKernelGGL((csc_bin_scatter<offset_t, type_t>), dim3(tiles), dim3(csc_tile_threads), 0, stream, cols, nnz, bins, tiles, offsets, row_indices, values, x,
                       counts, bounds, sorted_rows, sorted_products);
  } else {  // more rows than that: products, then one library radix pass per 8 bits of the bin number
    type_t* products = reinterpret_cast<type_t*>(p); p += up(sizeof(type_t) * n);
    void* sort_temp = p;
    constexpr int IPT = 8;
    hipLaunchKernelGGL((csc_products<IPT, int, offset_t, type_t>), dim3(math::ceil_div(math::ceil_div(nnz, IPT), 256)), dim3(256), 0, stream, cols, nnz, offsets,
                       values, x, products);
    int top = csc_bin_shift;  // bits of (rows - 1) above the bin
    while (top < 32 && ((static_cast<unsigned int>(rows - 1)) >> top) != 0u) ++top;
    if (top == csc_bin_shift) top = csc_bin_shift + 1;
    std::size_t sort_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, reinterpret_cast<const unsigned int*>(row_indices), sorted_rows, products,
                                                      sorted_products, nnz, csc_bin_shift, top, stream);
    if (e == hipSuccess)
      e = hipcub::DeviceRadixSort::SortPairs(sort_temp, sort_bytes, reinterpret_cast<const unsigned int*>(row_indices), sorted_rows, products, sorted_products,
                                             nnz, csc_bin_shift, top, stream);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((csc_bin_bounds<unsigned int>), dim3(math::ceil_div(bins + 1, 256)), dim3(256), 0, stream, nnz, bins, sorted_rows, bounds);
  }
  // (csc_bin_shares = 32 shares per bin: the host does not know the bins' sizes; a bin of more chunks than that has its chunks dealt to the
  //  shares in turn.  R-MAT scale 20, whose first bins hold the hub rows: 0.845 ms with 8 shares, 0.664 with 32 or 64; C2 unchanged.)
  hipLaunchKernelGGL((csc_reduce_bins<unsigned int, type_t>), dim3(bins, nnz > csc_bin_chunk ? csc_bin_shares : 1), dim3(512), 0, stream, rows, bounds, sorted_rows,
                     sorted_products, y);
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
