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

}  // namespace kernels
}  // namespace loops
