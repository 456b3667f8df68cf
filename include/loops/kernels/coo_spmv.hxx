/**
 * @file coo_spmv.hxx
 * @brief COO SpMV kernels.
 *
 *  - `coo_atom_spmv`: one lane per nonzero, one atomicAdd per nonzero -- the shape of the reference
 *    kernel (algorithms/spmv/coo_thread_mapped.cuh:37-60) on raw pointers.
 *  - `coo_runs_spmv`: a lane owns IPT consecutive nonzeros, read with 16-byte loads; runs of equal row
 *    indices are summed in registers, stitched across the 64 lanes with a segmented prefix sum, and
 *    ONE atomicAdd per run and wavefront reaches y.  Correct for any ordering of the triplets; for
 *    row-sorted COO (what the Matrix-Market loader and coo_t::sort_by_row produce) the number of
 *    atomics drops from nnz to about rows + nnz / (64 IPT).  y must be zero-filled (same
 *    precondition as the reference).
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/kernels/run_stitch.hxx>
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

template <typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
coo_atom_spmv(const std::size_t nnz, const index_t* __restrict__ row_indices, const index_t* __restrict__ col_indices,
              const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  const std::size_t i = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < nnz) atomicAdd(&y[row_indices[i]], values[i] * x[col_indices[i]]);
}

/// @tparam IPT nonzeros per lane (multiple of 4); VEC: all three arrays are 16-byte aligned.
template <int IPT, bool VEC, typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
coo_runs_spmv(const std::size_t nnz, const index_t* __restrict__ row_indices, const index_t* __restrict__ col_indices,
              const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  static_assert(IPT % 4 == 0, "IPT: multiple of 4");
  const std::size_t base = (static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * IPT;
  index_t r[IPT], c[IPT];
  type_t v[IPT];
  // (no early exit: every lane takes part in the wavefront stitch; lanes past the end carry row -1)
  const bool full = base + IPT <= nnz;
  if (VEC && full) {
#pragma unroll
    for (int k = 0; k < IPT; k += 4) {
      index_t r4[4], c4[4];
      type_t v4[4];
      detail::load4<index_t, false>(row_indices + base + k, r4);
      detail::load4<index_t, false>(col_indices + base + k, c4);
      detail::load4<type_t, false>(values + base + k, v4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[k + j] = r4[j];
        c[k + j] = c4[j];
        v[k + j] = v4[j];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const bool ok = base + k < nnz;
      r[k] = ok ? row_indices[base + k] : index_t(-1);  // base may be >= nnz: all lanes stay for the stitch
      c[k] = ok ? col_indices[base + k] : index_t(0);
      v[k] = ok ? values[base + k] : type_t(0);
    }
  }
  type_t p[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) p[k] = (full || base + k < nnz) ? v[k] * x[c[k]] : type_t(0);
  add_row_runs<IPT>(r, p, y);
}

template <typename index_t, typename type_t>
int launch_coo_atom(hipStream_t stream, std::size_t nnz, const index_t* row_indices, const index_t* col_indices,
                    const type_t* values, const type_t* x, type_t* y) {
  if (nnz == 0) return 0;
  hipLaunchKernelGGL((coo_atom_spmv<index_t, type_t>), dim3(static_cast<unsigned>(math::ceil_div(nnz, std::size_t(256)))),
                     dim3(256), 0, stream, nnz, row_indices, col_indices, values, x, y);
  return static_cast<int>(hipGetLastError());
}

template <typename index_t, typename type_t>
int launch_coo_runs(hipStream_t stream, std::size_t nnz, const index_t* row_indices, const index_t* col_indices,
                    const type_t* values, const type_t* x, type_t* y) {
  if (nnz == 0) return 0;
  constexpr int IPT = 8;
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(row_indices) | reinterpret_cast<std::uintptr_t>(col_indices) |
                         reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  const dim3 grid(static_cast<unsigned>(math::ceil_div(nnz, std::size_t(256) * IPT))), block(256);
  if (aligned)
    hipLaunchKernelGGL((coo_runs_spmv<IPT, true, index_t, type_t>), grid, block, 0, stream, nnz, row_indices, col_indices,
                       values, x, y);
  else
    hipLaunchKernelGGL((coo_runs_spmv<IPT, false, index_t, type_t>), grid, block, 0, stream, nnz, row_indices,
                       col_indices, values, x, y);
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
