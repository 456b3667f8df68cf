/**
 * @file dia_spmv.hxx
 * @brief Tuned DIA SpMV.  DIA as the reference stores it: `diag_offsets[d]` = (col - row) of stored
 * diagonal d, ascending; `values` column-major, values[d * stride + r], stride = rows (container/dia.hxx:69-230).
 *
 * The reference kernel (algorithms/spmv/dia_thread_mapped.cuh:36-58) is one lane per row walking the diagonals:
 * coalesced (neighbouring lanes read neighbouring cells of a diagonal) but 4 bytes per lane and load, one diagonal
 * in flight, and the diagonal's offset re-read by every lane.  `dia_row4_spmv`: a lane owns FOUR consecutive rows --
 * one 16-byte load per diagonal (`global_load_dwordx4`; stride and bases 16-byte aligned), the matching four x values
 * are consecutive too -- U (= 2) diagonals in flight, the offsets are wave-uniform scalar loads, y leaves as one 16-byte
 * store.  A pure stream: bound by HBM (4 or 8 B per cell + x from L2).
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/util/math.hxx>

namespace loops {
namespace kernels {

/// @tparam U diagonals in flight; VEC: stride % 4 == 0 and `values` / `y` 16-byte aligned (else 4 scalar loads).
template <int U, bool VEC, typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
dia_row4_spmv(const int rows, const int cols, const std::size_t stride, const int num_diagonals,
              const index_t* __restrict__ diag_offsets, const type_t* __restrict__ values, const type_t* __restrict__ x,
              type_t* __restrict__ y) {
  const int r0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (r0 >= rows) return;
  const bool full = r0 + 3 < rows;  // all four rows exist (false only for the last lane when rows % 4 != 0)
  type_t acc[4] = {type_t(0), type_t(0), type_t(0), type_t(0)};
  for (int d0 = 0; d0 < num_diagonals; d0 += U) {
    type_t v[U][4];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = d0 + u < num_diagonals ? d0 + u : num_diagonals - 1;  // clamped: surplus slots repeat the last diagonal
      off[u] = static_cast<int>(diag_offsets[d]);
      const type_t* cell = values + static_cast<std::size_t>(d) * stride + r0;
      if (VEC && full) {
        detail::load4<type_t, false>(cell, v[u]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = r0 + k < rows ? cell[k] : type_t(0);
      }
    }
    // the four x values of a lane are consecutive (x[r0 + off .. r0 + off + 3]): one 16-byte load at 4-byte alignment
    // when all four columns are inside the matrix, element-wise otherwise (cells outside are skipped, as in the reference)
    type_t xv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long c0 = static_cast<long long>(r0) + off[u];
      if (full && c0 >= 0 && c0 + 3 < cols) {
        using v4u = type_t __attribute__((ext_vector_type(4), aligned(sizeof(type_t))));
        const v4u t = *reinterpret_cast<const v4u*>(x + c0);
        xv[u][0] = t.x;
        xv[u][1] = t.y;
        xv[u][2] = t.z;
        xv[u][3] = t.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const long long c = c0 + k;
          const bool ok = c >= 0 && c < cols && r0 + k < rows;
          xv[u][k] = ok ? x[ok ? c : 0] : type_t(0);
          if (!ok) v[u][k] = type_t(0);  // (an x of inf / NaN outside the matrix must not meet a stored 0)
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (d0 + u < num_diagonals) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += v[u][k] * xv[u][k];
      }
    }
  }
  if (VEC && full) {
    using v4 = type_t __attribute__((ext_vector_type(4)));
    *reinterpret_cast<v4*>(y + r0) = v4{acc[0], acc[1], acc[2], acc[3]};
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (r0 + k < rows) y[r0 + k] = acc[k];
  }
}

/// Reference-shaped: one lane per row (dia_thread_mapped.cuh:36-58 semantics), raw pointers.
template <typename index_t, typename type_t>
__global__ void __launch_bounds__(128)
dia_thread_spmv(const int rows, const int cols, const std::size_t stride, const int num_diagonals,
                const index_t* __restrict__ diag_offsets, const type_t* __restrict__ values, const type_t* __restrict__ x,
                type_t* __restrict__ y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  type_t acc = type_t(0);
  for (int d = 0; d < num_diagonals; ++d) {
    const long long c = static_cast<long long>(r) + static_cast<long long>(diag_offsets[d]);
    if (c >= 0 && c < cols) acc += values[static_cast<std::size_t>(d) * stride + r] * x[c];
  }
  y[r] = acc;
}

template <typename index_t, typename type_t>
int launch_dia_thread(hipStream_t stream, int rows, int cols, std::size_t stride, int num_diagonals,
                      const index_t* diag_offsets, const type_t* values, const type_t* x, type_t* y) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL((dia_thread_spmv<index_t, type_t>), dim3(math::ceil_div(rows, 128)), dim3(128), 0, stream, rows, cols,
                     stride, num_diagonals, diag_offsets, values, x, y);
  return static_cast<int>(hipGetLastError());
}

template <typename index_t, typename type_t>
int launch_dia_row4(hipStream_t stream, int rows, int cols, std::size_t stride, int num_diagonals,
                    const index_t* diag_offsets, const type_t* values, const type_t* x, type_t* y) {
  if (rows == 0) return 0;
  const bool vec = stride % 4 == 0 && ((reinterpret_cast<std::uintptr_t>(values) | reinterpret_cast<std::uintptr_t>(y)) & 15u) == 0;
  const dim3 grid(math::ceil_div(math::ceil_div(rows, 4), 256)), block(256);
  // Diagonals in flight per lane: 2 -- measured 2 / 4 / 8 / 16 / 32 on 2^20 rows x 33 diagonals (138 MB): 5.9 / 5.7 / 5.0 /
  // 3.8 / 3.1 TB/s in f32 (registers, i.e. resident wavefronts, matter more than loads per lane; 2^23 rows, 1.1 GB: 5.6 / 5.3
  // / 5.1); f64 is within 3 % for 1 / 2 / 4.
  constexpr int U = 2;
  if (vec)
    hipLaunchKernelGGL((dia_row4_spmv<U, true, index_t, type_t>), grid, block, 0, stream, rows, cols, stride, num_diagonals,
                       diag_offsets, values, x, y);
  else
    hipLaunchKernelGGL((dia_row4_spmv<U, false, index_t, type_t>), grid, block, 0, stream, rows, cols, stride, num_diagonals,
                       diag_offsets, values, x, y);
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
