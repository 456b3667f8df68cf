/**
 * @file ell_spmv.hxx
 * @brief ELL SpMV kernels.  ELL here is the reference's layout: two ROW-major dense
 * arrays of rows * pitch entries, padding marked by a negative column index (container/ell.hxx:31-55).
 *
 *  - `ell_thread_spmv`: one lane per row walking its `pitch` slots (the shape of the reference kernel,
 *    algorithms/spmv/ell_thread_mapped.cuh:36-60) -- adjacent lanes are `pitch` entries apart, so every
 *    load instruction touches 64 different lines.
 *  - `ell_row_split_spmv<G>`: G lanes per row, each reading 4 consecutive slots with one 16-byte load
 *    (a row is read as contiguous G * 16-byte runs), partial sums combined with log2(G) cross-lane
 *    steps; a wavefront covers 64 / G rows.  G = the power of two covering pitch / 4, at most 64.
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

template <typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
ell_thread_spmv(const std::size_t rows, const std::size_t pitch, const index_t* __restrict__ indices,
                const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  const std::size_t row = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  type_t sum = type_t(0);
  for (std::size_t j = 0; j < pitch; ++j) {
    const index_t col = indices[row * pitch + j];
    if (col >= 0) sum += values[row * pitch + j] * x[col];
  }
  y[row] = sum;
}

/// @tparam G lanes per row (power of two <= 64); VEC: pitch % 4 == 0 and 16-byte aligned arrays.
template <int G, bool VEC, typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
ell_row_split_spmv(const std::size_t rows, const std::size_t pitch, const index_t* __restrict__ indices,
                   const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  const std::size_t gid = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const std::size_t row = gid / G;
  const int l = static_cast<int>(gid % G);
  type_t sum = type_t(0);
  if (row < rows) {
    const index_t* __restrict__ ri = indices + row * pitch;
    const type_t* __restrict__ rv = values + row * pitch;
    for (std::size_t j = static_cast<std::size_t>(l) * 4; j < pitch; j += static_cast<std::size_t>(G) * 4) {
      index_t c[4];
      type_t v[4];
      if (VEC) {
        detail::load4<index_t, false>(ri + j, c);
        detail::load4<type_t, false>(rv + j, v);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = j + k < pitch;
          c[k] = ok ? ri[j + k] : index_t(-1);
          v[k] = ok ? rv[j + k] : type_t(0);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c[k] >= 0) sum += v[k] * x[c[k]];
    }
  }
#pragma unroll
  for (int d = 1; d < G; d <<= 1) sum += __shfl_xor(sum, d);
  if (l == 0 && row < rows) y[row] = sum;
}

template <typename index_t, typename type_t>
int launch_ell_thread(hipStream_t stream, std::size_t rows, std::size_t pitch, const index_t* indices,
                      const type_t* values, const type_t* x, type_t* y) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL((ell_thread_spmv<index_t, type_t>), dim3(static_cast<unsigned>(math::ceil_div(rows, std::size_t(256)))),
                     dim3(256), 0, stream, rows, pitch, indices, values, x, y);
  return static_cast<int>(hipGetLastError());
}

template <typename index_t, typename type_t>
int launch_ell_row_split(hipStream_t stream, std::size_t rows, std::size_t pitch, const index_t* indices,
                         const type_t* values, const type_t* x, type_t* y) {
  if (rows == 0) return 0;
  const bool vec = pitch % 4 == 0 &&
                   ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  int g = 1;
  while (g < 64 && static_cast<std::size_t>(g) * 4 < pitch) g *= 2;
  auto go = [&](auto width) {
    constexpr int G = decltype(width)::value;
    const dim3 grid(static_cast<unsigned>(math::ceil_div(rows * G, std::size_t(256)))), block(256);
    if (vec) hipLaunchKernelGGL((ell_row_split_spmv<G, true, index_t, type_t>), grid, block, 0, stream, rows, pitch, indices, values, x, y);
    else hipLaunchKernelGGL((ell_row_split_spmv<G, false, index_t, type_t>), grid, block, 0, stream, rows, pitch, indices, values, x, y);
  };
  switch (g) {
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    case 8: go(std::integral_constant<int, 8>{}); break;
    case 16: go(std::integral_constant<int, 16>{}); break;
    case 32: go(std::integral_constant<int, 32>{}); break;
    default: go(std::integral_constant<int, 64>{}); break;
  }
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
