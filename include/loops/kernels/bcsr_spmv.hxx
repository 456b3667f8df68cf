/**
 * @file bcsr_spmv.hxx
 * @brief BCSR (R x C dense blocks) SpMV kernels.
 *
 *  - `bcsr_thread_mapped_spmv<R, C>`: the schedule-API kernel -- one thread per block-row over
 *    `layout::bcsr` with `acc[R]` in registers (semantics of the reference kernel,
 *    algorithms/spmv/bcsr_thread_mapped.cuh:36-74; guarded store for rows >= `rows`).
 *  - `bcsr4x4_mfma_spmv`: the CDNA4 path for 4 x 4 blocks.  A 64-lane wavefront owns 16
 *    block-rows; the 4 lanes of slot s own block-row s.  Per step each lane loads ONE 16-byte row
 *    of its slot's current block (row-major blocks, container/bcsr.hxx:13-17 -> lane (s, i) reads
 *    values[blk * 16 + 4 i .. 4 i + 3]) and the slot's 4 x-values; the 4 x 4 block times x[4] is
 *    issued as four chained `v_mfma_f32_4x4x1_16b_f32` (16 independent 4x4 blocks per
 *    instruction: A = column j of the block, B = x[j] broadcast along the output columns), so the
 *    accumulator D[i][*] holds y_i of the slot's block-row and no cross-lane reduction is
 *    needed.  fp32 MFMA is exact fp32 FMA; the kernel stays HBM-bound (68 B per 32 flop).
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#include <loops/schedule.hxx>
#include <loops/container/layout.hxx>
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

template <std::size_t R, std::size_t C, typename setup_t, typename index_t, typename type_t>
__global__ void bcsr_thread_mapped_spmv(setup_t config, std::size_t rows, const index_t* block_col_indices,
                                        const type_t* values, const type_t* x, type_t* y) {
  for (auto br : config.tiles()) {
    type_t acc[R];
#pragma unroll
    for (std::size_t i = 0; i < R; ++i) acc[i] = type_t{0};
    for (auto b : config.atoms(br)) {
      const std::size_t bc = static_cast<std::size_t>(block_col_indices[b]);
      const type_t* block = values + static_cast<std::size_t>(b) * R * C;
#pragma unroll
      for (std::size_t i = 0; i < R; ++i) {
#pragma unroll
        for (std::size_t j = 0; j < C; ++j) acc[i] += block[i * C + j] * x[bc * C + j];
      }
    }
    const std::size_t row_base = static_cast<std::size_t>(br) * R;
#pragma unroll
    for (std::size_t i = 0; i < R; ++i)
      if (row_base + i < rows) y[row_base + i] = acc[i];
  }
}

template <std::size_t R, std::size_t C>
int launch_bcsr_thread_mapped(hipStream_t stream, int rows, int num_block_rows, int num_blocks,
                              const int* block_offsets, const int* block_cols, const float* values, const float* x,
                              float* y) {
  using layout_t = layout::bcsr<int, int>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, int, int, std::size_t, std::size_t,
                                  layout_t>;
  setup_t config(layout_t(block_offsets, num_block_rows, num_blocks));
  constexpr int block = 128;  // bcsr_thread_mapped.cuh:110 (reference hard-codes 128)
  hipLaunchKernelGGL((bcsr_thread_mapped_spmv<R, C, setup_t, int, float>), dim3(math::ceil_div(num_block_rows, block)),
                     dim3(block), 0, stream, config, std::size_t(rows), block_cols, values, x, y);
  return static_cast<int>(hipGetLastError());
}

using f32x4 = float __attribute__((ext_vector_type(4)));

template <int TPB, int UNROLL>
__global__ void __launch_bounds__(TPB)
bcsr4x4_mfma_spmv(const int rows, const int num_block_rows, const int* __restrict__ block_offsets,
                  const int* __restrict__ block_cols, const float* __restrict__ values, const float* __restrict__ x,
                  float* __restrict__ y) {
  const int lane = wave::lane();
  const int slot = lane >> 2;  // which of the wavefront's 16 block-rows
  const int i = lane & 3;      // row of the 4 x 4 block this lane feeds
  const long long gwave = (static_cast<long long>(blockIdx.x) * TPB + threadIdx.x) / wave::size;
  const long long br = gwave * 16 + slot;
  int beg = 0, end = 0;
  if (br < num_block_rows) {
    beg = block_offsets[br];
    end = block_offsets[br + 1];
  }
  const int len = end - beg;
  int maxlen = len;
#pragma unroll
  for (int d = 32; d >= 4; d >>= 1) {
    const int o = __shfl_xor(maxlen, d);
    maxlen = o > maxlen ? o : maxlen;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int safe = len > 0 ? beg : 0;  // any valid block index for masked-off steps
  // Software pipeline over batches of UNROLL blocks: the 16-byte block rows and block columns of
  // batch n + 1 are requested BEFORE the x gathers and MFMAs of batch n, so a wavefront always has
  // one batch of HBM reads in flight behind the batch it is multiplying.
  f32x4 a_next[UNROLL];
  int bc_next[UNROLL];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const bool live = k0 + u < len;
      const int b = live ? beg + k0 + u : safe;
      a_next[u] = *reinterpret_cast<const f32x4*>(values + static_cast<size_t>(b) * 16 + i * 4);
      bc_next[u] = block_cols[b];
      if (!live) a_next[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (maxlen > 0) fetch(0);
  for (int k0 = 0; k0 < maxlen; k0 += UNROLL) {
    f32x4 a[UNROLL];
    f32x4 xv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      a[u] = a_next[u];
      xv[u] = *reinterpret_cast<const f32x4*>(x + static_cast<size_t>(bc_next[u]) * 4);
    }
    if (k0 + UNROLL < maxlen) fetch(k0 + UNROLL);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].x, xv[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].y, xv[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].z, xv[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].w, xv[u].w, acc, 0, 0, 0);
    }
  }
  // D layout of the 4x4x1 16-block form: lane (slot, col) register v = D[v][col]; every column
  // holds the same y (B was broadcast), so column 0's lane stores the block-row's 4 outputs.
  if (i == 0 && br < num_block_rows) {
    const long long r0 = br * 4;
    if (r0 + 3 < rows) {
      *reinterpret_cast<f32x4*>(y + r0) = acc;
    } else {
      if (r0 + 0 < rows) y[r0 + 0] = acc.x;
      if (r0 + 1 < rows) y[r0 + 1] = acc.y;
      if (r0 + 2 < rows) y[r0 + 2] = acc.z;
    }
  }
}

inline int launch_bcsr4x4_mfma(hipStream_t stream, int rows, int num_block_rows, int num_blocks,
                               const int* block_offsets, const int* block_cols, const float* values, const float* x,
                               float* y, int unroll = 8) {
  (void)num_blocks;
  constexpr int TPB = 256;                      // 4 wavefronts = 64 block-rows per workgroup
  constexpr int rows_per_block = TPB / 64 * 16;
  const dim3 grid(math::ceil_div(num_block_rows, rows_per_block)), block(TPB);
  switch (unroll) {
    case 1: hipLaunchKernelGGL((bcsr4x4_mfma_spmv<TPB, 1>), grid, block, 0, stream, rows, num_block_rows, block_offsets, block_cols, values, x, y); break;
    case 2: hipLaunchKernelGGL((bcsr4x4_mfma_spmv<TPB, 2>), grid, block, 0, stream, rows, num_block_rows, block_offsets, block_cols, values, x, y); break;
    default: hipLaunchKernelGGL((bcsr4x4_mfma_spmv<TPB, 8>), grid, block, 0, stream, rows, num_block_rows, block_offsets, block_cols, values, x, y); break;
    case 4: hipLaunchKernelGGL((bcsr4x4_mfma_spmv<TPB, 4>), grid, block, 0, stream, rows, num_block_rows, block_offsets, block_cols, values, x, y); break;
  }
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
