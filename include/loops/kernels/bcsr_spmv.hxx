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
#include <type_traits>

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

template <std::size_t R, std::size_t C, typename type_t>
int launch_bcsr_thread_mapped(hipStream_t stream, int rows, int num_block_rows, int num_blocks,
                              const int* block_offsets, const int* block_cols, const type_t* values, const type_t* x,
                              type_t* y) {
  using layout_t = layout::bcsr<int, int>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, int, int, std::size_t, std::size_t,
                                  layout_t>;
  setup_t config(layout_t(block_offsets, num_block_rows, num_blocks));
  constexpr int block = 128;  // bcsr_thread_mapped.cuh:110 (reference hard-codes 128)
  hipLaunchKernelGGL((bcsr_thread_mapped_spmv<R, C, setup_t, int, type_t>), dim3(math::ceil_div(num_block_rows, block)),
                     dim3(block), 0, stream, config, std::size_t(rows), block_cols, values, x, y);
  return static_cast<int>(hipGetLastError());
}

using f32x4 = float __attribute__((ext_vector_type(4)));

/**
 * @tparam H blocks of ONE block-row a wavefront multiplies per step: the 64 lanes are 16 MFMA
 *           batches of 4 lanes; batch q = (slot q / H, h = q % H) works on block (k * H + h) of block-row
 *           `slot`, so a slot reads H * 64 contiguous bytes per step (H = 1: 64-byte requests from 16
 *           different rows; H = 4: 256-byte requests from 4 rows; H = 16: 1 KB from one row) and the H
 *           partial products of a slot are added with log2(H) cross-lane steps after the loop.
 *
 * A wavefront owns `groups_per_wave` CONSECUTIVE groups of 16 / H block-rows and walks them as ONE software
 * pipeline: the block rows / block columns of batch n + 1 -- which is the first batch of the NEXT group when the
 * current group ends -- are requested before the x gathers and MFMAs of batch n, and the row offsets run two groups
 * ahead.  A wavefront therefore always has one batch of HBM reads in flight behind the batch it is multiplying; with
 * one group per wavefront (the first version of this kernel) a wavefront's life was offsets -> stream -> gather ->
 * MFMA, each waiting for the one before, and only a third of the resident wavefronts had stream requests outstanding
 * at any time (C4: 4.35 TB/s against 6.7 TB/s with the gather removed).
 */
template <int TPB, int UNROLL, int H>
__global__ void __launch_bounds__(TPB)
bcsr4x4_mfma_spmv(const int rows, const int num_block_rows, const int* __restrict__ block_offsets,
                  const int* __restrict__ block_cols, const float* __restrict__ values, const float* __restrict__ x,
                  float* __restrict__ y, const int groups_per_wave) {
  static_assert(H == 1 || H == 2 || H == 4 || H == 8 || H == 16, "H: power of two <= 16");
  constexpr int SLOTS = 16 / H;  // block-rows a wavefront works on at a time
  const int lane = wave::lane();
  const int q = lane >> 2;     // MFMA batch
  const int slot = q / H;      // which of the group's block-rows
  const int h = q % H;         // which of the H concurrent blocks of that block-row
  const int i = lane & 3;      // row of the 4 x 4 block this lane feeds
  const long long gwave = (static_cast<long long>(blockIdx.x) * TPB + threadIdx.x) / wave::size;
  const long long total_groups = (static_cast<long long>(num_block_rows) + SLOTS - 1) / SLOTS;
  long long g = gwave * groups_per_wave;
  long long g_end = g + groups_per_wave;
  g_end = g_end < total_groups ? g_end : total_groups;
  if (g >= g_end) return;  // (wave-uniform)

  struct range_t {
    int beg, len;
  };
  auto load_range = [&](long long grp) {
    const long long br = grp * SLOTS + slot;
    range_t r{0, 0};
    if (grp < g_end && br < num_block_rows) {
      r.beg = block_offsets[br];
      r.len = block_offsets[br + 1] - r.beg;
    }
    return r;
  };
  auto wave_steps = [&](const range_t& r) {  // steps the group needs = max over its slots, at least one
    int steps = (r.len + H - 1) / H;
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) {
      const int o = __shfl_xor(steps, d);
      steps = o > steps ? o : steps;
    }
    return steps > 1 ? steps : 1;
  };
  f32x4 a_next[UNROLL];
  int bc_next[UNROLL];
  auto fetch = [&](const range_t& r, const int k0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int k = (k0 + u) * H + h;
      const bool live = k < r.len;
      const int b = live ? r.beg + k : 0;  // block 0 for masked-off steps (an in-bounds index whenever a block exists)
      // The block stream is read once: with H >= 2 a slot consumes whole 128-byte lines per step, so
      // it is loaded non-temporally and stops evicting x from L1 / L2 (C4: 73.1 -> 67.7 us).  With
      // H = 1 the second half of a line is used by the NEXT step and must stay cached.
      if constexpr (H >= 2) {
        a_next[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(values + static_cast<size_t>(b) * 16 + i * 4));
        bc_next[u] = __builtin_nontemporal_load(block_cols + b);
      } else {
        a_next[u] = *reinterpret_cast<const f32x4*>(values + static_cast<size_t>(b) * 16 + i * 4);
        bc_next[u] = block_cols[b];
      }
      if (!live) a_next[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  range_t cur = load_range(g), nxt = load_range(g + 1);
  int steps = wave_steps(cur);
  fetch(cur, 0);
  for (; g < g_end; ++g) {
    const range_t after = load_range(g + 2);  // offsets run two groups ahead of the MFMAs
    const int next_steps = wave_steps(nxt);
    const bool has_next = g + 1 < g_end;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < steps; k0 += UNROLL) {
      f32x4 a[UNROLL];
      f32x4 xv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        a[u] = a_next[u];
        xv[u] = *reinterpret_cast<const f32x4*>(x + static_cast<size_t>(bc_next[u]) * 4);
      }
      // the next batch's HBM reads go out before this batch's MFMAs: this group's, or the next group's first
      if (k0 + UNROLL < steps) fetch(cur, k0 + UNROLL);
      else if (has_next) fetch(nxt, 0);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].x, xv[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].y, xv[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].z, xv[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].w, xv[u].w, acc, 0, 0, 0);
      }
    }
    // D layout of the 4x4x1 16-block form: lane (batch, col) register v = D[v][col]; every column
    // holds the same 4 partial outputs (B was broadcast).  Add the H partials of a slot.
#pragma unroll
    for (int d = 4; d < 4 * H; d <<= 1) {
      acc.x += __shfl_xor(acc.x, d);
      acc.y += __shfl_xor(acc.y, d);
      acc.z += __shfl_xor(acc.z, d);
      acc.w += __shfl_xor(acc.w, d);
    }
    const long long br = g * SLOTS + slot;
    if (i == 0 && h == 0 && br < num_block_rows) {
      const long long r0 = br * 4;
      if (r0 + 3 < rows) {
        *reinterpret_cast<f32x4*>(y + r0) = acc;
      } else {
        if (r0 + 0 < rows) y[r0 + 0] = acc.x;
        if (r0 + 1 < rows) y[r0 + 1] = acc.y;
        if (r0 + 2 < rows) y[r0 + 2] = acc.z;
      }
    }
    cur = nxt;
    nxt = after;
    steps = next_steps;
  }
}

/// `unroll`: steps in flight (1, 2, 4, 8); `h`: blocks of one block-row per step.  0 = automatic from the
/// mean blocks per block-row m: h = 1 (m < 1.5), 2 (m < 3), else 4; unroll = the power of two covering
/// m / (2 h), at most 8 (C4, m = 16: h = 4, unroll = 2 -- 69.0 us against 72.1 us for unroll = 4 by rocprofv3's kernel
/// duration, profiles/r02_bcsr_c4_kernel_stats.csv).
/// `groups_per_wave`: consecutive groups of 16 / h block-rows one wavefront pipelines through; 0 = automatic (one).
inline int launch_bcsr4x4_mfma(hipStream_t stream, int rows, int num_block_rows, int num_blocks,
                               const int* block_offsets, const int* block_cols, const float* values, const float* x,
                               float* y, int unroll = 0, int h = 0, int groups_per_wave = 0) {
  constexpr int TPB = 256;  // 4 wavefronts
  if (num_block_rows == 0) return 0;
  if (num_blocks == 0) {
    // no block at all: y = 0.  The kernel's masked-off steps read block 0 ("an in-bounds index whenever a block
    // exists"), and the caller may pass null block_cols / values / x for an empty matrix.
    return rows > 0 ? static_cast<int>(hipMemsetAsync(y, 0, sizeof(float) * static_cast<size_t>(rows), stream)) : 0;
  }
  const double mean = static_cast<double>(num_blocks) / num_block_rows;
  if (h == 0) h = mean < 1.5 ? 1 : mean < 3 ? 2 : 4;
  if (unroll == 0) {  // two batches per block-row: the second batch's HBM reads fly behind the first one's gathers + MFMAs
    unroll = 1;
    while (unroll < 8 && 2 * unroll * h < mean) unroll *= 2;
  }
  const long long groups = math::ceil_div(static_cast<long long>(num_block_rows), static_cast<long long>(16 / h));
  // Automatic: ONE group per wavefront.  Measured on C4 (tests/perf/bench_bcsr.py, profiles/r02_bcsr_c4_groups_per_wave.txt):
  // 1 or 2 groups per wavefront are best (70-72 us), 8 and more are slower (74-84 us) -- with 32 resident wavefronts per
  // CU the other wavefronts already cover one wavefront's dependent phases, and fewer, longer wavefronts balance worse.
  if (groups_per_wave <= 0) groups_per_wave = 1;
  if (groups_per_wave > 64) groups_per_wave = 64;
  const long long waves = math::ceil_div(groups, static_cast<long long>(groups_per_wave));
  auto go = [&](auto u_tag, auto h_tag) {
    constexpr int U = decltype(u_tag)::value, HH = decltype(h_tag)::value;
    hipLaunchKernelGGL((bcsr4x4_mfma_spmv<TPB, U, HH>), dim3(static_cast<unsigned>(math::ceil_div(waves, static_cast<long long>(TPB / 64)))),
                       dim3(TPB), 0, stream, rows, num_block_rows, block_offsets, block_cols, values, x, y, groups_per_wave);
  };
  auto with_h = [&](auto u_tag) {
    switch (h) {
      case 1: go(u_tag, std::integral_constant<int, 1>{}); break;
      case 2: go(u_tag, std::integral_constant<int, 2>{}); break;
      case 4: go(u_tag, std::integral_constant<int, 4>{}); break;
      case 8: go(u_tag, std::integral_constant<int, 8>{}); break;
      default: go(u_tag, std::integral_constant<int, 16>{}); break;
    }
  };
  switch (unroll) {
    case 1: with_h(std::integral_constant<int, 1>{}); break;
    case 2: with_h(std::integral_constant<int, 2>{}); break;
    case 4: with_h(std::integral_constant<int, 4>{}); break;
    default: with_h(std::integral_constant<int, 8>{}); break;
  }
  return static_cast<int>(hipGetLastError());
}


// ------------------------------------------------------------------------------------------------------------------
// Coalesced BCSR SpMV for every block shape and both precisions.  What made the 4 x 4 MFMA kernel fast
// was not the MFMA but the block stream: a slot of lanes reads WHOLE lines of consecutive blocks of one block-row, 16
// bytes per lane, non-temporally.  The two kernels below give that stream to the shapes the reference ships and tests
// (examples/spmv/bcsr_thread_mapped.cu:31-32: 2 x 2; unittests/test_spmv_bcsr.cu:24-36: 2 x 2, 3 x 3; every example also
// as .f64) with the block inner product on the VALU:
//
//  * bcsr_vector_mapped_spmv<R, C, T>   R * C * sizeof(T) a power-of-two multiple of 16 bytes (2x2, 4x4, 8x8, ...): the
//    values of a block-row are a flat run of 16-byte vectors; G = H * VPB consecutive lanes take the VPB vectors of H
//    consecutive blocks per step (lane (h, q) = vector q of block k * H + h), so a slot reads G * 16 contiguous bytes.  A
//    vector is a piece of ONE block row (C * sizeof(T) >= 16: row q / PPR, columns (q % PPR) * VW ...) or VW / C whole
//    block rows (2 x 2 fp32: the block IS the vector); the matching x piece is one aligned load.  Partial sums of a row
//    live in lanes that differ in h and in the piece index: log2(H * PPR) xor-shuffles after the loop.
//  * bcsr_block_mapped_spmv<R, C, T>    any other shape (3 x 3: 36-byte blocks): G = H lanes per block-row, lane h owns
//    block k * H + h entirely and reads it with the widest loads 4-byte (8-byte) alignment allows; consecutive lanes
//    read consecutive blocks, i.e. the wavefront still covers one contiguous span per step.
//
// Same result contract as bcsr_thread_mapped_spmv (rows of y >= `rows` are not written); summation order differs from
// the reference's (blocks of a block-row are dealt round-robin to H lanes): bit-exact on exactly summable inputs.
namespace detail {
constexpr bool bcsr_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

/// N consecutive elements at `p` (aligned to sizeof(T) only) with the widest loads: 16-byte pieces, then the rest.
template <int N, bool NT, typename T>
__device__ __forceinline__ void load_run(const T* __restrict__ p, T (&out)[N]) {
  constexpr int VW = 16 / static_cast<int>(sizeof(T));
  using vu = T __attribute__((ext_vector_type(VW), aligned(sizeof(T))));
  constexpr int full = N / VW;
#pragma unroll
  for (int v = 0; v < full; ++v) {
    vu t;
    if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const vu*>(p + v * VW));
    else t = *reinterpret_cast<const vu*>(p + v * VW);
#pragma unroll
    for (int e = 0; e < VW; ++e) out[v * VW + e] = t[e];
  }
  constexpr int rem = N - full * VW;
  if constexpr (rem == 1) {
    if constexpr (NT) out[N - 1] = __builtin_nontemporal_load(p + N - 1);
    else out[N - 1] = p[N - 1];
  } else if constexpr (rem > 1) {  // one 8- or 12-byte load for the tail
    using ru = T __attribute__((ext_vector_type(rem), aligned(sizeof(T))));
    ru t;
    if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const ru*>(p + full * VW));
    else t = *reinterpret_cast<const ru*>(p + full * VW);
#pragma unroll
    for (int e = 0; e < rem; ++e) out[full * VW + e] = t[e];
  }
}
}  // namespace detail

template <int R, int C, typename T>
struct bcsr_vector_shape {
  static constexpr int VW = 16 / static_cast<int>(sizeof(T));  ///< elements per 16-byte vector
  static constexpr int BE = R * C;
  static constexpr bool pieces = C % VW == 0;                   ///< a vector is a piece of one block row
  static constexpr bool whole_rows = !pieces && VW % C == 0;    ///< a vector holds VW / C whole block rows
  static constexpr bool ok = BE % VW == 0 && detail::bcsr_pow2(BE / VW) && BE / VW <= 64 &&
                             (pieces ? detail::bcsr_pow2(C / VW) : whole_rows);
  static constexpr int VPB = ok ? BE / VW : 1;                  ///< vectors per block
  static constexpr int PPR = pieces ? C / VW : 1;               ///< pieces per block row
  static constexpr int RPV = pieces ? 1 : VW / C;               ///< block rows per vector
};

template <int R, int C, typename T, int TPB, int H, int U>
__global__ void __launch_bounds__(TPB)
bcsr_vector_mapped_spmv(const int rows, const int num_block_rows, const int* __restrict__ block_offsets,
                        const int* __restrict__ block_cols, const T* __restrict__ values, const T* __restrict__ x,
                        T* __restrict__ y) {
  using shape = bcsr_vector_shape<R, C, T>;
  static_assert(shape::ok, "bcsr_vector_mapped_spmv: block bytes must be a power-of-two multiple of 16");
  constexpr int VW = shape::VW, VPB = shape::VPB, PPR = shape::PPR, RPV = shape::RPV, BE = shape::BE;
  constexpr int G = H * VPB;        // lanes per block-row
  static_assert(G <= wave::size && detail::bcsr_pow2(H), "H * vectors per block must fit a wavefront");
  constexpr int SLOTS = wave::size / G;
  constexpr bool NT = G >= 8;       // whole 128-byte lines per slot and step: read once, keep them out of the caches
  using vec_t = T __attribute__((ext_vector_type(VW)));
  using vec_ld_t = T __attribute__((ext_vector_type(VW), aligned(sizeof(T))));  // (bases are element-aligned at least)
  const int lane = wave::lane();
  const int q = lane % VPB;         // vector of the block
  const int h = (lane / VPB) % H;   // which of the H concurrent blocks
  const int slot = lane / G;
  const long long gwave = (static_cast<long long>(blockIdx.x) * TPB + threadIdx.x) / wave::size;
  const long long br = gwave * SLOTS + slot;
  if (gwave * SLOTS >= num_block_rows) return;  // (wave-uniform)
  int beg = 0, len = 0;
  if (br < num_block_rows) {
    beg = block_offsets[br];
    len = block_offsets[br + 1] - beg;
  }
  int steps = (len + H - 1) / H;
#pragma unroll
  for (int d = wave::size / 2; d >= G; d >>= 1) {
    const int o = __shfl_xor(steps, d);
    steps = o > steps ? o : steps;
  }
  T acc[RPV];
#pragma unroll
  for (int r = 0; r < RPV; ++r) acc[r] = T(0);
  for (int k0 = 0; k0 < steps; k0 += U) {
    vec_t a[U];
    int bc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (k0 + u) * H + h;
      const bool live = k < len;
      const int b = live ? beg + k : 0;  // block 0 for masked-off steps (in bounds: the launcher handles num_blocks == 0)
      const vec_ld_t* src = reinterpret_cast<const vec_ld_t*>(values + static_cast<size_t>(b) * BE + q * VW);
      if constexpr (NT) {
        a[u] = __builtin_nontemporal_load(src);
        bc[u] = __builtin_nontemporal_load(block_cols + b);
      } else {
        a[u] = *src;
        bc[u] = block_cols[b];
      }
      if (!live) a[u] = vec_t(T(0));
    }
    if constexpr (shape::pieces) {
      vec_t xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        xv[u] = *reinterpret_cast<const vec_ld_t*>(x + static_cast<size_t>(bc[u]) * C + (q % PPR) * VW);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[0] += a[u][e] * xv[u][e];
      }
    } else {
      T xs[U][C];
#pragma unroll
      for (int u = 0; u < U; ++u) detail::load_run<C, false>(x + static_cast<size_t>(bc[u]) * C, xs[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < RPV; ++r) {
#pragma unroll
          for (int j = 0; j < C; ++j) acc[r] += a[u][r * C + j] * xs[u][j];
        }
      }
    }
  }
  // partial sums of one row sit in the lanes of its PPR pieces (neighbouring q) and in the H concurrent blocks
#pragma unroll
  for (int r = 0; r < RPV; ++r) {
#pragma unroll
    for (int d = 1; d < PPR; d <<= 1) acc[r] += __shfl_xor(acc[r], d);
#pragma unroll
    for (int d = VPB; d < G; d <<= 1) acc[r] += __shfl_xor(acc[r], d);
  }
  if (h == 0 && q % PPR == 0 && br < num_block_rows) {
    const long long r0 = br * R + static_cast<long long>(q / PPR) * RPV;
#pragma unroll
    for (int r = 0; r < RPV; ++r)
      if (r0 + r < rows) y[r0 + r] = acc[r];
  }
}

template <int R, int C, typename T, int TPB, int H, int U>
__global__ void __launch_bounds__(TPB)
bcsr_block_mapped_spmv(const int rows, const int num_block_rows, const int* __restrict__ block_offsets,
                       const int* __restrict__ block_cols, const T* __restrict__ values, const T* __restrict__ x,
                       T* __restrict__ y) {
  static_assert(H <= wave::size && detail::bcsr_pow2(H), "H: power of two <= 64");
  constexpr int BE = R * C;
  constexpr int SLOTS = wave::size / H;
  constexpr bool NT = H * BE * static_cast<int>(sizeof(T)) >= 128;
  const int lane = wave::lane();
  const int h = lane % H;
  const int slot = lane / H;
  const long long gwave = (static_cast<long long>(blockIdx.x) * TPB + threadIdx.x) / wave::size;
  const long long br = gwave * SLOTS + slot;
  if (gwave * SLOTS >= num_block_rows) return;  // (wave-uniform)
  int beg = 0, len = 0;
  if (br < num_block_rows) {
    beg = block_offsets[br];
    len = block_offsets[br + 1] - beg;
  }
  int steps = (len + H - 1) / H;
#pragma unroll
  for (int d = wave::size / 2; d >= H; d >>= 1) {
    const int o = __shfl_xor(steps, d);
    steps = o > steps ? o : steps;
  }
  T acc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) acc[i] = T(0);
  for (int k0 = 0; k0 < steps; k0 += U) {
    T a[U][BE];
    int bc[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (k0 + u) * H + h;
      live[u] = k < len;
      const int b = live[u] ? beg + k : 0;
      detail::load_run<BE, NT>(values + static_cast<size_t>(b) * BE, a[u]);
      bc[u] = block_cols[b];
    }
    T xs[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u) detail::load_run<C, false>(x + static_cast<size_t>(bc[u]) * C, xs[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < C; ++j) s += a[u][i * C + j] * xs[u][j];
        acc[i] += live[u] ? s : T(0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
#pragma unroll
    for (int d = 1; d < H; d <<= 1) acc[i] += __shfl_xor(acc[i], d);
  }
  if (h == 0 && br < num_block_rows) {
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (br * R + i < rows) y[br * R + i] = acc[i];
  }
}

/// Coalesced BCSR SpMV for block shape R x C (vector-mapped where the block bytes allow, block-mapped otherwise).
/// `h`: blocks of one block-row per step (1, 4, 16; 0 = automatic, from the shape and the mean block-row length);
/// `unroll`: steps in flight (1, 2, 4; 0 = automatic).
template <int R, int C, typename T>
int launch_bcsr_coalesced(hipStream_t stream, int rows, int num_block_rows, int num_blocks, const int* block_offsets,
                          const int* block_cols, const T* values, const T* x, T* y, int h = 0, int unroll = 0) {
  constexpr int TPB = 256;
  if (num_block_rows == 0) return 0;
  if (num_blocks == 0)
    return rows > 0 ? static_cast<int>(hipMemsetAsync(y, 0, sizeof(T) * static_cast<size_t>(rows), stream)) : 0;
  using shape = bcsr_vector_shape<R, C, T>;
  constexpr bool vector_path = shape::ok;
  constexpr int lanes_per_block = vector_path ? shape::VPB : 1;
  constexpr int h_cap = wave::size / lanes_per_block < 16 ? wave::size / lanes_per_block : 16;
  const double mean = static_cast<double>(num_blocks) / num_block_rows;
  // Measured on C4-sized inputs of every compiled shape (tests/perf/bench_bcsr_shapes.py --explicit, profiles/r03_bcsr_shapes.json):
  // vector-mapped kernels are best with 4 blocks of a block-row per step and every step of the row in flight (h = 4, u = 4:
  // 4x4 fp32 62 us against 65-77 for the other shapes, 8x8 fp32 192 against 208-241, 4x4 fp64 119 against 121-138); blocks
  // of one vector (2x2 fp32) whose stream exceeds the Infinity Cache prefer whole 256-byte runs per slot (h = 16, u = 1:
  // 158 against 175 us); block-mapped kernels (3x3: a lane reads a whole 36- / 72-byte block) want 16 consecutive blocks per
  // step so that a wavefront instruction covers one contiguous span (h = 16, u = 2: 64 against 78-99 us).
  if (h == 0) {
    if constexpr (!vector_path) {
      h = mean >= 8 ? 16 : mean >= 2 ? 4 : 1;
      if (unroll == 0) unroll = 2;  // (also at h = 16 with 16 blocks per row: the masked second step costs less than the shorter pipeline)
    } else {
      const double stream_bytes = static_cast<double>(num_blocks) * (R * C * sizeof(T) + 4);
      if (lanes_per_block == 1 && stream_bytes > 256e6 && mean >= 12) h = 16;
      else h = mean >= 3 ? 4 : 1;
    }
  }
  if (h > h_cap) h = h_cap;
  h = h >= 16 ? 16 : h >= 4 ? 4 : 1;  // compiled: 1, 4, 16
  if (h > h_cap) h = h_cap >= 4 ? 4 : 1;
  if (unroll == 0) {
    const double steps = mean / h;
    unroll = steps > 2 ? 4 : steps > 1 ? 2 : 1;
  }
  unroll = unroll >= 4 ? 4 : unroll >= 2 ? 2 : 1;
  auto go = [&](auto h_tag, auto u_tag) {
    constexpr int HH = decltype(h_tag)::value, UU = decltype(u_tag)::value;
    constexpr int slots = wave::size / (HH * lanes_per_block);
    const long long waves = math::ceil_div(static_cast<long long>(num_block_rows), static_cast<long long>(slots));
    const dim3 grid(static_cast<unsigned>(math::ceil_div(waves, static_cast<long long>(TPB / wave::size))));
    if constexpr (vector_path)
      hipLaunchKernelGGL((bcsr_vector_mapped_spmv<R, C, T, TPB, HH, UU>), grid, dim3(TPB), 0, stream, rows, num_block_rows,
                         block_offsets, block_cols, values, x, y);
    else
      hipLaunchKernelGGL((bcsr_block_mapped_spmv<R, C, T, TPB, HH, UU>), grid, dim3(TPB), 0, stream, rows, num_block_rows,
                         block_offsets, block_cols, values, x, y);
  };
  auto with_u = [&](auto h_tag) {
    switch (unroll) {
      case 1: go(h_tag, std::integral_constant<int, 1>{}); break;
      case 2: go(h_tag, std::integral_constant<int, 2>{}); break;
      default: go(h_tag, std::integral_constant<int, 4>{}); break;
    }
  };
  if (h == 16) {
    if constexpr (h_cap >= 16) with_u(std::integral_constant<int, 16>{});
  } else if (h == 4) {
    if constexpr (h_cap >= 4) with_u(std::integral_constant<int, 4>{});
  } else {
    with_u(std::integral_constant<int, 1>{});
  }
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
