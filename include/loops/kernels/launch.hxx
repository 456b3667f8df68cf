/**
 * @file launch.hxx
 * @brief Raw-pointer launchers of the CSR SpMV kernels.  Both front ends -- the container
 * wrappers `algorithms::spmv::*` (the .cuh files under include/loops/algorithms/spmv) and the C ABI
 * (loops_amd/csrc/loops_c_abi.hip) -- go through these, so there is exactly one launch
 * configuration per kernel.  All launchers are asynchronous on `stream` and return the
 * hipError_t of the launch (0 = success).
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_runtime.h>

#include <loops/schedule.hxx>
#include <loops/algorithms/spmv/launch_box.hxx>
#include <loops/kernels/csr_spmv.hxx>
#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/util/wave.hxx>
#include <loops/kernels/merge_path_spmm.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/launch_box.hxx>
#include <loops/util/math.hxx>

namespace loops {
namespace kernels {

inline int launch_status() { return static_cast<int>(hipGetLastError()); }

/// Device-side scratch of the fused merge-path SpMV for one (matrix structure, tile shape).
struct merge_plan_view {
  coord_t* coords;      ///< M + 1 per-workgroup start coordinates
  int* carry_row;       ///< M carry-out rows
  void* carry_val;      ///< M carry-out partial sums (value type of the SpMV)
  int num_merge_tiles;  ///< M
  bool self_complete = false;  ///< every tile head <= TPB (launch_merge_path_head_check): no carries, no fix-up
  const int* head_start = nullptr;  ///< M first-nonzero-of-the-first-row entries (self-completing plans)
};

/// Sets *flag_dev (zeroed here) to 1 if some merge tile starts more than `limit` nonzeros inside a row.
template <typename offset_t>
int launch_merge_path_head_check(hipStream_t stream, const coord_t* coords, int num_merge_tiles, int rows,
                                 const offset_t* offsets, int limit, int* flag_dev, int* head_start) {
  hipError_t e = hipMemsetAsync(flag_dev, 0, sizeof(int), stream);
  if (e != hipSuccess) return static_cast<int>(e);
  if (num_merge_tiles == 0) return 0;
  hipLaunchKernelGGL(merge_path_head_check<offset_t>, dim3(math::ceil_div(num_merge_tiles, 256)), dim3(256), 0, stream,
                     coords, num_merge_tiles, rows, offsets, limit, flag_dev, head_start);
  return launch_status();
}

/// merge_path_coordinates_of with a WAVEFRONT per coordinate: a 64-ary search -- every lane probes one row end, a ballot narrows
/// the range 64-fold -- so a table over 2^20 rows costs 4 dependent memory round trips instead of 20 (the pre-pass of every
/// plan-less call: 7.2 -> ~3 us on C2, profiles/r06_oneshot_kernel_stats.csv).  Same split, same coordinates (search.hxx).
template <typename row_end_t>
__global__ void __launch_bounds__(256)
merge_path_coordinates_wide(const row_end_t row_end, const int rows, const int nnz, const int tile_items, const int num_merge_tiles,
                            coord_t* __restrict__ coords) {
  const int i = blockIdx.x * (256 / wave::size) + static_cast<int>(threadIdx.x) / wave::size;  // (wavefront-uniform)
  if (i > num_merge_tiles) return;
  const int lane = wave::lane();
  const int d = static_cast<int>(static_cast<long long>(i) * tile_items);  // int, like the reference (search.hxx:46-47)
  int lo = d - nnz > 0 ? d - nnz : 0;  // the first m in [lo, hi) with row_end(m) > d - m - 1, or hi
  int hi = d < rows ? d : rows;
  while (hi > lo) {
    const int span = hi - lo;
    const int step = (span + wave::size - 1) / wave::size;
    const int m = lo + lane * step;
    const bool before = m < hi && static_cast<int>(row_end(m)) <= d - m - 1;  // (true: the split lies behind m)
    const int trues = __popcll(__ballot(before));                             // monotone predicate: the trues are lanes 0 .. trues - 1
    const int new_lo = trues > 0 ? lo + (trues - 1) * step + 1 : lo;
    const int new_hi = trues < wave::size && lo + trues * step < hi ? lo + trues * step : hi;
    lo = new_lo;
    hi = step == 1 ? new_lo : new_hi;  // (step 1: every candidate was probed)
  }
  if (lane == 0) coords[i] = coord_t{static_cast<unsigned int>(lo < rows ? lo : rows), static_cast<unsigned int>(d - lo)};
}

/// coords[i] = merge-path split at diagonal i * tpb * ipt, i in [0, M].
template <typename offset_t>
int launch_merge_path_coordinates(hipStream_t stream, const offset_t* offsets, int rows, int nnz, int tile_items,
                                  int num_merge_tiles, coord_t* coords) {
  const int n = num_merge_tiles + 1;
  if (rows >= 4096)  // (a wavefront per coordinate pays once the lane-per-coordinate search is more than ~12 round trips deep)
    hipLaunchKernelGGL(merge_path_coordinates_wide<csr_row_end<offset_t>>, dim3(math::ceil_div(n, 256 / wave::size)), dim3(256), 0, stream,
                       csr_row_end<offset_t>{offsets}, rows, nnz, tile_items, num_merge_tiles, coords);
  else
    hipLaunchKernelGGL(merge_path_coordinates_of<csr_row_end<offset_t>>, dim3(math::ceil_div(n, 256)), dim3(256), 0, stream,
                       csr_row_end<offset_t>{offsets}, rows, nnz, tile_items, num_merge_tiles, coords);
  return launch_status();
}

/// The same table for an ELL matrix (row r ends at (r + 1) * pitch): no offsets array.
inline int launch_merge_path_coordinates_ell(hipStream_t stream, int rows, int pitch, int tile_items, int num_merge_tiles,
                                             coord_t* coords) {
  const int n = num_merge_tiles + 1;
  hipLaunchKernelGGL(merge_path_coordinates_of<ell_row_end>, dim3(math::ceil_div(n, 256)), dim3(256), 0, stream,
                     ell_row_end{pitch}, rows, rows * pitch, tile_items, num_merge_tiles, coords);
  return launch_status();
}

/// Fused merge-path SpMV (+ fix-up).  stages: bit 0 = tile kernel, bit 1 = fix-up.
/// `planned`: the product comes from an SpMV plan handle -- the same code under its own kernel symbol (profile attribution).
/// MASK (default): bit-mask split instead of the per-thread search (merge_tile_engine<..., MASK = true>):
/// 1-2 % faster on every input measured (C2 103.3 -> 102.5 us, band-8192 50.4 -> 49.5, runs 43.7 -> 42.7).
template <int TPB, int IPT, bool PAD, int NT, typename index_t, typename offset_t, typename T, bool MASK = true>
int launch_merge_path_fused(hipStream_t stream, const merge_plan_view& plan, int rows, int nnz,
                            const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y,
                            int stages = 3, bool planned = false) {
  const int m = plan.num_merge_tiles;
  if (m == 0) return 0;
  if (m == 1) stages &= ~2;  // one tile holds every row completely: no carry-out to add (launch-bound sizes: 1 kernel)
  T* carry_val = static_cast<T*>(plan.carry_val);
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  if (plan.self_complete && plan.head_start && m > 1) {
    // no row crosses more than one tile boundary with more than TPB nonzeros behind it: tiles complete their
    // rows themselves -- one kernel, no carry-outs (the "fix-up" stage has nothing to do)
    if (stages & 1) {
      if (aligned)
        hipLaunchKernelGGL((merge_path_spmv_fused_self<TPB, IPT, PAD, NT, true, index_t, offset_t, T, MASK>), dim3(m),
                           dim3(TPB), 0, stream, plan.coords, plan.head_start, rows, nnz, offsets, indices, values, x, y);
      else
        hipLaunchKernelGGL((merge_path_spmv_fused_self<TPB, IPT, PAD, NT, false, index_t, offset_t, T, MASK>), dim3(m),
                           dim3(TPB), 0, stream, plan.coords, plan.head_start, rows, nnz, offsets, indices, values, x, y);
    }
    return launch_status();
  }
  if (stages & 1) {
    auto go = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(m), dim3(TPB), 0, stream, plan.coords, rows, nnz, offsets, indices, values, x, y,
                         plan.carry_row, carry_val);
    };
    if (planned) {  // (same code under the symbol of SpMV-plan handles: profile attribution only)
      if (aligned) go(merge_path_spmv_fused_planned<TPB, IPT, PAD, NT, true, index_t, offset_t, T, MASK>);
      else go(merge_path_spmv_fused_planned<TPB, IPT, PAD, NT, false, index_t, offset_t, T, MASK>);
    } else {
      if (aligned) go(merge_path_spmv_fused<TPB, IPT, PAD, NT, true, index_t, offset_t, T, MASK>);
      else go(merge_path_spmv_fused<TPB, IPT, PAD, NT, false, index_t, offset_t, T, MASK>);
    }
  }
  if (stages & 2)
    hipLaunchKernelGGL(merge_path_spmv_fixup<T>, dim3(math::ceil_div(m, 256)), dim3(256), 0, stream, plan.carry_row,
                       carry_val, m, rows, y);
  return launch_status();
}

/// How the phased-gather kernels are run on a matrix of `cols` columns of `elem_bytes`-byte values: the number of column
/// parts M -- 8 up to |x| = 6 MB, 16 up to 24 MB, 32 beyond: parts of 0.5-2 MB (8-byte values: 8 up to 12 MB, 16 beyond) -- the
/// shift that maps a column to its part, min(col >> shift, M - 1), and the period the first part of a tile is chosen by: about
/// the time a pass takes, 3.75 us with 8 parts, 1.87 us with 16 or 32.  Measured optima over |x| = 4 ... 64 MB, flat within +- 25 %
/// (profiles/r04_phased_gather_experiments.txt, sections 2, 7 and 8).  LOOPS_PHASED_PARTS / LOOPS_PHASED_TICKS (environment, read
/// once) override M and the period in 10 ns ticks: measurement aids (scripts/sweep_phased.sh).
struct phased_config {
  int parts;
  detail::phase_args args;
};
inline phased_config phased_config_for(long long cols, int elem_bytes) {
  static const int env_parts = [] { const char* e = std::getenv("LOOPS_PHASED_PARTS"); return e ? std::atoi(e) : 0; }();
  static const int env_ticks = [] { const char* e = std::getenv("LOOPS_PHASED_TICKS"); return e ? std::atoi(e) : 0; }();
  const double x_mb = static_cast<double>(cols > 0 ? cols : 1) * elem_bytes / (1024.0 * 1024.0);
  int parts = x_mb <= 6.0 ? 8 : x_mb <= 24.0 ? 16 : 32;
  double ticks = parts == 8 ? 375.0 : 187.0;             // 10 ns ticks per pass
  if (elem_bytes == 8) {  // 8-byte values (the stream is twice as heavy per gather): 8 parts up to 12 MB, 16 beyond, never 32 --
    parts = x_mb <= 12.0 ? 8 : 16;   // |x| = 8 / 16 / 32 MB: 1.54 / 1.43 / 1.30 x (experiments file, section 16)
    ticks = x_mb <= 6.0 ? 375.0 : 187.0;
  }
  if (env_parts == 8 || env_parts == 16 || env_parts == 32) parts = env_parts;
  if (env_ticks > 0) ticks = env_ticks;
  int bits = 0;
  while (bits < 31 && ((cols > 0 ? cols : 1) - 1) >> bits) ++bits;  // bits needed for cols - 1
  int lg = 0;
  while ((1 << (lg + 1)) <= parts) ++lg;
  phased_config c;
  c.parts = parts;
  c.args.shift = static_cast<unsigned int>(bits > lg ? bits - lg : 0);
  c.args.inv_ticks = static_cast<unsigned int>(4294967296.0 / (ticks < 2.0 ? 2.0 : ticks));
  return c;
}

/// A STRUCTURAL guess at whether the phased gathers pay on this matrix, for callers that cannot measure (the plan-less
/// `algorithms::spmv::merge_path_flat(csr, x, y)`, whose set-up is untimed and synchronous anyway): true when x is at least 3 MB
/// (6 MB for 8-byte values; below that it fits an L2 next to the stream), the matrix has at least 2^20 nonzeros, of 16 384 sampled
/// pairs of nonzeros ONE TO TWO MERGE TILES APART fewer than 2 / parts share a part of x (uniform columns: 1 / parts) and fewer than
/// a quarter of the sampled ADJACENT pairs share a 128-byte line of x (kernels::column_scatter_sample, detail::scatter_counts_say_phase).
/// On the twelve structures of tests/perf/sweep_structures.py, the three C3 stand-ins and two R-MAT graphs of C3's size it agrees with
/// the measured choice (scripts/check_scatter_guess.py; until round 5 the threshold was one half whatever the number of parts, and an
/// R-MAT graph of 2^23 vertices in generator order -- 3.2 / parts -- was sent to the phased kernel, which loses 18 % there).  Two small kernels,
/// one 16-byte copy, one stream synchronisation; `scratch` = kernels::scatter_scratch_words device words.  Measuring
/// (loops_autotune_merge_path_variants_f32, LOOPS_PLAN_MEASURE) remains the reliable way.
/// (the size test alone -- no device work: callers allocate the scratch words only when it passes)
/// `timed_path`: the sample itself is paid by every call (the asynchronous plan-less C ABI entry: ~5 us for its two small
/// kernels) -- ask only from 6 MB on, where scattered columns gain 1.5 x and more, not at 4 MB, where they gain 12 %.
inline bool columns_worth_sampling(long long nnz, long long cols, int elem_bytes, bool timed_path = false) {
  const double x_mb = static_cast<double>(cols) * elem_bytes / (1024.0 * 1024.0);
  return x_mb >= (elem_bytes == 8 || timed_path ? 6.0 : 3.0) && nnz >= (1ll << 20);  // (fp64 at 4 MB: nothing to gain)
}
template <typename index_t>
inline bool columns_look_scattered(hipStream_t stream, const index_t* indices, long long nnz, long long cols, int elem_bytes,
                                   unsigned int* scratch) {
  if (!columns_worth_sampling(nnz, cols, elem_bytes) || !scratch) return false;
  const phased_config cfg = phased_config_for(cols, elem_bytes);
  constexpr long long far = 4096;  // nonzeros: about one 512 x 8 merge tile
  const long long stride = nnz / scatter_samples > 1 ? nnz / scatter_samples : 1;
  hipLaunchKernelGGL(column_scatter_sample<index_t>, dim3(scatter_blocks), dim3(256), 0, stream, indices, nnz, stride, far, cfg.args.shift,
                     static_cast<unsigned int>(cfg.parts), elem_bytes == 8 ? 4u : 5u, scratch + 4);
  hipLaunchKernelGGL(column_scatter_decide, dim3(1), dim3(64), 0, stream, scratch + 4, scratch);
  unsigned int host[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(host, scratch, sizeof(host), hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
  if (hipStreamSynchronize(stream) != hipSuccess) return false;
  return detail::scatter_counts_say_phase(host[0], host[1], host[2], host[3], static_cast<unsigned int>(cfg.parts));
}

/// Launches the sample of columns_look_scattered WITHOUT reading it back: `stats` (scatter_scratch_words device words) then holds in [0..3] what
/// merge_path_spmv_fused_auto decides by.  The caller has checked columns_worth_sampling.
template <typename index_t>
inline int launch_column_scatter_sample(hipStream_t stream, const index_t* indices, long long nnz, long long cols, int elem_bytes,
                                        unsigned int* stats) {
  const phased_config cfg = phased_config_for(cols, elem_bytes);
  constexpr long long far = 4096;
  const long long stride = nnz / scatter_samples > 1 ? nnz / scatter_samples : 1;
  hipLaunchKernelGGL(column_scatter_sample<index_t>, dim3(scatter_blocks), dim3(256), 0, stream, indices, nnz, stride, far, cfg.args.shift,
                     static_cast<unsigned int>(cfg.parts), elem_bytes == 8 ? 4u : 5u, stats + 4);
  hipLaunchKernelGGL(column_scatter_decide, dim3(1), dim3(64), 0, stream, stats + 4, stats);
  return launch_status();
}

/// The plan-less product on a matrix worth asking about: sample (async) -> merge_path_spmv_fused_auto (plain or phased gathers,
/// decided on the device) -> fix-up.  Two-kernel form only (plan-less plans are never classified); unaligned arrays or a single
/// tile run the plain kernel.  `resample` = false: `stats` still hold the sample of THIS matrix (callers that remember which
/// matrix they sampled last: the sample is then paid once per matrix and `timed_path` = false -- ask from 3 MB on -- is the right rule).
template <int TPB, int IPT, typename index_t, typename offset_t, typename T>
int launch_merge_path_fused_auto(hipStream_t stream, const merge_plan_view& plan, int rows, int cols, int nnz,
                                 const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y,
                                 unsigned int* stats, bool resample = true, bool timed_path = true) {
  const int m = plan.num_merge_tiles;
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  if (m <= 1 || !aligned || !stats || !columns_worth_sampling(nnz, cols, static_cast<int>(sizeof(T)), timed_path))
    return launch_merge_path_fused<TPB, IPT, true, 0, index_t, offset_t, T, true>(stream, plan, rows, nnz, offsets, indices, values, x, y);
  int err = resample ? launch_column_scatter_sample(stream, indices, static_cast<long long>(nnz), static_cast<long long>(cols), static_cast<int>(sizeof(T)), stats) : 0;
  if (err) return err;
  T* carry_val = static_cast<T*>(plan.carry_val);
  const phased_config cfg = phased_config_for(cols, static_cast<int>(sizeof(T)));
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(m), dim3(TPB), 0, stream, plan.coords, rows, nnz, offsets, indices, values, x, y, plan.carry_row,
                       carry_val, stats, cfg.args);
  };
  if (cfg.parts == 8) go(merge_path_spmv_fused_auto<TPB, IPT, 8, true, index_t, offset_t, T>);
  else if (cfg.parts == 16) go(merge_path_spmv_fused_auto<TPB, IPT, 16, true, index_t, offset_t, T>);
  else go(merge_path_spmv_fused_auto<TPB, IPT, 32, true, index_t, offset_t, T>);
  hipLaunchKernelGGL(merge_path_spmv_fixup<T>, dim3(math::ceil_div(m, 256)), dim3(256), 0, stream, plan.carry_row, carry_val, m, rows, y);
  return launch_status();
}

/// Fused merge-path SpMV with PHASED x gathers (merge_path_spmv_fused_phased + fix-up, or merge_path_spmv_fused_self_phased for
/// self-completing plans).  A single tile or arrays that are not 16-byte aligned run the plain kernel -- same result either way.
template <int TPB, int IPT, typename index_t, typename offset_t, typename T>
int launch_merge_path_fused_phased(hipStream_t stream, const merge_plan_view& plan, int rows, int cols, int nnz,
                                   const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y,
                                   int stages = 3, bool planned = false) {
  const int m = plan.num_merge_tiles;
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  if (m <= 1 || !aligned)
    return launch_merge_path_fused<TPB, IPT, true, 0, index_t, offset_t, T, true>(stream, plan, rows, nnz, offsets, indices, values, x, y,
                                                                                 stages, planned);
  T* carry_val = static_cast<T*>(plan.carry_val);
  const phased_config cfg = phased_config_for(cols, static_cast<int>(sizeof(T)));
  if (plan.self_complete && plan.head_start) {  // one kernel, no carry-outs (the "fix-up" stage has nothing to do)
    if (stages & 1) {
      auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(m), dim3(TPB), 0, stream, plan.coords, plan.head_start, rows, nnz, offsets, indices, values, x, y,
                           cfg.args);
      };
      if (cfg.parts == 8) go(merge_path_spmv_fused_self_phased<TPB, IPT, 8, true, index_t, offset_t, T>);
      else if (cfg.parts == 16) go(merge_path_spmv_fused_self_phased<TPB, IPT, 16, true, index_t, offset_t, T>);
      else go(merge_path_spmv_fused_self_phased<TPB, IPT, 32, true, index_t, offset_t, T>);
    }
    return launch_status();
  }
  if (stages & 1) {
    auto go = [&](auto plain, auto from_plan) {
      if (planned)
        hipLaunchKernelGGL(from_plan, dim3(m), dim3(TPB), 0, stream, plan.coords, rows, nnz, offsets, indices, values, x, y,
                           plan.carry_row, carry_val, cfg.args);
      else
        hipLaunchKernelGGL(plain, dim3(m), dim3(TPB), 0, stream, plan.coords, rows, nnz, offsets, indices, values, x, y,
                           plan.carry_row, carry_val, cfg.args);
    };
    if (cfg.parts == 8)
      go(merge_path_spmv_fused_phased<TPB, IPT, 8, true, index_t, offset_t, T>, merge_path_spmv_fused_phased_planned<TPB, IPT, 8, true, index_t, offset_t, T>);
    else if (cfg.parts == 16)
      go(merge_path_spmv_fused_phased<TPB, IPT, 16, true, index_t, offset_t, T>, merge_path_spmv_fused_phased_planned<TPB, IPT, 16, true, index_t, offset_t, T>);
    else
      go(merge_path_spmv_fused_phased<TPB, IPT, 32, true, index_t, offset_t, T>, merge_path_spmv_fused_phased_planned<TPB, IPT, 32, true, index_t, offset_t, T>);
  }
  if (stages & 2)
    hipLaunchKernelGGL(merge_path_spmv_fixup<T>, dim3(math::ceil_div(m, 256)), dim3(256), 0, stream, plan.carry_row,
                       carry_val, m, rows, y);
  return launch_status();
}

/// Fused merge-path SpMV (+ fix-up) whose finished rows also go to `peers.count` peer-mapped vectors: the allgatherv(y)
/// of a row-range sharded multi-GPU SpMV issued from the epilogue.  Always the two-kernel form (the
/// fix-up re-writes completed rows on every destination).
template <int TPB, int IPT, typename index_t, typename offset_t, typename T>
int launch_merge_path_fused_fanout(hipStream_t stream, const merge_plan_view& plan, int rows, int nnz,
                                   const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y,
                                   const peer_fanout<T>& peers) {
  const int m = plan.num_merge_tiles;
  if (m == 0) return 0;
  if (peers.count < 0 || peers.count > max_peers) return static_cast<int>(hipErrorInvalidValue);
  T* carry_val = static_cast<T*>(plan.carry_val);
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  constexpr bool PAD = IPT % 2 == 0;
  if (aligned)
    hipLaunchKernelGGL((merge_path_spmv_fused_fanout<TPB, IPT, PAD, true, index_t, offset_t, T>), dim3(m), dim3(TPB), 0, stream,
                       plan.coords, rows, nnz, offsets, indices, values, x, y, peers, plan.carry_row, carry_val);
  else
    hipLaunchKernelGGL((merge_path_spmv_fused_fanout<TPB, IPT, PAD, false, index_t, offset_t, T>), dim3(m), dim3(TPB), 0, stream,
                       plan.coords, rows, nnz, offsets, indices, values, x, y, peers, plan.carry_row, carry_val);
  if (m > 1)
    hipLaunchKernelGGL(merge_path_spmv_fixup_fanout<T>, dim3(math::ceil_div(m, 256)), dim3(256), 0, stream, plan.carry_row,
                       carry_val, m, rows, y, peers);
  return launch_status();
}

/// merge_path_flat over an ELL matrix (row-major rows x pitch cells, negative column = padding) on the fused engine
/// (+ fix-up): no atomics, y needs no zero-fill.  `plan` = coordinates of the merge path of (row ends (r + 1) * pitch,
/// cells) for tile shape TPB x IPT -- schedule::merge_path::preprocess_t over layout::ell, or
/// launch_merge_path_coordinates over an explicit offsets array.
template <int TPB, int IPT, typename index_t, typename T>
int launch_ell_merge_path_fused(hipStream_t stream, const merge_plan_view& plan, int rows, int pitch, const index_t* indices,
                                const T* values, const T* x, T* y) {
  const int m = plan.num_merge_tiles;
  if (m == 0) return 0;
  if (static_cast<long long>(rows) * pitch + rows >= (1ll << 31) - 4096) return static_cast<int>(hipErrorInvalidValue);
  T* carry_val = static_cast<T*>(plan.carry_val);
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  constexpr bool PAD = IPT % 2 == 0;
  if (aligned)
    hipLaunchKernelGGL((ell_merge_path_spmv_fused<TPB, IPT, PAD, true, index_t, T>), dim3(m), dim3(TPB), 0, stream, plan.coords,
                       rows, pitch, indices, values, x, y, plan.carry_row, carry_val);
  else
    hipLaunchKernelGGL((ell_merge_path_spmv_fused<TPB, IPT, PAD, false, index_t, T>), dim3(m), dim3(TPB), 0, stream, plan.coords,
                       rows, pitch, indices, values, x, y, plan.carry_row, carry_val);
  if (m > 1)
    hipLaunchKernelGGL(merge_path_spmv_fixup<T>, dim3(math::ceil_div(m, 256)), dim3(256), 0, stream, plan.carry_row, carry_val,
                       m, rows, y);
  return launch_status();
}

/// Merge-path SpMM C = A * B (+ fix-up).  `carry_mat` holds plan.num_merge_tiles * n values.
/// V (columns per lane) = the widest of 4 / 2 / 1 that divides n and the base alignments allow;
/// G (lanes per sub-group) = the smallest power of two >= n / V, capped at 64 -- wider B is
/// covered by grid.y slabs of 64 * V columns.
template <int TPB, int IPT, typename index_t, typename offset_t, typename T>
int launch_merge_path_spmm(hipStream_t stream, const merge_plan_view& plan, T* carry_mat, int rows, int cols, int nnz,
                           const offset_t* offsets, const index_t* indices, const T* values, const T* B, int n,
                           std::size_t ldb, T* C, std::size_t ldc) {
  const int m = plan.num_merge_tiles;
  if (m == 0 || n <= 0) return 0;
  // B-row offsets are staged as 32-bit counts of per-lane vectors: cols * (ldb / V) must fit
  if (static_cast<unsigned long long>(cols) * ldb >= (1ull << 32)) return static_cast<int>(hipErrorInvalidValue);
#ifndef LOOPS_SPMM_U
#define LOOPS_SPMM_U 8
#endif
  constexpr int U = LOOPS_SPMM_U;
  auto go = [&](auto width, auto per_lane) {
    constexpr int G = decltype(width)::value;
    constexpr int V = decltype(per_lane)::value;
    hipLaunchKernelGGL((merge_path_spmm<TPB, IPT, G, V, U, index_t, offset_t, T>), dim3(m, math::ceil_div(n, G * V)),
                       dim3(TPB), 0, stream, plan.coords, rows, nnz, offsets, indices, values, B, n, ldb, C, ldc,
                       plan.carry_row, carry_mat);
  };
  auto pick_width = [&](auto per_lane) {
    constexpr int V = decltype(per_lane)::value;
    const int w = n / V;  // lanes needed for one row of B
    if (w <= 2) go(std::integral_constant<int, 2>{}, per_lane);
    else if (w <= 4) go(std::integral_constant<int, 4>{}, per_lane);
    else if (w <= 8) go(std::integral_constant<int, 8>{}, per_lane);
    else if (w <= 16) go(std::integral_constant<int, 16>{}, per_lane);
    else if (w <= 32) go(std::integral_constant<int, 32>{}, per_lane);
    else go(std::integral_constant<int, 64>{}, per_lane);
  };
  const std::uintptr_t bases = reinterpret_cast<std::uintptr_t>(B) | reinterpret_cast<std::uintptr_t>(C) |
                               reinterpret_cast<std::uintptr_t>(carry_mat);
  auto fits = [&](int v) {
    const std::size_t bytes = sizeof(T) * v;
    return n % v == 0 && ldb % v == 0 && ldc % v == 0 && bases % (bytes < 16 ? bytes : 16) == 0;
  };
  if (fits(4)) pick_width(std::integral_constant<int, 4>{});
  else if (fits(2)) pick_width(std::integral_constant<int, 2>{});
  else pick_width(std::integral_constant<int, 1>{});
  const std::size_t work = static_cast<std::size_t>(m) * n;
  hipLaunchKernelGGL(merge_path_spmm_fixup<T>, dim3(static_cast<unsigned>(math::ceil_div(work, std::size_t(256)))),
                     dim3(256), 0, stream, plan.carry_row, carry_mat, m, rows, n, C, ldc);
  return launch_status();
}

/// Reference-shaped SpMM: thread per row of A, columns of B in the outer loop.
template <typename index_t, typename offset_t, typename T>
int launch_thread_mapped_spmm(hipStream_t stream, int rows, const offset_t* offsets, const index_t* indices,
                              const T* values, const T* B, int n, std::size_t ldb, T* C, std::size_t ldc) {
  if (rows == 0 || n <= 0) return 0;
  hipLaunchKernelGGL((thread_mapped_spmm<index_t, offset_t, T>), dim3(math::ceil_div(rows, 128)), dim3(128), 0, stream,
                     rows, offsets, indices, values, B, n, ldb, C, ldc);
  return launch_status();
}

/// Merge tiles per workgroup of the persistent `work_oriented` kernels: min(4, ceil(tiles / (8 x resident workgroups))), at least 1.
/// Shares of a few tiles, not one share per resident workgroup: with one long share each, the workgroups of a CU run in step
/// (their stream and gather phases coincide) and the launch ends when the slowest share does; short shares are balanced by the
/// dispatcher and still chain their carries in registers.  C3 stand-ins (47 K tiles of 512 x 8; profiles/
/// r05_work_oriented_shares_experiment.txt): 1 024 / 2 048 / 8 192 / 16 384 shares = 1 332 / 1 176 / 1 068 / 1 062 us (host-blocked),
/// 936 / 814 / 748 / 741 us (65 536-wide band; one tile per workgroup: 787); C2: 1 024 / 2 048 / 4 352 shares = 115 / 102 / 94 us.
inline int work_oriented_share(int tiles, int resident) {
  const int per_resident = math::ceil_div(tiles, 8 * (resident < 1 ? 1 : resident));
  return per_resident < 1 ? 1 : (per_resident > 4 ? 4 : per_resident);
}

/// Tuned work_oriented: contiguous shares of 1-4 plan tiles per workgroup (at least eight shares per resident workgroup).
template <int TPB, int IPT, bool PAD, typename index_t, typename offset_t, typename T, bool MASK = true>
int launch_work_oriented_fused(hipStream_t stream, const merge_plan_view& plan, int rows, int nnz,
                               const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y) {
  const int m = plan.num_merge_tiles;
  if (m == 0) return 0;
  T* carry_val = static_cast<T*>(plan.carry_val);
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  auto k_vec = work_oriented_spmv_fused<TPB, IPT, PAD, false, true, index_t, offset_t, T, MASK>;
  auto k_scl = work_oriented_spmv_fused<TPB, IPT, PAD, false, false, index_t, offset_t, T, MASK>;
  static const int resident = static_cast<int>(launch_box::occupancy_grid(k_vec, TPB));  // blocks per CU x CUs
  const int tiles_per_group = work_oriented_share(m, resident);
  const int groups = math::ceil_div(m, tiles_per_group);
  if (aligned)
    hipLaunchKernelGGL(k_vec, dim3(groups), dim3(TPB), 0, stream, plan.coords, m, tiles_per_group, rows, nnz, offsets,
                       indices, values, x, y, plan.carry_row, carry_val);
  else
    hipLaunchKernelGGL(k_scl, dim3(groups), dim3(TPB), 0, stream, plan.coords, m, tiles_per_group, rows, nnz, offsets,
                       indices, values, x, y, plan.carry_row, carry_val);
  hipLaunchKernelGGL(merge_path_spmv_fixup<T>, dim3(math::ceil_div(groups, 256)), dim3(256), 0, stream, plan.carry_row,
                     carry_val, groups, rows, y);
  return launch_status();
}

/// Tuned group_mapped: one workgroup per TPB consecutive rows, no plan, no atomics.
template <int TPB, int IPT, bool PAD, typename index_t, typename offset_t, typename T, bool MASK = true>
int launch_group_mapped_fused(hipStream_t stream, int rows, int nnz, const offset_t* offsets, const index_t* indices,
                              const T* values, const T* x, T* y) {
  if (rows == 0) return 0;
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  const dim3 grid(math::ceil_div(rows, TPB)), block(TPB);
  if (aligned)
    hipLaunchKernelGGL((group_mapped_spmv_fused<TPB, IPT, PAD, false, true, index_t, offset_t, T, MASK>), grid, block, 0,
                       stream, rows, nnz, offsets, indices, values, x, y);
  else
    hipLaunchKernelGGL((group_mapped_spmv_fused<TPB, IPT, PAD, false, false, index_t, offset_t, T, MASK>), grid, block, 0,
                       stream, rows, nnz, offsets, indices, values, x, y);
  return launch_status();
}

/// `reference_shape`: the plain loop over `config.atoms(row)` (the reference's kernel, algorithms/spmv/thread_mapped.cuh:27-44)
/// instead of the tuned one (same schedule, same bits; rows 16 / 4 at a time and long rows read by the whole wavefront: 5-25 x faster
/// on this GPU, see thread_mapped_batched_spmv / thread_mapped_assisted_spmv).
template <typename index_t, typename offset_t, typename T>
int launch_thread_mapped(hipStream_t stream, std::size_t rows, std::size_t cols, std::size_t nnz,
                         const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y,
                         bool reference_shape = false) {
  if (rows == 0) return 0;
  constexpr std::size_t block = algorithms::spmv::launch_t<T>::block_size;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, offset_t>;
  setup_t config(const_cast<offset_t*>(offsets), rows, nnz);
  const dim3 grid(static_cast<unsigned>(math::ceil_div(rows, block)));
  if (reference_shape)
    launch::non_cooperative(stream, thread_mapped_spmv<setup_t, index_t, offset_t, T>, grid, dim3(block), config, rows, cols, nnz,
                            offsets, indices, values, x, y);
  else if (rows < (std::size_t(1) << 31))  // the same rows per thread, long rows read by the whole wavefront (thread_mapped_assisted_spmv)
    hipLaunchKernelGGL((thread_mapped_assisted_spmv<static_cast<int>(block), index_t, offset_t, T>), grid, dim3(block), 0, stream,
                       static_cast<int>(rows), offsets, indices, values, x, y);
  else
    launch::non_cooperative(stream, thread_mapped_batched_spmv<setup_t, index_t, offset_t, T>, grid, dim3(block), config, offsets,
                            indices, values, x, y);
  return launch_status();
}

template <typename index_t, typename offset_t, typename T>
int launch_original(hipStream_t stream, std::size_t rows, std::size_t cols, std::size_t nnz, const offset_t* offsets,
                    const index_t* indices, const T* values, const T* x, T* y) {
  if (rows == 0) return 0;
  launch::non_cooperative(stream, original_spmv<index_t, offset_t, T>,
                          dim3(static_cast<unsigned>(math::ceil_div(rows, std::size_t(128)))), dim3(128), rows, cols,
                          nnz, offsets, indices, values, x, y);
  return launch_status();
}

/// Reference-shaped group_mapped (block_mapped) kernel: atomics, y must be zero-filled.
template <typename index_t, typename offset_t, typename T>
int launch_group_mapped_atomic(hipStream_t stream, std::size_t rows, std::size_t cols, std::size_t nnz,
                               const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y) {
  if (rows == 0) return 0;
  constexpr std::size_t block = algorithms::spmv::launch_t<T>::block_size;
  launch::non_cooperative(stream, group_mapped_atomic_spmv<block, block, index_t, offset_t, T>,
                          dim3(static_cast<unsigned>(math::ceil_div(rows, block))), dim3(block), rows, cols, nnz,
                          const_cast<offset_t*>(offsets), const_cast<index_t*>(indices), values, x, y);
  return launch_status();
}

/// Reference-shaped work_oriented kernel: store/atomic mix, y must be zero-filled.
template <typename index_t, typename offset_t, typename T>
int launch_work_oriented_atomic(hipStream_t stream, std::size_t rows, std::size_t cols, std::size_t nnz,
                                const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y) {
  if (rows == 0) return 0;
  constexpr std::size_t block = algorithms::spmv::launch_t<T>::block_size;
  auto kernel = work_oriented_atomic_spmv<block, index_t, offset_t, T>;
  const std::size_t grid = launch_box::occupancy_grid(kernel, block);
  launch::non_cooperative(stream, kernel, dim3(static_cast<unsigned>(grid)), dim3(block), rows, cols, nnz,
                          const_cast<offset_t*>(offsets), const_cast<index_t*>(indices), values, x, y);
  return launch_status();
}

/// thread_mapped over flat_uniform_occupancy<K, csr>: atomics, y must be zero-filled.
/// `shape` 0 (default): the tuned kernel -- 16-byte loads, one tile_of per lane, runs stitched across the wavefront,
/// one atomic per row and wavefront (flat_partitioned_stitched_spmv); 1: the reference-shaped kernel (one tile_of
/// search + one atomic per nonzero); 2: row runs per thread (one search per thread, one atomic per run; round 1).
template <std::size_t K, typename index_t, typename offset_t, typename T>
int launch_flat_partitioned(hipStream_t stream, std::size_t rows, std::size_t nnz, const offset_t* offsets,
                            const index_t* indices, const T* values, const T* x, T* y, int shape = 0) {
  const bool per_atom = shape == 1;
  using base_t = layout::csr<index_t, offset_t>;
  using part_t = layout::flat_uniform_occupancy<K, base_t>;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, offset_t, std::size_t,
                                  std::size_t, part_t>;
  part_t part(base_t(offsets, static_cast<index_t>(rows), static_cast<offset_t>(nnz)));
  const std::size_t chunks = static_cast<std::size_t>(part.num_tiles());
  if (chunks == 0) return 0;
  constexpr std::size_t block = algorithms::spmv::launch_t<T>::block_size;
  setup_t config(part);
  const dim3 grid(static_cast<unsigned>(math::ceil_div(chunks, block)));
  if (shape == 0) {
    const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
    const dim3 wide(static_cast<unsigned>(math::ceil_div(chunks, std::size_t(256))));
    if (aligned)
      hipLaunchKernelGGL((flat_partitioned_stitched_spmv<static_cast<int>(K), true, part_t, index_t, T>), wide, dim3(256), 0,
                         stream, part, indices, values, x, y);
    else
      hipLaunchKernelGGL((flat_partitioned_stitched_spmv<static_cast<int>(K), false, part_t, index_t, T>), wide, dim3(256), 0,
                         stream, part, indices, values, x, y);
    return launch_status();
  }
  if (per_atom)
    launch::non_cooperative(stream, flat_partitioned_spmv<setup_t, index_t, T>, grid, dim3(block), config, indices,
                            values, x, y);
  else
    launch::non_cooperative(stream, flat_partitioned_runs_spmv<setup_t, index_t, T>, grid, dim3(block), config,
                            indices, values, x, y);
  return launch_status();
}

/// Reference-shaped merge_path_flat kernel through the schedule API: atomics, y zero-filled.
/// Blocks until the kernel is done (the temporary preprocess_t owns device scratch).
template <std::size_t TPB, std::size_t IPT, typename index_t, typename offset_t, typename T>
int launch_merge_path_atomic(hipStream_t stream, std::size_t rows, std::size_t cols, std::size_t nnz,
                             const offset_t* offsets, const index_t* indices, const T* values, const T* x, T* y) {
  using pre_t = schedule::merge_path::preprocess_t<TPB, IPT, index_t, offset_t, std::size_t, std::size_t>;
  pre_t meta(const_cast<offset_t*>(offsets), rows, nnz, stream);
  const std::size_t m = meta.merge_tiles();
  if (m == 0) return 0;
  launch::non_cooperative(stream, merge_path_flat_atomic_spmv<TPB, IPT, pre_t, index_t, offset_t, T>,
                          dim3(static_cast<unsigned>(m)), dim3(TPB), meta, rows, cols, nnz,
                          const_cast<offset_t*>(offsets), const_cast<index_t*>(indices), values, x, y);
  (void)hipStreamSynchronize(stream);
  return launch_status();
}

}  // namespace kernels
}  // namespace loops
