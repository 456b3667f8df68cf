/**
 * @file csr_spmv.hxx
 * @brief CSR SpMV kernels written against the public schedule API (raw-pointer interface).
 *
 * These are the "a user would write this" kernels: each is a couple of range-for loops over
 * what a `schedule::setup<...>` hands out.  They double as the executable specification of
 * the schedules (tests dump their (tile, atom) -> thread assignment and compare it with the
 * oracle / the reference's device code).  The tuned, atomics-free paths live next door:
 * merge_path_spmv.hxx (merge_path_flat, work_oriented, group_mapped) and, for flat_partitioned,
 * flat_partitioned_runs_spmv at the end of this file.
 *
 * Kernel semantics follow the reference kernels:
 *   thread_mapped   algorithms/spmv/thread_mapped.cuh:27-56     y[row] = sum
 *   original        algorithms/spmv/original.cuh:26-46          y[row] = sum (no schedule)
 *   merge_path_flat algorithms/spmv/merge_path_flat.cuh:38-83   atomicAdd per nonzero, y pre-zeroed
 *   work_oriented   algorithms/spmv/work_oriented.cuh:33-89     store / atomicAdd mix, y pre-zeroed
 *   group_mapped    algorithms/spmv/group_mapped.cuh:27-61      atomicAdd per nonzero, y pre-zeroed
 *   flat_partitioned algorithms/spmv/flat_partitioned.cuh:46-60 atomicAdd per nonzero, y pre-zeroed
 */
#pragma once

#include <type_traits>

#include <cstddef>

#include <hip/hip_runtime.h>

#include <loops/schedule.hxx>
#include <loops/container/layout.hxx>
#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/kernels/run_stitch.hxx>

namespace loops {
namespace kernels {

template <typename setup_t, typename index_t, typename offset_t, typename type_t>
__global__ void thread_mapped_spmv(setup_t config, const std::size_t rows, const std::size_t cols,
                                   const std::size_t nnz, const offset_t* offsets, const index_t* indices,
                                   const type_t* values, const type_t* x, type_t* y) {
  for (auto row : config.tiles()) {
    type_t sum = 0;
    for (auto nz : config.atoms(row)) sum += values[nz] * x[indices[nz]];
    y[row] = sum;
  }
}

/// The same schedule (a thread owns whole rows, `config.tiles()` says which) with the row's atoms taken B at a time: B index
/// and value loads, then B gathers, are in flight before the first product is added, and the products are added in the
/// row's order with the same fused multiply-adds -- the result is bit for bit the plain loop's, the long rows that set this
/// schedule's time (one lane walks a 16 384-nonzero row of C2 while 63 wait) cost B x fewer memory round trips.
/// Measured (C2 / 2^20 rows of 16 in a 64-wide band): plain loop 3.20 / 0.291 ms, batches of 8: 0.89 / 0.026, of 16:
/// 0.66 / 0.023, of 32: 0.60 / 0.280 (rows of 16 never fill a batch of 32: hence the second, 4-wide level).
namespace detail {
/// a * b + c with ONE rounding -- what the compiler makes of `sum += a * b` in the plain loops (v_fmac_f32 / v_fmac_f64).
template <typename type_t>
__device__ __forceinline__ type_t fused_multiply_add(type_t a, type_t b, type_t c) {
  if constexpr (std::is_same_v<type_t, float>) return __builtin_fmaf(a, b, c);
  else if constexpr (std::is_same_v<type_t, double>) return __builtin_fma(a, b, c);
  else return a * b + c;
}

/// Lane j's value, in scalar registers (wavefront-uniform).
__device__ __forceinline__ long long uniform_of_lane(const long long v, const int j) {
  const int lo = __builtin_amdgcn_readfirstlane(__shfl(static_cast<int>(v), j));
  const int hi = __builtin_amdgcn_readfirstlane(__shfl(static_cast<int>(v >> 32), j));
  return (static_cast<long long>(hi) << 32) | static_cast<long long>(static_cast<unsigned int>(lo));
}

template <int B, typename index_t, typename offset_t, typename type_t>
__device__ __forceinline__ void row_batches(offset_t& k, const offset_t end, const index_t* __restrict__ indices,
                                            const type_t* __restrict__ values, const type_t* __restrict__ x, type_t& sum) {
  while (k + B <= end) {
    index_t c[B];
    type_t v[B], xv[B];
#pragma unroll
    for (int b = 0; b < B; ++b) { c[b] = indices[k + b]; v[b] = values[k + b]; }
#pragma unroll
    for (int b = 0; b < B; ++b) xv[b] = x[c[b]];
#pragma unroll
    for (int b = 0; b < B; ++b) sum = fused_multiply_add(v[b], xv[b], sum);  // (left to itself the compiler packs the
    k += B;                                                                   //  multiplies -- v_pk_mul_f32 -- and adds unfused)
  }
}
}  // namespace detail

template <typename setup_t, typename index_t, typename offset_t, typename type_t>
__global__ void thread_mapped_batched_spmv(setup_t config, const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                                           const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
  for (auto row : config.tiles()) {
    offset_t k = offsets[row];
    const offset_t end = offsets[row + 1];
    type_t sum = 0;
    detail::row_batches<16>(k, end, indices, values, x, sum);
    detail::row_batches<4>(k, end, indices, values, x, sum);
    for (; k < end; ++k) sum = detail::fused_multiply_add(values[k], x[indices[k]], sum);
    y[row] = sum;
  }
}

/// thread_mapped with the MEMORY reads of its long rows shared by the wavefront.  Ownership and arithmetic are the schedule's: a
/// thread owns whole rows (row = global thread id, + grid size per round: `config.tiles()`), a row's products are added in the
/// row's order, starting from 0, with the plain loop's fused multiply-adds -- bit for bit the reference loop
/// (algorithms/spmv/thread_mapped.cuh:27-44).  What changes is who LOADS: a lane walking a 16 384-nonzero row alone has 16 loads
/// in flight and 63 idle neighbours (C2: the few rows at the degree cap set the whole kernel's 0.64 ms).  Rows of `LONG` nonzeros
/// or more are therefore read by the whole wavefront, 256 consecutive nonzeros per step (coalesced index / value loads, 256
/// gathers in flight, the next step's loads issued before this step is summed), handed over through LDS, and summed by every lane
/// in sequence (the owner keeps the result) -- the chain of dependent fused multiply-adds, 11.5 clocks per nonzero with its LDS
/// reads, is what remains.  Shorter rows run as before, 16 / 4 / 1 at a time per lane.  Measured against the batched kernel
/// (tests/perf/exp_thread_mapped_assisted.py, profiles/r06_thread_mapped_assisted.txt): C2 0.645 -> 0.204 ms (fp64 0.34), one row of
/// 2^19 nonzeros 16.9 -> 2.5 ms, one row of 900 per wavefront 0.096 -> 0.043, rows of 16 / 200 and 16 rows of 400 per wavefront unchanged.
template <int TPB, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(TPB)
thread_mapped_assisted_spmv(const int rows, const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                            const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y) {
    constexpr int W = wave::size, WAVES = TPB / W, U = 4, CHUNK = U * W, LONG = 128;
  __shared__ __attribute__((aligned(16))) type_t s_v[WAVES][CHUNK];
  __shared__ __attribute__((aligned(16))) type_t s_x[WAVES][CHUNK];
  const int lane = wave::lane();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / W);
  const long long stride = static_cast<long long>(gridDim.x) * TPB;
  for (long long base = static_cast<long long>(blockIdx.x) * TPB + w * W; base < rows; base += stride) {  // (wavefront-uniform)
    const long long row = base + lane;
    offset_t k = 0, end = 0;
    if (row < rows) {
      k = offsets[row];
      end = offsets[row + 1];
    }
    // Which rows of this wavefront go the assisted way: those of LONG nonzeros or more -- unless there are so many of them that their
    // start-up (two dependent memory round trips each, one row after the other: ~3 us) outweighs what the lockstep walk of all 64 rows
    // costs (~37 ns per nonzero of the longest row; the assisted chain: ~4 ns per nonzero).  Rough constants, generous margins:
    // 64 rows of 200 stay in lockstep (7 us against 250), one row of 900 among short ones is assisted (7 against 33).
    bool is_long = end - k >= static_cast<offset_t>(LONG);
    {
      const unsigned long long longs = __ballot(is_long);
      if (longs != 0ull) {  // (wavefront-uniform)
        long long mine = end - k, longest = mine, long_sum = is_long ? mine : 0, short_longest = is_long ? 0 : mine;
        for (int d = 1; d < W; d <<= 1) {
          longest = max(longest, __shfl_xor(longest, d));
          short_longest = max(short_longest, __shfl_xor(short_longest, d));
          long_sum += __shfl_xor(long_sum, d);
        }
        const long long assisted_ns = 3000ll * __popcll(longs) + 4ll * long_sum + 37ll * short_longest, lockstep_ns = 37ll * longest;
        if (assisted_ns >= lockstep_ns) is_long = false;
      }
    }
    type_t sum = 0;
    if (!is_long) {
      detail::row_batches<16>(k, end, indices, values, x, sum);
      detail::row_batches<4>(k, end, indices, values, x, sum);
      for (; k < end; ++k) sum = detail::fused_multiply_add(values[k], x[indices[k]], sum);
    }
    unsigned long long todo = __ballot(is_long);
    while (todo) {  // the wavefront's long rows, one after the other
      const int j = __builtin_ctzll(todo);
      todo &= todo - 1;
      const long long rk = detail::uniform_of_lane(static_cast<long long>(k), j), rend = detail::uniform_of_lane(static_cast<long long>(end), j);
      index_t c_next[U];
      type_t v_next[U], v_cur[U], x_cur[U];
      auto load = [&](const long long b, index_t (&c)[U], type_t (&v)[U]) {  // nonzeros b + lane + 64 u; past the row: column 0, value 0
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long long at = b + lane + W * u;
          const bool ok = at < rend;
          c[u] = ok ? indices[at] : index_t(0);
          v[u] = ok ? values[at] : type_t(0);
        }
      };
      load(rk, c_next, v_cur);
#pragma unroll
      for (int u = 0; u < U; ++u) x_cur[u] = x[c_next[u]];
      load(rk + CHUNK, c_next, v_next);
      type_t s = 0;
      for (long long b = rk; b < rend; b += CHUNK) {
#pragma unroll
        for (int u = 0; u < U; ++u) {  // (past the row's end: value 0 AND x 0 -- the gather read x[0], which may be anything)
          s_v[w][lane + W * u] = v_cur[u];
          s_x[w][lane + W * u] = b + lane + W * u < rend ? x_cur[u] : type_t(0);
        }
        // the next step's gathers and the loads of the one after it leave before this step is summed
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v_cur[u] = v_next[u];
          x_cur[u] = x[c_next[u]];
        }
        load(b + 2 * CHUNK, c_next, v_next);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n = rend - b < CHUNK ? static_cast<int>(rend - b) : CHUNK;
        // (32 at a time: the LDS reads of a block leave together, ahead of its chain of dependent fused multiply-adds -- one item at a
        //  time every step waits out an LDS round trip: 120 clocks per nonzero instead of ~11.5; two register sets with the next block's
        //  reads behind the current chain spilled and ran slower, the chain on the owner lane alone changed nothing)
        int t = 0;
        for (; t + 32 <= n; t += 32) {
          type_t a[32], b32[32];
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            a[i] = s_v[w][t + i];
            b32[i] = s_x[w][t + i];
          }
#pragma unroll
          for (int i = 0; i < 32; ++i) s = detail::fused_multiply_add(a[i], b32[i], s);
        }
        for (; t < n; ++t) s = detail::fused_multiply_add(s_v[w][t], s_x[w][t], s);
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == j) sum = s;
    }
    if (row < rows) y[row] = sum;
  }
}

template <typename index_t, typename offset_t, typename type_t>
__global__ void original_spmv(const std::size_t rows, const std::size_t cols, const std::size_t nnz,
                              const offset_t* offsets, const index_t* indices, const type_t* values,
                              const type_t* x, type_t* y) {
  for (std::size_t row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
    type_t sum = 0;
    for (offset_t nz = offsets[row]; nz < offsets[row + 1]; ++nz) sum += values[nz] * x[indices[nz]];
    y[row] = sum;
  }
}

/// Reference-shaped merge-path kernel through the public schedule (one atomic per nonzero).
template <std::size_t threads_per_block, std::size_t items_per_thread, typename meta_t, typename index_t,
          typename offset_t, typename type_t>
__global__ void __launch_bounds__(int(threads_per_block))
merge_path_flat_atomic_spmv(meta_t meta, std::size_t rows, std::size_t cols, std::size_t nnz, offset_t* offsets,
                            index_t* indices, const type_t* values, const type_t* x, type_t* y) {
  using setup_t = schedule::setup<schedule::algorithms_t::merge_path_flat, threads_per_block, items_per_thread,
                                  index_t, offset_t, std::size_t, std::size_t>;
  __shared__ typename setup_t::storage_t temporary_storage;
  setup_t config(meta, temporary_storage, offsets, rows, nnz);
  auto map = config.init();
  if (!config.is_valid_accessor(map)) return;
#pragma unroll
  for (auto item : config.virtual_idx()) {
    auto nz = config.atom_idx(item, map);
    auto row = config.tile_idx(map);
    if (config.atoms_counting_it[map.y] < temporary_storage.tile_end_offset[map.x]) {
      atomicAdd(&(y[row]), values[nz] * x[indices[nz]]);
      map.y++;
    } else {
      map.x++;
    }
  }
}

/// Records which thread the merge_path_flat schedule gives every atom to (test / parity hook).
template <std::size_t threads_per_block, std::size_t items_per_thread, typename meta_t, typename offset_t>
__global__ void __launch_bounds__(int(threads_per_block))
merge_path_flat_dump(meta_t meta, std::size_t rows, std::size_t nnz, offset_t* offsets, unsigned int* thread_start,
                     int* atom_owner, int* atom_row, int* atom_visits) {
  using setup_t = schedule::setup<schedule::algorithms_t::merge_path_flat, threads_per_block, items_per_thread, int,
                                  offset_t, std::size_t, std::size_t>;
  __shared__ typename setup_t::storage_t temporary_storage;
  setup_t config(meta, temporary_storage, offsets, rows, nnz);
  auto map = config.init();
  if (!config.is_valid_accessor(map)) return;
  const std::size_t gt = (static_cast<std::size_t>(blockIdx.x) * gridDim.y + blockIdx.y) * threads_per_block + threadIdx.x;
  thread_start[2 * gt] = map.x;
  thread_start[2 * gt + 1] = map.y;
  for (auto item : config.virtual_idx()) {
    auto nz = config.atom_idx(item, map);
    auto row = config.tile_idx(map);
    if (config.atoms_counting_it[map.y] < temporary_storage.tile_end_offset[map.x]) {
      atom_owner[nz] = static_cast<int>(gt);
      atom_row[nz] = row;
      atomicAdd(&atom_visits[nz], 1);
      map.y++;
    } else {
      map.x++;
    }
  }
}

/// Reference-shaped work_oriented kernel (first complete row and the remainder are folded in
/// atomically, rows owned outright are stored).  y must be zero-filled.
template <std::size_t threads_per_block, typename index_t, typename offset_t, typename type_t>
__global__ void __launch_bounds__(threads_per_block)
work_oriented_atomic_spmv(std::size_t rows, std::size_t cols, std::size_t nnz, offset_t* offsets, index_t* indices,
                          const type_t* values, const type_t* x, type_t* y) {
  using setup_t = schedule::setup<schedule::algorithms_t::work_oriented, threads_per_block, 1, index_t, offset_t,
                                  std::size_t, std::size_t>;
  setup_t config(offsets, rows, nnz);
  auto map = config.init();
  type_t sum = 0;
  bool first_tile = true;
  for (auto row : config.tiles(map)) {
    for (auto nz : config.atoms(row, map)) sum += values[nz] * x[indices[nz]];
    if (first_tile) {
      if (sum != 0) atomicAdd(&(y[row]), sum);
      first_tile = false;
    } else {
      y[row] = sum;
    }
    sum = 0;
  }
  __syncthreads();
  for (auto row : config.remainder_tiles(map)) {
    for (auto nz : config.remainder_atoms(map)) sum += values[nz] * x[indices[nz]];
    if (sum != 0) atomicAdd(&(y[row]), sum);
  }
}

template <std::size_t threads_per_block, typename offset_t>
__global__ void __launch_bounds__(threads_per_block)
work_oriented_dump(std::size_t rows, std::size_t nnz, offset_t* offsets, int* thread_map, int* atom_owner,
                   int* atom_row, int* atom_visits) {
  using setup_t = schedule::setup<schedule::algorithms_t::work_oriented, threads_per_block, 1, int, offset_t,
                                  std::size_t, std::size_t>;
  setup_t config(offsets, rows, nnz);
  auto map = config.init();
  const int g = threadIdx.x + blockIdx.x * blockDim.x;
  thread_map[4 * g + 0] = static_cast<int>(map.first.first);
  thread_map[4 * g + 1] = static_cast<int>(map.first.second);
  thread_map[4 * g + 2] = static_cast<int>(map.second.first);
  thread_map[4 * g + 3] = static_cast<int>(map.second.second);
  for (auto row : config.tiles(map)) {
    for (auto nz : config.atoms(row, map)) {
      atom_owner[nz] = g;
      atom_row[nz] = row;
      atomicAdd(&atom_visits[nz], 1);
    }
  }
  for (auto row : config.remainder_tiles(map)) {
    for (auto nz : config.remainder_atoms(map)) {
      atom_owner[nz] = g;
      atom_row[nz] = row;
      atomicAdd(&atom_visits[nz], 1);
    }
  }
}

/// Reference-shaped group_mapped kernel (block_mapped: the workgroup is the group).
template <std::size_t threads_per_block, std::size_t threads_per_tile, typename index_t, typename offset_t,
          typename type_t>
__global__ void __launch_bounds__(threads_per_block)
group_mapped_atomic_spmv(std::size_t rows, std::size_t cols, std::size_t nnz, offset_t* offsets, index_t* indices,
                         const type_t* values, const type_t* x, type_t* y) {
  using setup_t = schedule::setup<schedule::algorithms_t::group_mapped, threads_per_block, threads_per_tile, index_t,
                                  offset_t>;
  __shared__ typename setup_t::storage_t temporary_storage;
  setup_t config(temporary_storage, offsets, rows, nnz);
  auto p = config.partition();
  for (auto virtual_atom : config.atom_accessor(p)) {
    auto virtual_tile = config.tile_accessor(virtual_atom, p);
    if (!(config.is_valid_accessor(virtual_tile, p))) continue;
    auto row = config.tile_id(virtual_tile, p);
    auto nz_idx = config.atom_id(virtual_atom, row, virtual_tile, p);
    atomicAdd(&(y[row]), values[nz_idx] * x[indices[nz_idx]]);
  }
}

template <std::size_t threads_per_block, std::size_t threads_per_tile, typename offset_t>
__global__ void __launch_bounds__(threads_per_block)
group_mapped_dump(std::size_t rows, std::size_t nnz, offset_t* offsets, int* atom_owner, int* atom_row,
                  int* atom_visits) {
  using setup_t = schedule::setup<schedule::algorithms_t::group_mapped, threads_per_block, threads_per_tile, int,
                                  offset_t>;
  __shared__ typename setup_t::storage_t temporary_storage;
  setup_t config(temporary_storage, offsets, rows, nnz);
  auto p = config.partition();
  for (auto virtual_atom : config.atom_accessor(p)) {
    auto virtual_tile = config.tile_accessor(virtual_atom, p);
    if (!(config.is_valid_accessor(virtual_tile, p))) continue;
    auto row = config.tile_id(virtual_tile, p);
    auto nz_idx = config.atom_id(virtual_atom, row, virtual_tile, p);
    atom_owner[nz_idx] = static_cast<int>(p.grid_rank());
    atom_row[nz_idx] = row;
    atomicAdd(&atom_visits[nz_idx], 1);
  }
}

/// thread_mapped over flat_uniform_occupancy<K, csr>: perfectly balanced K-atom tiles, the
/// original row of every atom recovered with base().tile_of (y must be zero-filled).
template <typename setup_t, typename index_t, typename type_t>
__global__ void flat_partitioned_spmv(setup_t config, const index_t* indices, const type_t* values, const type_t* x,
                                      type_t* y) {
  const auto& part = config.layout();
  const auto& base = part.base();
  for (auto chunk : config.tiles()) {
    for (auto nz : config.atoms(chunk)) {
      const auto row = base.tile_of(nz);
      atomicAdd(&y[row], values[nz] * x[indices[nz]]);
    }
  }
}

/// Tuned form of the same schedule: a thread still owns one K-atom tile of the partitioned layout,
/// but it looks up the original row ONCE (base().tile_of of its first atom) and then follows the
/// row boundaries while it walks its K consecutive atoms, adding each run of same-row atoms with
/// one atomicAdd -- K x fewer binary searches and up to K x fewer atomics than the per-atom form.
/// y must be zero-filled.
template <typename setup_t, typename index_t, typename type_t>
__global__ void flat_partitioned_runs_spmv(setup_t config, const index_t* indices, const type_t* values,
                                           const type_t* x, type_t* y) {
  const auto& part = config.layout();
  const auto& base = part.base();
  for (auto chunk : config.tiles()) {
    auto nz = part.tile_begin(chunk);
    const auto nz_end = part.tile_end(chunk);
    if (nz >= nz_end) continue;
    auto row = base.tile_of(nz);
    auto row_end = base.tile_end(row);
    type_t run = type_t(0);
    for (; nz < nz_end; ++nz) {
      while (nz >= row_end) {  // leave the row (skipping empty ones): flush its partial sum
        if (run != type_t(0)) atomicAdd(&y[row], run);
        run = type_t(0);
        ++row;
        row_end = base.tile_end(row);
      }
      run += values[nz] * x[indices[nz]];
    }
    if (run != type_t(0)) atomicAdd(&y[row], run);
  }
}

/// The tuned flat_partitioned kernel.  Same schedule -- thread t owns tile t of
/// flat_uniform_occupancy<K, csr>, i.e. the K consecutive nonzeros [t K, (t + 1) K) -- but
///  * the lane's nonzeros arrive as 16-byte loads (K / 4 per array; consecutive lanes, consecutive vectors),
///  * the row is looked up ONCE per lane (base().tile_of of its first atom) and then followed along the row ends,
///  * runs of same-row products are summed in registers and stitched across the 64 lanes (add_row_runs): one
///    atomicAdd per row and wavefront instead of one per nonzero (reference) or per run and thread (flat_partitioned_runs_spmv).
/// One tile per thread: the grid covers every tile in one pass and no lane leaves before the wavefront stitch.
/// y must be zero-filled (the reference's precondition, flat_partitioned.cuh:99-101).
template <int K, bool VEC, typename part_t, typename index_t, typename type_t>
__global__ void __launch_bounds__(256)
flat_partitioned_stitched_spmv(const part_t part, const index_t* __restrict__ indices, const type_t* __restrict__ values,
                               const type_t* __restrict__ x, type_t* __restrict__ y) {
  using atom_t = typename part_t::atom_id_t;
  const auto& base = part.base();
  const long long chunk = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long nnz = static_cast<long long>(part.num_atoms());
  const long long first = chunk * K;  // == part.tile_begin(chunk) (container/partitioning.hxx: fixed_tiling::first)
  const bool any = first < nnz;
  const bool full = first + K <= nnz;
  index_t c[K];
  type_t v[K];
  if constexpr (VEC && K % 4 == 0) {
    if (full) {
#pragma unroll
      for (int k = 0; k < K; k += 4) {
        index_t c4[4];
        type_t v4[4];
        detail::load4<index_t, false>(indices + first + k, c4);
        detail::load4<type_t, false>(values + first + k, v4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c[k + j] = c4[j];
          v[k + j] = v4[j];
        }
      }
    }
  }
  if (!(VEC && K % 4 == 0 && full)) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool ok = first + k < nnz;
      c[k] = ok ? indices[first + k] : index_t(0);
      v[k] = ok ? values[first + k] : type_t(0);
    }
  }
  type_t p[K];
#pragma unroll
  for (int k = 0; k < K; ++k) p[k] = (full || first + k < nnz) ? v[k] * x[c[k]] : type_t(0);
  // The row of every atom: base().tile_of of the lane's first atom, then a walk along the row ends (empty rows are
  // skipped).  tile_of is evaluated in two levels: the rows of the WAVEFRONT's first and last atom are found with
  // wave-uniform searches -- scalar loads, tracked by their own counter, so the two ~20-step chains run underneath the
  // vector loads above instead of in front of them -- and a lane then searches only the few rows in between.
  int r[K];
  int row = -1;
  atom_t row_end = 0;
  {
    const int wave_first = __builtin_amdgcn_readfirstlane(static_cast<int>(first - static_cast<long long>(wave::lane()) * K));
    const int num_rows = static_cast<int>(base.num_tiles());
    int wave_last = wave_first + wave::size * K - 1;
    wave_last = wave_last < static_cast<int>(nnz) - 1 ? wave_last : static_cast<int>(nnz) - 1;
    // smallest t with tile_end(t) > a (layout::csr::tile_of, container/layout.hxx), both targets in one loop
    int lo = 0, lo_n = num_rows, hi = 0, hi_n = num_rows;
    while (lo_n > 0 || hi_n > 0) {
      if (lo_n > 0) {
        const int half = lo_n >> 1;
        if (static_cast<int>(base.tile_end(lo + half)) <= wave_first) { lo += half + 1; lo_n -= half + 1; }
        else lo_n = half;
      }
      if (hi_n > 0) {
        const int half = hi_n >> 1;
        if (static_cast<int>(base.tile_end(hi + half)) <= wave_last) { hi += half + 1; hi_n -= half + 1; }
        else hi_n = half;
      }
    }
    if (any) {
      row = lo;
      int count = hi - lo;  // the lane's row lies in [lo, hi]
      while (count > 0) {
        const int half = count >> 1;
        if (static_cast<long long>(base.tile_end(row + half)) <= first) { row += half + 1; count -= half + 1; }
        else count = half;
      }
      row_end = base.tile_end(row);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool ok = first + k < nnz;
    if (ok) {
      while (static_cast<atom_t>(first + k) >= row_end) row_end = base.tile_end(++row);
    }
    r[k] = ok ? row : -1;
  }
  add_row_runs<K>(r, p, y);
}

}  // namespace kernels
}  // namespace loops
