/**
 * @file bcsr_merge_path.hxx
 * @brief 4 x 4 fp32 BCSR SpMV on the merge-path schedule: the load-balanced one-shot form of
 *        algorithms::spmv::bcsr_thread_mapped<4, 4> (reference algorithms/spmv/bcsr_thread_mapped.cuh:36-123).
 *
 * Why.  The thread_mapped schedule gives a block-row to ONE owner -- a thread in the reference's kernel, a slot of a wavefront in
 * `bcsr4x4_mfma_spmv` -- so a block-row of 16 384 blocks is walked by that owner alone: 64 such block-rows among 2^17 short ones
 * cost 1.83 ms (tests/perf/exp_bcsr_band_hubs.py), a cliff of the kind merge_path_flat exists to remove for CSR.  Here the merge
 * path of (block-row ends, blocks) is cut into equal tiles exactly as merge_path_flat cuts (row ends, nonzeros)
 * (schedule/merge_path_flat.hxx:45-76; the same split of diagonal t * TILE over `block_offsets`, found inside the tile kernel by a
 * 64-ary search per wavefront instead of a pre-pass launch), one workgroup per tile:
 *   - the tile's block-row ends go to LDS, 4 fp64 sums per block-row of the tile are zeroed in LDS;
 *   - the tile's blocks are streamed 16 at a time per wavefront (1 KB contiguous: lane (q, i) loads row i of block q); the
 *     block-row of a block is found by a halving search over the LDS row ends (<= 11 probes); x[4] is gathered per block;
 *     the block product runs on the matrix core (four chained v_mfma_f32_4x4x1, x as the A operand: lane i's accumulator holds
 *     row i); every lane adds its row into the block-row's sums with ds_add_f64;
 *   - block-rows whose END lies inside the tile are stored (exactly once each); the sums of the block-row still open at the
 *     tile's end leave as a 4-wide carry-out that `bcsr_merge_path_fixup` adds -- merge_path_flat's fix-up with four values
 *     per tile.
 * y needs no zero-fill; no global atomics; rows of y >= `rows` are not written.  Sums inside a tile are fp64 LDS atomics in no
 * fixed order, rounded to fp32 per tile, carry-outs added in tile order: bit-identical to bcsr_thread_mapped on exactly summable
 * inputs (what the tests pin).
 *
 * Measured (MI355X): 64 hub block-rows of 16 384 blocks among 2^17 of 8 -- 39 us against 1 840 us for `bcsr4x4_mfma_spmv`; BASELINE
 * C4 (every block-row 16 blocks: nothing to balance) 81.5 us against 66-70 us -- the search, the LDS sums and the tile set-up (its coordinates by a 64-ary
 * search, row ends, zeroing) and the fix-up launch cost 20 % where the lengths are uniform.  Hence an explicit entry (`loops_spmv_bcsr_f32` mode 4,
 * `algorithms::spmv::bcsr_merge_path`), not what the thread_mapped wrapper launches; callers that multiply one matrix many times
 * hold a block-band plan (bcsr_band.hxx: 56 us on C4, 31 us on the hub case).
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

/// Merge items (block-row ends + blocks) per tile.  A tile may hold that many block-rows: 4 fp64 sums each + the row ends live in
/// LDS (36 bytes per item: 1024 -> 37 KB, four workgroups of 8 wavefronts per CU).  Measured on MI355X (C4 / the hub case of
/// tests/perf/exp_bcsr_band_hubs.py, us; with the coordinate pre-pass still a launch of its own): 1024 x 8 wavefronts 84.8 / 41.6,
/// 2048 x 16 86.6 / 42.2, 4096 x 16 103.4 / 51.8; two steps
/// per batch 10 % slower; without the software pipeline 86.4 / 43.3.
#ifndef LOOPS_BCSR_MERGE_TILE  // (tuning builds only)
#define LOOPS_BCSR_MERGE_TILE 1024
#endif
#ifndef LOOPS_BCSR_MERGE_WAVES
#define LOOPS_BCSR_MERGE_WAVES 8
#endif
#ifndef LOOPS_BCSR_MERGE_U
#define LOOPS_BCSR_MERGE_U 1
#endif
constexpr int bcsr_merge_tile = LOOPS_BCSR_MERGE_TILE;

inline int bcsr_merge_tiles(int num_block_rows, int num_blocks) {
  return static_cast<int>((static_cast<long long>(num_block_rows) + num_blocks + bcsr_merge_tile - 1) / bcsr_merge_tile);
}
/// Scratch of one product: carry values [4 M], carry rows [M].
inline std::size_t bcsr_merge_scratch_bytes(int num_block_rows, int num_blocks) {
  const std::size_t m = static_cast<std::size_t>(bcsr_merge_tiles(num_block_rows, num_blocks));
  return sizeof(int) * m + sizeof(float) * 4 * m + 64;
}
/// Dynamic LDS of the tile kernel: the row ends, then the sums of the tile's block-rows, of the open one and a dump group.
constexpr std::size_t bcsr_merge_lds_bytes(int tile) { return sizeof(double) * 4 * (static_cast<std::size_t>(tile) + 2) + sizeof(int) * static_cast<std::size_t>(tile); }

/// Software pipeline as bcsr_band_accumulate's: the stream loads of the NEXT batch go out behind the gathers of the current one and
/// fly while the current batch is multiplied and added up; streams as buffer loads relative to the tile (a step past the tile's end
/// reads zeros: buffer semantics) -- the block-band kernel with the block-row of a block found by a search instead of read off a word.
template <int TILE, int WAVES, int U>
__global__ void __launch_bounds__(WAVES * wave::size)
bcsr4x4_mfma_merge_path(const int rows, const int num_block_rows, const int num_blocks, const int* __restrict__ block_offsets,
                        const int* __restrict__ block_cols, const float* __restrict__ values, const float* __restrict__ x, float* __restrict__ y,
                        int* __restrict__ carry_row, float* __restrict__ carry_val) {
  using f32x4 = float __attribute__((ext_vector_type(4)));
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  constexpr int TPB = WAVES * wave::size;
  static_assert(WAVES >= 2, "two wavefronts split the tile's two diagonals");
  extern __shared__ __attribute__((aligned(16))) unsigned char bcsr_merge_lds[];
  double* s_acc = reinterpret_cast<double*>(bcsr_merge_lds);                   // [4 (TILE + 2)]
  int* s_re = reinterpret_cast<int*>(bcsr_merge_lds + sizeof(double) * 4 * (TILE + 2));  // [TILE] end (block index) of block-row row0 + r
  const int tid = threadIdx.x;
  const int lane = wave::lane();
  const int q = lane >> 2, i = lane & 3;
  const int w = __builtin_amdgcn_readfirstlane(tid / wave::size);
  const int t = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  // The tile's two merge-path coordinates, found HERE: wavefront 0 splits diagonal t * TILE, wavefront 1 diagonal (t + 1) * TILE, each
  // with a 64-ary search (every lane probes one block-row end, a ballot narrows the range 64-fold: three dependent rounds for 2^18
  // block-rows instead of eighteen) -- no coordinate pre-pass launch.  Same split as search.hxx / merge_path_coordinates_of.
  __shared__ int s_split[2];
  if (w < 2) {
    const long long dl = static_cast<long long>(t + w) * TILE;
    const long long total = static_cast<long long>(num_block_rows) + num_blocks;
    const int d = static_cast<int>(dl < total ? dl : total);
    int lo = d - num_blocks > 0 ? d - num_blocks : 0;       // the first m in [lo, hi) with block_offsets[m + 1] > d - m - 1, or hi
    int hi = d < num_block_rows ? d : num_block_rows;
    while (hi > lo) {
      const int span = hi - lo;
      const int step = (span + wave::size - 1) / wave::size;
      const int m = lo + lane * step;
      const bool before = m < hi && block_offsets[m + 1] <= d - m - 1;  // (true: the split lies behind m)
      const unsigned long long votes = __ballot(before);
      const int trues = __popcll(votes);                               // the predicate is monotone: the trues are lanes 0 .. trues - 1
      const int new_lo = trues > 0 ? lo + (trues - 1) * step + 1 : lo;
      const int new_hi = trues < wave::size && lo + trues * step < hi ? lo + trues * step : hi;
      lo = new_lo;
      hi = step == 1 ? new_lo : new_hi;                                 // (step 1: every candidate was probed)
    }
    if (lane == 0) s_split[w] = lo;
  }
  __syncthreads();
  const long long total_items = static_cast<long long>(num_block_rows) + num_blocks;
  const long long dl0 = static_cast<long long>(t) * TILE, dl1 = dl0 + TILE;
  const int d0 = static_cast<int>(dl0 < total_items ? dl0 : total_items), d1 = static_cast<int>(dl1 < total_items ? dl1 : total_items);
  const int row0 = s_split[0], blk0 = d0 - row0;
  const int nrows = s_split[1] - row0, nblocks = (d1 - s_split[1]) - blk0;
  for (int r = tid; r < nrows; r += TPB) s_re[r] = block_offsets[row0 + r + 1];
  for (int j = tid; j < 4 * (nrows + 1); j += TPB) s_acc[j] = 0.0;
  if (tid < 4) s_acc[4 * (TILE + 1) + tid] = 0.0;
  __syncthreads();
  const int nsteps = (nblocks + 15) >> 4;
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(values + static_cast<std::size_t>(blk0) * 16), 0, nblocks * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(block_cols + blk0), 0, nblocks * 4, 0x00020000);
  const int voff_a = lane * 16, voff_c = q * 4;
  struct batch_t {
    f32x4 a[U];
    unsigned int col[U];
    int row[U];
  };
  auto load = [&](batch_t& b, const int k) {  // the U steps k, k + WAVES, ... of this wavefront; columns first (the gathers wait for them only)
#pragma unroll
    for (int u = 0; u < U; ++u) b.col[u] = __builtin_amdgcn_raw_buffer_load_b32(rc, voff_c, (k + u * WAVES) * 64, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) b.a[u] = __builtin_bit_cast(f32x4, static_cast<u32x4>(__builtin_amdgcn_raw_buffer_load_b128(rv, voff_a, (k + u * WAVES) * 1024, 0)));
#pragma unroll
    for (int u = 0; u < U; ++u) {  // the block-row of the lane's block: first block-row of the tile that ends behind it (nrows: the open one)
      const int kb = (k + u * WAVES) * 16 + q;
      const int blk = blk0 + kb;
      int lo = 0, count = nrows;
      while (count > 0) {
        const int half = count >> 1;
        const int mid = lo + half;
        if (s_re[mid] <= blk) {
          lo = mid + 1;
          count -= half + 1;
        } else {
          count = half;
        }
      }
      b.row[u] = kb < nblocks ? lo : TILE + 1;  // (surplus lanes add into the dump group)
    }
  };
  auto gather = [&](const batch_t& b, f32x4 (&xv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const f32x4*>(x + static_cast<std::size_t>(b.col[u]) * 4);
  };
  auto update = [&](const batch_t& b, const f32x4 (&xv)[U]) {
    f32x4 d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // x as the A operand: every accumulator register of lane i holds row i's product (bcsr_band.hxx)
      d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].x, b.a[u].x, d[u], 0, 0, 0);
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].y, b.a[u].y, d[u], 0, 0, 0);
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].z, b.a[u].z, d[u], 0, 0, 0);
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].w, b.a[u].w, d[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) atomicAdd(&s_acc[4 * b.row[u] + i], static_cast<double>(d[u].x));
  };
  constexpr int STRIDE = WAVES * U;
  int k = w;
  if (k < nsteps) {  // (wave-uniform)
    batch_t a, b;
    f32x4 xa[U], xb[U];
    load(a, k);
    gather(a, xa);
    for (;;) {
      if (k + STRIDE >= nsteps) { update(a, xa); break; }
      load(b, k + STRIDE);
      __builtin_amdgcn_sched_barrier(0);
      update(a, xa);
      gather(b, xb);
      k += STRIDE;
      if (k + STRIDE >= nsteps) { update(b, xb); break; }
      load(a, k + STRIDE);
      __builtin_amdgcn_sched_barrier(0);
      update(b, xb);
      gather(a, xa);
      k += STRIDE;
    }
  }
  __syncthreads();
  for (int j = tid; j < 4 * nrows; j += TPB) {
    const long long r = static_cast<long long>(row0) * 4 + j;
    if (r < rows) y[r] = static_cast<float>(s_acc[j]);
  }
  if (tid < 4) carry_val[static_cast<std::size_t>(t) * 4 + tid] = static_cast<float>(s_acc[4 * nrows + tid]);
  if (tid == 0) carry_row[t] = row0 + nrows;  // (== num_block_rows behind the last block-row: nothing open)
}

/// y[4 r + i] += the carry-outs of the run of tiles that ended inside block-row r (tile order); thread per (tile, i).
__global__ void __launch_bounds__(256)
bcsr_merge_path_fixup(const int* __restrict__ carry_row, const float* __restrict__ carry_val, const int num_tiles, const int num_block_rows,
                      const int rows, float* __restrict__ y) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = idx >> 2, i = idx & 3;
  if (t >= num_tiles) return;
  const int r = carry_row[t];
  if (r >= num_block_rows || (t > 0 && carry_row[t - 1] == r)) return;  // nothing open / not the first tile of the run
  float s = carry_val[static_cast<std::size_t>(t) * 4 + i];
  for (int j = t + 1; j < num_tiles && carry_row[j] == r; ++j) s += carry_val[static_cast<std::size_t>(j) * 4 + i];
  const long long at = static_cast<long long>(r) * 4 + i;
  if (at < rows) y[at] = y[at] + s;
}

/// y = A x for a 4 x 4 fp32 BCSR on the merge-path schedule.  `scratch`: bcsr_merge_scratch_bytes(...) bytes (no initial
/// contents needed); calls that share it must be ordered by the stream.
inline int launch_bcsr4x4_merge_path(hipStream_t stream, int rows, int num_block_rows, int num_blocks, const int* block_offsets,
                                     const int* block_cols, const float* values, const float* x, float* y, void* scratch) {
  if (num_block_rows == 0) return 0;
  if (num_blocks == 0) return rows > 0 ? static_cast<int>(hipMemsetAsync(y, 0, sizeof(float) * static_cast<std::size_t>(rows), stream)) : 0;
  const int m = bcsr_merge_tiles(num_block_rows, num_blocks);
  char* p = static_cast<char*>(scratch);
  float* carry_val = reinterpret_cast<float*>(p);
  int* carry_row = reinterpret_cast<int*>(reinterpret_cast<char*>(carry_val) + sizeof(float) * 4 * static_cast<std::size_t>(m));
  {
    constexpr int W = LOOPS_BCSR_MERGE_WAVES, UU = LOOPS_BCSR_MERGE_U;
    auto* kernel = bcsr4x4_mfma_merge_path<bcsr_merge_tile, W, UU>;
    constexpr std::size_t lds = bcsr_merge_lds_bytes(bcsr_merge_tile);
    if constexpr (lds > 65536) {
      static unsigned long long opted_devices = 0;
      int dev = 0;
      const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
      if (!known || !((opted_devices >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (known) opted_devices |= 1ull << dev;
      }
    }
    hipLaunchKernelGGL(kernel, dim3(m), dim3(W * wave::size), lds, stream, rows, num_block_rows, num_blocks, block_offsets, block_cols, values, x, y,
                       carry_row, carry_val);
  }
  if (m > 1)
    hipLaunchKernelGGL(bcsr_merge_path_fixup, dim3(math::ceil_div(4 * m, 256)), dim3(256), 0, stream, carry_row, carry_val, m, num_block_rows, rows, y);
  return static_cast<int>(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// Which of the two plan-less 4 x 4 fp32 products a matrix should get (callers that cannot measure: the
// bcsr_thread_mapped<4, 4> wrapper, loops_spmv_bcsr_f32 mode "tuned").  The thread_mapped MFMA kernel (bcsr_spmv.hxx) walks the
// block-rows of a wavefront in lockstep, a few blocks of each per step: 17 % faster than the merge-path tiles where the
// lengths are even (2^18 block-rows of 16 blocks, scattered columns: 70 against 84 us), but a block-row of L blocks is a serial
// chain of 0.043 us x L, and a wavefront's 4 block-rows cost 4 x the longest of them.  Measured (tests/perf/exp_bcsr_skew_rule.py,
// profiles/r06_bcsr_skew_rule.txt; MFMA / tiles in us): lengths uniform in 16 +- 4 / 8 / 16: 84 / 83, 89 / 83, 99 / 82 (lockstep
// blocks 1.17 / 1.32 / 1.62 x the blocks); geometric 122 / 81; Pareto tails capped at 64 / 4 096: 84 / 43, 469 / 54; ONE block-row
// of 256 / 512 / 2 048 blocks among even ones: 70 / 83, 75 / 82, 156 / 81 in the middle of the matrix, 79 / 82, 92 / 81 (512) as the
// last block-row; 64 block-rows of 16 384: 1 830 / 39.  Rule: SKEWED when the lockstep walk (sum over groups of 4 consecutive
// block-rows of 4 x the group's longest) touches more than 1.2 x the blocks, or the longest block-row exceeds
// max(64, blocks / 14 000) blocks (its chain alone then costs what the tiles cost extra, should it come last).
constexpr int bcsr_rows_even = 1, bcsr_rows_skewed = 2;  ///< (0 = not known yet)
__host__ __device__ inline int bcsr_row_length_class(unsigned long long lockstep_blocks, unsigned int longest, unsigned long long num_blocks) {
  const unsigned long long cap = num_blocks / 14000ull > 64ull ? num_blocks / 14000ull : 64ull;
  return (longest > cap || 10ull * lockstep_blocks > 12ull * num_blocks) ? bcsr_rows_skewed : bcsr_rows_even;
}
/// Host form over a host copy of the offsets (container/bcsr.hxx builds them on the host anyway).
template <typename offset_t>
inline int bcsr_row_length_class_of(const offset_t* block_offsets, std::size_t num_block_rows) {
  unsigned long long lockstep = 0;
  unsigned int longest = 0;
  for (std::size_t g = 0; g < num_block_rows; g += 4) {
    unsigned int m = 0;
    for (std::size_t r = g; r < g + 4 && r < num_block_rows; ++r) {
      const unsigned int len = static_cast<unsigned int>(block_offsets[r + 1] - block_offsets[r]);
      m = len > m ? len : m;
    }
    lockstep += 4ull * m;
    longest = m > longest ? m : longest;
  }
  return bcsr_row_length_class(lockstep, longest, num_block_rows ? static_cast<unsigned long long>(block_offsets[num_block_rows]) : 0ull);
}
struct bcsr_skew_ctl {
  unsigned long long lockstep;
  unsigned int longest, done;
};
/// Device form: a few workgroups stride over the block-rows (a thread per block-row and step), sum up inside the workgroup and
/// post ONE set of atomics each (a wavefront per set was 87 us on 2^18 block-rows: ~9 000 atomics on two words, ~88 per us); the
/// last workgroup through stores the class to `report` (may be host memory mapped into the device) and leaves `ctl` (ZERO before
/// the first launch) zero again.
__global__ void __launch_bounds__(256)
bcsr_skew_probe(const int num_block_rows, const int num_blocks, const int* __restrict__ block_offsets, bcsr_skew_ctl* __restrict__ ctl,
                unsigned int* __restrict__ report) {
  unsigned long long sum = 0;
  unsigned int top = 0;
  // (the stride is a multiple of 4: the groups of 4 consecutive block-rows stay with 4 neighbouring lanes)
  for (long long r0 = static_cast<long long>(blockIdx.x) * 256; r0 < num_block_rows; r0 += static_cast<long long>(gridDim.x) * 256) {
    const long long r = r0 + threadIdx.x;
    unsigned int m = r < num_block_rows ? static_cast<unsigned int>(block_offsets[r + 1] - block_offsets[r]) : 0u;
    m = max(m, static_cast<unsigned int>(__shfl_xor(static_cast<int>(m), 1)));
    m = max(m, static_cast<unsigned int>(__shfl_xor(static_cast<int>(m), 2)));  // longest of the group of 4 (in every lane of the group)
    sum += m;  // 4 lanes x the group's longest = the group's lockstep blocks
    top = max(top, m);
  }
  for (int d = 1; d < wave::size; d <<= 1) {
    sum += static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(sum), d));
    top = max(top, static_cast<unsigned int>(__shfl_xor(static_cast<int>(top), d)));
  }
  __shared__ unsigned long long s_sum[256 / wave::size];
  __shared__ unsigned int s_top[256 / wave::size];
  if (wave::lane() == 0) {
    s_sum[threadIdx.x / wave::size] = sum;
    s_top[threadIdx.x / wave::size] = top;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 256 / wave::size; ++w) {
      sum += s_sum[w];
      top = max(top, s_top[w]);
    }
    atomicAdd(&ctl->lockstep, sum);
    atomicMax(&ctl->longest, top);
    __threadfence();
    if (atomicAdd(&ctl->done, 1u) == gridDim.x - 1) {
      __threadfence();
      const unsigned long long lockstep = __hip_atomic_load(&ctl->lockstep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int longest = __hip_atomic_load(&ctl->longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(report, static_cast<unsigned int>(bcsr_row_length_class(lockstep, longest, static_cast<unsigned long long>(num_blocks))),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ctl->lockstep = 0ull;
      ctl->longest = 0u;
      ctl->done = 0u;
    }
  }
}
inline int launch_bcsr_skew_probe(hipStream_t stream, int num_block_rows, int num_blocks, const int* block_offsets, bcsr_skew_ctl* ctl,
                                  unsigned int* report) {
  if (num_block_rows <= 0) return 0;
  const int blocks = math::ceil_div(num_block_rows, 256);
  hipLaunchKernelGGL(bcsr_skew_probe, dim3(blocks < 128 ? blocks : 128), dim3(256), 0, stream, num_block_rows, num_blocks, block_offsets, ctl, report);
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
