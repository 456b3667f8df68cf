/**
 * @file bcsr_band.hxx
 * @brief Block-band layout for 4 x 4 fp32 BCSR: the row-band idea (rowband.hxx) carried over to dense blocks.
 *
 * Why.  `bcsr4x4_mfma_spmv` (bcsr_spmv.hxx) issues one 16-byte x gather per block, and on BASELINE C4 (2^18 block-rows x 16
 * blocks, uniformly random block columns over a 4 MB x) every one of the 4.2 M gathers is its own L2 request: 2.23 M stream
 * lines + 4.2 M gather requests share the ~94 reads a CU keeps in flight (profiles/r05_bcsr_c4_pmc_summary.json), 51 us of
 * stream + 16 us of gathers.  Here the blocks of a BAND of HB consecutive block-rows are sorted by block column, so the 16
 * gathers of one wavefront instruction fall on neighbouring 128-byte lines of x and successive instructions of the
 * workgroup's wavefronts re-use them out of the CU's L1: one L2 request per LINE a band touches (C4, HB = 4096: 65 536 blocks
 * over 32 768 lines of x -> ~28 k requests per band instead of 65 k).  The sort scatters a block-row's blocks over the band,
 * which is repaired in LDS: the 4 HB sums of the band live there as fp64 words and every block's four row products are
 * added with `ds_add_f64`.  The block inner product stays on the matrix core: four chained `v_mfma_f32_4x4x1_16b_f32` per
 * 16 blocks, started from a zero accumulator (the chain is the fp32 fma sequence of bcsr_thread_mapped's inner loop over
 * one block).
 *
 * Layout (a re-ordered COPY, built once on the device: one radix sort of (band, block column) keys over the blocks):
 *   blocks sorted by (band = block-row / HB, block column, BCSR position); every band padded to whole STEPS of 16 blocks (one
 *   MFMA batch = one 1 KB wavefront load).  Per block 68 bytes, as in BCSR: the 16 cells (row-major, container/bcsr.hxx:13-17)
 *   and ONE 32-bit word (row code << cbits) | block column; the row code is the block-row inside the band, HB for a padding block
 *   (zero cells, column 0: a dump accumulator), or -- for the blocks of a HUB block-row (>= 1 / 32 of its band's blocks) -- one of
 *   the hub's 16 replicated accumulator groups, picked by the block's place in its step, so that the lanes of one ds_add
 *   instruction never meet in one LDS word (rowband.hxx: hub rows).
 *
 * y = A x:
 *   A  bcsr_band_accumulate  one workgroup per CHUNK = a run of steps of one band: zero 4 (HB + 1) fp64 words of LDS, stream
 *                            the chunk, per block D = cells * x[4] on the MFMA, acc[4 row + i] += double(D[i]); store the
 *                            band's rows -- straight to y when the band is one chunk, else as an fp32 partial vector;
 *   B  rowband_combine       (rowband.hxx, shared) rows of the bands that were cut: partial vectors added in chunk order.
 * y needs no zero-fill; no global atomics.  Products are fp32 fma chains per block, sums across blocks fp64 (LDS atomics of
 * different wavefronts arrive in no fixed order: exactly summable inputs give bcsr_thread_mapped's bits -- the contract the
 * tests pin; otherwise the last bit of a row may differ between runs, see include/loops_amd.h).
 *
 * Replaces the product of algorithms::spmv::bcsr_thread_mapped<4, 4> (reference bcsr_thread_mapped.cuh:36-74) for callers
 * that hold a plan.  No reference counterpart for the layout.
 */
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <loops/kernels/rowband.hxx>
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

namespace bcsr_band {
constexpr int step_blocks = 16;            ///< blocks per step: one MFMA batch, one 16-byte load per lane
constexpr int block_cells = 16;            ///< 4 x 4
constexpr int max_band_block_rows = 4096;  ///< 4 (HB + 1) fp64 accumulators in the 160 KB LDS of a CU
constexpr int min_band_block_rows = 16;
constexpr int max_hubs = rowband::max_hubs;          ///< block-rows per band that get replicated accumulators ("hubs")
constexpr int hub_replicas = rowband::hub_replicas;  ///< accumulator groups per hub: the 16 blocks of a step spread over them
constexpr int hub_share_div = 32;                    ///< a block-row is a hub from 1 / 32 of its band's blocks (at least 64 blocks) on
/// LDS fp64 words of a workgroup of kernel A: the band's block-rows, the dump block-row, the hubs' replicas (4 rows each).
constexpr int lds_words(int HB) { return 4 * (HB + 1 + max_hubs * hub_replicas); }
/// Bits of a row code: block-row inside the band, HB = padding, above HB a hub's replica.
constexpr int row_code_bits(int HB) {
  int b = 0;
  while ((1 << b) <= HB + max_hubs * hub_replicas) ++b;
  return b;
}
}  // namespace bcsr_band

/// Device arrays of a block-band matrix (owned by bcsr_band_storage).
struct bcsr_band_view {
  int rows, num_block_rows, num_block_cols, num_blocks;
  int HB, B;                 ///< block-rows per band (power of two), bands
  int cbits;                 ///< bits of the block column inside a meta word
  int steps;                 ///< 16-block steps incl. padding
  int num_chunks, num_partials, num_multi;
  const float* val;          ///< [steps * 256] blocks in sorted order, cells row-major
  const unsigned int* meta;  ///< [steps * 16] (row code << cbits) | block column
  const int* chunks;         ///< [4 * num_chunks] {band, first step, end step, partial slot or -1}
  const int* multi;          ///< [3 * num_multi] {band, first partial slot, chunks}
  const unsigned short* hubs;  ///< [B * (max_hubs + 1)] per band: the number of hubs, then their block-rows inside the band
  float* partial;            ///< [num_partials * 4 HB]
  int waves, unroll, nt;     ///< kernel A's shape: wavefronts per workgroup (8 | 16), steps per batch (1 | 2 | 4), non-temporal streams
  int max_pieces;
};

namespace bcsr_band {

using f32x4 = float __attribute__((ext_vector_type(4)));

/// Kernel A.  Software pipeline as rowband_accumulate's: the stream loads of the NEXT batch go out behind the gathers of the
/// current one and fly while the current batch is multiplied and added up; loads are branch-free (a step past the chunk's end
/// re-reads the chunk's first step and adds into the dump accumulators).
template <int WAVES, int U, bool NT, typename store_t>
__global__ void __launch_bounds__(WAVES * wave::size)
bcsr_band_accumulate(const int* __restrict__ chunks, const float* __restrict__ val, const unsigned int* __restrict__ meta,
                     const unsigned short* __restrict__ hubs, const float* __restrict__ x, const int HB, const int cbits, const int rows,
                     float* __restrict__ partial, const store_t out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bcsr_band_lds[];
  double* acc = reinterpret_cast<double*>(bcsr_band_lds);  // [lds_words(HB)]: the band's rows, the dump block-row, the hubs' replicas
  constexpr int TPB = WAVES * wave::size;
  const int lane = wave::lane();
  const int q = lane >> 2;  // block of the step
  const int i = lane & 3;   // row of the block this lane loads and adds
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / wave::size);
  const int c = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int band = chunks[4 * c], sb = chunks[4 * c + 1], se = chunks[4 * c + 2], slot = chunks[4 * c + 3];
  const int words = lds_words(HB);
  for (int j = threadIdx.x; j < words; j += TPB) acc[j] = 0.0;
  __syncthreads();
  const unsigned int cmask = (1u << cbits) - 1u;
  struct batch_t {
    f32x4 a[U];
    unsigned int m[U];
    bool live[U];
  };
  // The chunk's two streams as BUFFER loads: a constant per-lane offset + a scalar step offset, so that no load needs a vector
  // address computed per batch (with 64-bit global addresses the compiler recycled registers of loads still in flight for them
  // and waited, vmcnt(0), at the loop header).  Offsets are relative to the chunk: 1 KB / 64 B per step, below 2^31 for any chunk.
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  constexpr int AUX = NT ? 2 : 0;  // (gfx950 cache-policy bits of a buffer load: 2 = nt)
  const int nsteps = se - sb;
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(val + static_cast<long long>(sb) * (step_blocks * block_cells)), 0, nsteps * (step_blocks * block_cells * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(meta + static_cast<long long>(sb) * step_blocks), 0,
                                                                      nsteps * (step_blocks * 4), 0x00020000);
  const int voff_a = lane * 16, voff_m = q * 4;
  auto load = [&](batch_t& t, const int k) {  // the U steps k, k + WAVES, ... of this wavefront (wave-uniform k, relative to the chunk)
    // the words first: the gathers wait for them only (loads return in order), the cells may still be in flight.  Steps past the
    // chunk's end re-read its first step (branch-free) and add into the dump accumulators.
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int s = k + u * WAVES;
      t.live[u] = s < nsteps;
      s = t.live[u] ? s : 0;
      t.m[u] = __builtin_amdgcn_raw_buffer_load_b32(rm, voff_m, s * (step_blocks * 4), AUX);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int s = k + u * WAVES;
      s = s < nsteps ? s : 0;
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rv, voff_a, s * (step_blocks * block_cells * 4), AUX);
      t.a[u] = __builtin_bit_cast(f32x4, raw);
    }
  };
  auto gather = [&](const batch_t& t, f32x4 (&xv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const f32x4*>(x + static_cast<size_t>(t.m[u] & cmask) * 4);
  };
  auto update = [&](const batch_t& t, const f32x4 (&xv)[U]) {
    f32x4 d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // 16 blocks per instruction.  A = x[j] (the same in the block's four lanes), B = cells[lane's row][j]: D[m][n] = sum_j x[j] cells[n][j]
      // for every m, i.e. EVERY accumulator register of lane n holds row n's product -- no cross-lane step, no select
      d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].x, t.a[u].x, d[u], 0, 0, 0);
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].y, t.a[u].y, d[u], 0, 0, 0);
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].z, t.a[u].z, d[u], 0, 0, 0);
      d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u].w, t.a[u].w, d[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned int row = t.live[u] ? (t.m[u] >> cbits) : static_cast<unsigned int>(HB);
      atomicAdd(&acc[row * 4u + static_cast<unsigned int>(i)], static_cast<double>(d[u].x));
    }
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
      __builtin_amdgcn_sched_barrier(0);  // (keep every stream load of the batch ahead of the first wait for a gather)
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
  // hubs: replicas -> the block-row's own four words (one thread per (hub, row of the block), fixed order)
  const unsigned short* hb = hubs + static_cast<long long>(band) * (max_hubs + 1);
  const int nh = hb[0];
  if (static_cast<int>(threadIdx.x) < 4 * nh) {
    const int h = static_cast<int>(threadIdx.x) >> 2, r4 = static_cast<int>(threadIdx.x) & 3;
    const double* rep = acc + 4 * (HB + 1 + h * hub_replicas) + r4;
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < hub_replicas; ++r) sum += rep[4 * r];
    acc[4 * hb[1 + h] + r4] += sum;
  }
  if (nh > 0) __syncthreads();  // (workgroup-uniform)
  const int H = 4 * HB;
  if (slot < 0) {
    const long long row0 = static_cast<long long>(band) * H;
    for (int j = threadIdx.x; j < H && row0 + j < rows; j += TPB) out(static_cast<int>(row0 + j), static_cast<float>(acc[j]));
  } else {  // the chunk's partial vector; rowband_combine adds the band's up
    float* to = partial + static_cast<long long>(slot) * H;
    for (int j = threadIdx.x; j < H; j += TPB) to[j] = static_cast<float>(acc[j]);
  }
}

// ------------------------------------------------------------------------------------------------ plan-time kernels

/// key[b] = band << cbits | block column, item[b] = b, rin[b] = block-row inside the band, or 0x8000 | hub number for the blocks of a
/// hub block-row (one lane per block; its block-row by binary search over the offsets).
template <typename key_t>
__global__ void __launch_bounds__(256)
make_keys(const int* __restrict__ block_offsets, const int* __restrict__ block_cols, const short* __restrict__ hubidx, const int nbr, const int nb,
          const int hshift, const int cbits, const int nbc, key_t* __restrict__ keys, int* __restrict__ item, unsigned short* __restrict__ rin,
          int* __restrict__ bad) {
  const long long b_ll = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b_ll >= nb) return;
  const int b = static_cast<int>(b_ll);
  int br = 0, count = nbr;  // first block-row whose end lies behind b
  while (count > 0) {
    const int half = count >> 1;
    const int mid = br + half;
    if (block_offsets[mid + 1] <= b) {
      br = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  unsigned int col = static_cast<unsigned int>(block_cols[b]);
  if (col >= static_cast<unsigned int>(nbc) || br >= nbr) {  // (also a negative index) flagged, then clamped
    *bad = 1;
    col = 0;
    br = br < nbr ? br : nbr - 1;
  }
  const unsigned int band = static_cast<unsigned int>(br) >> hshift;
  keys[b] = (static_cast<key_t>(band) << cbits) | static_cast<key_t>(col);
  item[b] = b;
  const int hub = hubidx[br];
  rin[b] = hub >= 0 ? static_cast<unsigned short>(0x8000u | static_cast<unsigned int>(hub))
                    : static_cast<unsigned short>(static_cast<unsigned int>(br) - (band << hshift));
}

/// band_start[b] = the first sorted position whose band is >= b (b <= B: band_start[B] = nb).
template <typename key_t>
__global__ void __launch_bounds__(256)
band_starts(const key_t* __restrict__ sorted, const int nb, const int B, const int cbits, int* __restrict__ band_start) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  const key_t want = static_cast<key_t>(static_cast<unsigned int>(b)) << cbits;
  int lo = 0, count = nb;
  if (b == B) lo = nb, count = 0;  // (B << cbits may not fit a 32-bit key)
  while (count > 0) {
    const int half = count >> 1;
    if (sorted[lo + half] < want) {
      lo += half + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  band_start[b] = lo;
}

/// steps[b] = whole steps of band b, steps[B] = 0.
__global__ void __launch_bounds__(256)
band_step_counts(const int* __restrict__ band_start, const int B, int* __restrict__ steps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= B) steps[b] = b < B ? (band_start[b + 1] - band_start[b] + step_blocks - 1) / step_blocks : 0;
}

/// Every slot starts as a padding block: zero cells, row code HB, column 0, no BCSR position.
__global__ void __launch_bounds__(256)
fill_padding(const long long slots, const unsigned int pad_word, float* __restrict__ val, unsigned int* __restrict__ meta, int* __restrict__ perm) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;  // one thread per 16 bytes of cells
  if (t >= slots * 4) return;
  *reinterpret_cast<f32x4*>(val + t * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  if ((t & 3) == 0) {
    meta[t >> 2] = pad_word;
    perm[t >> 2] = -1;
  }
}

/// Sorted block j -> its slot: four threads move the block's 64 bytes, the first writes its word.
template <typename key_t>
__global__ void __launch_bounds__(256)
place(const key_t* __restrict__ sorted, const int* __restrict__ item, const unsigned short* __restrict__ rin, const int* __restrict__ band_start,
      const int* __restrict__ band_step, const float* __restrict__ values, const int nb, const int cbits, const int HB, float* __restrict__ val,
      unsigned int* __restrict__ meta, int* __restrict__ perm) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long j = t >> 2;
  if (j >= nb) return;
  const int part = static_cast<int>(t & 3);
  const key_t key = sorted[j];
  const unsigned int col = static_cast<unsigned int>(key & ((static_cast<key_t>(1) << cbits) - 1));
  const int b = static_cast<int>(key >> cbits);
  const long long s = static_cast<long long>(band_step[b]) * step_blocks + (j - band_start[b]);
  const int src = item[j];
  *reinterpret_cast<f32x4*>(val + s * block_cells + part * 4) = *reinterpret_cast<const f32x4*>(values + static_cast<long long>(src) * block_cells + part * 4);
  if (part == 0) {
    // row code: the block-row inside the band, or -- a hub's block -- one of the hub's replicas, picked by the block's place in its
    // step (the 16 blocks of one ds_add instruction go to 16 different accumulator groups)
    const unsigned int c = rin[src];
    const unsigned int code = (c & 0x8000u) ? static_cast<unsigned int>(HB) + 1u + (c & 0x7FFFu) * hub_replicas + static_cast<unsigned int>(s % step_blocks) : c;
    meta[s] = (code << cbits) | col;
    perm[s] = src;
  }
}

__global__ void __launch_bounds__(256)
refresh_values(const int* __restrict__ perm, const float* __restrict__ values, const long long slots, float* __restrict__ val) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= slots * 4) return;
  const int src = perm[t >> 2];
  const int part = static_cast<int>(t & 3);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (src >= 0) v = *reinterpret_cast<const f32x4*>(values + static_cast<long long>(src) * block_cells + part * 4);
  *reinterpret_cast<f32x4*>(val + t * 4) = v;
}

}  // namespace bcsr_band

/// The device arrays of one block-band matrix, OWNED.
struct bcsr_band_storage {
  int rows = 0, num_block_rows = 0, num_block_cols = 0, num_blocks = 0;
  int HB = 0, B = 0, cbits = 0, steps = 0, num_chunks = 0, num_partials = 0, num_multi = 0, target_chunks = 0, cus = 0;
  int waves = 16, unroll = 1, nt = 0, max_pieces = 1;  ///< (16 wavefronts x 1 step: best or within 2 % of the best shape on every size measured)
  float *val = nullptr, *partial = nullptr;
  unsigned int* meta = nullptr;
  unsigned short* hubs = nullptr;
  int *perm = nullptr, *chunks = nullptr, *multi = nullptr, *band_step = nullptr;

  bcsr_band_storage() = default;
  bcsr_band_storage(const bcsr_band_storage&) = delete;
  bcsr_band_storage& operator=(const bcsr_band_storage&) = delete;
  ~bcsr_band_storage() { release(); }
  void release() {
    (void)hipFree(val); (void)hipFree(partial); (void)hipFree(meta); (void)hipFree(perm); (void)hipFree(chunks); (void)hipFree(multi);
    (void)hipFree(band_step); (void)hipFree(hubs);
    val = partial = nullptr; meta = nullptr; hubs = nullptr; perm = chunks = multi = band_step = nullptr;
  }
  bcsr_band_view view() const {
    return bcsr_band_view{rows, num_block_rows, num_block_cols, num_blocks, HB, B, cbits, steps, num_chunks, num_partials, num_multi,
                          val, meta, chunks, multi, hubs, partial, waves, unroll, nt, max_pieces};
  }
  /// bytes one product streams (cells + words + partial vectors both ways)
  double stream_bytes() const {
    return static_cast<double>(steps) * bcsr_band::step_blocks * 68.0 + 8.0 * static_cast<double>(num_partials) * 4.0 * HB;
  }
};

/// Block-rows per band: the tallest power of two that still leaves one band per compute unit (no band is cut then: no partial
/// vectors, no second kernel), between 64 and 4096 (the fp64 sums of a band fill the LDS), halved while a word cannot hold row
/// code + column.  Measured on MI355X, 16 blocks per block-row, tests/perf/bench_bcsr_band.py (us, uncut height | 4096 cut):
/// 2^15 block-rows 8.8 | 18.0, 2^16 13.5 | 20.3, 2^17 23.9 | 29.3, 2^18 (C4) 56.0 | 57.4, 2^19 116.3 | 115.7 -- the taller band shares
/// more lines of x (A alone: 50.7 against 55.5 us on C4) but pays 5-7 us for the partial vectors' round trip.
inline int bcsr_band_block_rows(int num_block_rows, int cbits, int cus) {
  const int c = cus > 0 ? cus : 256;
  int h = bcsr_band::max_band_block_rows;
  while (h > 64 && num_block_rows / h < c) h /= 2;
  while (h > bcsr_band::min_band_block_rows && bcsr_band::row_code_bits(h) + cbits > 32) h /= 2;
  return h;
}

inline int bcsr_band_set_chunks(bcsr_band_storage& out, const std::vector<int>& band_step_host, int target_chunks) {
  out.target_chunks = target_chunks > 0 ? target_chunks : rowband_target_chunks(out.B, out.cus);
  if (target_chunks <= 0 && out.target_chunks == out.B && out.B > 0) {
    // one chunk per band is right while the bands are about equally long; a band of more than twice the mean (hub block-rows) is
    // cut into pieces of at most that: 64 hub block-rows of 16 384 blocks among 2^17 of 8 -- 56 us uncut (the bands that hold a hub
    // run five times as long as the others), 32 us with equal chunks (tests/perf/exp_bcsr_band_hubs.py)
    const long long total = band_step_host[static_cast<std::size_t>(out.B)];
    const long long cap = std::max<long long>(2 * ((total + out.B - 1) / out.B), 1);
    long long extra = 0;
    for (int b = 0; b < out.B; ++b) {
      const long long n = band_step_host[static_cast<std::size_t>(b) + 1] - band_step_host[static_cast<std::size_t>(b)];
      if (n > cap) extra += (n + cap - 1) / cap - 1;
    }
    out.target_chunks += static_cast<int>(std::min<long long>(extra, 1 << 20));
  }
  std::vector<int> chunks, multi;
  rowband_chunk_list(band_step_host, out.B, out.target_chunks, chunks, multi, out.num_partials);
  {
    // kernel A addresses a chunk's streams with 32-bit byte offsets (1 KB per step): no chunk may pass 2^21 steps.  A cut that
    // leaves one (an explicit target on an enormous, skewed matrix) is refined until none does.
    constexpr long long max_chunk_steps = (1ll << 21) - 1;
    auto longest = [&]() {
      long long m = 0;
      for (std::size_t c = 0; c + 3 < chunks.size(); c += 4) m = std::max<long long>(m, static_cast<long long>(chunks[c + 2]) - chunks[c + 1]);
      return m;
    };
    for (int round = 0; round < 8 && longest() > max_chunk_steps; ++round) {
      const long long total = out.B > 0 ? band_step_host[static_cast<std::size_t>(out.B)] : 0;
      out.target_chunks = static_cast<int>(std::min<long long>(std::max<long long>(2ll * out.target_chunks, total / (max_chunk_steps / 2) + out.B), 1 << 24));
      rowband_chunk_list(band_step_host, out.B, out.target_chunks, chunks, multi, out.num_partials);
    }
    if (longest() > max_chunk_steps) return rowband_e_range;
  }
  out.num_chunks = static_cast<int>(chunks.size() / 4);
  out.num_multi = static_cast<int>(multi.size() / 3);
  out.max_pieces = 1;
  for (std::size_t m = 0; m < multi.size() / 3; ++m) out.max_pieces = std::max(out.max_pieces, multi[3 * m + 2]);
  (void)hipFree(out.chunks); (void)hipFree(out.multi); (void)hipFree(out.partial);
  out.chunks = out.multi = nullptr;
  out.partial = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&out.chunks), sizeof(int) * (chunks.empty() ? 4 : chunks.size()));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&out.multi), sizeof(int) * (multi.empty() ? 3 : multi.size()));
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void**>(&out.partial), sizeof(float) * (out.num_partials > 0 ? static_cast<std::size_t>(out.num_partials) * 4 * out.HB : 4));
  if (e == hipSuccess && !chunks.empty()) e = hipMemcpy(out.chunks, chunks.data(), sizeof(int) * chunks.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !multi.empty()) e = hipMemcpy(out.multi, multi.data(), sizeof(int) * multi.size(), hipMemcpyHostToDevice);
  // non-temporal streams once a product's working set is well beyond the Infinity Cache (C4, 295 MB: plain 56.5 us, non-temporal
  // 57.4; 2^19 block-rows, 590 MB: non-temporal ahead)
  out.nt = out.stream_bytes() + 4.0 * (static_cast<double>(out.rows) + 4.0 * out.num_block_cols) > 400e6 ? 1 : 0;
  return static_cast<int>(e);
}

/// Builds the block-band copy of a 4 x 4 fp32 BCSR on the device.  band_block_rows: 0 = automatic, else a power of two in
/// [16, 4096]; target_chunks: 0 = automatic.  Returns 0, a hipError_t, rowband_e_badarg (also: a block column outside
/// [0, num_block_cols)) or rowband_e_range (row code + block column do not fit 32 bits at the band height asked for).
inline int bcsr_band_create(hipStream_t stream, int rows, int num_block_rows, int num_block_cols, int num_blocks, const int* block_offsets,
                            const int* block_cols, const float* values, int band_block_rows, int target_chunks, bcsr_band_storage& out) {
  namespace bb = bcsr_band;
  if (!block_offsets || rows < 0 || num_block_rows < 0 || num_block_cols < 0 || num_blocks < 0 || (num_blocks > 0 && (!block_cols || !values)) ||
      target_chunks < 0 || static_cast<long long>(rows) > 4ll * num_block_rows)
    return rowband_e_badarg;
  out.release();
  out.rows = rows; out.num_block_rows = num_block_rows; out.num_block_cols = num_block_cols; out.num_blocks = num_blocks;
  int cbits = 1;  // bits of the largest block column
  while (cbits < 31 && ((static_cast<long long>(num_block_cols) - 1) >> cbits) > 0) ++cbits;
  out.cbits = cbits;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&out.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) out.cus = 256;
  }
  out.HB = band_block_rows != 0 ? band_block_rows : bcsr_band_block_rows(num_block_rows, cbits, out.cus);
  if (out.HB < bb::min_band_block_rows || out.HB > bb::max_band_block_rows || (out.HB & (out.HB - 1))) return rowband_e_badarg;
  int hshift = 0;
  while ((1 << hshift) < out.HB) ++hshift;
  if (bb::row_code_bits(out.HB) + cbits > 32) return rowband_e_range;
  out.B = num_block_rows > 0 ? (num_block_rows + out.HB - 1) / out.HB : 0;
  out.steps = out.num_chunks = out.num_partials = out.num_multi = 0;
  if (num_block_rows == 0) return 0;
  const int B = out.B, nb = num_blocks;
  if (static_cast<long long>(nb) + static_cast<long long>(B) * bb::step_blocks >= (1ll << 31) - 4096) return rowband_e_range;
  int bbits = 1;
  while (bbits < 31 && (static_cast<long long>(B) >> bbits) != 0) ++bbits;
  const bool wide = cbits + bbits > 32;

  auto up = [](std::size_t v) { return (v + 255) & ~std::size_t(255); };
  std::size_t sort_bytes = 0, scan_bytes = 0;
  {
    unsigned long long* k8 = nullptr;
    unsigned int* k4 = nullptr;
    int* ci = nullptr;
    if (wide) (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k8, k8, ci, ci, nb, 0, cbits + bbits);
    else (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k4, k4, ci, ci, nb, 0, cbits + bbits);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, ci, ci, B + 1);
  }
  const std::size_t cub_bytes_total = up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
  const std::size_t key_bytes = up((static_cast<std::size_t>(nb) + 1) * (wide ? 8 : 4)), item_bytes = up((static_cast<std::size_t>(nb) + 1) * 4);
  const std::size_t rin_bytes = up((static_cast<std::size_t>(nb) + 1) * 2), band_bytes = up((static_cast<std::size_t>(B) + 1) * 4);
  const std::size_t hub_bytes = up((static_cast<std::size_t>(num_block_rows) + 1) * 2);
  const std::size_t temp_bytes = 2 * key_bytes + 2 * item_bytes + rin_bytes + hub_bytes + 2 * band_bytes + 256 + cub_bytes_total;
  char* base = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), temp_bytes);
  struct guard_t {
    char*& a;
    ~guard_t() { (void)hipFree(a); }
  } guard{base};
  if (e != hipSuccess) return static_cast<int>(e);
  char* at = base;
  auto carve = [&](std::size_t bytes) { char* p = at; at += bytes; return p; };
  void* keys_in = carve(key_bytes);
  void* keys_out = carve(key_bytes);
  int* item_in = reinterpret_cast<int*>(carve(item_bytes));
  int* item_out = reinterpret_cast<int*>(carve(item_bytes));
  auto* rin = reinterpret_cast<unsigned short*>(carve(rin_bytes));
  auto* hubidx = reinterpret_cast<short*>(carve(hub_bytes));
  int* band_start = reinterpret_cast<int*>(carve(band_bytes));
  int* steps_of = reinterpret_cast<int*>(carve(band_bytes));
  int* bad = reinterpret_cast<int*>(carve(256));
  void* cub_temp = carve(cub_bytes_total);
  std::size_t cub_bytes = cub_bytes_total;

  const std::size_t hubs_n = static_cast<std::size_t>(B) * (bb::max_hubs + 1);
  e = hipMalloc(reinterpret_cast<void**>(&out.hubs), sizeof(unsigned short) * hubs_n);
  if (e == hipSuccess) e = hipMemsetAsync(out.hubs, 0, sizeof(unsigned short) * hubs_n, stream);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(int), stream);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  // hub block-rows per band (rowband's kernel over the block offsets: a block-row holding >= 1 / 32 of its band's blocks)
  hipLaunchKernelGGL((rowband::find_hubs<int>), dim3(B), dim3(256), 0, stream, block_offsets, num_block_rows, out.HB, hubidx, out.hubs, bb::hub_share_div);
  auto sorted_phase = [&](auto key_tag) -> hipError_t {
    using key_t = decltype(key_tag);
    auto* kin = static_cast<key_t*>(keys_in);
    auto* kout = static_cast<key_t*>(keys_out);
    if (nb > 0) {
      hipLaunchKernelGGL((bb::make_keys<key_t>), dim3(math::ceil_div(nb, 256)), dim3(256), 0, stream, block_offsets, block_cols, hubidx, num_block_rows,
                         nb, hshift, cbits, num_block_cols, kin, item_in, rin, bad);
      const hipError_t se = hipcub::DeviceRadixSort::SortPairs(cub_temp, cub_bytes, kin, kout, item_in, item_out, nb, 0, cbits + bbits, stream);
      if (se != hipSuccess) return se;
    }
    hipLaunchKernelGGL((bb::band_starts<key_t>), dim3(math::ceil_div(B + 1, 256)), dim3(256), 0, stream, kout, nb, B, cbits, band_start);
    return hipSuccess;
  };
  e = wide ? sorted_phase(static_cast<unsigned long long>(0)) : sorted_phase(static_cast<unsigned int>(0));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&out.band_step), sizeof(int) * (static_cast<std::size_t>(B) + 1));
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  hipLaunchKernelGGL(bb::band_step_counts, dim3(math::ceil_div(B + 1, 256)), dim3(256), 0, stream, band_start, B, steps_of);
  cub_bytes = cub_bytes_total;
  e = hipcub::DeviceScan::ExclusiveSum(cub_temp, cub_bytes, steps_of, out.band_step, B + 1, stream);
  std::vector<int> bs(static_cast<std::size_t>(B) + 1, 0);
  int h_bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(bs.data(), out.band_step, sizeof(int) * bs.size(), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  if (h_bad != 0) { out.release(); return rowband_e_badarg; }
  out.steps = bs[B];
  const std::size_t slots = static_cast<std::size_t>(out.steps > 0 ? out.steps : 1) * bb::step_blocks;
  auto alloc = [&](auto** ptr, std::size_t bytes) { if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(ptr), bytes); };
  alloc(&out.val, sizeof(float) * slots * bb::block_cells);
  alloc(&out.meta, sizeof(unsigned int) * slots);
  alloc(&out.perm, sizeof(int) * slots);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  hipLaunchKernelGGL(bb::fill_padding, dim3(static_cast<unsigned int>((slots * 4 + 255) / 256)), dim3(256), 0, stream, static_cast<long long>(slots),
                     static_cast<unsigned int>(out.HB) << cbits, out.val, out.meta, out.perm);
  if (nb > 0) {
    const dim3 grid(static_cast<unsigned int>((static_cast<long long>(nb) * 4 + 255) / 256));
    if (wide)
      hipLaunchKernelGGL((bb::place<unsigned long long>), grid, dim3(256), 0, stream, static_cast<const unsigned long long*>(keys_out), item_out, rin,
                         band_start, out.band_step, values, nb, cbits, out.HB, out.val, out.meta, out.perm);
    else
      hipLaunchKernelGGL((bb::place<unsigned int>), grid, dim3(256), 0, stream, static_cast<const unsigned int*>(keys_out), item_out, rin, band_start,
                         out.band_step, values, nb, cbits, out.HB, out.val, out.meta, out.perm);
  }
  e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = static_cast<hipError_t>(bcsr_band_set_chunks(out, bs, target_chunks));
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  if (band_block_rows == 0 && target_chunks == 0 && out.B > 0) {
    // Automatic height, one band per CU -- right while the bands are about equally long.  When some band is more than twice the
    // mean (hub block-rows), equal CHUNKS of the tallest band balance better than whole bands of unequal length: the same matrix
    // again at the tallest height the words allow, cut into one round of chunks (64 hubs of 16 384 blocks among 2^17 block-rows
    // of 8: 47.8 us at the uncut height with its long bands cut, 31.4 us tall; tests/perf/exp_bcsr_band_hubs.py)
    long long longest = 0;
    for (int b = 0; b < out.B; ++b) longest = std::max<long long>(longest, bs[static_cast<std::size_t>(b) + 1] - bs[static_cast<std::size_t>(b)]);
    int tall = bb::max_band_block_rows;
    while (tall > bb::min_band_block_rows && bb::row_code_bits(tall) + cbits > 32) tall /= 2;
    if (longest * out.B > 2 * static_cast<long long>(bs[static_cast<std::size_t>(B)]) && tall > out.HB)
      return bcsr_band_create(stream, rows, num_block_rows, num_block_cols, num_blocks, block_offsets, block_cols, values, tall, 0, out);
  }
  return 0;
}

namespace bcsr_band {
/// One launch of kernel A in shape <WAVES, U, NT>.  Bands taller than 64 KB of accumulators need the kernel opted into the
/// large LDS: once per instantiation and device, not per launch.
template <int WAVES, int U, bool NT, typename store_t>
inline void launch_accumulate(hipStream_t stream, const bcsr_band_view& m, const float* x, const store_t out) {
  const std::size_t lds = static_cast<std::size_t>(lds_words(m.HB)) * sizeof(double);
  auto* kernel = bcsr_band_accumulate<WAVES, U, NT, store_t>;
  if (lds > 65536) {
    static unsigned long long opted_devices = 0;  // (one per instantiation of this function template)
    kernels::detail::allow_large_lds(reinterpret_cast<const void*>(kernel), opted_devices);
  }
  hipLaunchKernelGGL(kernel, dim3(m.num_chunks), dim3(WAVES * wave::size), lds, stream, m.chunks, m.val, m.meta, m.hubs, x, m.HB, m.cbits, m.rows, m.partial,
                     out);
}
}  // namespace bcsr_band

/// y = A x over a block-band matrix: kernel A, then kernel B if some band was cut.  stages: bit 0 = accumulate, bit 1 = combine.
template <typename store_t>
int launch_bcsr_band_to(hipStream_t stream, const bcsr_band_view& m, const float* x, const store_t out, int stages = 3) {
  namespace bb = bcsr_band;
  if (m.rows == 0 || m.num_block_rows == 0) return 0;
  if ((stages & 1) && m.num_chunks > 0) {
    auto with_nt = [&](auto w_tag, auto u_tag) {
      constexpr int W = decltype(w_tag)::value, UU = decltype(u_tag)::value;
      if (m.nt) bb::launch_accumulate<W, UU, true>(stream, m, x, out);
      else bb::launch_accumulate<W, UU, false>(stream, m, x, out);
    };
    auto with_u = [&](auto w_tag) {
      switch (m.unroll) {
        case 1: with_nt(w_tag, std::integral_constant<int, 1>{}); break;
        case 4: with_nt(w_tag, std::integral_constant<int, 4>{}); break;
        default: with_nt(w_tag, std::integral_constant<int, 2>{}); break;
      }
    };
    if (m.waves == 16) with_u(std::integral_constant<int, 16>{});
    else with_u(std::integral_constant<int, 8>{});
  }
  if ((stages & 2) && m.num_multi > 0) launch_rowband_combine<float>(stream, m.multi, m.num_multi, m.max_pieces, m.partial, 4 * m.HB, m.rows, out);
  return static_cast<int>(hipGetLastError());
}

inline int launch_bcsr_band(hipStream_t stream, const bcsr_band_view& m, const float* x, float* y, int stages = 3) {
  return launch_bcsr_band_to(stream, m, x, plain_store<float>{y}, stages);
}

/// Measures the product in every compiled shape of kernel A (wavefronts per workgroup x steps per batch x stream policy; x =
/// zeros: the time does not depend on the values) and keeps the fastest in `m`.  ms12 (may be null) receives the times in the
/// order (waves 8 | 16) x (unroll 1 | 2 | 4) x (plain | non-temporal).  Synchronous.
inline int bcsr_band_tune(hipStream_t stream, bcsr_band_storage& m, int repeats, float* ms12) {
  if (ms12) for (int i = 0; i < 12; ++i) ms12[i] = -1.f;
  if (m.rows == 0 || m.steps == 0) return 0;
  if (repeats < 1) repeats = 10;
  float *x = nullptr, *y = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const std::size_t xn = static_cast<std::size_t>(m.num_block_cols > 0 ? m.num_block_cols : 1) * 4;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&x), sizeof(float) * xn);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&y), sizeof(float) * static_cast<std::size_t>(m.rows));
  if (e == hipSuccess) e = hipMemsetAsync(x, 0, sizeof(float) * xn, stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  int err = static_cast<int>(e);
  // the shape the plan came with is measured first and stays unless another one is MEASURABLY (2 %) faster: the shapes differ by
  // a microsecond or two, which is also what two measurements of ONE shape differ by
  const int w0 = m.waves, u0 = m.unroll, nt0 = m.nt;
  float best = 0.f, first = 0.f;
  int best_w = w0, best_u = u0, best_nt = nt0, n = 0;
  auto time_shape = [&](int w, int u, int nt, float& ms) {
    m.waves = w; m.unroll = u; m.nt = nt;
    const bcsr_band_view v = m.view();
    for (int it = 0; !err && it < 3; ++it) err = launch_bcsr_band(stream, v, x, y);
    if (!err) err = static_cast<int>(hipEventRecord(e0, stream));
    for (int it = 0; !err && it < repeats; ++it) err = launch_bcsr_band(stream, v, x, y);
    if (!err) err = static_cast<int>(hipEventRecord(e1, stream));
    if (!err) err = static_cast<int>(hipEventSynchronize(e1));
    ms = 0.f;
    if (!err) err = static_cast<int>(hipEventElapsedTime(&ms, e0, e1));
    ms /= static_cast<float>(repeats);
  };
  if (!err) time_shape(w0, u0, nt0, first);
  best = first;
  for (int w : {8, 16})
    for (int u : {1, 2, 4})
      for (int nt : {0, 1}) {
        if (err) break;
        float ms = first;
        if (!(w == w0 && u == u0 && nt == nt0)) time_shape(w, u, nt, ms);
        if (err) break;
        if (ms12) ms12[n] = ms;
        if (ms < 0.98f * first && ms < best) { best = ms; best_w = w; best_u = u; best_nt = nt; }
        ++n;
      }
  m.waves = best_w; m.unroll = best_u; m.nt = best_nt;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(x);
  (void)hipFree(y);
  return err;
}

}  // namespace kernels
}  // namespace loops
