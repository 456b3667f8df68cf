/**
 * @file rowband.hxx
 * @brief Row-band layout: SpMV with the y accumulators of a band of rows in LDS and x read through sorted, coalescing
 *        gathers -- the inverse of panel_binned.hxx (there: x in LDS, the products travel through memory).
 *
 * Why.  Every CSR kernel on this chip issues one L2 request per nonzero for its x gather, and the L2 serves ~266 G of them
 * per second at best: 63 us on BASELINE C2 whatever the kernel does, 0.29 of the HBM roofline.  The
 * panel-binned layout removes the gather but pays for it with 17 bytes of traffic per nonzero.  This layout keeps 8 bytes
 * per nonzero (7 since the columns are stored as deltas) AND removes most of the requests: inside a BAND of H consecutive rows the nonzeros are sorted by COLUMN, so the
 * 64 gathers of one wavefront instruction fall on a handful of neighbouring 128-byte lines of x (the band of a C2-like
 * matrix holds one nonzero per 4-8 columns) and the CU's L1 turns them into one L2 request per LINE.  What the sort destroys
 * -- a row's nonzeros are no longer adjacent -- is repaired where it is cheap: the band's H sums live in LDS as fp64 words
 * and every product is one `ds_add_f64` (3-8 lanes per clock and CU on gfx950, profiles/r03_lds_update_rates.txt).
 *
 * Layout (a re-ordered COPY of the matrix, built once on the device, O(nnz): one stable radix sort of (band, column) keys):
 *   items sorted by (band b = row / H, column, CSR order); every band padded to whole STEPS of 256 slots (one wavefront load: 64
 *   lanes x 4 slots).  Per slot 7 BYTES: the value (4), a row code (2: the row inside the band; H = padding, a dump accumulator)
 *   and the column DELTA to the previous slot (1: the columns of a band ascend, C2's by 4 on average); a gap of more than 255
 *   columns is bridged by padding slots of delta 255; per step and group of 64 slots the absolute column of the group's first
 *   slot (`stepbase`, 16 B per step; that slot's own delta is stored as 0), so a lane's column is the group's base + the 64-lane prefix sum of the deltas (DPP).
 *   Inside a step the slots are INTERLEAVED: sorted position q sits at lane (q % 64), element (q / 64) of the lane's vectors, so
 *   the streams are read with 16- / 8- / 4-byte loads AND the 64 lanes of gather instruction e hold 64 CONSECUTIVE sorted slots
 *   (neighbouring columns: few lines per instruction, quads of lanes share a line).
 *
 * y = A x:
 *   A  rowband_accumulate   one workgroup per CHUNK = a run of steps of ONE band: zero H fp64 words of LDS, stream the chunk,
 *                           acc[row] += double(val * x[column]), then store the H sums -- straight to y when the band is
 *                           one chunk, else as an fp32 partial vector;
 *   B  rowband_combine      bands cut into several chunks (few, long bands: the chunk is the unit of parallelism): y[r] = the
 *                           partial vectors of r's band (each a chunk's fp64 sums rounded to the value type) added in chunk
 *                           order in fp64 and rounded to the value type again.  8 H / chunk_items bytes
 *                           per nonzero of extra traffic; not launched when every band is one chunk.
 * y needs no zero-fill; no global atomics.  Products are fp32 (one rounding each, as in every other kernel here), all sums
 * fp64: exactly summable inputs give the CSR kernels' bits; the LDS atomics of different wavefronts arrive in no fixed order,
 * which cannot change an EXACT fp64 sum (fp32 products spanning < 53 - 24 - log2 n binary orders of magnitude per row).
 *
 * When it pays: nonzeros per band / columns spanned >= ~1/8, i.e. x of a few MB (C2) or column locality at band scale
 * (web graphs in crawl order, FEM bands).  With scattered columns over an x of tens of MB every gather is its own line
 * again (C5 shards, uniform C3 stand-in): panel-binned territory.  The SpMV plan adopts it by measurement (LOOPS_PLAN_MEASURE) or,
 * unmeasured, by size (x of 2-6 MB under rows of >= 8 nonzeros) -- never under LOOPS_PLAN_DETERMINISTIC: the LDS atomics of
 * different wavefronts arrive in no fixed order (see "y needs no zero-fill" above for when that cannot matter).
 * No reference counterpart (the reference's merge_path_flat.cuh:71-82 pays one global atomic per nonzero).
 */
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <queue>
#include <type_traits>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/util/math.hxx>
#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

namespace rowband {
constexpr int step_items = wave::size * 4;  ///< items of one step: one 16-byte load per lane
constexpr int max_delta = 255;              ///< column deltas are one byte: longer gaps are bridged by padding slots
constexpr int max_band_rows = 16384;        ///< (H + 1) fp64 accumulators in the 160 KB LDS of a CU
constexpr int max_hubs = 32;                ///< rows per band that get replicated accumulators ("hubs")
constexpr int hub_replicas = 16;            ///< accumulators per hub: the lanes of one instruction spread over them
/// LDS words of a workgroup of kernel A: H accumulators, the dump word, the hubs' replicas.
constexpr int lds_words(int H) { return H + 1 + max_hubs * hub_replicas; }
}  // namespace rowband

/// Device arrays of a row-band matrix (owned by rowband_storage).
template <typename type_t>
struct rowband_view {
  int rows, cols, nnz;
  int H, B;                    ///< rows per band (power of two), bands
  int steps;                   ///< 256-item steps incl. padding
  int num_chunks;              ///< workgroups of kernel A
  int num_partials;            ///< partial vectors = chunks of bands that hold more than one
  int num_multi;               ///< bands that hold more than one chunk
  const type_t* val;           ///< [steps * 256] interleaved inside a step
  const unsigned int* meta;    ///< [steps * 192] per lane of a step 3 words: its 4 row codes (16 bits each: the row inside the band; H =
                               ///< padding; above H: a hub's replica), then its 4 column deltas (8 bits each, to the previous sorted slot)
  const int* stepbase;         ///< [steps * 4] absolute column of the first slot of each group of 64 sorted slots
  const int* chunks;           ///< [4 * num_chunks] {band, first step, end step, partial slot or -1}
  const int* multi;            ///< [3 * num_multi] {band, first partial slot, chunks}
  const unsigned short* hubs;  ///< [B * (max_hubs + 1)] per band: the number of hubs, then their rows inside the band
  type_t* partial;             ///< [num_partials * H]
  int waves;                   ///< wavefronts per workgroup of kernel A: 8 or 16
  int max_pieces;              ///< most chunks any band is cut into (picks the form of kernel B)
};

namespace rowband {

/// Kernel A.  WAVES wavefronts per workgroup, U steps per wavefront and batch (2 U 16-byte stream loads, then 4 U gathers).
/// Software-pipelined: the stream loads of the NEXT batch are issued behind the gathers of the current one (the memory counter
/// retires in order: waiting for the gathers then does not wait for the stream) and fly while the current batch is added up.
/// Loads are branch-free (a step past the chunk's end re-reads the chunk's first step and adds into the dump word).
/// Hub rows (rowband.hxx, file comment): row codes above H address one of hub_replicas accumulators per hub, folded into the
/// hub's own word before the band's rows are stored.
template <int WAVES, int U, bool NT, typename type_t, typename store_t>
__global__ void __launch_bounds__(WAVES * wave::size)
rowband_accumulate(const int* __restrict__ chunks, const type_t* __restrict__ val, const unsigned int* __restrict__ meta,
                   const int* __restrict__ stepbase, const unsigned short* __restrict__ hubs, const type_t* __restrict__ x, const int H,
                   const int rows, type_t* __restrict__ partial, const store_t out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rowband_lds[];
  double* acc = reinterpret_cast<double*>(rowband_lds);  // [lds_words(H)]: rows, the dump word H, the hubs' replicas
  constexpr int TPB = WAVES * wave::size;
  const int lane = wave::lane();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / wave::size);
  const int c = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int band = chunks[4 * c], sb = chunks[4 * c + 1], se = chunks[4 * c + 2], slot = chunks[4 * c + 3];
  const int words = lds_words(H);
  for (int j = threadIdx.x; j < words; j += TPB) acc[j] = 0.0;
  __syncthreads();
  using u32x3 = unsigned int __attribute__((ext_vector_type(3)));
  using u32x3_ld = unsigned int __attribute__((ext_vector_type(3), aligned(4)));
  using i32x4 = int __attribute__((ext_vector_type(4)));
  struct batch_t {
    type_t v[U][4];
    u32x3 m[U];  // row codes 0 | 1, row codes 2 | 3, four one-byte deltas
    i32x4 base[U];
    bool live[U];
  };
  // the group bases of the batch AFTER the one being loaded are requested one batch ahead: a scalar load's wait (lgkmcnt(0)) also
  // waits for every LDS atomic in flight, so it must find its data long there
  i32x4 base_ahead[U];
  auto prefetch_bases = [&](const int k) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = k + u * WAVES;
      base_ahead[u] = *reinterpret_cast<const i32x4*>(stepbase + 4 * static_cast<long long>(s < se ? s : sb));
    }
  };
  auto load = [&](batch_t& t, const int k) {  // the U steps k, k + WAVES, ... of this wavefront (wave-uniform k)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int s = k + u * WAVES;
      t.live[u] = s < se;
      s = t.live[u] ? s : sb;
      const long long at = static_cast<long long>(s) * step_items + lane * 4;
      const unsigned int* mp = meta + static_cast<long long>(s) * (3 * wave::size) + lane * 3;  // one 12-byte load per lane
      if constexpr (NT) t.m[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x3_ld*>(mp));
      else t.m[u] = *reinterpret_cast<const u32x3_ld*>(mp);
      detail::load4<type_t, NT>(val + at, t.v[u]);
      t.base[u] = base_ahead[u];
    }
    prefetch_bases(k + WAVES * U);
  };
  auto gather = [&](const batch_t& t, type_t (&xv)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // column of sorted slot 64 e + lane = the group's base + the deltas of lanes 1 .. lane (the first slot of a group carries
      // delta 0: the base IS its column).  Two groups per prefix sum: their running sums stay below 64 * 255 < 2^16.
      const unsigned int d = t.m[u].z;
      const unsigned int s01 = wave::inclusive_sum((d & 0xFFu) | ((d & 0xFF00u) << 8));
      const unsigned int s23 = wave::inclusive_sum(((d >> 16) & 0xFFu) | ((d >> 24) << 16));
      xv[u][0] = x[static_cast<unsigned int>(t.base[u][0]) + (s01 & 0xFFFFu)];
      xv[u][1] = x[static_cast<unsigned int>(t.base[u][1]) + (s01 >> 16)];
      xv[u][2] = x[static_cast<unsigned int>(t.base[u][2]) + (s23 & 0xFFFFu)];
      xv[u][3] = x[static_cast<unsigned int>(t.base[u][3]) + (s23 >> 16)];
    }
  };
  auto update = [&](const batch_t& t, const type_t (&xv)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int code = ((e < 2 ? t.m[u].x : t.m[u].y) >> (16 * (e & 1))) & 0xFFFFu;
        const unsigned int row = t.live[u] ? code : static_cast<unsigned int>(H);
        atomicAdd(&acc[row], static_cast<double>(t.v[u][e] * xv[u][e]));
      }
  };
  constexpr int STRIDE = WAVES * U;
  int k = sb + w;
  if (k < se) {  // (wave-uniform)
    batch_t a, b;
    type_t xa[U][4], xb2[U][4];
    prefetch_bases(k);
    load(a, k);
    gather(a, xa);
    // (no exit between a batch's load and its gather: the compiler can neither sink the load past the update before it nor
    //  lose count of the loads in flight at a wait)
    for (;;) {
      if (k + STRIDE >= se) { update(a, xa); break; }
      load(b, k + STRIDE);
      __builtin_amdgcn_sched_barrier(0);  // (keep every stream load of the batch ahead of the first wait for a gather)
      update(a, xa);
      gather(b, xb2);
      k += STRIDE;
      if (k + STRIDE >= se) { update(b, xb2); break; }
      load(a, k + STRIDE);
      __builtin_amdgcn_sched_barrier(0);
      update(b, xb2);
      gather(a, xa);
      k += STRIDE;
    }
  }
  __syncthreads();
  // hubs: replicas -> the row's own word (one thread per hub, fixed order)
  const unsigned short* hb = hubs + static_cast<long long>(band) * (max_hubs + 1);
  const int nh = hb[0];
  if (static_cast<int>(threadIdx.x) < nh) {
    const double* rep = acc + H + 1 + static_cast<int>(threadIdx.x) * hub_replicas;
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < hub_replicas; ++r) sum += rep[r];
    acc[hb[1 + threadIdx.x]] += sum;
  }
  if (nh > 0) __syncthreads();  // (workgroup-uniform)
  if (slot < 0) {
    const long long row0 = static_cast<long long>(band) * H;
    for (int j = threadIdx.x; j < H && row0 + j < rows; j += TPB) out(static_cast<int>(row0 + j), static_cast<type_t>(acc[j]));
  } else {  // the chunk's partial vector; rowband_combine adds the band's up
    type_t* to = partial + static_cast<long long>(slot) * H;
    for (int j = threadIdx.x; j < H; j += TPB) to[j] = static_cast<type_t>(acc[j]);
  }
}

/// Kernel B: rows of the bands that were cut into several chunks.  grid = (H / 1024, num_multi), 256 threads x 4 rows.
/// DEPTH = partial vectors a thread has in flight (4 or 8: the launcher takes the smallest that covers the longest cut).
template <int DEPTH, typename type_t, typename store_t>
__global__ void __launch_bounds__(256)
rowband_combine(const int* __restrict__ multi, const type_t* __restrict__ partial, const int H, const int rows, const store_t out) {
  const int m = blockIdx.y;
  const int band = multi[3 * m], first = multi[3 * m + 1], count = multi[3 * m + 2];
  const int j = (static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x)) * 4;
  if (j >= H) return;
  double sum[4] = {0.0, 0.0, 0.0, 0.0};
  // DEPTH partial vectors in flight per thread (branch-free: a surplus load repeats the band's last vector and is not added) --
  // with one load in flight the kernel was a chain of `count` memory round trips (C2 / C4, 4 chunks per band: 6.8 us for 21 MB)
  for (int k0 = 0; k0 < count; k0 += DEPTH) {
    type_t p[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = k0 + d < count ? k0 + d : count - 1;
      detail::load4<type_t, false>(partial + static_cast<long long>(first + k) * H + j, p[d]);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const bool live = k0 + d < count;
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] += live ? static_cast<double>(p[d][e]) : 0.0;
    }
  }
  const long long row0 = static_cast<long long>(band) * H + j;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (row0 + e < rows) out(static_cast<int>(row0 + e), static_cast<type_t>(sum[e]));
}

/// Kernel B for bands cut into MANY chunks (a few bands hold most nonzeros: R-MAT graphs in generator or degree order cut one band
/// into 50-150 chunks, and one load in flight per thread makes the loop above the longest stage): PL threads share a group of 4
/// rows, thread p adds the chunks p, p + PL, ... (two loads in flight), the PL sums are added in LDS in the order of p.
/// grid = (H / (1024 / PL), num_multi), 256 threads.  A fixed order, another one than rowband_combine's: the same bits
/// wherever the fp64 sums are exact.
template <int PL, typename type_t, typename store_t>
__global__ void __launch_bounds__(256)
rowband_combine_wide(const int* __restrict__ multi, const type_t* __restrict__ partial, const int H, const int rows, const store_t out) {
  constexpr int QUADS = 256 / PL;  // groups of 4 rows per workgroup
  __shared__ double part[PL][QUADS][4];
  const int m = blockIdx.y;
  const int band = multi[3 * m], first = multi[3 * m + 1], count = multi[3 * m + 2];
  const int quad = static_cast<int>(threadIdx.x) % QUADS, p = static_cast<int>(threadIdx.x) / QUADS;
  const int j = (static_cast<int>(blockIdx.x) * QUADS + quad) * 4;
  double sum[4] = {0.0, 0.0, 0.0, 0.0};
  if (j < H) {
    for (int k = p; k < count; k += 2 * PL) {
      type_t a[4], b[4];
      const int k2 = k + PL < count ? k + PL : k;  // (branch-free second load; not added when it repeats the first)
      detail::load4<type_t, false>(partial + static_cast<long long>(first + k) * H + j, a);
      detail::load4<type_t, false>(partial + static_cast<long long>(first + k2) * H + j, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] += static_cast<double>(a[e]);
      if (k + PL < count) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += static_cast<double>(b[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[p][quad][e] = sum[e];
  __syncthreads();
  if (p != 0 || j >= H) return;
#pragma unroll
  for (int q = 1; q < PL; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] += part[q][quad][e];
  }
  const long long row0 = static_cast<long long>(band) * H + j;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (row0 + e < rows) out(static_cast<int>(row0 + e), static_cast<type_t>(sum[e]));
}

// ------------------------------------------------------------------------------------------------ plan-time kernels

/// A row is a HUB of its band when it holds at least hub_threshold(items of the band) nonzeros: with a share p of the band's
/// items, 64 p lanes of every update instruction meet in its accumulator and the LDS serialises them (C2: stream + updates
/// 25-30 us with uniform row lengths, 37-40 us with its power-law rows).
__host__ __device__ constexpr int hub_threshold(long long band_items, int share_div = 128) {
  return band_items / share_div > 64 ? static_cast<int>(band_items / share_div) : 64;
}

/// One workgroup per band: the first max_hubs rows (in row order) that reach the band's hub threshold.
/// hubidx[row] = the row's hub number inside its band, or -1; hubs[b * (max_hubs + 1)] = count, then the rows inside the band.
template <typename offset_t>
__global__ void __launch_bounds__(256)
find_hubs(const offset_t* __restrict__ offsets, const int rows, const int H, short* __restrict__ hubidx, unsigned short* __restrict__ hubs,
          const int share_div = 128) {
  using scan_t = hipcub::BlockScan<int, 256>;
  __shared__ typename scan_t::TempStorage temp;
  __shared__ int carry;
  const int b = blockIdx.x;
  const long long row0 = static_cast<long long>(b) * H;
  const int n = rows - row0 < H ? static_cast<int>(rows - row0) : H;
  const int threshold = hub_threshold(static_cast<long long>(offsets[row0 + n]) - static_cast<long long>(offsets[row0]), share_div);
  unsigned short* hb = hubs + static_cast<long long>(b) * (max_hubs + 1);
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + static_cast<int>(threadIdx.x);
    const bool hub = j < n && static_cast<int>(offsets[row0 + j + 1] - offsets[row0 + j]) >= threshold;
    int pos = 0, total = 0;
    scan_t(temp).ExclusiveSum(hub ? 1 : 0, pos, total);
    const int base = carry;
    if (j < n) {
      const int h = base + pos;
      const bool taken = hub && h < max_hubs;
      hubidx[row0 + j] = taken ? static_cast<short>(h) : static_cast<short>(-1);
      if (taken) hb[1 + h] = static_cast<unsigned short>(j);
    }
    __syncthreads();
    if (threadIdx.x == 0) carry = base + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) hb[0] = static_cast<unsigned short>(carry < max_hubs ? carry : max_hubs);
}

/// key[i] = band << cbits | column, item[i] = i, rin[i] = row inside the band, or 0x8000 | hub number for the rows of hubs.
/// One workgroup per TILE consecutive nonzeros: the rows they belong to are found ONCE per workgroup (two searches over the
/// offsets), their offsets staged in LDS, and every lane then handles nonzeros i, i + 256, ... -- coalesced reads of the columns,
/// coalesced writes of the three outputs, the row of a nonzero by a search in LDS.  (The first version gave a lane 8 CONSECUTIVE
/// nonzeros and walked the offsets: every access of a wavefront instruction 32 bytes apart, 0.32 ms of C2's 2.1 ms build.)
/// A tile that spans more rows than the stage holds (runs of empty rows) searches the offsets in memory instead.
template <int TILE, typename key_t, typename index_t, typename offset_t>
__global__ void __launch_bounds__(256)
make_keys(const offset_t* __restrict__ offsets, const index_t* __restrict__ indices, const short* __restrict__ hubidx, const int rows,
          const int nnz, const int hshift, const int cbits, const int cols, key_t* __restrict__ keys, int* __restrict__ item,
          unsigned short* __restrict__ rin, int* __restrict__ bad) {
  constexpr int STAGE = 2 * TILE;  // row starts of the tile's rows
  __shared__ offset_t s_off[STAGE + 1];
  __shared__ int s_row0, s_rows;
  const long long base_ll = static_cast<long long>(blockIdx.x) * TILE;
  if (base_ll >= nnz) return;
  const int base = static_cast<int>(base_ll);
  const int end = nnz - base < TILE ? nnz : base + TILE;
  auto row_of = [&](const int i, int lo, int count) {  // last row in [lo, lo + count) that starts at or before nonzero i
    while (count > 1) {
      const int half = count >> 1;
      if (offsets[lo + half] <= i) {
        lo += half;
        count -= half;
      } else {
        count = half;
      }
    }
    return lo;
  };
  if (threadIdx.x == 0) {
    const int r0 = row_of(base, 0, rows), r1 = row_of(end - 1, r0, rows - r0);
    s_row0 = r0;
    s_rows = r1 - r0 + 1;
  }
  __syncthreads();
  const int row0 = s_row0, nrows = s_rows;
  const bool staged = nrows <= STAGE;
  if (staged) {
    for (int j = threadIdx.x; j <= nrows; j += 256) s_off[j] = offsets[row0 + j];
    __syncthreads();
  }
  for (int i = base + static_cast<int>(threadIdx.x); i < end; i += 256) {
    int row;
    if (staged) {
      int lo = 0, count = nrows;
      while (count > 1) {
        const int half = count >> 1;
        if (s_off[lo + half] <= i) {
          lo += half;
          count -= half;
        } else {
          count = half;
        }
      }
      row = row0 + lo;
    } else {
      row = row_of(i, row0, nrows);
    }
    const int hub = hubidx[row];
    unsigned int col = static_cast<unsigned int>(indices[i]);
    if (col >= static_cast<unsigned int>(cols)) {  // (also a negative index) flagged, then clamped
      *bad = 1;
      col = 0;
    }
    const unsigned int b = static_cast<unsigned int>(row) >> hshift;
    keys[i] = (static_cast<key_t>(b) << cbits) | static_cast<key_t>(col);
    item[i] = i;
    rin[i] = hub >= 0 ? static_cast<unsigned short>(0x8000u | static_cast<unsigned int>(hub))
                      : static_cast<unsigned short>(static_cast<unsigned int>(row) - (b << hshift));
  }
}

/// band_start[b] = the first sorted position whose band is >= b (b <= B: band_start[B] = nnz).
template <typename key_t>
__global__ void __launch_bounds__(256)
band_starts(const key_t* __restrict__ sorted, const int nnz, const int B, const int cbits, int* __restrict__ band_start) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  const key_t want = static_cast<key_t>(static_cast<unsigned int>(b)) << cbits;
  int lo = 0, count = nnz;
  if (b == B) lo = nnz, count = 0;  // (B << cbits may not fit a 32-bit key)
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

/// Column gap of sorted item j to its predecessor in the band (0 for a band's first item) and the padding slots that bridge it.
template <typename key_t>
__device__ __forceinline__ void gap_of(const key_t* __restrict__ sorted, const int j, const int cbits, unsigned int& gap,
                                       unsigned int& pads) {
  gap = 0;
  if (j > 0) {
    const key_t key = sorted[j], prev = sorted[j - 1];
    if ((key >> cbits) == (prev >> cbits)) gap = static_cast<unsigned int>(key - prev);  // same band: the keys differ in the column only
  }
  pads = gap > static_cast<unsigned int>(max_delta) ? (gap - 1u) / static_cast<unsigned int>(max_delta) : 0u;
}

/// slots[j] = 1 + the padding slots in front of sorted item j (j < nnz), slots[nnz] = 0: the exclusive scan numbers the slots.
/// *any_pads is set when some gap needs padding slots at all (rare: a column gap of more than 255 inside a band).
template <typename key_t>
__global__ void __launch_bounds__(256)
count_slots(const key_t* __restrict__ sorted, const int nnz, const int cbits, int* __restrict__ slots, int* __restrict__ any_pads) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > nnz) return;
  unsigned int gap = 0, pads = 0;
  if (j < nnz) gap_of(sorted, j, cbits, gap, pads);
  slots[j] = j < nnz ? static_cast<int>(1u + pads) : 0;
  if (pads) *any_pads = 1;
}

/// steps[b] of a layout WITHOUT gap padding: the band's items rounded up to whole steps, steps[B] = 0.
__global__ void __launch_bounds__(256)
band_step_counts_dense(const int* __restrict__ band_start, const int B, int* __restrict__ steps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= B) steps[b] = b < B ? (band_start[b + 1] - band_start[b] + step_items - 1) / step_items : 0;
}

/// steps[b] = whole steps of band b (its slots rounded up to 256), steps[B] = 0.
__global__ void __launch_bounds__(256)
band_step_counts(const int* __restrict__ band_start, const int* __restrict__ slot_pos, const int B, int* __restrict__ steps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= B) steps[b] = b < B ? (slot_pos[band_start[b + 1]] - slot_pos[band_start[b]] + step_items - 1) / step_items : 0;
}

/// Every slot of the layout starts as padding: value 0, row code H, delta 0, no CSR position; every group base 0.
template <typename type_t>
__global__ void __launch_bounds__(256)
fill_padding(const long long n, const int H, type_t* __restrict__ val, unsigned int* __restrict__ meta, int* __restrict__ perm,
             int* __restrict__ stepbase) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < n) {
    val[j] = type_t(0);
    perm[j] = -1;
    if ((j & 63) == 0) stepbase[j >> 6] = 0;
    if ((j & 3) == 0) {  // the 3 meta words of the lane that owns slots j .. j + 3
      unsigned int* m = meta + (j >> 2) * 3;
      m[0] = m[1] = static_cast<unsigned int>(H) | (static_cast<unsigned int>(H) << 16);
      m[2] = 0u;
    }
  }
}

/// Row code / column delta of the slot at memory position `at` (= 4 * (lane of its step, counted over all steps) + element): the
/// lane's words hold 16-bit codes and 8-bit deltas; slots of one lane are written by different threads, hence the atomics.
__device__ __forceinline__ void set_meta(unsigned int* __restrict__ meta, const long long at, const unsigned int code, const unsigned int delta,
                                         const bool with_code, const unsigned int H) {
  unsigned int* m = meta + (at >> 2) * 3;
  const int e = static_cast<int>(at & 3);
  if (with_code) {  // (the word was filled with the padding code H in both halves: replace this half)
    const unsigned int shift = 16u * (e & 1);
    atomicXor(m + (e >> 1), ((code ^ H) & 0xFFFFu) << shift);
  }
  if (delta) atomicOr(m + 2, (delta & 0xFFu) << (8 * e));
}

/// Memory position of sorted slot s (counted from the start of the arrays): inside its step lane (q % 64), element (q / 64).
__device__ __forceinline__ long long slot_at(const long long s) {
  const long long q = s % step_items;
  return (s / step_items) * step_items + (q % wave::size) * 4 + q / wave::size;
}

/// Sorted item j -> its slot (and the padding slots in front of it that bridge a long column gap).
/// Row code: the row inside the band, or -- for a hub's item -- H + 1 + hub * hub_replicas + q % hub_replicas (q = slot in the step).
template <typename key_t, typename type_t>
__global__ void __launch_bounds__(256)
place(const key_t* __restrict__ sorted, const int* __restrict__ item, const unsigned short* __restrict__ rin,
      const int* __restrict__ band_start, const int* __restrict__ slot_pos, const int* __restrict__ band_step,
      const type_t* __restrict__ values, const int nnz, const int cbits, const int H, type_t* __restrict__ val,
      unsigned int* __restrict__ meta, int* __restrict__ perm, int* __restrict__ stepbase) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nnz) return;
  const key_t key = sorted[j];
  const unsigned int col = static_cast<unsigned int>(key & ((static_cast<key_t>(1) << cbits) - 1));
  const int b = static_cast<int>(key >> cbits);
  unsigned int gap, pads;
  gap_of(sorted, j, cbits, gap, pads);
  const long long s = static_cast<long long>(band_step[b]) * step_items + (slot_pos[j] - slot_pos[band_start[b]]) + pads;
  for (unsigned int t = 0; t < pads; ++t) {  // (rare: a gap of more than 255 columns)
    const long long ps = s - pads + t;
    set_meta(meta, slot_at(ps), 0u, (ps & 63) == 0 ? 0u : static_cast<unsigned int>(max_delta), false, static_cast<unsigned int>(H));
    if ((ps & 63) == 0) stepbase[ps >> 6] = static_cast<int>(col - gap + static_cast<unsigned int>(max_delta) * (t + 1u));
  }
  const long long at = slot_at(s);
  const int q = static_cast<int>(s % step_items);
  const int i = item[j];
  const unsigned int code = rin[i];
  const unsigned int row = (code & 0x8000u) ? static_cast<unsigned int>(H) + 1u + (code & 0x7FFFu) * hub_replicas + static_cast<unsigned int>(q % hub_replicas)
                                            : code;
  val[at] = values[i];
  set_meta(meta, at, row, (s & 63) == 0 ? 0u : gap - static_cast<unsigned int>(max_delta) * pads, true, static_cast<unsigned int>(H));
  perm[at] = i;
  if ((s & 63) == 0) stepbase[s >> 6] = static_cast<int>(col);
}

/// The layout WITHOUT gap padding (no column gap of more than 255 inside a band: the common case), written lane record by lane
/// record: slot s of band b is sorted item band_start[b] + (s - 256 band_step[b]), so the thread of (step, lane) fetches its four
/// items itself and writes its 16 bytes of values, 12 bytes of codes + deltas and 16 bytes of CSR positions whole -- no atomics on
/// the packed words, no padding pre-fill, and no scan over the nonzeros in front of it (`place` needs all three: C2 368 + 28 +
/// 110 us of a 1.6 ms build).  Same arrays as fill_padding + place produce.
template <typename key_t, typename type_t>
__global__ void __launch_bounds__(256)
place_dense(const key_t* __restrict__ sorted, const int* __restrict__ item, const unsigned short* __restrict__ rin,
            const int* __restrict__ band_start, const int* __restrict__ band_step, const type_t* __restrict__ values, const int steps, const int B,
            const int cbits, const int H, type_t* __restrict__ val, unsigned int* __restrict__ meta, int* __restrict__ perm, int* __restrict__ stepbase) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int step = static_cast<int>(t >> 6), lane = static_cast<int>(t & 63);
  if (step >= steps) return;
  int b = 0, count = B;  // the band of the step: the last b with band_step[b] <= step (bands without steps share a start: the last one owns it)
  while (count > 1) {
    const int half = count >> 1;
    if (band_step[b + half] <= step) {
      b += half;
      count -= half;
    } else {
      count = half;
    }
  }
  const int j0 = band_start[b] + (step - band_step[b]) * step_items, jend = band_start[b + 1];
  const key_t cmask = (static_cast<key_t>(1) << cbits) - 1;
  type_t v[4];
  int at[4];
  unsigned int code[4], delta[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = 64 * e + lane;
    const int j = j0 + q;
    v[e] = type_t(0);
    at[e] = -1;
    code[e] = static_cast<unsigned int>(H);
    delta[e] = 0u;
    unsigned int col = 0u;
    if (j < jend) {
      col = static_cast<unsigned int>(sorted[j] & cmask);
      const int i = item[j];
      const unsigned int c = rin[i];
      v[e] = values[i];
      at[e] = i;
      code[e] = (c & 0x8000u) ? static_cast<unsigned int>(H) + 1u + (c & 0x7FFFu) * hub_replicas + static_cast<unsigned int>(q % hub_replicas) : c;
      if (lane != 0) delta[e] = col - static_cast<unsigned int>(sorted[j - 1] & cmask);  // (same band: j - 1 >= j0; <= 255 here)
    }
    if (lane == 0) stepbase[static_cast<long long>(step) * 4 + e] = static_cast<int>(col);
  }
  const long long rec = static_cast<long long>(step) * wave::size + lane;
  type_t* vp = val + rec * 4;
  int* pp = perm + rec * 4;
  unsigned int* mp = meta + rec * 3;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    vp[e] = v[e];
    pp[e] = at[e];
  }
  mp[0] = code[0] | (code[1] << 16);
  mp[1] = code[2] | (code[3] << 16);
  mp[2] = delta[0] | (delta[1] << 8) | (delta[2] << 16) | (delta[3] << 24);
}

template <typename type_t>
__global__ void __launch_bounds__(256)
refresh_values(const int* __restrict__ perm, const type_t* __restrict__ values, const long long n, type_t* __restrict__ val) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < n) {
    const int i = perm[j];
    val[j] = i >= 0 ? values[i] : type_t(0);
  }
}

}  // namespace rowband

constexpr int rowband_e_badarg = -1, rowband_e_range = -2;

/// Rows per band: 16384 (the tallest band the LDS holds: the taller the band, the denser its column-sorted nonzeros and the fewer
/// lines a wavefront's 64 gathers touch), halved while the matrix would be left with fewer than 64 bands; 8192 at most for
/// 8-byte values (C2 in fp64, tests/perf/bench_rowband.py --f64: 56.4 us with 8192 rows, 82.4 with 16384; CSR fp64 189).
inline int rowband_rows(int rows, int /*cols*/, int /*nnz*/, int vbytes = 4) {
  int h = vbytes > 4 ? rowband::max_band_rows / 2 : rowband::max_band_rows;
  while (h > 256 && rows / h < 64) h /= 2;
  return h;
}

/// Kernel A's work list from the bands' step ranges (B + 1 entries).  The bands are cut into `target_chunks` chunks (every band at
/// least one; the next cut always goes to the band whose chunks are the longest), a band's chunks of equal size.  {band, first step, end step, partial slot or -1} per chunk, {band, first slot,
/// chunks} per cut band.
inline void rowband_chunk_list(const std::vector<int>& band_step, int B, int target_chunks, std::vector<int>& chunks, std::vector<int>& multi,
                               int& num_partials) {
  chunks.clear();
  multi.clear();
  num_partials = 0;
  // every band one chunk (a band without nonzeros too: its chunk stores zeros), then cut by cut: the next cut goes to the band whose
  // chunks are the longest (ties: the lower band) until the list has `target_chunks` entries -- exactly, so that one round of
  // workgroups covers it (proportional shares rounded per band overshoot when a few bands hold most of the steps: 288 chunks
  // on 256 CUs for a degree-ordered R-MAT graph)
  std::vector<int> pieces(static_cast<std::size_t>(B), 1);
  long long sum = B;
  {
    using entry = std::pair<double, int>;
    std::priority_queue<entry> heap;
    for (int b = 0; b < B; ++b) {
      const int n = band_step[b + 1] - band_step[b];
      if (n > pieces[b]) heap.push({static_cast<double>(n) / pieces[b], -b});
    }
    while (sum < target_chunks && !heap.empty()) {
      const int b = -heap.top().second;
      heap.pop();
      const int n = band_step[b + 1] - band_step[b];
      ++pieces[b];
      ++sum;
      if (n > pieces[b]) heap.push({static_cast<double>(n) / pieces[b], -b});
    }
  }
  for (int b = 0; b < B; ++b) {
    const int s0 = band_step[b], n = band_step[b + 1] - s0;
    if (n <= 0) {  // a band without nonzeros still owns rows of y: an empty chunk stores its zeros
      chunks.insert(chunks.end(), {b, s0, s0, -1});
      continue;
    }
    const int size = (n + pieces[b] - 1) / pieces[b];
    const int count = (n + size - 1) / size;
    if (count > 1) multi.insert(multi.end(), {b, num_partials, count});
    for (int k = 0; k < count; ++k) {
      const int begin = s0 + k * size, end = begin + size < s0 + n ? begin + size : s0 + n;
      chunks.insert(chunks.end(), {b, begin, end, count > 1 ? num_partials + k : -1});
    }
    if (count > 1) num_partials += count;
  }
  // Order of the work list: by piece number first, bands ascending inside -- the workgroups an XCD runs together (it walks one
  // contiguous run of the list, xcd_contiguous) then sweep the SAME share of the columns of neighbouring bands, so the XCD's
  // L2 has to hold that share of x only (C2, 4 pieces per band: 1 MB instead of the 4 MB that fill it).  Uncut bands are piece 0.
  const std::size_t n_chunks = chunks.size() / 4;
  std::vector<int> piece(n_chunks, 0), order(n_chunks);
  for (std::size_t c = 0; c < n_chunks; ++c) {
    order[c] = static_cast<int>(c);
    if (c > 0 && chunks[4 * c] == chunks[4 * (c - 1)]) piece[c] = piece[c - 1] + 1;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return piece[a] < piece[b]; });
  std::vector<int> sorted(chunks.size());
  for (std::size_t c = 0; c < n_chunks; ++c)
    for (int f = 0; f < 4; ++f) sorted[4 * c + f] = chunks[4 * static_cast<std::size_t>(order[c]) + f];
  chunks.swap(sorted);
}

/// The device arrays of one row-band matrix, OWNED.  Type-erased over the value type (`vbytes`).
struct rowband_storage {
  int rows = 0, cols = 0, nnz = 0, vbytes = 0;
  int H = 0, B = 0, steps = 0, num_chunks = 0, num_partials = 0, num_multi = 0, target_chunks = 0, cus = 0, waves = 8, max_pieces = 1;
  long long gap_pads = 0;      ///< padding slots that bridge column gaps of more than 255
  void *val = nullptr, *partial = nullptr;
  unsigned short* hubs = nullptr;
  unsigned int* meta = nullptr;
  int *stepbase = nullptr, *perm = nullptr, *chunks = nullptr, *multi = nullptr, *band_step = nullptr;

  rowband_storage() = default;
  rowband_storage(const rowband_storage&) = delete;
  rowband_storage& operator=(const rowband_storage&) = delete;
  ~rowband_storage() { release(); }
  void release() {
    (void)hipFree(val); (void)hipFree(partial); (void)hipFree(meta); (void)hipFree(hubs); (void)hipFree(stepbase);
    (void)hipFree(perm); (void)hipFree(chunks); (void)hipFree(multi); (void)hipFree(band_step);
    val = partial = nullptr; hubs = nullptr; meta = nullptr; stepbase = perm = chunks = multi = band_step = nullptr;
  }
  template <typename type_t>
  rowband_view<type_t> view() const {
    return rowband_view<type_t>{rows, cols, nnz, H, B, steps, num_chunks, num_partials, num_multi, static_cast<const type_t*>(val), meta,
                                stepbase, chunks, multi, hubs, static_cast<type_t*>(partial), waves, max_pieces};
  }
};

/// Default number of chunks: one per band where there are more than half as many bands as compute units (the dispatcher
/// balances them; cutting ONE band of 255 on 256 CUs would buy a partial-vector round trip and a second launch for nothing),
/// else one round of equal chunks (C2: 64 bands of 16384 rows -> 256 chunks).  Measured (profiles/r05_rowband_experiments.txt,
/// the 8-byte layout): band C3 stand-in, 453 bands, uncut 268 us, 512 / 1024 chunks 275 / 305; C2 256 chunks 37 us, 512 chunks
/// 44-54 (two workgroups per CU, twice the partial vectors).
inline int rowband_target_chunks(int B, int cus) {
  const int c = cus > 0 ? cus : 256;
  return 2 * B > c ? B : c;
}

/// (Re)builds the work lists of a built layout for about `target_chunks` chunks (0 = automatic); `band_step_host` = the B + 1 step
/// starts.  Used by the builder and by tuning code that sweeps the cut without re-sorting.
inline int rowband_set_chunks(rowband_storage& out, const std::vector<int>& band_step_host, int target_chunks) {
  out.target_chunks = target_chunks > 0 ? target_chunks : rowband_target_chunks(out.B, out.cus);
  std::vector<int> chunks, multi;
  rowband_chunk_list(band_step_host, out.B, out.target_chunks, chunks, multi, out.num_partials);
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
    e = hipMalloc(&out.partial, static_cast<std::size_t>(out.vbytes) * (out.num_partials > 0 ? static_cast<std::size_t>(out.num_partials) * out.H : 4));
  if (e == hipSuccess && !chunks.empty()) e = hipMemcpy(out.chunks, chunks.data(), sizeof(int) * chunks.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !multi.empty()) e = hipMemcpy(out.multi, multi.data(), sizeof(int) * multi.size(), hipMemcpyHostToDevice);
  return static_cast<int>(e);
}

/// Builds the row-band copy of a CSR on the device.  band_rows: 0 = automatic, else a power of two in [64, 16384];
/// target_chunks: 0 = automatic.  Returns 0, a hipError_t, rowband_e_badarg (also: a column index outside [0, cols)) or
/// rowband_e_range (the padded layout may not fit 32-bit positions).
template <typename key_t, typename index_t, typename offset_t, typename type_t>
int rowband_create_keyed(hipStream_t stream, int rows, int cols, int nnz, const offset_t* offsets, const index_t* indices, const type_t* values,
                   int band_rows, int target_chunks, rowband_storage& out) {
  static_assert(sizeof(index_t) == 4 && sizeof(offset_t) == 4, "rowband_create: 32-bit indices and offsets");
  if (!offsets || rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && (!indices || !values)) || target_chunks < 0) return rowband_e_badarg;
  out.release();
  out.rows = rows; out.cols = cols; out.nnz = nnz; out.vbytes = static_cast<int>(sizeof(type_t));
  out.H = band_rows != 0 ? band_rows : rowband_rows(rows, cols, nnz, static_cast<int>(sizeof(type_t)));
  if (out.H < 64 || out.H > rowband::max_band_rows || (out.H & (out.H - 1))) return rowband_e_badarg;
  int hshift = 0;
  while ((1 << hshift) < out.H) ++hshift;
  out.B = rows > 0 ? static_cast<int>((static_cast<long long>(rows) + out.H - 1) / out.H) : 0;
  out.steps = out.num_chunks = out.num_partials = out.num_multi = 0;
  out.gap_pads = 0;
  if (rows == 0) return 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&out.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) out.cus = 256;
  }
  const int B = out.B;
  // upper bound of the slots: the nonzeros, the padding slots that bridge column gaps (a band's gaps sum to less than `cols`),
  // the padding of every band to whole steps
  if (B > (1 << 26) ||
      static_cast<long long>(nnz) + static_cast<long long>(B) * (static_cast<long long>(cols) / rowband::max_delta + rowband::step_items) >= (1ll << 31) - 4096)
    return rowband_e_range;
  int cbits = 1;
  while (cbits < 31 && (static_cast<long long>(cols) >> cbits) != 0) ++cbits;
  int bbits = 1;
  while (bbits < 31 && (static_cast<long long>(B) >> bbits) != 0) ++bbits;

  auto up = [](std::size_t v) { return (v + 255) & ~std::size_t(255); };
  std::size_t sort_bytes = 0, scan_bytes = 0;
  {
    key_t* k = nullptr;
    int* ci = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k, k, ci, ci, nnz, 0, cbits + bbits);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, ci, ci, (nnz > B ? nnz : B) + 1);
  }
  const std::size_t cub_bytes_total = up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
  const std::size_t key_bytes = up((static_cast<std::size_t>(nnz) + 1) * sizeof(key_t)), item_bytes = up((static_cast<std::size_t>(nnz) + 1) * 4);
  const std::size_t rin_bytes = up((static_cast<std::size_t>(nnz) + 1) * 2), band_bytes = up((static_cast<std::size_t>(B) + 1) * 4);
  const std::size_t hub_bytes = up((static_cast<std::size_t>(rows) + 1) * 2);
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
  auto* keys_in = reinterpret_cast<key_t*>(carve(key_bytes));
  auto* keys_out = reinterpret_cast<key_t*>(carve(key_bytes));
  int* item_in = reinterpret_cast<int*>(carve(item_bytes));
  int* item_out = reinterpret_cast<int*>(carve(item_bytes));
  auto* rin = reinterpret_cast<unsigned short*>(carve(rin_bytes));
  auto* hubidx = reinterpret_cast<short*>(carve(hub_bytes));
  int* band_start = reinterpret_cast<int*>(carve(band_bytes));
  int* steps_of = reinterpret_cast<int*>(carve(band_bytes));
  int* bad = reinterpret_cast<int*>(carve(256));
  void* cub_temp = carve(cub_bytes_total);
  int* slots = reinterpret_cast<int*>(keys_in);    // (over keys_in once the sort is done) [nnz + 1]: slots of item j incl. its gap pads
  int* slot_pos = item_in;                         // (over item_in once the sort is done) [nnz + 1]: their exclusive scan
  std::size_t cub_bytes = cub_bytes_total;

  const std::size_t hubs_n = static_cast<std::size_t>(B) * (rowband::max_hubs + 1);
  e = hipMalloc(reinterpret_cast<void**>(&out.hubs), sizeof(unsigned short) * hubs_n);
  if (e == hipSuccess) e = hipMemsetAsync(out.hubs, 0, sizeof(unsigned short) * hubs_n, stream);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(int), stream);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  hipLaunchKernelGGL((rowband::find_hubs<offset_t>), dim3(B), dim3(256), 0, stream, offsets, rows, out.H, hubidx, out.hubs);
  constexpr int KEY_TILE = 2048;
  if (nnz > 0) {
    hipLaunchKernelGGL((rowband::make_keys<KEY_TILE, key_t, index_t, offset_t>), dim3(math::ceil_div(nnz, KEY_TILE)), dim3(256), 0, stream, offsets,
                       indices, hubidx, rows, nnz, hshift, cbits, cols, keys_in, item_in, rin, bad);
    e = hipcub::DeviceRadixSort::SortPairs(cub_temp, cub_bytes, keys_in, keys_out, item_in, item_out, nnz, 0, cbits + bbits, stream);
    if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  }
  hipLaunchKernelGGL((rowband::band_starts<key_t>), dim3(math::ceil_div(B + 1, 256)), dim3(256), 0, stream, keys_out, nnz, B, cbits, band_start);
  // First the layout WITHOUT gap padding (no column gap of more than 255 inside a band: the common case) -- its band steps follow
  // from the band starts alone; only when count_slots finds a gap that needs padding slots is the scan over the nonzeros run.
  int* any_pads = bad + 1;
  e = hipMemsetAsync(any_pads, 0, sizeof(int), stream);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&out.band_step), sizeof(int) * (static_cast<std::size_t>(B) + 1));
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  hipLaunchKernelGGL((rowband::count_slots<key_t>), dim3(math::ceil_div(nnz + 1, 256)), dim3(256), 0, stream, keys_out, nnz, cbits, slots, any_pads);
  hipLaunchKernelGGL(rowband::band_step_counts_dense, dim3(math::ceil_div(B + 1, 256)), dim3(256), 0, stream, band_start, B, steps_of);
  cub_bytes = cub_bytes_total;
  e = hipcub::DeviceScan::ExclusiveSum(cub_temp, cub_bytes, steps_of, out.band_step, B + 1, stream);
  std::vector<int> bs(static_cast<std::size_t>(B) + 1, 0);
  int h_bad = 0, h_pads = 0, h_slots = nnz;
  if (e == hipSuccess) e = hipMemcpyAsync(bs.data(), out.band_step, sizeof(int) * bs.size(), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h_pads, any_pads, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  if (h_bad != 0) { out.release(); return rowband_e_badarg; }
  if (h_pads != 0) {  // gaps to bridge: number the slots by a scan, redo the band steps from it
    cub_bytes = cub_bytes_total;
    e = hipcub::DeviceScan::ExclusiveSum(cub_temp, cub_bytes, slots, slot_pos, nnz + 1, stream);
    if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
    hipLaunchKernelGGL(rowband::band_step_counts, dim3(math::ceil_div(B + 1, 256)), dim3(256), 0, stream, band_start, slot_pos, B, steps_of);
    cub_bytes = cub_bytes_total;
    e = hipcub::DeviceScan::ExclusiveSum(cub_temp, cub_bytes, steps_of, out.band_step, B + 1, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(bs.data(), out.band_step, sizeof(int) * bs.size(), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_slots, slot_pos + nnz, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  }
  out.steps = bs[B];
  out.gap_pads = static_cast<long long>(h_slots) - nnz;
  const std::size_t n = static_cast<std::size_t>(out.steps > 0 ? out.steps : 1) * rowband::step_items;
  auto alloc = [&](auto** ptr, std::size_t bytes) { if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(ptr), bytes); };
  alloc(&out.val, sizeof(type_t) * n);
  alloc(&out.meta, sizeof(unsigned int) * (n / 4) * 3);
  alloc(&out.perm, sizeof(int) * n);
  alloc(&out.stepbase, sizeof(int) * (n / 64));
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  if (h_pads == 0 && out.steps > 0) {
    hipLaunchKernelGGL((rowband::place_dense<key_t, type_t>), dim3(static_cast<unsigned int>((static_cast<long long>(out.steps) * wave::size + 255) / 256)),
                       dim3(256), 0, stream, keys_out, item_out, rin, band_start, out.band_step, values, out.steps, B, cbits, out.H,
                       static_cast<type_t*>(out.val), out.meta, out.perm, out.stepbase);
  } else {
    hipLaunchKernelGGL((rowband::fill_padding<type_t>), dim3(static_cast<unsigned int>((n + 255) / 256)), dim3(256), 0, stream,
                       static_cast<long long>(n), out.H, static_cast<type_t*>(out.val), out.meta, out.perm, out.stepbase);
    if (nnz > 0)
      hipLaunchKernelGGL((rowband::place<key_t, type_t>), dim3(math::ceil_div(nnz, 256)), dim3(256), 0, stream, keys_out, item_out, rin, band_start,
                         slot_pos, out.band_step, values, nnz, cbits, out.H, static_cast<type_t*>(out.val), out.meta, out.perm,
                         out.stepbase);
  }
  e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = static_cast<hipError_t>(rowband_set_chunks(out, bs, target_chunks));
  if (e != hipSuccess) { out.release(); return static_cast<int>(e); }
  return 0;
}

/// The builder with the narrowest sort key that holds band and column: 32 bits wherever they fit (C2: 6 + 20 bits -- half the
/// radix passes of the 64-bit key the first version always sorted), 64 otherwise.  Same layout arrays either way (the sort is
/// stable, the key order the same).
template <typename index_t, typename offset_t, typename type_t>
int rowband_create(hipStream_t stream, int rows, int cols, int nnz, const offset_t* offsets, const index_t* indices, const type_t* values,
                   int band_rows, int target_chunks, rowband_storage& out) {
  int H = band_rows != 0 ? band_rows : rowband_rows(rows, cols, nnz, static_cast<int>(sizeof(type_t)));
  if (H < 64) H = 64;  // (the keyed builder refuses it; only the key width is decided here)
  const long long B = rows > 0 ? (static_cast<long long>(rows) + H - 1) / H : 0;
  int cbits = 1;
  while (cbits < 31 && (static_cast<long long>(cols) >> cbits) != 0) ++cbits;
  int bbits = 1;
  while (bbits < 31 && (B >> bbits) != 0) ++bbits;
  if (cbits + bbits <= 32)
    return rowband_create_keyed<unsigned int, index_t, offset_t, type_t>(stream, rows, cols, nnz, offsets, indices, values, band_rows, target_chunks, out);
  return rowband_create_keyed<unsigned long long, index_t, offset_t, type_t>(stream, rows, cols, nnz, offsets, indices, values, band_rows, target_chunks,
                                                                            out);
}

namespace detail {
/// Opts `kernel` into the 160 KB LDS of a CU once per device: `done` is the calling instantiation's own static mask of the devices
/// already served (function attributes are per device; a process may drive several).  Not synchronised: two host threads may both
/// set the attribute, which is harmless.
inline void allow_large_lds(const void* kernel, unsigned long long& done) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
  if (known && ((done >> dev) & 1ull)) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (known) done |= 1ull << dev;
}
}  // namespace detail

/// Kernel B's launch: the rows of `num_multi` cut bands of H rows each (`multi` = {band, first partial slot, chunks} per band).
/// One thread per group of 4 rows while no band is cut into more than a handful of chunks, 4 or 16 threads beyond.  The band
/// index is the grid's y dimension (at most 65 535 per launch: longer lists go out in slabs).
template <typename type_t, typename store_t>
void launch_rowband_combine(hipStream_t stream, const int* multi, int num_multi, int max_pieces, const type_t* partial, int H, int rows,
                            const store_t out) {
  constexpr int max_y = 65535;
  for (int m0 = 0; m0 < num_multi; m0 += max_y) {
    const int my = num_multi - m0 < max_y ? num_multi - m0 : max_y;
    const int* mm = multi + 3 * static_cast<std::size_t>(m0);
    if (max_pieces <= 4)
      hipLaunchKernelGGL((rowband::rowband_combine<4, type_t, store_t>), dim3(math::ceil_div(H, 1024), my), dim3(256), 0, stream, mm, partial, H, rows, out);
    else if (max_pieces <= 8)
      hipLaunchKernelGGL((rowband::rowband_combine<8, type_t, store_t>), dim3(math::ceil_div(H, 1024), my), dim3(256), 0, stream, mm, partial, H, rows, out);
    else if (max_pieces <= 48)
      hipLaunchKernelGGL((rowband::rowband_combine_wide<4, type_t, store_t>), dim3(math::ceil_div(H, 256), my), dim3(256), 0, stream, mm, partial, H,
                         rows, out);
    else
      hipLaunchKernelGGL((rowband::rowband_combine_wide<16, type_t, store_t>), dim3(math::ceil_div(H, 64), my), dim3(256), 0, stream, mm, partial, H,
                         rows, out);
  }
}

/// y = A x over a row-band matrix: kernel A, then kernel B if some band was cut.  stages: bit 0 = accumulate, bit 1 = combine.
/// Kernel A runs 8 or 16 wavefronts per workgroup (`m.waves`), one step per wavefront and batch.  Measured on MI355X
/// (tests/perf/bench_rowband.py, profiles/r05_rowband_experiments.txt): C2 32.6 / 35.7 us with 8 / 16 wavefronts (more wavefronts
/// lengthen the queues of the CU's memory path without adding requests in flight), 2 / 4 steps per batch +3 / +9 us; a matrix
/// whose gathers hit the L1 (band C3 stand-in) 372 / 291 us.
template <typename type_t, typename store_t>
int launch_rowband_to(hipStream_t stream, const rowband_view<type_t>& m, const type_t* x, const store_t out, int stages = 3) {
  if (m.rows == 0) return 0;
  // Non-temporal streams unless a product's working set (7 B per slot with 4-byte values, x, y, partials) fits the Infinity Cache
  // (C2, 135 MB: plain 33 us, non-temporal 49 with the 8-byte layout)
  const double items = static_cast<double>(m.steps) * rowband::step_items;
  const bool nt = items * (sizeof(type_t) + 3.0) + (static_cast<double>(m.rows) + m.cols + 2.0 * m.num_partials * m.H) * sizeof(type_t) > 240e6;
  if ((stages & 1) && m.num_chunks > 0) {
    const std::size_t lds = static_cast<std::size_t>(rowband::lds_words(m.H)) * sizeof(double);
    auto go = [&](auto w_tag, auto nt_tag) {
      constexpr int W = decltype(w_tag)::value;
      constexpr bool NT = decltype(nt_tag)::value;
      auto* kernel = rowband::rowband_accumulate<W, 1, NT, type_t, store_t>;
      if (lds > 65536) {  // opt into the large LDS once per instantiation (this lambda body is instantiated per <W, NT>) and device, not per launch
        static unsigned long long opted_devices = 0;
        detail::allow_large_lds(reinterpret_cast<const void*>(kernel), opted_devices);
      }
      hipLaunchKernelGGL(kernel, dim3(m.num_chunks), dim3(W * wave::size), lds, stream, m.chunks, m.val, m.meta, m.stepbase, m.hubs, x, m.H, m.rows,
                         m.partial, out);
    };
    using w8 = std::integral_constant<int, 8>;
    using w16 = std::integral_constant<int, 16>;
    if (m.waves == 16) {
      if (nt) go(w16{}, std::true_type{});
      else go(w16{}, std::false_type{});
    } else {
      if (nt) go(w8{}, std::true_type{});
      else go(w8{}, std::false_type{});
    }
  }
  if ((stages & 2) && m.num_multi > 0) launch_rowband_combine<type_t>(stream, m.multi, m.num_multi, m.max_pieces, m.partial, m.H, m.rows, out);
  return static_cast<int>(hipGetLastError());
}

template <typename type_t>
int launch_rowband(hipStream_t stream, const rowband_view<type_t>& m, const type_t* x, type_t* y, int stages = 3) {
  return launch_rowband_to(stream, m, x, plain_store<type_t>{y}, stages);
}

/// Measures the product with 8 and with 16 wavefronts per workgroup of kernel A on this device (x = zeros: the time does not
/// depend on the values) and keeps the faster shape in `m.waves`.  ms2 (may be null) receives the two times per product.
/// Synchronous.
template <typename type_t>
int rowband_tune(hipStream_t stream, rowband_storage& m, int repeats, float* ms2) {
  if (ms2) ms2[0] = ms2[1] = -1.f;
  if (m.rows == 0 || m.steps == 0) return 0;
  if (repeats < 1) repeats = 10;
  type_t *x = nullptr, *y = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&x), sizeof(type_t) * static_cast<std::size_t>(m.cols > 0 ? m.cols : 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&y), sizeof(type_t) * static_cast<std::size_t>(m.rows));
  if (e == hipSuccess) e = hipMemsetAsync(x, 0, sizeof(type_t) * static_cast<std::size_t>(m.cols > 0 ? m.cols : 1), stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  int err = static_cast<int>(e);
  const int shapes[2] = {8, 16};
  float best = 0.f;
  int best_waves = m.waves;
  for (int i = 0; !err && i < 2; ++i) {
    m.waves = shapes[i];
    const rowband_view<type_t> v = m.view<type_t>();
    for (int it = 0; !err && it < 2; ++it) err = launch_rowband<type_t>(stream, v, x, y);
    if (!err) err = static_cast<int>(hipEventRecord(e0, stream));
    for (int it = 0; !err && it < repeats; ++it) err = launch_rowband<type_t>(stream, v, x, y);
    if (!err) err = static_cast<int>(hipEventRecord(e1, stream));
    if (!err) err = static_cast<int>(hipEventSynchronize(e1));
    float ms = 0.f;
    if (!err) err = static_cast<int>(hipEventElapsedTime(&ms, e0, e1));
    if (err) break;
    ms /= static_cast<float>(repeats);
    if (ms2) ms2[i] = ms;
    if (i == 0 || ms < 0.98f * best) { best = ms; best_waves = shapes[i]; }  // (16 wavefronts must be measurably faster)
  }
  m.waves = best_waves;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(x);
  (void)hipFree(y);
  return err;
}

template <typename type_t>
int launch_rowband_fanout(hipStream_t stream, const rowband_view<type_t>& m, const type_t* x, type_t* y, const peer_fanout<type_t>& peers) {
  if (peers.count < 0 || peers.count > max_peers) return static_cast<int>(hipErrorInvalidValue);
  return launch_rowband_to(stream, m, x, fanout_store<type_t>{y, peers}, 3);
}

}  // namespace kernels
}  // namespace loops
