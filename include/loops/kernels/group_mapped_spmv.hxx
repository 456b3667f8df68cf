/**
 * @file group_mapped_spmv.hxx
 * @brief group_mapped with heavy groups shared out: a workgroup still OWNS 256 consecutive rows (schedule::setup<group_mapped>,
 *        reference schedule/group_mapped.hxx:104-192), but a group of more than `group_heavy_tiles` merge tiles is no longer
 *        swept by its owner alone.
 *
 * Why.  `group_mapped_spmv_fused` (merge_path_spmv.hxx) walks the tiles of a group one after the other inside the owning
 * workgroup.  On a graph whose first rows are hubs (R-MAT in generator order: the group of rows 0-255 is 1 535 tiles long, 576
 * groups of 49 tiles or more hold 40 % of the nonzeros) that one workgroup runs for 7.7 ms while 255 compute units idle: 7.9 ms against 1.3 ms for
 * work_oriented on the 2^23-row stand-in -- the cliff the reference's kernel has too (algorithms/spmv/group_mapped.cuh:27-61:
 * one block loops over the group's atoms).
 *
 * How.  Three launches on the caller's stream, no host decision in between, no zero-fill of y, no floating-point atomics:
 *   1. `group_mapped_spmv_publish`: one workgroup per group.  A group of at most `group_heavy_tiles` tiles is processed as
 *      before (tiles in sequence, the open row's partial sum carried in a register).  The owner of a HEAVY group only
 *      PUBLISHES it: one 64-bit atomicAdd on a packed {groups, claims} counter hands it a slot and a contiguous range of CLAIM
 *      numbers (a claim = `group_claim_tiles` consecutive tiles), recorded as {group, first claim, claims};
 *   2. `group_mapped_spmv_claims`: a fixed grid strides over the claim numbers (the host does not know how many were handed
 *      out).  A workgroup reads its claim's record (the owner left the slot number per claim), loads the group's 256 row ends into LDS, cuts its tiles out
 *      of the group's own merge path and runs them through the merge-tile engine, carrying the open row inside the claim and
 *      leaving a carry-out {row, partial sum} at the end -- exactly what merge_path_flat's tiles do;
 *   3. `group_mapped_fixup`: per published group, the carry-outs of its claims are added in claim order to the rows they
 *      belong to (merge_path_flat's fix-up restricted to one group); the last workgroup through clears the counter for the
 *      next call and reports the number of published groups (see `launch_group_mapped_shared`).
 * Launches 2 and 3 cost ~3 us each when nothing was published; callers that run the same matrix repeatedly can skip them with
 * the reported count (the C ABI's one-shot entry does: abi_csr.inc).
 *
 * Gather order.  Over an x of 6 MB or more (kernels::columns_worth_sampling, the plan-less merge_path_flat entry's rule) the
 * launcher puts column_scatter_sample in front and runs the `policy::phased_auto` builds of the same kernels: where the sampled
 * columns are scattered every workgroup gathers x part by part off the shared clock (merge_tile_engine), otherwise
 * exactly as before -- decided on the device, same loads in another order, same bits.
 *
 * Deterministic: a claim's result does not depend on which workgroup ran it, carry-outs are added in claim order.
 * Bit-identical to group_mapped_spmv_fused wherever fp32 sums are exact; otherwise the two differ as merge_path_flat and
 * work_oriented do (a hub row is summed per claim, then across claims).
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <type_traits>

#include <hip/hip_runtime.h>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/kernels/launch.hxx>
#include <loops/util/math.hxx>

namespace loops {
namespace kernels {

/// A group is shared out when it has more merge tiles than this.  24 tiles of 2 048 items: the power-law classes with rows
/// capped at 2^14 (C2, the C3 stand-ins: a hub row is 8 tiles, a group with one of them ~10) publish nothing and -- through the
/// one-shot entry's memo -- keep running the one-kernel form; R-MAT 2^23 publishes its 576 groups of 49 tiles or more (40 % of the
/// nonzeros), the 16-tile class stays with its owners.  Measured (tests/perf/bench_group_mapped.py, profiles/r06_group_mapped_*):
/// thresholds of 4 / 8 take R-MAT to 1.37 / 1.59 ms but cost the host-blocked and band stand-ins 9-15 % / 3-6 % (two kernels, two
/// tails, claims of one group on different XCDs).
#ifndef LOOPS_GROUP_HEAVY_TILES  // (tuning builds only: tests/perf/bench_group_mapped.py with LOOPS_AMD_LIB)
#define LOOPS_GROUP_HEAVY_TILES 24
#endif
constexpr int group_heavy_tiles = LOOPS_GROUP_HEAVY_TILES;  ///< (in tiles of 2 048 items)
/// Tiles per claim: the open row is carried in a register through `group_claim_tiles` consecutive tiles (one carry-out and one
/// load of the group's row ends per claim).  A claim is a serial chain of ~5 us per tile: with few claims it IS the second
/// kernel's duration (4 tiles: 25 us on a matrix that publishes a handful of groups), hence 2.
#ifndef LOOPS_GROUP_CLAIM_TILES
#define LOOPS_GROUP_CLAIM_TILES 2
#endif
constexpr int group_claim_tiles = LOOPS_GROUP_CLAIM_TILES;  ///< (in tiles of 2 048 items)
/// The two thresholds for a tile of `tile` items: the same ITEM counts (and the same scratch layout) whatever the tile shape.
constexpr int group_heavy_tiles_of(int tile) { return group_heavy_tiles * 2048 / tile > 1 ? group_heavy_tiles * 2048 / tile : 1; }
constexpr int group_claim_tiles_of(int tile) { return group_claim_tiles * 2048 / tile > 1 ? group_claim_tiles * 2048 / tile : 1; }

/// Control words of one group_mapped call (zero between calls: the fix-up kernel leaves them so).
struct group_share_ctl {
  unsigned long long packed;  ///< published groups << 32 | claims handed out
  unsigned int fix_done;      ///< fix-up workgroups that are through
  unsigned int pad;
};
struct group_share_rec {
  int group, first, claims, pad;
};

namespace detail {
struct group_share_layout {
  std::size_t G, U, off_rec, off_val, off_row, off_slot, bytes;
};
/// ctl (256 B) | rec [heavy groups] | carry_val [U] | carry_row [U] | claim_slot [U]; U = claims any matrix of this size can publish.
inline group_share_layout group_share_layout_of(int rows, int nnz, int TPB, int IPT, std::size_t vbytes) {
  group_share_layout l;
  const std::size_t items = static_cast<std::size_t>(rows) + static_cast<std::size_t>(nnz), tile = static_cast<std::size_t>(TPB) * IPT;
  l.G = items / (tile * group_heavy_tiles_of(static_cast<int>(tile))) + 1;  // groups that can be heavy
  l.U = items / (tile * group_claim_tiles_of(static_cast<int>(tile))) + 2 * l.G + 1;  // (sum of ceil(tiles / claim) over the heavy groups)
  l.off_rec = 256;
  l.off_val = l.off_rec + sizeof(group_share_rec) * l.G;
  l.off_row = l.off_val + ((vbytes * l.U + 15) & ~std::size_t(15));
  l.off_slot = l.off_row + ((4 * l.U + 15) & ~std::size_t(15));
  l.bytes = l.off_slot + 4 * l.U + 64;
  return l;
}
}  // namespace detail

/// Bytes of scratch `launch_group_mapped_shared` needs for a matrix.  Its first 256 bytes must be ZERO before the first call
/// (every call leaves them zero again); calls that share a block must be ordered by the stream.
template <typename type_t>
inline std::size_t group_share_scratch_bytes(int rows, int nnz, int TPB, int IPT) {
  return detail::group_share_layout_of(rows, nnz, TPB, IPT, sizeof(type_t)).bytes;
}

template <typename type_t>
struct group_share_view {
  group_share_ctl* ctl;
  group_share_rec* rec;  ///< [published groups] sorted by `first`
  int* claim_slot;       ///< [claims] record of the claim's group
  int* carry_row;        ///< [claims]
  type_t* carry_val;     ///< [claims]
  int max_claims;        ///< claims a matrix of this size can hand out
  static group_share_view carve(void* base, int rows, int nnz, int TPB, int IPT) {
    const auto l = detail::group_share_layout_of(rows, nnz, TPB, IPT, sizeof(type_t));
    char* p = static_cast<char*>(base);
    return group_share_view{reinterpret_cast<group_share_ctl*>(p), reinterpret_cast<group_share_rec*>(p + l.off_rec),
                            reinterpret_cast<int*>(p + l.off_slot), reinterpret_cast<int*>(p + l.off_row), reinterpret_cast<type_t*>(p + l.off_val),
                            static_cast<int>(l.U)};
  }
};

namespace detail {
/// Group owned by workgroup `i` of `m` in the publish kernel: runs of 37 consecutive groups dealt to the 8 XCDs in turn
/// (workgroups reach the XCDs round-robin).  A group's work is NOT uniform, and on a Kronecker / R-MAT graph it is a product over
/// the BITS of the group's index: any assignment of groups to XCDs by a bit field of the index -- one contiguous eighth each
/// (xcd_contiguous: right for uniform tiles), the hardware's own i mod 8, power-of-two runs -- gives one XCD 32 times the work of
/// another (0.76 / 0.24 per bit, three bits), and it then runs alone: R-MAT 2^23, groups of up to 16 tiles left to their owners,
/// 1.75 ms for 60 % of the tiles against 0.61 ms for the other 40 % as uniform claims.  A run length that is not a power of two
/// cuts across the bits: 37 -> 0.78 ms (whole call 2.40 -> 1.39 ms; runs of 100 / 1000: 1.71 / 2.19 ms), while 37 x 256 consecutive
/// rows still share an L2's neighbourhood of x (band stand-in: + 2.4 % against one contiguous eighth per XCD).
__device__ __forceinline__ int xcd_chunk_cyclic(const int i, const int m) {
#ifndef LOOPS_GROUP_RUN
#define LOOPS_GROUP_RUN 37
#endif
  constexpr int XCDS = 8, RUN = LOOPS_GROUP_RUN;
  const int full = m / (XCDS * RUN) * (XCDS * RUN);
  if (i >= full) return i;
  const int k = i % XCDS, j = i / XCDS;
  return (j / RUN * XCDS + k) * RUN + j % RUN;
}
}  // namespace detail

/// What the kernels share: the rows of ONE group in LDS and tiles cut out of the group's own merge path.
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t, bool MASK>
struct group_tiles {
  using engine_t = merge_tile_engine<TPB, IPT, PAD, NT, VEC, index_t, offset_t, type_t, MASK>;
  static constexpr int TILE = TPB * IPT;
  struct storage_t {
    typename engine_t::storage_t engine;
    offset_t off[TPB + 1 + IPT];  // offsets of the group's rows (+ clamped slack)
  };
  storage_t& s;
  int group_row0 = 0, group_rows = 0, nz_begin = 0, group_atoms = 0, total = 0;
  __device__ __forceinline__ explicit group_tiles(storage_t& storage) : s(storage) {}

  /// Collective; contains a barrier behind the loads.
  __device__ __forceinline__ void load_group(const int g, const int rows, const offset_t* __restrict__ offsets) {
    group_row0 = g * TPB;
    group_rows = rows - group_row0;
    group_rows = group_rows < TPB ? group_rows : TPB;
    for (int i = threadIdx.x; i < TPB + 1 + IPT; i += TPB) {
      int r = group_row0 + i;
      r = r < rows ? r : rows;
      s.off[i] = offsets[r];
    }
    __syncthreads();
    nz_begin = s.off[0];
    group_atoms = s.off[group_rows] - nz_begin;
    total = group_rows + group_atoms;
  }
  __device__ __forceinline__ int tiles() const { return (total + TILE - 1) / TILE; }
  /// Rows consumed when the group's merge path reaches diagonal d: every lane does the same <= log2(TPB) LDS probes.
  __device__ __forceinline__ int split(const int d) const {
    const offset_t* row_end = s.off + 1;
    int lo = d - group_atoms > 0 ? d - group_atoms : 0;
    int count = (d < group_rows ? d : group_rows) - lo;
    while (count > 0) {
      const int half = count >> 1;
      const int mid = lo + half;
      if (row_end[mid] <= nz_begin + (d - mid - 1)) {
        lo = mid + 1;
        count -= half + 1;
      } else {
        count = half;
      }
    }
    return lo < group_rows ? lo : group_rows;
  }
  /// The tiles of diagonals [d_begin, d_end) in sequence, the open row carried in a register.  Returns the partial sum of the
  /// row open at d_end and leaves that row (inside the group; == group_rows when nothing is open) in `open_row`.
  __device__ __forceinline__ type_t run(const int d_begin, const int d_end, const int nnz, const index_t* __restrict__ indices,
                                        const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y, int& open_row,
                                        const detail::phase_args phase = {}) {
    const offset_t* row_end = s.off + 1;
    type_t carry = type_t(0);
    int tx0 = split(d_begin);
    for (int d0 = d_begin; d0 < d_end; d0 += TILE) {
      const int d1 = d0 + TILE < d_end ? d0 + TILE : d_end;
      const int tx1 = split(d1);
      const int ty0 = d0 - tx0, ty1 = d1 - tx1;
      if constexpr (MASK) {  // row ends of the tile (already in LDS) -> marks of the engine's bit mask
        engine_t::clear_marks(s.engine);
        __syncthreads();
        for (int i = threadIdx.x; i < tx1 - tx0; i += TPB) engine_t::mark_row_end(s.engine, i, static_cast<int>(row_end[tx0 + i]), nz_begin + ty0);
      }
      carry = engine_t::run_to(s.engine, row_end + tx0, group_row0 + tx0, nz_begin + ty0, tx1 - tx0, ty1 - ty0, nnz, indices, values, x,
                               plain_store<type_t>{y}, carry, typename engine_t::no_marks{}, phase);
      tx0 = tx1;
    }
    open_row = tx0;
    return carry;
  }
};

namespace detail {
/// policy::phased_auto builds: plain or phased gathers by what column_scatter_sample left in `stats` (merge_path_spmv_fused_auto's rule).
template <int NT>
__device__ __forceinline__ phase_args group_phase(const unsigned int* __restrict__ stats, phase_args phase) {
  if constexpr (policy::phased_is_auto(NT)) {
    const unsigned int far_same = stats[0], far_seen = stats[1], near_same = stats[2], near_seen = stats[3];
    phase.enabled = scatter_counts_say_phase(far_same, far_seen, near_same, near_seen, policy::phases(NT)) ? 1u : 0u;
  }
  return phase;
}
}  // namespace detail

/// SHARE = false: every group swept by its owner (what the one-shot entry launches alone once its memo says the matrix has no
/// heavy group -- group_mapped_spmv_fused with a choice of gather order).
template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t, bool MASK = false, bool SHARE = true>
__global__ void __launch_bounds__(TPB, (TPB == 256 && sizeof(type_t) == 4 && NT == 0 ? 8 : 1))  // (fp32: 64 VGPRs, 8 workgroups per CU, as group_mapped_spmv_fused compiles)
group_mapped_spmv_publish(const int rows, const int nnz, const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                          const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                          const group_share_view<type_t> sc, const unsigned int* __restrict__ stats, detail::phase_args phase) {
  using tiles_t = group_tiles<TPB, IPT, PAD, NT, VEC, index_t, offset_t, type_t, MASK>;
  __shared__ typename tiles_t::storage_t storage;
  tiles_t gt(storage);
  phase = detail::group_phase<NT>(stats, phase);
#ifdef LOOPS_GROUP_CONTIGUOUS  // (tuning builds only)
  const int g = detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
#else
  const int g = detail::xcd_chunk_cyclic(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
#endif
  gt.load_group(g, rows, offsets);
  const int tiles = gt.tiles();
  if (!SHARE || tiles <= group_heavy_tiles_of(tiles_t::TILE)) {
    int open_row;
    (void)gt.run(0, gt.total, nnz, indices, values, x, y, open_row, phase);
  } else {
    __shared__ unsigned long long s_got;
    constexpr int per_claim = group_claim_tiles_of(tiles_t::TILE);
    const int claims = (tiles + per_claim - 1) / per_claim;
    if (threadIdx.x == 0) {
      s_got = atomicAdd(&sc.ctl->packed, (1ull << 32) | static_cast<unsigned long long>(claims));
      sc.rec[s_got >> 32] = group_share_rec{g, static_cast<int>(s_got & 0xFFFFFFFFull), claims, 0};
    }
    __syncthreads();
    const int slot = static_cast<int>(s_got >> 32), first = static_cast<int>(s_got & 0xFFFFFFFFull);
    for (int c = threadIdx.x; c < claims; c += TPB) sc.claim_slot[first + c] = slot;
  }
}

template <int TPB, int IPT, bool PAD, int NT, bool VEC, typename index_t, typename offset_t, typename type_t, bool MASK = false>
__global__ void __launch_bounds__(TPB, (TPB == 256 && sizeof(type_t) == 4 && NT == 0 ? 8 : 1))
group_mapped_spmv_claims(const int rows, const int nnz, const offset_t* __restrict__ offsets, const index_t* __restrict__ indices,
                         const type_t* __restrict__ values, const type_t* __restrict__ x, type_t* __restrict__ y,
                         const group_share_view<type_t> sc, const unsigned int* __restrict__ stats, detail::phase_args phase) {
  using tiles_t = group_tiles<TPB, IPT, PAD, NT, VEC, index_t, offset_t, type_t, MASK>;
  constexpr int CLAIM_ITEMS = group_claim_tiles_of(tiles_t::TILE) * tiles_t::TILE;
  __shared__ typename tiles_t::storage_t storage;
  const unsigned long long packed = sc.ctl->packed;
  const int total_claims = static_cast<int>(packed & 0xFFFFFFFFull);
  tiles_t gt(storage);
  phase = detail::group_phase<NT>(stats, phase);
  int loaded = -1;
  // (a fixed grid strides over the claim numbers: the host does not know how many were handed out, and an upper-bound grid of
  //  workgroups that exit at once still costs ~0.5 us per thousand)
  for (int base = 0; base < total_claims; base += static_cast<int>(gridDim.x)) {
    // this round's claim numbers, dealt so that each XCD works on one contiguous run of them (neighbouring claims -- tiles of one
    // group, of neighbouring groups -- share an L2's view of x, as merge_path_flat's tiles do)
    const int round = total_claims - base < static_cast<int>(gridDim.x) ? total_claims - base : static_cast<int>(gridDim.x);
    if (static_cast<int>(blockIdx.x) >= round) break;  // (workgroup-uniform)
    const int b = base + detail::xcd_contiguous(static_cast<int>(blockIdx.x), round);
    const int lo = sc.claim_slot[b];
    const group_share_rec r = sc.rec[lo];
    if (r.group != loaded) {
      __syncthreads();  // (the previous claim's tiles no longer read the row ends)
      gt.load_group(r.group, rows, offsets);
      loaded = r.group;
    }
    const int c = b - r.first;
    const int d0 = c * CLAIM_ITEMS;
    const int d1 = d0 + CLAIM_ITEMS < gt.total ? d0 + CLAIM_ITEMS : gt.total;
    int open_row;
    const type_t carry = gt.run(d0, d1, nnz, indices, values, x, y, open_row, phase);
    if (threadIdx.x == 0) {
      sc.carry_row[b] = open_row < gt.group_rows ? gt.group_row0 + open_row : rows;  // (`rows`: nothing open -- the group ends with this claim)
      sc.carry_val[b] = carry;
    }
  }
}

/// Fix-up of the published groups: y[row] += the carry-outs of the run of claims that ended inside `row`, in claim order (one
/// thread per claim, grid-stride; the claims of a group are consecutive numbers and a group's last claim leaves nothing open, so
/// a run never crosses into another group -- merge_path_flat's fix-up over the claim numbers).  The last workgroup through clears
/// the control words and, when `report` is not null, stores 1 + the number of published groups there (`report` may be host
/// memory mapped into the device).  Any grid size; 256 threads.
template <typename type_t>
__global__ void __launch_bounds__(256)
group_mapped_fixup(const int rows, type_t* __restrict__ y, const group_share_view<type_t> sc, unsigned int* __restrict__ report) {
  const unsigned long long packed = sc.ctl->packed;
  const int n = static_cast<int>(packed >> 32), total_claims = static_cast<int>(packed & 0xFFFFFFFFull);
  const int* __restrict__ cr = sc.carry_row;
  const type_t* __restrict__ cv = sc.carry_val;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total_claims; i += gridDim.x * 256) {
    const int row = cr[i];
    const int before = i > 0 ? cr[i - 1] : -1;
    const int after = i + 1 < total_claims ? cr[i + 1] : -1;
    if (row >= rows || before == row) continue;  // nothing open / not the first claim of the run
    type_t sum = cv[i];
    if (after == row) {
      sum += cv[i + 1];
      for (int j = i + 2; j < total_claims && cr[j] == row; ++j) sum += cv[j];
    }
    y[row] = y[row] + sum;
  }
  // (every workgroup has read the counter by the time it takes a ticket: the last one may clear it)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = atomicAdd(&sc.ctl->fix_done, 1u);
    if (done == gridDim.x - 1) {
      if (report) __hip_atomic_store(report, static_cast<unsigned int>(n) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      sc.ctl->packed = 0ull;
      sc.ctl->fix_done = 0u;
    }
  }
}

/// Tuned group_mapped with heavy groups shared out (file comment).  `scratch`: group_share_scratch_bytes<T>(rows, nnz, TPB, IPT)
/// bytes whose first 256 are zero before the first call.  `report` (may be null; device-visible): receives 1 + the number of
/// groups the call published when its last kernel ends.  `stats` (may be null: plain gathers): kernels::scatter_scratch_words
/// device words -- with them, and `cols` columns worth asking about, the gather order is decided on the device (file comment).
/// `share` = false: the publish kernel alone with every group swept by its owner (callers that KNOW the matrix has no heavy group).
/// `resample` = false: `stats` still hold the sample of this matrix; `timed_path`: columns_worth_sampling's (false for callers
/// that pay the sample once per matrix).
template <int TPB, int IPT, bool PAD, typename index_t, typename offset_t, typename T, bool MASK = true>
int launch_group_mapped_shared(hipStream_t stream, int rows, int nnz, const offset_t* offsets, const index_t* indices, const T* values, const T* x,
                               T* y, void* scratch, unsigned int* report = nullptr, int cols = 0, unsigned int* stats = nullptr, bool share = true,
                               bool resample = true, bool timed_path = true) {
  if (rows == 0) return 0;
  const auto sc = share ? group_share_view<T>::carve(scratch, rows, nnz, TPB, IPT) : group_share_view<T>{};
  const bool aligned = ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) == 0;
  const dim3 groups(math::ceil_div(rows, TPB)), claims(sc.max_claims < 8192 ? sc.max_claims : 8192), block(TPB);
  const bool sampled = aligned && stats && columns_worth_sampling(static_cast<long long>(nnz), static_cast<long long>(cols), static_cast<int>(sizeof(T)), timed_path);
  auto go = [&](auto publish, auto owner, auto claim, const detail::phase_args ph) {
    if (!share) {
      hipLaunchKernelGGL(owner, groups, block, 0, stream, rows, nnz, offsets, indices, values, x, y, sc, stats, ph);
      return;
    }
    hipLaunchKernelGGL(publish, groups, block, 0, stream, rows, nnz, offsets, indices, values, x, y, sc, stats, ph);
    hipLaunchKernelGGL(claim, claims, block, 0, stream, rows, nnz, offsets, indices, values, x, y, sc, stats, ph);
  };
  if (sampled) {
    const int err = resample ? launch_column_scatter_sample(stream, indices, static_cast<long long>(nnz), static_cast<long long>(cols), static_cast<int>(sizeof(T)), stats) : 0;
    if (err) return err;
    const phased_config cfg = phased_config_for(cols, static_cast<int>(sizeof(T)));
    // (tiles of 16 items per lane for 16 / 32 parts, the plan-less merge_path_flat entry's shape, measured here: uniform C3 stand-in
    //  2.65 against 2.78 ms, but R-MAT 2^23 -- whose sample says "plain" -- 1.44 against 1.385: a group of 256 rows is 1.6 such tiles,
    //  the second one mostly empty and as many passes long.  One shape, then.)
    auto with = [&](auto parts) {
      constexpr int NT = detail::policy::phased_auto(decltype(parts)::value);
      go(group_mapped_spmv_publish<TPB, IPT, PAD, NT, true, index_t, offset_t, T, MASK, true>,
         group_mapped_spmv_publish<TPB, IPT, PAD, NT, true, index_t, offset_t, T, MASK, false>,
         group_mapped_spmv_claims<TPB, IPT, PAD, NT, true, index_t, offset_t, T, MASK>, cfg.args);
    };
    if (cfg.parts == 8) with(std::integral_constant<int, 8>{});
    else if (cfg.parts == 16) with(std::integral_constant<int, 16>{});
    else with(std::integral_constant<int, 32>{});
  } else if (aligned) {
    go(group_mapped_spmv_publish<TPB, IPT, PAD, 0, true, index_t, offset_t, T, MASK, true>,
       group_mapped_spmv_publish<TPB, IPT, PAD, 0, true, index_t, offset_t, T, MASK, false>,
       group_mapped_spmv_claims<TPB, IPT, PAD, 0, true, index_t, offset_t, T, MASK>, detail::phase_args{});
  } else {
    go(group_mapped_spmv_publish<TPB, IPT, PAD, 0, false, index_t, offset_t, T, MASK, true>,
       group_mapped_spmv_publish<TPB, IPT, PAD, 0, false, index_t, offset_t, T, MASK, false>,
       group_mapped_spmv_claims<TPB, IPT, PAD, 0, false, index_t, offset_t, T, MASK>, detail::phase_args{});
  }
  if (share) hipLaunchKernelGGL((group_mapped_fixup<T>), dim3(64), dim3(256), 0, stream, rows, y, sc, report);
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
