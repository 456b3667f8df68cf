/**
 * @file column_blocked.hxx
 * @brief Column-blocked ("stacked") CSR: the layout the row-range shards of a multi-GPU SpMV --
 * and any matrix whose x does not fit the 4 MB per-XCD L2 -- are held in.
 *
 * The columns are cut into K blocks; stacked row (k * rows + r) holds the nonzeros of row r whose
 * column lies in block k, in their original order.  That is an ordinary CSR of K * rows rows, so the
 * SpMV over it IS the fused merge_path_flat kernel, unchanged -- and because that kernel hands
 * every XCD one CONTIGUOUS run of merge tiles (detail::xcd_contiguous), XCD j works on (about)
 * column block j * K / 8 only: its private L2 has to hold x[block] (cols * 4 / K bytes) instead of
 * all of x.  A K-way row reduce y[r] = sum_k ys[k * rows + r] finishes the product.
 *
 * Why: the x gather, not HBM, bounds CSR SpMV with scattered columns; its rate drops
 * from 180 G/s (x = 4 MB, mostly L2 hits) to ~55 G/s once x lives in Infinity Cache / HBM, which is
 * where every rank of an N-GPU run is (x = N * 4 MB).  Measured on one MI355X, shard of 2^20 rows /
 * 2^24 nnz with x = 32 MB: 282 us plain -> 129 us blocked (K = 8).
 *
 * In a multi-GPU run the natural blocks are the owners' row ranges (x[block k] = the y slice rank k
 * produces), which is also what lets a rank start on block k as soon as that slice has arrived.
 *
 * Build (one-time, on the device, O(nnz)): row of every nonzero (search over offsets) -> 64-bit key
 * (stacked row << 32 | position) -> radix sort on the high bits only (stable: original order kept
 * inside a stacked row) -> gather col_idx / values, histogram + scan for the stacked offsets.
 */
#pragma once

#include <cstddef>
#include <cstdint>

#include <type_traits>

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <loops/kernels/merge_path_spmv.hxx>
#include <loops/util/math.hxx>

namespace loops {
namespace kernels {

namespace colblock {

/// key[i] = (stacked row of nonzero i) << 32 | i ; counts[stacked row + 1] += (nonzeros in it).
/// A lane owns IPT consecutive nonzeros: ONE search over the offsets for the row of the first, then a
/// walk along the offsets; consecutive nonzeros of the same stacked row share one atomic.
/// `bounds` (K + 1 ascending column boundaries, bounds[0] = 0, bounds[K] = cols) in global memory.
template <int IPT, typename index_t, typename offset_t>
__global__ void __launch_bounds__(256)
make_keys(const offset_t* __restrict__ offsets, const index_t* __restrict__ indices, const int rows, const int nnz,
          const int* __restrict__ bounds, const int K, unsigned long long* __restrict__ keys,
          int* __restrict__ counts) {
  const long long base_ll = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * IPT;
  if (base_ll >= nnz) return;
  const int base = static_cast<int>(base_ll);
  // row of nonzero `base`: last r with offsets[r] <= base  (upper_bound over offsets[1..rows])
  int row = 0, count = rows;
  while (count > 0) {
    const int half = count >> 1;
    const int mid = row + half;
    if (offsets[mid + 1] <= base) {
      row = mid + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  offset_t row_end = offsets[row + 1];
  unsigned int run_row = 0xffffffffu;
  int run_len = 0;
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int i = base + j;
    if (i >= nnz) break;
    while (i >= row_end) row_end = offsets[++row + 1];  // skip empty rows
    const int c = static_cast<int>(indices[i]);
    int k = 0;
    while (k + 1 < K && bounds[k + 1] <= c) ++k;
    const unsigned int srow = static_cast<unsigned int>(k) * static_cast<unsigned int>(rows) + static_cast<unsigned int>(row);
    keys[i] = (static_cast<unsigned long long>(srow) << 32) | static_cast<unsigned int>(i);
    if (srow != run_row) {
      if (run_len) atomicAdd(counts + run_row + 1, run_len);
      run_row = srow;
      run_len = 0;
    }
    ++run_len;
  }
  if (run_len) atomicAdd(counts + run_row + 1, run_len);
}

/// perm[j] = original position of the j-th nonzero of the stacked matrix; gathers col_idx.
template <typename index_t>
__global__ void __launch_bounds__(256)
apply_keys(const unsigned long long* __restrict__ sorted, const index_t* __restrict__ indices, const int nnz,
           int* __restrict__ perm, index_t* __restrict__ sidx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nnz) return;
  const int src = static_cast<int>(sorted[j] & 0xffffffffull);
  perm[j] = src;
  sidx[j] = indices[src];
}

template <typename type_t>
__global__ void __launch_bounds__(256)
gather_values(const int* __restrict__ perm, const type_t* __restrict__ values, const int nnz,
              type_t* __restrict__ sval) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nnz) sval[j] = values[perm[j]];
}

/// y[r] = sum_k ys[k * rows + r], k ascending (deterministic).
template <typename type_t>
__global__ void __launch_bounds__(256)
reduce_blocks(const type_t* __restrict__ ys, const int rows, const int K, type_t* __restrict__ y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  type_t s = ys[r];
  for (int k = 1; k < K; ++k) s += ys[static_cast<std::size_t>(k) * rows + r];
  y[r] = s;
}

/// Same, 4 rows per lane (4-byte values, rows % 4 == 0, 16-byte aligned bases).  A template so that the header may be
/// included from several translation units of one binary.
template <typename type_t>
__global__ void __launch_bounds__(256)
reduce_blocks_x4(const type_t* __restrict__ ys, const int rows, const int K, type_t* __restrict__ y) {
  static_assert(sizeof(type_t) == 4, "reduce_blocks_x4: 16-byte vectors of four 4-byte values");
  using f4 = type_t __attribute__((ext_vector_type(4)));
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (r >= rows) return;
  f4 s = *reinterpret_cast<const f4*>(ys + r);
  for (int k = 1; k < K; ++k) s += *reinterpret_cast<const f4*>(ys + static_cast<std::size_t>(k) * rows + r);
  *reinterpret_cast<f4*>(y + r) = s;
}

/// reduce_blocks_x4 with the multi-GPU allgatherv(y) fused in: the finished 16 bytes also go to the same
/// place of every peer-mapped vector (non-temporal 16-byte stores: write-combined over xGMI, nothing kept in L2).
template <typename type_t>
__global__ void __launch_bounds__(256)
reduce_blocks_x4_fanout(const type_t* __restrict__ ys, const int rows, const int K, type_t* __restrict__ y,
                        const peer_fanout<type_t> peers) {
  static_assert(sizeof(type_t) == 4, "reduce_blocks_x4_fanout: 16-byte vectors of four 4-byte values");
  using f4 = type_t __attribute__((ext_vector_type(4)));
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (r >= rows) return;
  f4 s = *reinterpret_cast<const f4*>(ys + r);
  for (int k = 1; k < K; ++k) s += *reinterpret_cast<const f4*>(ys + static_cast<std::size_t>(k) * rows + r);
  *reinterpret_cast<f4*>(y + r) = s;
  for (int p = 0; p < peers.count; ++p) __builtin_nontemporal_store(s, reinterpret_cast<f4*>(peers.base[p] + r));
}

/// Scalar form (any row count / alignment).
template <typename type_t>
__global__ void __launch_bounds__(256)
reduce_blocks_fanout(const type_t* __restrict__ ys, const int rows, const int K, type_t* __restrict__ y,
                     const peer_fanout<type_t> peers) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  type_t s = ys[r];
  for (int k = 1; k < K; ++k) s += ys[static_cast<std::size_t>(k) * rows + r];
  fanout_store<type_t>{y, peers}(r, s);
}

}  // namespace colblock

/// Device arrays of a column-blocked CSR (all owned by the caller).
template <typename index_t, typename offset_t, typename type_t>
struct column_blocked_view {
  int rows, cols, nnz, K;
  offset_t* soff;  ///< K * rows + 1 stacked offsets
  index_t* sidx;   ///< nnz column indices (global ids), stacked order
  type_t* sval;    ///< nnz values, stacked order
  int* perm;       ///< nnz: original position of stacked nonzero j
};

/// Bytes of temporary device storage `build_column_blocked` needs.
inline std::size_t column_blocked_temp_bytes(int nnz, int stacked_rows) {
  std::size_t sort_bytes = 0, scan_bytes = 0;
  unsigned long long* k = nullptr;
  int* c = nullptr;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, k, k, nnz, 32, 64);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, c, c, stacked_rows + 1);
  const std::size_t a = (sort_bytes + 255) & ~std::size_t(255), b = (scan_bytes + 255) & ~std::size_t(255);
  return 2 * (static_cast<std::size_t>(nnz) * 8 + 256) + (a > b ? a : b);
}

/// Builds the stacked offsets / indices / permutation of `out` from a CSR; asynchronous on `stream`.
/// `bounds_dev`: K + 1 column boundaries on the device.  `temp`: column_blocked_temp_bytes() bytes.
template <typename index_t, typename offset_t, typename type_t>
int build_column_blocked(hipStream_t stream, const offset_t* offsets, const index_t* indices, const type_t* values,
                         const int* bounds_dev, column_blocked_view<index_t, offset_t, type_t> out, void* temp,
                         std::size_t temp_bytes) {
  const int nnz = out.nnz, srows = out.K * out.rows;
  hipError_t e = hipMemsetAsync(out.soff, 0, sizeof(offset_t) * (static_cast<std::size_t>(srows) + 1), stream);
  if (e != hipSuccess) return static_cast<int>(e);
  if (nnz == 0) return 0;
  const std::size_t key_bytes = (static_cast<std::size_t>(nnz) * 8 + 255) & ~std::size_t(255);
  auto* keys_in = static_cast<unsigned long long*>(temp);
  auto* keys_out = reinterpret_cast<unsigned long long*>(static_cast<char*>(temp) + key_bytes);
  void* cub_temp = static_cast<char*>(temp) + 2 * key_bytes;
  std::size_t cub_bytes = temp_bytes - 2 * key_bytes;
  const dim3 grid(math::ceil_div(nnz, 256)), block(256);
  static_assert(sizeof(offset_t) == sizeof(int), "stacked offsets are accumulated with 32-bit atomics");
  constexpr int KEYS_PER_LANE = 8;
  hipLaunchKernelGGL((colblock::make_keys<KEYS_PER_LANE, index_t, offset_t>),
                     dim3(math::ceil_div(nnz, 256 * KEYS_PER_LANE)), block, 0, stream, offsets, indices, out.rows, nnz,
                     bounds_dev, out.K, keys_in, reinterpret_cast<int*>(out.soff));
  int end_bit = 33;
  while (end_bit < 64 && (static_cast<unsigned long long>(srows) >> (end_bit - 32)) != 0) ++end_bit;
  e = hipcub::DeviceRadixSort::SortKeys(cub_temp, cub_bytes, keys_in, keys_out, nnz, 32, end_bit, stream);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((colblock::apply_keys<index_t>), grid, block, 0, stream, keys_out, indices, nnz, out.perm,
                     out.sidx);
  hipLaunchKernelGGL((colblock::gather_values<type_t>), grid, block, 0, stream, out.perm, values, nnz, out.sval);
  cub_bytes = temp_bytes - 2 * key_bytes;
  e = hipcub::DeviceScan::InclusiveSum(cub_temp, cub_bytes, reinterpret_cast<int*>(out.soff),
                                       reinterpret_cast<int*>(out.soff), srows + 1, stream);
  if (e != hipSuccess) return static_cast<int>(e);
  return static_cast<int>(hipGetLastError());
}

template <typename type_t>
int launch_reduce_blocks(hipStream_t stream, const type_t* ys, int rows, int K, type_t* y) {
  if (rows == 0) return 0;
  if constexpr (std::is_same<type_t, float>::value) {
    const bool vec = rows % 4 == 0 && ((reinterpret_cast<std::uintptr_t>(ys) | reinterpret_cast<std::uintptr_t>(y)) & 15u) == 0;
    if (vec) {
      hipLaunchKernelGGL((colblock::reduce_blocks_x4<type_t>), dim3(math::ceil_div(rows / 4, 256)), dim3(256), 0, stream, ys, rows,
                         K, y);
      return static_cast<int>(hipGetLastError());
    }
  }
  hipLaunchKernelGGL((colblock::reduce_blocks<type_t>), dim3(math::ceil_div(rows, 256)), dim3(256), 0, stream, ys, rows,
                     K, y);
  return static_cast<int>(hipGetLastError());
}

/// launch_reduce_blocks with the peer fan-out of the finished y (see reduce_blocks_x4_fanout).
template <typename type_t>
int launch_reduce_blocks_fanout(hipStream_t stream, const type_t* ys, int rows, int K, type_t* y,
                                const peer_fanout<type_t>& peers) {
  if (peers.count < 0 || peers.count > max_peers) return static_cast<int>(hipErrorInvalidValue);  // as launch_merge_path_fused_fanout
  if (rows == 0) return 0;
  if constexpr (std::is_same<type_t, float>::value) {
    std::uintptr_t bits = reinterpret_cast<std::uintptr_t>(ys) | reinterpret_cast<std::uintptr_t>(y);
    for (int p = 0; p < peers.count; ++p) bits |= reinterpret_cast<std::uintptr_t>(peers.base[p]);
    if (rows % 4 == 0 && (bits & 15u) == 0) {
      hipLaunchKernelGGL((colblock::reduce_blocks_x4_fanout<type_t>), dim3(math::ceil_div(rows / 4, 256)), dim3(256), 0, stream,
                         ys, rows, K, y, peers);
      return static_cast<int>(hipGetLastError());
    }
  }
  hipLaunchKernelGGL((colblock::reduce_blocks_fanout<type_t>), dim3(math::ceil_div(rows, 256)), dim3(256), 0, stream, ys, rows,
                     K, y, peers);
  return static_cast<int>(hipGetLastError());
}

}  // namespace kernels
}  // namespace loops
