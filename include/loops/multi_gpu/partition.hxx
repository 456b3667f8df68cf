/**
 * @file partition.hxx
 * @brief Row-range partition of a CSR for the GPUs of one node: y = A x is independent per row, so the matrix is cut into
 * `parts` contiguous row ranges balanced by (rows + nonzeros) -- the merge-path diagonal split the kernels use per
 * workgroup (`search::_binary_search`, util/search.hxx, the function reference include/loops/util/search.hxx:35-60 defines),
 * applied once more at GPU granularity: boundary p is where diagonal p (rows + nnz) / parts crosses the merge path of
 * (row ends, nonzero indices); a boundary that falls inside a row moves to that row's start (rows are never split).
 * Every rank keeps x replicated, runs the single-GPU kernels on its slice (offsets rebased to 0, global column ids) and the
 * slices of y are exchanged by ONE allgatherv (multi_gpu/allgatherv.hxx).
 *
 * The reference is single-GPU (no collective anywhere); this is the multi-GPU leg BASELINE.json asks for
 * (config C5).  The Python host side (loops_amd/partition.py) calls the same code through the C ABI (loops_row_ranges).
 */
#pragma once

#include <cstddef>
#include <vector>

#include <hip/hip_runtime.h>

#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/error.hxx>
#include <loops/util/search.hxx>

namespace loops {
namespace multi_gpu {

/// Host-side counting iterator for the "nonzero indices" list of the merge path (0, 1, 2, ...).
struct counting_t {
  long long operator[](long long i) const { return i; }
};

/// parts + 1 row boundaries over HOST offsets (rows + 1 entries): range p = [b[p], b[p + 1]) holds ~ (rows + nnz) / parts merge
/// items.  rows + nnz must stay below 2^31 (the int arithmetic of the search).  Returns false on bad arguments.
template <typename offset_t>
inline bool row_ranges(const offset_t* offsets, std::size_t rows, int parts, long long* bounds) {
  if (!offsets || !bounds || parts < 1) return false;
  const long long nnz = static_cast<long long>(offsets[rows]);
  const long long total = static_cast<long long>(rows) + nnz;
  if (total >= (1ll << 31)) return false;
  bounds[0] = 0;
  for (int p = 1; p < parts; ++p) {
    const int diagonal = static_cast<int>(total * p / parts);
    // the list of row ENDS (offsets + 1) against the nonzero indices: x = rows consumed before the diagonal
    const auto c = search::_binary_search(diagonal, offsets + 1, counting_t{}, static_cast<int>(rows), static_cast<int>(nnz));
    const long long r = static_cast<long long>(c.x);
    bounds[p] = r > bounds[p - 1] ? r : bounds[p - 1];
  }
  bounds[parts] = static_cast<long long>(rows);
  return true;
}

template <typename offset_t>
inline std::vector<long long> row_ranges(const offset_t* offsets, std::size_t rows, int parts) {
  std::vector<long long> b(static_cast<std::size_t>(parts > 0 ? parts : 0) + 1, 0);
  error::throw_if_exception(!row_ranges(offsets, rows, parts, b.data()), "multi_gpu::row_ranges: bad arguments or rows + nnz >= 2^31");
  return b;
}

/// The same for a CSR held on the device (copies the offsets to the host once).
template <typename index_t, typename offset_t, typename type_t>
inline std::vector<long long> row_ranges(const csr_t<index_t, offset_t, type_t>& csr, int parts) {
  std::vector<offset_t> off(csr.rows + 1);
  error::throw_if_exception(
      hipMemcpy(off.data(), csr.offsets.data().get(), sizeof(offset_t) * off.size(), hipMemcpyDeviceToHost) != hipSuccess,
      "multi_gpu::row_ranges: cannot read the offsets");
  return row_ranges(off.data(), csr.rows, parts);
}

namespace detail {
template <typename offset_t>
__global__ void __launch_bounds__(256) rebase_offsets(const offset_t* __restrict__ in, const std::size_t n, offset_t* __restrict__ out) {
  const std::size_t i = static_cast<std::size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] - in[0];
}
}  // namespace detail

/// Rows [row_begin, row_end) of `csr` as a standalone device CSR (offsets rebased to 0, global column ids): one rank's shard.
/// Work is issued on `stream` (the copies are ordered behind it); returns after the shard is complete.  Throws on any failed copy.
template <typename index_t, typename offset_t, typename type_t>
inline csr_t<index_t, offset_t, type_t> slice_rows(const csr_t<index_t, offset_t, type_t>& csr, std::size_t row_begin, std::size_t row_end,
                                                    hipStream_t stream = 0) {
  error::throw_if_exception(row_begin > row_end || row_end > csr.rows, "multi_gpu::slice_rows: bad row range");
  offset_t ends[2] = {0, 0};
  hipError_t e = hipMemcpyAsync(&ends[0], csr.offsets.data().get() + row_begin, sizeof(offset_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&ends[1], csr.offsets.data().get() + row_end, sizeof(offset_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  error::throw_if_exception(e != hipSuccess, "multi_gpu::slice_rows: cannot read the row range's offsets");
  error::throw_if_exception(ends[1] < ends[0], "multi_gpu::slice_rows: offsets are not ascending");
  const std::size_t rows = row_end - row_begin, nnz = static_cast<std::size_t>(ends[1] - ends[0]);
  csr_t<index_t, offset_t, type_t> out(rows, csr.cols, nnz);
  hipLaunchKernelGGL(detail::rebase_offsets<offset_t>, dim3(static_cast<unsigned>((rows + 256) / 256)), dim3(256), 0, stream,
                     csr.offsets.data().get() + row_begin, rows + 1, out.offsets.data().get());
  e = hipGetLastError();
  if (nnz) {
    if (e == hipSuccess)
      e = hipMemcpyAsync(out.indices.data().get(), csr.indices.data().get() + ends[0], sizeof(index_t) * nnz, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(out.values.data().get(), csr.values.data().get() + ends[0], sizeof(type_t) * nnz, hipMemcpyDeviceToDevice, stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  error::throw_if_exception(e != hipSuccess, "multi_gpu::slice_rows: copy failed");
  return out;
}

}  // namespace multi_gpu
}  // namespace loops
