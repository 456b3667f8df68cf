/**
 * @file rowband.cuh
 * @brief `algorithms::spmv::rowband_t<index_t, offset_t, float | double>`: a CSR held in the row-band layout (loops/kernels/rowband.hxx)
 * -- the y accumulators of a band of rows live in LDS, the band's nonzeros are sorted by column so that a wavefront's x gathers
 * fall on a few neighbouring lines; 3 bytes per nonzero streamed next to the value (row code, column as a one-byte delta).  For an x of a few MB or column locality at band scale.  The
 * header-API twin of loops_rowband_plan_* (include/loops_amd.h).  No reference counterpart (its merge_path_flat.cuh:71-82 pays
 * one global atomic per nonzero, its CSR kernels one scattered gather).
 *
 *   algorithms::spmv::rowband_t<int, int, float> A(csr);
 *   A.spmv(x, y);                                                       // y = csr * x
 *
 * One product in flight per object (it owns the partial-vector scratch).  4- and 8-byte values.  The band sums are fp64 LDS
 * atomics that arrive in no fixed order: see include/loops_amd.h (LOOPS_PLAN_DETERMINISTIC) for when that cannot matter.
 */
#pragma once

#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/error.hxx>
#include <loops/kernels/rowband.hxx>
#include <loops/util/timer.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
struct rowband_t {
  static_assert(sizeof(index_t) == 4 && sizeof(offset_t) == 4, "rowband_t: 32-bit indices and offsets");
  static_assert(sizeof(type_t) == 4 || sizeof(type_t) == 8, "rowband_t: 4- or 8-byte values");
  std::size_t rows, cols, nnzs;
  kernels::rowband_storage arrays;   ///< H, B, steps, chunks, waves and the owned device arrays

  /// @param band_rows 0 = automatic (kernels::rowband_rows), or a power of two in [64, 16384]
  /// @param target_chunks 0 = automatic (kernels::rowband_target_chunks)
  explicit rowband_t(csr_t<index_t, offset_t, type_t>& csr, int band_rows = 0, int target_chunks = 0, xpu::stream_t stream = 0)
      : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs) {
    const int err = kernels::rowband_create<index_t, offset_t, type_t>(
        stream, static_cast<int>(rows), static_cast<int>(cols), static_cast<int>(nnzs), csr.offsets.data().get(), csr.indices.data().get(),
        csr.values.data().get(), band_rows, target_chunks, arrays);
    if (err == kernels::rowband_e_badarg)
      throw error::bad_argument_t("rowband_t: band_rows must be a power of two in [64, 16384] and every column index inside [0, cols)");
    error::throw_if_exception(err == kernels::rowband_e_range, "rowband_t: nnz + padding (bands x (cols / 255 + 256)) must stay below 2^31");
    error::throw_if_exception(err != 0, "rowband_t: build failed");
  }

  /// Whether rowband_create can take the matrix at all (the bounds it checks before it sorts).
  static bool fits(const csr_t<index_t, offset_t, type_t>& csr) {
    const int h = kernels::rowband_rows(static_cast<int>(csr.rows), static_cast<int>(csr.cols), static_cast<int>(csr.nnzs), static_cast<int>(sizeof(type_t)));
    const long long bands = (static_cast<long long>(csr.rows) + h - 1) / h;
    return bands <= (1ll << 26) &&
           static_cast<long long>(csr.nnzs) + bands * (static_cast<long long>(csr.cols) / kernels::rowband::max_delta + kernels::rowband::step_items) < (1ll << 31) - 4096;
  }

  kernels::rowband_view<type_t> view() const { return arrays.template view<type_t>(); }

  /// Times the product with 8 and 16 wavefronts per workgroup and keeps the faster (kernels::rowband_tune).
  void tune(int repeats = 10, xpu::stream_t stream = 0) {
    error::throw_if_exception(kernels::rowband_tune<type_t>(stream, arrays, repeats, nullptr) != 0, "rowband_t::tune failed");
  }

  /// y = A x; asynchronous on `stream`.
  void spmv_async(vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    if (rows == 0) return;
    kernels::launch_rowband<type_t>(stream, view(), x.data().get(), y.data().get());
  }

  /// The same product for one rank of a row-range sharded multi-GPU SpMV: the finished rows of y also go to `peers`.
  void spmv_fanout_async(vector_t<type_t>& x, vector_t<type_t>& y, const kernels::peer_fanout<type_t>& peers, xpu::stream_t stream = 0) {
    error::throw_if_exception(peers.count < 0 || peers.count > kernels::max_peers, "rowband_t::spmv_fanout_async: peers.count must be 0 .. 7");
    if (rows == 0) return;
    kernels::launch_rowband_fanout<type_t>(stream, view(), x.data().get(), y.data().get(), peers);
  }

  util::timer_t spmv(vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    util::timer_t timer(stream);
    timer.start();
    spmv_async(x, y, stream);
    (void)xpu::stream_synchronize(stream);
    timer.stop();
    return timer;
  }
};

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
