/**
 * @file panel_binned.cuh
 * @brief `algorithms::spmv::panel_binned_t<index_t, offset_t, type_t>`: a CSR held panel-binned (loops/kernels/panel_binned.hxx)
 * -- the layout for matrices whose x is far larger than the 4 MB per-XCD L2: x is read in panels that fit a CU's LDS, the
 * x value of a nonzero is an LDS read instead of a memory gather, products and their row-wise sums stream.  The header-API
 * twin of loops_panel_plan_* (include/loops_amd.h).  No reference counterpart (the reference leaves the gather to the cache).
 *
 *   algorithms::spmv::panel_binned_t<int, int, float> A(csr);
 *   A.spmv(x, y);                                                       // y = csr * x
 *
 * One product in flight per object (it owns the products scratch).
 */
#pragma once

#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/error.hxx>
#include <loops/kernels/panel_binned.hxx>
#include <loops/util/timer.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
struct panel_binned_t {
  static_assert(sizeof(index_t) == 4 && sizeof(offset_t) == 4, "panel_binned_t: 32-bit indices and offsets");
  std::size_t rows, cols, nnzs;
  kernels::panel_binned_storage arrays;   ///< W, Hw, P, S, padded, padded_b, compact, runs and the owned device arrays

  /// @param subband_rows 0 = automatic (kernels::panel_subband_rows), or a power of two in [64, kernels::panel_subband_rows_max<type_t>()]
  /// @param compact -1 = automatic, 0 = one B-order slot per nonzero, 1 = one per run of equal (row, panel) (pre-summed by kernel A)
  explicit panel_binned_t(csr_t<index_t, offset_t, type_t>& csr, int subband_rows = 0, xpu::stream_t stream = 0, int compact = -1)
      : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs) {
    const int err = kernels::panel_binned_create<index_t, offset_t, type_t>(
        stream, static_cast<int>(rows), static_cast<int>(cols), static_cast<int>(nnzs), csr.offsets.data().get(), csr.indices.data().get(),
        csr.values.data().get(), subband_rows, 0, compact, arrays);
    error::throw_if_exception(err == kernels::panel_e_badarg,
                              "panel_binned_t: subband_rows must be a power of two in [64, panel_subband_rows_max<type_t>()] and every column index inside [0, cols)");
    error::throw_if_exception(err == kernels::panel_e_range,
                              "panel_binned_t: panels x sub-bands must stay below 2^26 and nnz + padding below 2^31");
    error::throw_if_exception(err != 0, "panel_binned_t: build failed");
  }

  kernels::panel_binned_view<type_t> view() const { return arrays.template view<type_t>(); }

  /// y = A x; asynchronous on `stream`.
  void spmv_async(vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    if (rows == 0) return;
    kernels::launch_panel_binned<type_t>(stream, view(), x.data().get(), y.data().get());
  }

  /// The same product for one rank of a row-range sharded multi-GPU SpMV: the finished rows of y also go to `peers`.
  void spmv_fanout_async(vector_t<type_t>& x, vector_t<type_t>& y, const kernels::peer_fanout<type_t>& peers, xpu::stream_t stream = 0) {
    error::throw_if_exception(peers.count < 0 || peers.count > kernels::max_peers, "panel_binned_t::spmv_fanout_async: peers.count must be 0 .. 7");
    if (rows == 0) return;
    kernels::launch_panel_binned_fanout<type_t>(stream, view(), x.data().get(), y.data().get(), peers);
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
