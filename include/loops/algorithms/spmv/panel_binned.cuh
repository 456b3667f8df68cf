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

#include <vector>

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
  int W, Hw, P, S, padded = 0, num_chunks = 0;
  vector_t<type_t> values, products;
  vector_t<unsigned short> col16, row16;
  vector_t<int> dst4, perm, segb, bstart, chunks, wins, wstart;

  /// @param subband_rows 0 = automatic (kernels::panel_subband_rows), or a power of two in [64, kernels::panel_subband_rows_max<type_t>()]
  explicit panel_binned_t(csr_t<index_t, offset_t, type_t>& csr, int subband_rows = 0, xpu::stream_t stream = 0)
      : rows(csr.rows), cols(csr.cols), nnzs(csr.nnzs) {
    W = kernels::panel_columns<type_t>(static_cast<int>(rows), static_cast<int>(cols), static_cast<int>(nnzs));
    P = cols ? static_cast<int>((cols + W - 1) / W) : 1;
    Hw = subband_rows ? subband_rows : kernels::panel_subband_rows<type_t>(static_cast<int>(rows), static_cast<int>(nnzs), P);
    error::throw_if_exception(Hw < 64 || Hw > kernels::panel_subband_rows_max<type_t>() || (Hw & (Hw - 1)),
                              "panel_binned_t: subband_rows must be a power of two in [64, panel_subband_rows_max<type_t>()]");
    S = rows ? static_cast<int>((rows + Hw - 1) / Hw) : 1;
    const long long segments = static_cast<long long>(P) * S;
    error::throw_if_exception(segments > (1ll << 26) || static_cast<long long>(nnzs) + 3 * segments >= (1ll << 31) - 4096,
                              "panel_binned_t: panels x sub-bands must stay below 2^26 and nnz + padding below 2^31");
    if (rows == 0) return;
    const std::size_t temp_bytes = kernels::panel_binned_temp_bytes(static_cast<int>(nnzs), segments);
    vector_t<char> temp(temp_bytes);
    vector_t<int> panel_start(static_cast<std::size_t>(P) + 1);
    const int* padded_dev = nullptr;
    error::throw_if_exception(
        kernels::build_panel_binned_stage1(stream, csr.offsets.data().get(), csr.indices.data().get(), static_cast<int>(rows),
                                           static_cast<int>(nnzs), W, Hw, P, S, temp.data().get(), temp_bytes, &padded_dev) != 0,
        "panel_binned_t: build (sizes) failed");
    (void)xpu::stream_synchronize(stream);
    error::throw_if_exception(hipMemcpy(&padded, padded_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess,
                              "panel_binned_t: cannot read the padded size");
    const std::size_t n = static_cast<std::size_t>(padded > 0 ? padded : 4);
    values = vector_t<type_t>(n);
    products = vector_t<type_t>(n);
    col16 = vector_t<unsigned short>(n);
    row16 = vector_t<unsigned short>(n);
    perm = vector_t<int>(n);
    dst4 = vector_t<int>(n / 4 + 1);
    segb = vector_t<int>(static_cast<std::size_t>(segments) + 1);
    bstart = vector_t<int>(static_cast<std::size_t>(S) + 1);
    wstart = vector_t<int>(static_cast<std::size_t>(S) + 1);
    wins = vector_t<int>(2 * kernels::panel_window_capacity(padded, segments));
    chunks = vector_t<int>(3);
    error::throw_if_exception(kernels::build_panel_binned_stage2<index_t, type_t>(stream, csr.indices.data().get(), csr.values.data().get(),
                                                                                  view(), temp.data().get(), temp_bytes, panel_start.data().get()) != 0,
                              "panel_binned_t: build (placement) failed");
    (void)xpu::stream_synchronize(stream);
    std::vector<int> ps(static_cast<std::size_t>(P) + 1);
    error::throw_if_exception(hipMemcpy(ps.data(), panel_start.data().get(), sizeof(int) * ps.size(), hipMemcpyDeviceToHost) != hipSuccess,
                              "panel_binned_t: cannot read the panel starts");
    const std::vector<int> list = kernels::panel_chunk_list(ps, P);  // kernel A's work list
    num_chunks = static_cast<int>(list.size() / 3);
    if (!list.empty()) chunks = vector_t<int>(list.begin(), list.end());
  }

  kernels::panel_binned_view<type_t> view() {
    return kernels::panel_binned_view<type_t>{static_cast<int>(rows), static_cast<int>(cols), static_cast<int>(nnzs), W, Hw, P, S, padded,
                                              values.data().get(), col16.data().get(), dst4.data().get(), row16.data().get(),
                                              perm.data().get(), segb.data().get(), bstart.data().get(), chunks.data().get(), num_chunks,
                                              products.data().get(), wins.data().get(), wstart.data().get()};
  }

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
