/**
 * @file reference.hxx
 * @brief `loops::reference` -- the host-side validation utilities the example drivers call for
 * `--validate` / `--rigorous`: a plain CSR SpMV (value_t accumulation, index order), an
 * f64-accumulated variant, per-row L1 mass, the default tolerance predicate, the Wilkinson-style
 * rigorous validator and a mismatch counter.
 *
 * Same API and arithmetic as the reference's include/loops/util/reference.hxx (spmv :57-76,
 * default_tolerance :115-131, spmv_f64 :146-166, row_l1_products :178-198, unit_roundoff
 * :203-214, rigorous_report :226-243, rigorously_validate_spmv :274-337, count_errors :357-388).
 * This is a library feature (checking a GPU result on the host), not a compute fallback: no
 * `algorithms::` entry point ever routes here.
 */
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <vector>

#include <loops/backend/xpu.hxx>
#include <loops/container/csr.hxx>
#include <loops/container/vector.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace reference {
using namespace memory;

namespace detail {
/// y[r] = (out_t) sum_k f(values[k], x[indices[k]]) accumulated in acc_t, rows in order.
template <typename acc_t, typename index_t, typename offset_t, typename value_t, memory_space_t space, typename fn_t>
vector_t<value_t, memory_space_t::host> row_reduce(const csr_t<index_t, offset_t, value_t, space>& csr,
                                                   const vector_t<value_t, space>& x, fn_t term) {
  csr_t<index_t, offset_t, value_t, memory_space_t::host> a(csr);
  vector_t<value_t, memory_space_t::host> xh(x);
  vector_t<value_t, memory_space_t::host> y(a.rows, value_t{0});
  for (std::size_t r = 0; r < a.rows; ++r) {
    acc_t acc = acc_t{0};
    for (auto k = a.offsets[r]; k < a.offsets[r + 1]; ++k) acc += term(a.values[k], xh[a.indices[k]]);
    y[r] = static_cast<value_t>(acc);
  }
  return y;
}
}  // namespace detail

/// Host CSR SpMV with value_t accumulation (what `--validate` compares against).
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
vector_t<value_t, memory_space_t::host> spmv(const csr_t<index_t, offset_t, value_t, space>& csr,
                                             const vector_t<value_t, space>& x) {
  return detail::row_reduce<value_t>(csr, x, [](value_t a, value_t b) { return a * b; });
}

/// Same loop accumulated in double, cast back to value_t.
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
vector_t<value_t, memory_space_t::host> spmv_f64(const csr_t<index_t, offset_t, value_t, space>& csr,
                                                 const vector_t<value_t, space>& x) {
  return detail::row_reduce<double>(csr, x,
                                    [](value_t a, value_t b) { return static_cast<double>(a) * static_cast<double>(b); });
}

/// sum_k |a_k x_k| per row (double accumulation): the scale of the row's rounding error.
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
vector_t<value_t, memory_space_t::host> row_l1_products(const csr_t<index_t, offset_t, value_t, space>& csr,
                                                        const vector_t<value_t, space>& x) {
  return detail::row_reduce<double>(
      csr, x, [](value_t a, value_t b) { return std::abs(static_cast<double>(a) * static_cast<double>(b)); });
}

template <typename value_t>
struct default_tolerance {
  static constexpr value_t atol() { return value_t{1e-2}; }
  static constexpr value_t rtol() { return value_t{1e-3}; }
  /// true when a and b differ by more than atol + rtol * |b|.
  static __host__ __device__ bool ne(value_t a, value_t b) {
    using std::abs;
    return abs(a - b) > atol() + rtol() * abs(b);
  }
};

template <typename value_t>
constexpr value_t unit_roundoff();
template <>
constexpr float unit_roundoff<float>() { return 5.96046447753906e-08f; }  // 2^-24
template <>
constexpr double unit_roundoff<double>() { return 1.1102230246251565e-16; }  // 2^-53

struct rigorous_report {
  std::size_t total_rows = 0;
  std::size_t naive_mismatches = 0;       ///< rows flagged by default_tolerance vs the f32 host result
  std::size_t f32_baseline_overruns = 0;  ///< rows where the host f32 result itself exceeds the bound
  std::size_t gpu_overruns = 0;           ///< rows where the device result exceeds the bound
  double max_gpu_abs_error = 0.0;
  double max_gpu_rel_error = 0.0;         ///< |y - y64| / max(|y64|, 1)
  double wilkinson_k = 0.0;
};

/// Per-row bound max(atol_floor, K * nnz_row * u * L1_row) against the f64-accumulated result.
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
rigorous_report rigorously_validate_spmv(const csr_t<index_t, offset_t, value_t, space>& csr,
                                         const vector_t<value_t, space>& x, const value_t* d_y_gpu,
                                         double wilkinson_k = 8.0, double atol_floor = 1e-3, bool verbose = false) {
  csr_t<index_t, offset_t, value_t, memory_space_t::host> a(csr);
  vector_t<value_t, memory_space_t::host> xh(x);
  const auto y32 = spmv(a, xh);
  const auto y64 = spmv_f64(a, xh);
  const auto l1 = row_l1_products(a, xh);
  std::vector<value_t> y_gpu(a.rows);
  (void)xpu::memcpy(y_gpu.data(), d_y_gpu, a.rows * sizeof(value_t), xpu::memcpy_device_to_host);

  rigorous_report rep;
  rep.total_rows = a.rows;
  rep.wilkinson_k = wilkinson_k;
  const double u = static_cast<double>(unit_roundoff<value_t>());
  for (std::size_t r = 0; r < a.rows; ++r) {
    const std::size_t nnz_r = static_cast<std::size_t>(a.offsets[r + 1] - a.offsets[r]);
    const double bound = std::max(atol_floor, wilkinson_k * static_cast<double>(nnz_r) * u * static_cast<double>(l1[r]));
    const double ref = static_cast<double>(y64[r]);
    const double err32 = std::abs(static_cast<double>(y32[r]) - ref);
    const double err = std::abs(static_cast<double>(y_gpu[r]) - ref);
    if (default_tolerance<value_t>::ne(y_gpu[r], y32[r])) ++rep.naive_mismatches;
    if (err32 > bound) ++rep.f32_baseline_overruns;
    if (err > bound) {
      ++rep.gpu_overruns;
      if (verbose)
        std::printf("GPU_OVERRUN row=%zu nnz=%zu L1=%.6g y_gpu=%.8g y_f64=%.8g abs_err=%.6g bound=%.6g\n", r, nnz_r,
                    static_cast<double>(l1[r]), static_cast<double>(y_gpu[r]), ref, err, bound);
    }
    rep.max_gpu_abs_error = std::max(rep.max_gpu_abs_error, err);
    rep.max_gpu_rel_error = std::max(rep.max_gpu_rel_error, err / std::max(std::abs(ref), 1.0));
  }
  return rep;
}

/// Number of i with ne(d_y[i], h_ref[i]); d_y is a DEVICE pointer, h_ref a host pointer.
template <typename value_t, typename ne_t>
std::size_t count_errors(const value_t* d_y, const value_t* h_ref, std::size_t n, ne_t ne, bool verbose = false) {
  std::vector<value_t> y(n);
  (void)xpu::memcpy(y.data(), d_y, n * sizeof(value_t), xpu::memcpy_device_to_host);
  std::size_t errors = 0;
  for (std::size_t i = 0; i < n; ++i) {
    if (ne(y[i], h_ref[i])) {
      if (verbose) std::printf("Error[%zu]: %.8g != %.8g\n", i, static_cast<double>(y[i]), static_cast<double>(h_ref[i]));
      ++errors;
    }
  }
  return errors;
}

template <typename value_t>
std::size_t count_errors(const value_t* d_y, const value_t* h_ref, std::size_t n, bool verbose = false) {
  return count_errors(d_y, h_ref, n, [](value_t a, value_t b) { return default_tolerance<value_t>::ne(a, b); }, verbose);
}

}  // namespace reference
}  // namespace loops
