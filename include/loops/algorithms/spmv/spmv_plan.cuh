/**
 * @file spmv_plan.cuh
 * @brief `algorithms::spmv::spmv_plan_t<index_t, offset_t, type_t>`: what an iterative caller holds for ONE matrix -- the
 * merge-path plan of the unmodified CSR in the tile shape that suits its structure, or, if the caller allows a copy and it
 * is measurably faster, a re-ordered copy of the matrix: row-band (rowband.cuh: y accumulators in LDS, column-sorted gathers that
 * coalesce; 4-byte values) or panel-binned (panel_binned.cuh: x panels in LDS, no memory gather at all).  The
 * header-API twin of loops_spmv_plan_* (include/loops_amd.h).  The reference fixes the tile shape per architecture at
 * compile time (algorithms/spmv/launch_box.hxx:56-90) and always runs the CSR as given; no counterpart there.
 *
 *   algorithms::spmv::spmv_plan_t<int, int, float> plan(csr);            // measures 256 x 8, 512 x 8 (+ its phased-gather twin) and the copies
 *   for (...) plan.spmv_async(csr, x, y, stream);                         // y = csr * x with whatever won
 *
 * One product in flight per plan (it owns one set of carry-out / partial-result buffers).
 */
#pragma once

#include <exception>
#include <memory>

#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/algorithms/spmv/panel_binned.cuh>
#include <loops/algorithms/spmv/rowband.cuh>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename index_t, typename offset_t, typename type_t>
struct spmv_plan_t {
  enum layout_kind { csr_layout = 0, panel_binned_layout = 2, row_band_layout = 3 };  // (LOOPS_LAYOUT_* of include/loops_amd.h)
  using small_t = merge_path_small_plan_t<index_t, offset_t, type_t>;  // 256 x 8 (256 x 4 for 8-byte values)
  using large_t = merge_path_plan_t<index_t, offset_t, type_t>;        // 512 x 8 (512 x 4)
  using band_t = rowband_t<index_t, offset_t, type_t>;
  using panel_t = panel_binned_t<index_t, offset_t, type_t>;
  static constexpr std::size_t large_block = merge_path_launch_t<type_t>::block_size, large_items = merge_path_launch_t<type_t>::items_per_thread;

  layout_kind layout = csr_layout;
  bool phased = false;  ///< csr_layout over `large`: the product runs the phased-gather kernel (merge_path_flat_phased_async_with)
  float ms_small = -1.f, ms_large = -1.f, ms_band = -1.f, ms_panel = -1.f;  ///< measured ms per product (-1: not timed)
  float ms_phased = -1.f;
  std::unique_ptr<small_t> small;
  std::unique_ptr<large_t> large;
  std::unique_ptr<band_t> band;
  std::unique_ptr<panel_t> panel;

  /// @param allow_copy the plan may keep a re-ordered copy of `csr` (adopted when >= 5 % faster than the best CSR shape;
  ///        without `measure`: panel-binned when cols * sizeof(type_t) > 6 MB, row-band from 2 MB under a mean row of >= 8 nonzeros)
  /// @param measure time the candidates (`repeats` products each) instead of choosing by structure alone
  /// @param deterministic only layouts whose summation order is fixed: no row-band copy (LOOPS_PLAN_DETERMINISTIC of the C ABI)
  explicit spmv_plan_t(csr_t<index_t, offset_t, type_t>& csr, bool allow_copy = true, bool measure = true, int repeats = 10,
                       xpu::stream_t stream = 0, bool deterministic = false) {
    const typename small_t::layout_t lay(csr.offsets.data().get(), static_cast<index_t>(csr.rows), static_cast<offset_t>(csr.nnzs));
    small = std::make_unique<small_t>(lay, stream, small_t::prepass_always);
    const bool short_rows = small->classify(stream) || small->merge_tiles() <= 1;
    const bool work = csr.rows > 0 && csr.nnzs > 0;
    const std::size_t x_bytes = csr.cols * sizeof(type_t);
    if (!measure || !work) {
      if (!short_rows) {
        large = std::make_unique<large_t>(lay, stream, large_t::prepass_always);
        large->classify(stream);
        small.reset();
      }
      if (allow_copy && work && x_bytes >= (std::size_t(2) << 20)) {   // (the rule of loops_spmv_plan_create_* without MEASURE)
        // (the copy is OPTIONAL: a build that fails -- no memory for it -- leaves the plan on the CSR, as the measured path does)
        try {
          if (x_bytes > (std::size_t(6) << 20)) {
            if (fits_panel(csr)) {
              panel = std::make_unique<panel_t>(csr, 0, stream);
              layout = panel_binned_layout;
            }
          } else if (!deterministic && csr.nnzs / csr.rows >= 8 && band_t::fits(csr)) {
            band = std::make_unique<band_t>(csr, 0, 0, stream);
            layout = row_band_layout;
          }
        } catch (const error::bad_argument_t&) {
          throw;  // (the caller's mistake -- a column outside the matrix --, not a missing resource: LOOPS_E_BADARG of the C ABI)
        } catch (const std::exception&) {
          (void)hipGetLastError();  // clear the sticky error of the failed allocation
          panel.reset();
          band.reset();
          layout = csr_layout;
        }
        if (layout != csr_layout) {
          small.reset();
          large.reset();
        }
      }
      // still on the CSR: columns that look scattered over an x of 3 MB or more get 512 x 8 tiles with phased x gathers (the
      // structural guess of kernels::columns_look_scattered, as loops_spmv_plan_create_* without MEASURE)
      if constexpr (large_block == 512 && large_items == 8) {
        if (layout == csr_layout && work &&
            kernels::columns_worth_sampling(static_cast<long long>(csr.nnzs), static_cast<long long>(csr.cols), static_cast<int>(sizeof(type_t)))) {
          vector_t<unsigned int> scratch(kernels::scatter_scratch_words);
          if (kernels::columns_look_scattered(stream, csr.indices.data().get(), static_cast<long long>(csr.nnzs), static_cast<long long>(csr.cols),
                                              static_cast<int>(sizeof(type_t)), scratch.data().get())) {
            if (!large) {
              large = std::make_unique<large_t>(lay, stream, large_t::prepass_always);
              large->classify(stream);
            }
            if (large->merge_tiles() > 1) {
              small.reset();
              phased = true;
            } else if (small) {
              large.reset();
            }
          }
        }
      }
      return;
    }
    vector_t<type_t> x(csr.cols), y(csr.rows);
    large = std::make_unique<large_t>(lay, stream, large_t::prepass_always);
    large->classify(stream);
    ms_small = time_ms(repeats, stream, [&] {
      merge_path_flat_async_with<launch_t<type_t>::block_size, launch_t<type_t>::items_per_thread>(*small, csr, x, y, stream, true);
    });
    ms_large = time_ms(repeats, stream, [&] {
      merge_path_flat_async_with<merge_path_launch_t<type_t>::block_size, merge_path_launch_t<type_t>::items_per_thread>(*large, csr, x, y, stream, true);
    });
    // (a shape must be measurably -- > 1 % -- faster to displace the structural choice)
    const bool keep_small = ms_small < 0.99f * ms_large || (short_rows && ms_small <= 1.01f * ms_large);
    float best = keep_small ? ms_small : ms_large;
    // the same CSR with PHASED x gathers (no copy, same bits): a candidate where it can pay at all -- an x of at least a quarter
    // of one XCD's L2 -- adopted when > 2 % faster than the best plain shape (as loops_spmv_plan_create_*)
    if constexpr (large_block == 512 && large_items == 8) {
      if (large->merge_tiles() > 1 && x_bytes >= (std::size_t(1) << 20)) {
        ms_phased = time_ms(repeats, stream, [&] { merge_path_flat_phased_async_with<512, 8>(*large, csr, x, y, stream, true); });
        if (ms_phased < 0.98f * best) {
          phased = true;
          best = ms_phased;
        }
      }
    }
    if (keep_small && !phased) large.reset();
    else small.reset();
    {
      if (allow_copy && !deterministic && x_bytes >= (std::size_t(1) << 20) && band_t::fits(csr)) {  // second candidate: the row-band copy, its shape tuned
        auto rb = std::make_unique<band_t>(csr, 0, 0, stream);
        rb->tune(repeats, stream);
        ms_band = time_ms(repeats, stream, [&] { rb->spmv_async(x, y, stream); });
        if (ms_band < 0.95f * best) {
          band = std::move(rb);
          layout = row_band_layout;
          phased = false;
          small.reset();
          large.reset();
        }
      }
    }
    if (allow_copy && x_bytes >= (std::size_t(2) << 20) && fits_panel(csr)) {  // third candidate: x panels in LDS, no gather
      auto pb = std::make_unique<panel_t>(csr, 0, stream);
      ms_panel = time_ms(repeats, stream, [&] { pb->spmv_async(x, y, stream); });
      const float incumbent = band ? ms_band : best;
      if (ms_panel < 0.95f * incumbent && ms_panel < 0.95f * best) {
        panel = std::move(pb);
        layout = panel_binned_layout;
        phased = false;
        small.reset();
        large.reset();
        band.reset();
      }
    }
  }

  /// y = csr * x on `stream` (asynchronous).  `csr` must be the matrix the plan was built from.
  void spmv_async(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    if (panel) panel->spmv_async(x, y, stream);
    else if (band) band->spmv_async(x, y, stream);
    else if (small)
      merge_path_flat_async_with<launch_t<type_t>::block_size, launch_t<type_t>::items_per_thread>(*small, csr, x, y, stream, true);
    else if (phased) {
      if constexpr (large_block == 512 && large_items == 8) merge_path_flat_phased_async_with<512, 8>(*large, csr, x, y, stream, true);
    } else
      merge_path_flat_async_with<merge_path_launch_t<type_t>::block_size, merge_path_launch_t<type_t>::items_per_thread>(*large, csr, x, y, stream, true);
  }

  util::timer_t spmv(csr_t<index_t, offset_t, type_t>& csr, vector_t<type_t>& x, vector_t<type_t>& y, xpu::stream_t stream = 0) {
    util::timer_t timer(stream);
    timer.start();
    spmv_async(csr, x, y, stream);
    (void)xpu::stream_synchronize(stream);
    timer.stop();
    return timer;
  }

 private:
  static bool fits_panel(const csr_t<index_t, offset_t, type_t>& csr) {
    const int w = kernels::panel_columns<type_t>(static_cast<int>(csr.rows), static_cast<int>(csr.cols), static_cast<int>(csr.nnzs));
    const long long P = (static_cast<long long>(csr.cols) + w - 1) / w;
    const int hw = kernels::panel_subband_rows<type_t>(static_cast<int>(csr.rows), static_cast<int>(csr.nnzs), static_cast<int>(P > 0 ? P : 1));
    const long long segments = (P > 0 ? P : 1) * ((static_cast<long long>(csr.rows) + hw - 1) / hw);
    return segments <= (1ll << 26) && static_cast<long long>(csr.nnzs) + 3 * segments < (1ll << 31) - 4096;
  }
  template <typename fn_t>
  static float time_ms(int repeats, xpu::stream_t stream, fn_t&& run) {
    if (repeats < 1) repeats = 10;
    run();
    run();
    util::timer_t timer(stream);
    timer.start();
    for (int i = 0; i < repeats; ++i) run();
    return timer.stop() / static_cast<float>(repeats);
  }
};

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
