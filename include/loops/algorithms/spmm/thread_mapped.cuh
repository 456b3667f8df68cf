/**
 * @file thread_mapped.cuh
 * @brief `algorithms::spmm::thread_mapped(csr, B, C, stream)`: C = A * B with A in CSR and dense
 * row-major B, C (reference include/loops/algorithms/spmm/thread_mapped.cuh:68-94: same signature, C overwritten, the call
 * returns after the stream has drained).  Since round 4 the drop-in entry runs the merge-path SpMM
 * (loops/kernels/merge_path_spmm.hxx: merge tiles over rows and nonzeros, B rows streamed 16 bytes per lane; C2, 8 columns:
 * 0.29 ms where the per-thread loop needs 44 ms).  The reference's kernel -- one thread per row through
 * `schedule::setup<thread_mapped>`, reference :28-52 -- stays as `__thread_mapped` for user code and behind
 * `thread_mapped_schedule_api`, the executable statement of what the schedule API computes.
 */
#pragma once

#include <loops/schedule.hxx>
#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/device.hxx>
#include <loops/util/math.hxx>
#include <loops/util/timer.hxx>
#include <loops/algorithms/spmv/launch_box.hxx>
#include <loops/memory.hxx>
#include <loops/container/matrix.cuh>
#include <loops/algorithms/spmm/merge_path_flat.cuh>

namespace loops {
namespace algorithms {
namespace spmm {

template <typename setup_t, typename index_t, typename offset_t, typename type_t>
__global__ void __thread_mapped(setup_t config, const std::size_t a_rows, const std::size_t a_cols,
                                const std::size_t a_nnz, const offset_t* offsets, const index_t* indices,
                                const type_t* values, const matrix_t<type_t> B, matrix_t<type_t> C) {
  for (auto row : config.tiles()) {
    for (auto col : custom_stride_range(std::size_t(0), B.cols, std::size_t(1))) {
      type_t sum = 0;
      for (auto nz : config.atoms(row)) sum += values[nz] * B(indices[nz], col);
      C(row, col) = sum;
    }
  }
}

/// The drop-in entry: the merge-path SpMM (plan + carry-out scratch built per call, as the reference's callers expect a
/// self-contained call); synchronises the stream like the reference.
template <typename index_t, typename offset_t, typename type_t>
void thread_mapped(csr_t<index_t, offset_t, type_t>& csr, matrix_t<type_t>& B, matrix_t<type_t>& C,
                   xpu::stream_t stream = 0) {
  if (csr.rows == 0 || B.cols == 0) {
    (void)xpu::stream_synchronize(stream);
    return;
  }
  (void)merge_path_flat(csr, B, C, stream);
}

/// The reference's own shape -- one thread per row of A, a loop over B's columns -- through the public schedule API.
template <typename index_t, typename offset_t, typename type_t>
void thread_mapped_schedule_api(csr_t<index_t, offset_t, type_t>& csr, matrix_t<type_t>& B, matrix_t<type_t>& C,
                                xpu::stream_t stream = 0) {
  constexpr std::size_t block_size = 128;
  using setup_t = schedule::setup<schedule::algorithms_t::thread_mapped, 1, 1, index_t, offset_t>;
  setup_t config(csr.offsets.data().get(), csr.rows, csr.nnzs);
  if (csr.rows > 0)
    launch::non_cooperative(stream, __thread_mapped<setup_t, index_t, offset_t, type_t>,
                            dim3(static_cast<unsigned>(math::ceil_div(csr.rows, block_size))), dim3(block_size), config,
                            csr.rows, csr.cols, csr.nnzs, csr.offsets.data().get(), csr.indices.data().get(),
                            csr.values.data().get(), B, C);
  (void)xpu::stream_synchronize(stream);
}

}  // namespace spmm
}  // namespace algorithms
}  // namespace loops
