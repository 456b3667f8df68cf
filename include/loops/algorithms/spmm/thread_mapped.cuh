/**
 * @file thread_mapped.cuh
 * @brief `algorithms::spmm::thread_mapped(csr, B, C, stream)`: C = A * B with A in CSR and dense
 * row-major B, C; one thread per row of A through `schedule::setup<thread_mapped>`
 * (reference include/loops/algorithms/spmm/thread_mapped.cuh:28-90).
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

template <typename index_t, typename offset_t, typename type_t>
void thread_mapped(csr_t<index_t, offset_t, type_t>& csr, matrix_t<type_t>& B, matrix_t<type_t>& C,
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
