/**
 * @file merge_path_flat.cu
 * @brief CSR SpMM C = A * B on the merge-path schedule (algorithms::spmm::merge_path_flat), checked
 * against the per-thread loop the reference ships (algorithms::spmm::thread_mapped).
 *
 *   loops.spmm.merge_path_flat <matrix.mtx> [columns of B = 32]
 */
#include <cmath>
#include <cstdlib>
#include <iostream>

#include <loops/container/formats.hxx>
#include <loops/container/market.hxx>
#include <loops/util/generate.hxx>
#include <loops/algorithms/spmm/thread_mapped.cuh>
#include <loops/algorithms/spmm/merge_path_flat.cuh>

using namespace loops;

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "usage: " << argv[0] << " <matrix.mtx> [columns of B]" << std::endl;
    return 0;
  }
  const std::size_t n = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 32;
  matrix_market_t<int, int, float> mtx;
  csr_t<int, int, float> csr(mtx.load(argv[1]));
  matrix_t<float> B(csr.cols, n), C(csr.rows, n), C_ref(csr.rows, n);
  generate::random::uniform_distribution(B.m_data.begin(), B.m_data.end(), 1, 10);

  auto timer = algorithms::spmm::merge_path_flat(csr, B, C);
  util::timer_t ref_timer;
  ref_timer.start();
  algorithms::spmm::thread_mapped(csr, B, C_ref);
  ref_timer.stop();

  vector_t<float, memory_space_t::host> c(C.m_data), r(C_ref.m_data);
  std::size_t errors = 0;
  for (std::size_t i = 0; i < c.size(); ++i) errors += std::fabs(c[i] - r[i]) > 1e-4f * (1.f + std::fabs(r[i]));
  std::cout << "Elapsed (ms):\t" << timer.milliseconds() << std::endl;
  std::cout << "thread_mapped (ms):\t" << ref_timer.milliseconds() << std::endl;
  std::cout << "Matrix:\t\t" << csr.rows << " x " << csr.cols << ", " << csr.nnzs << " nnz; B: " << csr.cols << " x " << n << std::endl;
  std::cout << "Errors:\t\t" << errors << std::endl;
  return errors != 0;
}
