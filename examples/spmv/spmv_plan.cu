/**
 * @file spmv_plan.cu
 * @brief What an iterative caller does with this library: build ONE plan for the matrix (algorithms::spmv::spmv_plan_t picks
 * the merge-tile shape and, if a copy is allowed, the layout -- unmodified CSR, row-band or panel-binned -- by timing
 * the candidates on the device), then run many products through it.  Checked against merge_path_flat on the CSR.
 *
 *   loops.spmv.spmv_plan <matrix.mtx> [iterations = 20] [allow a re-ordered copy = 1] [measure = 1]
 */
#include <cmath>
#include <cstdlib>
#include <iostream>

#include <loops/container/formats.hxx>
#include <loops/container/market.hxx>
#include <loops/util/generate.hxx>
#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/algorithms/spmv/spmv_plan.cuh>

using namespace loops;

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "usage: " << argv[0] << " <matrix.mtx> [iterations] [allow_copy] [measure]" << std::endl;
    return 0;
  }
  const int iterations = argc > 2 ? std::atoi(argv[2]) : 20;
  const bool allow_copy = argc > 3 ? std::atoi(argv[3]) != 0 : true;
  const bool measure = argc > 4 ? std::atoi(argv[4]) != 0 : true;
  matrix_market_t<int, int, float> mtx;
  csr_t<int, int, float> csr(mtx.load(argv[1]));
  vector_t<float> x(csr.cols), y(csr.rows), y_ref(csr.rows);
  generate::random::uniform_distribution(x.begin(), x.end(), 1, 10);

  using plan_t = algorithms::spmv::spmv_plan_t<int, int, float>;
  plan_t plan(csr, allow_copy, measure);
  util::timer_t timer;
  timer.start();
  for (int it = 0; it < iterations; ++it) plan.spmv_async(csr, x, y);
  (void)xpu::stream_synchronize(0);
  const float total_ms = timer.stop();
  (void)algorithms::spmv::merge_path_flat(csr, x, y_ref);

  vector_t<float, memory_space_t::host> a(y), b(y_ref);
  std::size_t errors = 0;
  for (std::size_t i = 0; i < a.size(); ++i) errors += std::fabs(a[i] - b[i]) > 1e-4f * (1.f + std::fabs(b[i]));
  const char* layouts[] = {"unmodified CSR", "-", "panel-binned copy", "row-band copy"};
  std::cout << "Layout:\t\t" << layouts[plan.layout] << (plan.layout == plan_t::csr_layout ? (plan.small ? ", 256 x 8 tiles" : plan.phased ? ", 512 x 8 tiles, phased x gathers" : ", 512 x 8 tiles") : "")
            << std::endl;
  std::cout << "Measured (ms):\tcsr 256x8 " << plan.ms_small << ", csr 512x8 " << plan.ms_large << ", csr 512x8 phased " << plan.ms_phased
            << ", row-band " << plan.ms_band
            << ", panel-binned " << plan.ms_panel << "  (-1 = not a candidate)" << std::endl;
  std::cout << "Elapsed (ms):\t" << total_ms / static_cast<float>(iterations > 0 ? iterations : 1) << " per product, " << iterations
            << " products" << std::endl;
  std::cout << "Errors:\t\t" << errors << std::endl;
  return errors != 0;
}
