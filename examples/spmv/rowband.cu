/**
 * @file rowband.cu
 * @brief SpMV over the row-band copy of a CSR (algorithms::spmv::rowband_t: the y accumulators of a band of rows in LDS, the
 * band's nonzeros sorted by column) checked against merge_path_flat on the CSR.
 *
 *   loops.spmv.rowband <matrix.mtx> [rows per band, 0 = automatic] [target chunks, 0 = automatic]
 */
#include <cmath>
#include <cstdlib>
#include <iostream>

#include <loops/container/formats.hxx>
#include <loops/container/market.hxx>
#include <loops/util/generate.hxx>
#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/algorithms/spmv/rowband.cuh>

using namespace loops;

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "usage: " << argv[0] << " <matrix.mtx> [band rows] [target chunks]" << std::endl;
    return 0;
  }
  const int band_rows = argc > 2 ? std::atoi(argv[2]) : 0, chunks = argc > 3 ? std::atoi(argv[3]) : 0;
  matrix_market_t<int, int, float> mtx;
  csr_t<int, int, float> csr(mtx.load(argv[1]));
  vector_t<float> x(csr.cols), y(csr.rows), y_ref(csr.rows);
  generate::random::uniform_distribution(x.begin(), x.end(), 1, 10);

  algorithms::spmv::rowband_t<int, int, float> banded(csr, band_rows, chunks);
  banded.tune(5);
  auto timer = banded.spmv(x, y);
  auto ref_timer = algorithms::spmv::merge_path_flat(csr, x, y_ref);

  vector_t<float, memory_space_t::host> a(y), b(y_ref);
  std::size_t errors = 0;
  for (std::size_t i = 0; i < a.size(); ++i) errors += std::fabs(a[i] - b[i]) > 1e-4f * (1.f + std::fabs(b[i]));
  std::cout << "Elapsed (ms):\t" << timer.milliseconds() << std::endl;
  std::cout << "merge_path_flat on the CSR (ms):\t" << ref_timer.milliseconds() << std::endl;
  std::cout << "Bands:\t\t" << banded.arrays.B << " of " << banded.arrays.H << " rows, " << banded.arrays.num_chunks << " chunks, "
            << banded.arrays.waves << " wavefronts" << std::endl;
  std::cout << "Errors:\t\t" << errors << std::endl;
  return errors != 0;
}
