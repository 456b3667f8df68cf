// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" surface over the *real* gunrock/loops host code, compiled from the
// reference headers where they lie under /root/reference/include (nothing is copied
// into this repository).  Output goes to oracle/_ref/libloops_ref.so (git-ignored, but
// shipped to the GPU box like any other built .so).  Used to
//   * pin oracle/loops_oracle.c (the CPU restatement) against the reference itself,
//   * generate tests/golden/ fixtures (tests/golden/make_golden.py),
//   * optionally serve as bench.py's cpu_baseline (kind "reference").
//
// Reference entry points exercised (file:line under /root/reference):
//   matrix_market_t::load            include/loops/container/market.hxx:100-177
//   csr_t(coo_t) COO->CSR            include/loops/container/csr.hxx:86-94
//   generate::random::uniform_distribution   include/loops/util/generate.hxx:54-79
//   reference::spmv / spmv_f64 / row_l1_products   include/loops/util/reference.hxx:57-198
//   reference::default_tolerance::ne include/loops/util/reference.hxx:115-131
//   bcsr_t(csr_t) host builder     include/loops/container/bcsr.hxx:111-194
//   layout::csr / flat_uniform_occupancy   include/loops/container/layout.hxx:88-149,
//                                           include/loops/container/partitioning.hxx:72-141
#include <loops/container/formats.hxx>
#include <loops/container/market.hxx>
#include <loops/container/layout.hxx>
#include <loops/util/generate.hxx>
#include <loops/util/reference.hxx>
#include <loops/util/math.hxx>
#include <loops/container/bcsr.hxx>

#include <cstdlib>
#include <cstring>

using namespace loops;
using host_csr_f32 = csr_t<int, int, float, memory_space_t::host>;
using host_vec_f32 = vector_t<float, memory_space_t::host>;
using host_csr_f64 = csr_t<int, int, double, memory_space_t::host>;
using host_vec_f64 = vector_t<double, memory_space_t::host>;

template <typename csr_type, typename T>
static csr_type make_csr(long rows, long cols, long nnz, const int* off, const int* idx, const T* val) {
  csr_type c(rows, cols, nnz);
  std::copy(off, off + rows + 1, c.offsets.begin());
  std::copy(idx, idx + nnz, c.indices.begin());
  std::copy(val, val + nnz, c.values.begin());
  return c;
}

template <std::size_t K>
static int flat_q(const int* off, int ntiles, int natoms, int what, int arg) {
  layout::flat_uniform_occupancy<K, layout::csr<int, int>> L(layout::csr<int, int>(off, ntiles, natoms));
  switch (what) {
    case 0: return L.tile_begin(arg);
    case 1: return L.tile_end(arg);
    case 2: return L.tile_size(arg);
    case 3: return L.tile_end_iter()[arg];
    case 4: return L.tile_of(arg);
    case 5: return L.num_tiles();
    case 7: return L.base().tile_of(arg);
    default: return L.num_atoms();
  }
}
template <std::size_t R, std::size_t C>
static long bcsr_build(long rows, long cols, long nnz, const int* off, const int* idx, const float* val,
                       int* block_offsets, int* block_cols, float* block_values) {
  auto csr = make_csr<host_csr_f32>(rows, cols, nnz, off, idx, val);
  bcsr_t<R, C, int, int, float, memory_space_t::host> b(csr);
  std::copy(b.block_offsets.begin(), b.block_offsets.end(), block_offsets);
  if (block_cols) {
    std::copy(b.block_col_indices.begin(), b.block_col_indices.end(), block_cols);
    std::copy(b.values.begin(), b.values.end(), block_values);
  }
  return (long)b.num_blocks;
}
extern "C" {

void ref_free(void* p) { std::free(p); }

// Load a Matrix-Market file through the reference loader and COO->CSR conversion.
int ref_mtx_load_csr_f32(const char* path, long* rows, long* cols, long* nnz,
                         int** offsets, int** indices, float** values) {
  try {
    matrix_market_t<int, int, float> mtx;
    host_csr_f32 csr(mtx.load(path));
    *rows = csr.rows; *cols = csr.cols; *nnz = csr.nnzs;
    *offsets = (int*)std::malloc(sizeof(int) * (csr.rows + 1));
    *indices = (int*)std::malloc(sizeof(int) * (csr.nnzs ? csr.nnzs : 1));
    *values = (float*)std::malloc(sizeof(float) * (csr.nnzs ? csr.nnzs : 1));
    std::copy(csr.offsets.begin(), csr.offsets.end(), *offsets);
    std::copy(csr.indices.begin(), csr.indices.end(), *indices);
    std::copy(csr.values.begin(), csr.values.end(), *values);
    return 0;
  } catch (...) { return 1; }
}

// The examples' x generator with int bounds (examples/spmv/merge_path.cu:33).
void ref_xgen_int_f32(long n, int lo, int hi, unsigned seed, float* out) {
  host_vec_f32 x(n);
  generate::random::uniform_distribution(x.begin(), x.end(), lo, hi, seed);
  std::copy(x.begin(), x.end(), out);
}
void ref_xgen_int_f64(long n, int lo, int hi, unsigned seed, double* out) {
  host_vec_f64 x(n);
  generate::random::uniform_distribution(x.begin(), x.end(), lo, hi, seed);
  std::copy(x.begin(), x.end(), out);
}
unsigned ref_hash(unsigned a) { return generate::random::hash(a); }

void ref_spmv_f32(long rows, long cols, long nnz, const int* off, const int* idx,
                  const float* val, const float* x, float* y) {
  auto csr = make_csr<host_csr_f32>(rows, cols, nnz, off, idx, val);
  host_vec_f32 xv(x, x + cols);
  auto yv = reference::spmv(csr, xv);
  std::copy(yv.begin(), yv.end(), y);
}
void ref_spmv_f64(long rows, long cols, long nnz, const int* off, const int* idx,
                  const double* val, const double* x, double* y) {
  auto csr = make_csr<host_csr_f64>(rows, cols, nnz, off, idx, val);
  host_vec_f64 xv(x, x + cols);
  auto yv = reference::spmv(csr, xv);
  std::copy(yv.begin(), yv.end(), y);
}
void ref_spmv_f64acc_f32(long rows, long cols, long nnz, const int* off, const int* idx,
                         const float* val, const float* x, float* y) {
  auto csr = make_csr<host_csr_f32>(rows, cols, nnz, off, idx, val);
  host_vec_f32 xv(x, x + cols);
  auto yv = reference::spmv_f64(csr, xv);
  std::copy(yv.begin(), yv.end(), y);
}
void ref_row_l1_f32(long rows, long cols, long nnz, const int* off, const int* idx,
                    const float* val, const float* x, float* l1) {
  auto csr = make_csr<host_csr_f32>(rows, cols, nnz, off, idx, val);
  host_vec_f32 xv(x, x + cols);
  auto lv = reference::row_l1_products(csr, xv);
  std::copy(lv.begin(), lv.end(), l1);
}
int ref_default_ne_f32(float a, float b) { return reference::default_tolerance<float>::ne(a, b); }
long ref_ceil_div(long n, long d) { return math::ceil_div(n, d); }

// Layout-view contract, evaluated on the host by the reference's own PODs.
// what: 0 tile_begin, 1 tile_end, 2 tile_size, 3 tile_end_iter()[k], 4 tile_of
int ref_layout_csr(const int* off, int ntiles, int natoms, int what, int arg) {
  layout::csr<int, int> L(off, ntiles, natoms);
  switch (what) {
    case 0: return L.tile_begin(arg);
    case 1: return L.tile_end(arg);
    case 2: return L.tile_size(arg);
    case 3: return L.tile_end_iter()[arg];
    case 4: return L.tile_of(arg);
    case 5: return L.num_tiles();
    default: return L.num_atoms();
  }
}
int ref_layout_flat(int K, const int* off, int ntiles, int natoms, int what, int arg) {
  switch (K) {
    case 2: return flat_q<2>(off, ntiles, natoms, what, arg);
    case 4: return flat_q<4>(off, ntiles, natoms, what, arg);
    case 8: return flat_q<8>(off, ntiles, natoms, what, arg);
    case 16: return flat_q<16>(off, ntiles, natoms, what, arg);
    default: return -1;
  }
}
int ref_layout_ell(int ntiles, int pitch, int what, int arg) {
  layout::ell<int, int> L(ntiles, pitch);
  switch (what) {
    case 0: return L.tile_begin(arg);
    case 1: return L.tile_end(arg);
    case 2: return L.tile_size(arg);
    case 3: return L.tile_end_iter()[arg];
    case 4: return L.tile_of(arg);
    case 5: return L.num_tiles();
    default: return L.num_atoms();
  }
}
int ref_layout_coo(int nnz, int what, int arg) {
  layout::coo<int, int> L(nnz);
  switch (what) {
    case 0: return L.tile_begin(arg);
    case 1: return L.tile_end(arg);
    case 2: return L.tile_size(arg);
    case 3: return L.tile_end_iter()[arg];
    case 4: return L.tile_of(arg);
    case 5: return L.num_tiles();
    default: return L.num_atoms();
  }
}

// Two-call protocol like oracle_csr_to_bcsr_f32 (block_cols == NULL: count only).
long ref_csr_to_bcsr_f32(int R, int C, long rows, long cols, long nnz, const int* off, const int* idx,
                         const float* val, int* block_offsets, int* block_cols, float* block_values) {
  if (R == 2 && C == 2) return bcsr_build<2, 2>(rows, cols, nnz, off, idx, val, block_offsets, block_cols, block_values);
  if (R == 3 && C == 3) return bcsr_build<3, 3>(rows, cols, nnz, off, idx, val, block_offsets, block_cols, block_values);
  if (R == 4 && C == 4) return bcsr_build<4, 4>(rows, cols, nnz, off, idx, val, block_offsets, block_cols, block_values);
  if (R == 2 && C == 4) return bcsr_build<2, 4>(rows, cols, nnz, off, idx, val, block_offsets, block_cols, block_values);
  return -1;
}

}  // extern "C"
