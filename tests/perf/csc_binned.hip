// Harness of tests/perf/exp_csc_binned.py: the binned one-shot CSC product next to the atomic kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DLOOPS_TARGET_GFX=0x950 -Iinclude tests/perf/csc_binned.hip -o build/variants/libcsc_binned.so
#include <hip/hip_runtime.h>
#include <loops/kernels/csc_spmv.hxx>
using namespace loops;
extern "C" long long csc_binned_bytes(int rows, int nnz) { return (long long)kernels::csc_binned_scratch_bytes<int, float>(rows, nnz); }
extern "C" int csc_binned(int rows, int cols, int nnz, const int* off, const int* ridx, const float* val, const float* x, float* y, void* scratch, void* st) {
  (void)hipMemsetAsync(y, 0, sizeof(float) * rows, static_cast<hipStream_t>(st));  // (the product ADDS to y, like the atomic kernels)
  return kernels::launch_csc_binned(static_cast<hipStream_t>(st), rows, cols, nnz, off, ridx, val, x, y, scratch);
}
extern "C" int csc_atomic(int rows, int cols, int nnz, const int* off, const int* ridx, const float* val, const float* x, float* y, void* st) {
  (void)hipMemsetAsync(y, 0, sizeof(float) * rows, static_cast<hipStream_t>(st));
  return kernels::launch_csc_nonzero_split(static_cast<hipStream_t>(st), cols, nnz, off, ridx, val, x, y);
}
