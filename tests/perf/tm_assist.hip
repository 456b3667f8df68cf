// Harness of tests/perf/exp_thread_mapped_assisted.py: the tuned thread_mapped launch, the assisted kernel instantiated directly (fp32, fp64)
// and the reference-shaped loop behind a C ABI.  Build (dev container, no GPU needed):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DLOOPS_TARGET_GFX=0x950 -Iinclude tests/perf/tm_assist.hip -o build/variants/libtm_assist.so
#include <hip/hip_runtime.h>
#include <loops/kernels/csr_spmv.hxx>
#include <loops/kernels/launch.hxx>
using namespace loops;
extern "C" int tm_old(int rows, int cols, int nnz, const int* off, const int* idx, const float* val, const float* x, float* y, void* st) {
  return kernels::launch_thread_mapped(static_cast<hipStream_t>(st), std::size_t(rows), std::size_t(cols), std::size_t(nnz), off, idx, val, x, y);
}
extern "C" int tm_new(int rows, int cols, int nnz, const int* off, const int* idx, const float* val, const float* x, float* y, void* st) {
  hipLaunchKernelGGL((kernels::thread_mapped_assisted_spmv<256, int, int, float>), dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(st), rows, off, idx, val, x, y);
  return (int)hipGetLastError();
}
extern "C" int tm_new_f64(int rows, int cols, int nnz, const int* off, const int* idx, const double* val, const double* x, double* y, void* st) {
  hipLaunchKernelGGL((kernels::thread_mapped_assisted_spmv<256, int, int, double>), dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(st), rows, off, idx, val, x, y);
  return (int)hipGetLastError();
}
extern "C" int tm_ref(int rows, int cols, int nnz, const int* off, const int* idx, const float* val, const float* x, float* y, void* st) {
  return kernels::launch_thread_mapped(static_cast<hipStream_t>(st), std::size_t(rows), std::size_t(cols), std::size_t(nnz), off, idx, val, x, y, true);
}
