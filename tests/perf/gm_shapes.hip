// Harness of tests/perf/exp_group_mapped_shapes.py: group_mapped (heavy groups shared out) for several group sizes / tile shapes behind a C ABI.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DLOOPS_TARGET_GFX=0x950 -Iinclude tests/perf/gm_shapes.hip -o build/variants/libgm_shapes.so
#include <hip/hip_runtime.h>
#include <loops/kernels/launch.hxx>
#include <loops/kernels/group_mapped_spmv.hxx>
using namespace loops;
#define GM(NAME, TPB, IPT)                                                                                                              \
  extern "C" long long NAME##_bytes(int rows, int nnz) { return (long long)kernels::group_share_scratch_bytes<float>(rows, nnz, TPB, IPT); } \
  extern "C" int NAME(int rows, int cols, int nnz, const int* off, const int* idx, const float* val, const float* x, float* y, void* scratch, void* st) { \
    return kernels::launch_group_mapped_shared<TPB, IPT, (IPT % 2 == 0)>(static_cast<hipStream_t>(st), rows, nnz, off, idx, val, x, y, scratch);    \
  }
GM(gm_256x8, 256, 8)
GM(gm_512x4, 512, 4)
GM(gm_512x8, 512, 8)
GM(gm_1024x4, 1024, 4)
GM(gm_128x8, 128, 8)
