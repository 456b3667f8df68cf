// loops_c_abi.hip -- extern "C" surface of libloops_amd.so (declared in include/loops_amd.h).
//
// Thin wrappers: every entry point instantiates the C++ templates of include/loops/ for
// index_t = offset_t = int, type_t = float | double and launches them on the caller's
// stream.  No CPU fallbacks live here: if a kernel cannot be launched the hipError_t is
// returned.  Built by __graft_entry__.build() with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude loops_c_abi.hip
#include <loops_amd.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <new>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>

#include <loops/schedule.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/launch_box.hxx>
#include <loops/util/math.hxx>
#include <loops/kernels/launch.hxx>
#include <loops/kernels/group_mapped_spmv.hxx>
#include <loops/kernels/panel_binned.hxx>
#include <loops/kernels/rowband.hxx>
#include <loops/multi_gpu/partition.hxx>
#include <loops/kernels/coo_spmv.hxx>
#include <loops/kernels/ell_spmv.hxx>
#include <loops/kernels/dia_spmv.hxx>
#include <loops/kernels/csc_spmv.hxx>
#include <loops/kernels/bcsr_spmv.hxx>
#include <loops/kernels/bcsr_band.hxx>
#include <loops/kernels/bcsr_merge_path.hxx>

using namespace loops;
using kernels::coord_t;

namespace {

constexpr int kSpmvBlock = 256;  // launch_t<T>::block_size on gfx950 (launch_box.hxx:75-77)

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }
inline int last_error() { return static_cast<int>(hipGetLastError()); }

struct tile_shape {
  int tpb, ipt;
};
inline bool shape_of(int cfg, tile_shape* s) {
  switch (cfg) {
    case LOOPS_TILE_256x8: *s = {256, 8}; return true;
    case LOOPS_TILE_128x7: *s = {128, 7}; return true;
    case LOOPS_TILE_4x2: *s = {4, 2}; return true;
    case LOOPS_TILE_256x7: *s = {256, 7}; return true;
    case LOOPS_TILE_512x8: *s = {512, 8}; return true;
    case LOOPS_TILE_256x16: *s = {256, 16}; return true;
    default: return false;
  }
}

inline int check_csr(int rows, int cols, int nnz, const void* off, const void* idx, const void* val, const void* x,
                     const void* y) {
  if (rows < 0 || cols < 0 || nnz < 0) return LOOPS_E_BADARG;
  if (!off || !y || (nnz > 0 && (!idx || !val || !x))) return LOOPS_E_BADARG;
  if (static_cast<long long>(rows) + nnz >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
  return 0;
}

}  // namespace

// The entry points, by family (each part: the helpers it needs in an unnamed namespace, then its extern "C" block).  The export
// table is checked against include/loops_amd.h by tests/test_c_abi.py.
#include "abi_csr.inc"
#include "abi_formats.inc"
#include "abi_panel.inc"
#include "abi_rowband.inc"
#include "abi_bcsr_band.inc"
#include "abi_plans.inc"
#include "abi_multi_gpu.inc"
