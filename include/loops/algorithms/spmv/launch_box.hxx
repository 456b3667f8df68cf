/**
 * @file launch_box.hxx
 * @brief `launch_t<type_t>`: (workgroup size, merge items per thread) for the SpMV kernels, keyed
 * by the compile target (reference include/loops/algorithms/spmv/launch_box.hxx:63-90).
 *
 * gfx950 (MI355X): 256 threads x 8 items (fp32) / 4 items (fp64) = 2048 / 1024 merge items per
 * workgroup: 8 workgroups of 16.7 KB LDS are resident per CU (32 wavefronts), which keeps
 * ~128 KB of col_idx / value loads in flight per CU.  On this chip the tile shape is NOT the
 * bottleneck of SpMV -- the x gather is (profiles/) -- 256x7, 128x7 and 512x8 measure
 * within 2 % of 256x8 on the C2 workload, so the reference's gfx950 entry is kept.
 */
#pragma once

#include <cstddef>

#include <loops/util/launch_box.hxx>

namespace loops {
namespace algorithms {
namespace spmv {

template <typename type_t>
using launch_t = launch_box::launch_box_t<
    launch_box::launch_params_t<launch_box::gfx942 | launch_box::gfx950, 256, (sizeof(type_t) > 4 ? 4 : 8)>,
    launch_box::launch_params_t<launch_box::gfx906 | launch_box::gfx908 | launch_box::gfx90a, 256,
                                (sizeof(type_t) > 4 ? 4 : 7)>,
    launch_box::launch_params_t<launch_box::fallback, 256, (sizeof(type_t) > 4 ? 4 : 8)>>;

}  // namespace spmv
}  // namespace algorithms
}  // namespace loops
