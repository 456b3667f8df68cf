/**
 * @file schedule.hxx
 * @brief Load-balancing schedules: `schedule::setup<scheme, ...>` maps threads of a launch to
 * (tile, atom) ranges of a layout view.  The user kernel stays a couple of range-for loops;
 * the schedule decides who visits what.
 *
 *   thread_mapped    one thread <-> one tile (grid-stride), atoms sequential
 *   group_mapped     a wavefront / workgroup shares its tiles' atoms (LDS scan + lane-strided atoms)
 *   work_oriented    even share of (tiles + atoms) per THREAD (merge-path split per thread)
 *   merge_path_flat  even share of (tiles + atoms) per WORKGROUP, tile ends staged in LDS,
 *                    then an even share per thread inside the workgroup
 *
 * API shape (template parameter list, member names, storage_t, aliases) follows the reference
 * (include/loops/schedule.hxx:26-63) so kernels written against gunrock/loops compile
 * unchanged; the implementations are written for 64-lane CDNA4 wavefronts.
 */
#pragma once

#include <cstddef>

#include <loops/backend/xpu.hxx>
#include <loops/container/layout.hxx>

namespace loops {
namespace schedule {

enum algorithms_t {
  merge_path_flat,  ///< even share per workgroup + per thread (merge path)
  work_oriented,    ///< even share per thread (merge path)
  thread_mapped,    ///< tile per thread
  group_mapped,     ///< tiles per wavefront/workgroup, atoms lane-strided
  bucketing,        ///< declared by the reference (schedule.hxx:31), never implemented there either
};

template <algorithms_t scheme,
          std::size_t threads_per_block,
          std::size_t threads_per_tile,
          typename tiles_t,
          typename atoms_t,
          typename tile_size_t = std::size_t,
          typename atom_size_t = std::size_t,
          typename layout_type = layout::csr<tiles_t, atoms_t>>
class setup;

}  // namespace schedule
}  // namespace loops

#include <loops/schedule/thread_mapped.hxx>
#include <loops/schedule/group_mapped.hxx>
#include <loops/schedule/work_oriented.hxx>
#include <loops/schedule/merge_path_flat.hxx>
