/**
 * @file run_stitch.hxx
 * @brief `add_row_runs`: a lane holds IPT consecutive products and the row each belongs to; runs of equal rows are
 * summed in registers, the open runs at both ends of a lane are stitched across the 64 lanes with the segmented
 * prefix sum (wave::segmented_inclusive_sum), and ONE atomicAdd per run and wavefront reaches y.  Shared by the tuned
 * COO kernel (coo_runs_spmv: rows come from the triplets) and the tuned flat_partitioned kernel
 * (flat_partitioned_stitched_spmv: rows come from ONE tile_of per lane and a walk along the row ends).  Correct for
 * any order of the rows; non-decreasing rows are the fast case.  A row index < 0 marks a slot past the end.  y must be
 * zero-filled (the precondition of the reference kernels these replace: coo_thread_mapped.cuh:95-97,
 * flat_partitioned.cuh:99-101).  Every lane of the wavefront must call it.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <loops/util/wave.hxx>

namespace loops {
namespace kernels {

template <int IPT, typename index_t, typename type_t>
__device__ __forceinline__ void add_row_runs(const index_t (&r)[IPT], const type_t (&p)[IPT], type_t* __restrict__ y) {
  // runs of equal row indices inside the lane: the first one and the open last one are combined
  // across the wavefront below; the ones in between go straight to y
  index_t row = r[0];
  type_t sum = p[0];
  index_t first_row = row;
  type_t first_sum = type_t(0);
  bool closed = false;
#pragma unroll
  for (int k = 1; k < IPT; ++k) {
    if (r[k] != row) {
      if (!closed) {
        first_sum = sum;
        closed = true;
      } else if (row >= 0) {
        atomicAdd(&y[row], sum);
      }
      row = r[k];
      sum = type_t(0);
    }
    sum += p[k];
  }
  // Wavefront stitch (one atomicAdd per run and wavefront instead of per lane): segmented prefix
  // sum of the lanes' open tails; a lane's tail starts a new segment if the lane closed a run or its
  // first row differs from the previous lane's last row.
  const int lane = wave::lane();
  const index_t prev_last = wave::shift_up1(row, index_t(-2));  // lane 0: never equal
  const bool continues = first_row == prev_last;                 // my first run continues the previous lane's tail
  type_t run = sum;
  bool head = closed || !continues;
  wave::segmented_inclusive_sum(run, head);
  const type_t prev_run = wave::shift_up1(run, type_t(0));
  const int next_continues = __shfl_down(static_cast<int>(continues), 1);
  if (closed && first_row >= 0) atomicAdd(&y[first_row], first_sum + (continues ? prev_run : type_t(0)));
  const bool tail_ends_here = lane == wave::size - 1 || !next_continues;
  if (tail_ends_here && row >= 0) atomicAdd(&y[row], run);
}

}  // namespace kernels
}  // namespace loops
