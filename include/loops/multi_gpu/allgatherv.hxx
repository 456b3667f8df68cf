/**
 * @file allgatherv.hxx
 * @brief allgatherv of the row-range slices of y over RCCL: on entry rank r has written y_full[bounds[r] .. bounds[r + 1]);
 * on return (stream-ordered) every rank holds the whole vector.
 *
 * RCCL has no native allgatherv.  The exchange is ONE group of point-to-point operations on the caller's communicator --
 * ncclGroupStart, a ncclSend of the own slice to and a ncclRecv of the peer's slice from every other rank, ncclGroupEnd --
 * so every slice crosses every xGMI link exactly once and all 7 links of a GPU are busy at the same time (xGMI is
 * point-to-point: a ring would be 7 serial per-link-bound hops).  In place: the receive buffers are the peers'
 * ranges of y_full itself.
 *
 * Header form for C++ callers that link RCCL themselves (`#include <rccl/rccl.h>`, -lrccl).  The C ABI twins
 * (loops_allgatherv_f32 / _f64, include/loops_amd.h) take the communicator as `void*` and resolve the RCCL entry points of
 * the process at run time, so libloops_amd.so carries no link-time dependency on a particular RCCL build.
 * Unmeasured on multi-GPU hardware (no such box in this project's test pool): world size 1 on an MI355X, the
 * partition logic with gloo on CPU.  No reference counterpart (the reference is single-GPU).
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace loops {
namespace multi_gpu {

template <typename type_t>
struct nccl_type;
template <>
struct nccl_type<float> { static constexpr ncclDataType_t value = ncclFloat32; };
template <>
struct nccl_type<double> { static constexpr ncclDataType_t value = ncclFloat64; };
template <>
struct nccl_type<int> { static constexpr ncclDataType_t value = ncclInt32; };

/// @param bounds host array of world + 1 row boundaries (multi_gpu::row_ranges); asynchronous on `stream`.
template <typename type_t>
inline ncclResult_t allgatherv(ncclComm_t comm, int rank, int world, type_t* y_full, const long long* bounds, hipStream_t stream) {
  if (world <= 1) return ncclSuccess;
  if (!y_full || !bounds || rank < 0 || rank >= world) return ncclInvalidArgument;
  const std::size_t mine = static_cast<std::size_t>(bounds[rank + 1] - bounds[rank]);
  ncclResult_t r = ncclGroupStart();
  for (int peer = 0; peer < world && r == ncclSuccess; ++peer) {
    if (peer == rank) continue;
    const std::size_t theirs = static_cast<std::size_t>(bounds[peer + 1] - bounds[peer]);
    if (mine) r = ncclSend(y_full + bounds[rank], mine, nccl_type<type_t>::value, peer, comm, stream);
    if (theirs && r == ncclSuccess) r = ncclRecv(y_full + bounds[peer], theirs, nccl_type<type_t>::value, peer, comm, stream);
  }
  const ncclResult_t e = ncclGroupEnd();
  return r != ncclSuccess ? r : e;
}

}  // namespace multi_gpu
}  // namespace loops
