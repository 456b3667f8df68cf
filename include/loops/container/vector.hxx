/**
 * @file vector.hxx
 * @brief `vector_t<T, space>`: owning 1-D storage in host or device (HBM) memory.  These are the
 * rocThrust vectors -- the callers rely on that surface (`.data().get()`, `thrust::raw_pointer_cast`,
 * cross-space copy construction; reference include/loops/container/vector.hxx:33-44).
 * Containers are plumbing: nothing on the timed SpMV path allocates or copies them.
 */
#pragma once

#include <type_traits>

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <loops/memory.hxx>

namespace loops {
using namespace memory;

template <typename type_t, memory_space_t space = memory_space_t::device>
using vector_t = std::conditional_t<space == memory_space_t::host, thrust::host_vector<type_t>,
                                    thrust::device_vector<type_t>>;

template <typename type_t>
using host_vector_t = thrust::host_vector<type_t>;
template <typename type_t>
using device_vector_t = thrust::device_vector<type_t>;

}  // namespace loops
