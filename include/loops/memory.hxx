/**
 * @file memory.hxx
 * @brief Memory spaces and `raw_pointer_cast` (reference: include/loops/memory.hxx:33-67).
 */
#pragma once

#include <thrust/device_ptr.h>

#include <hip/hip_runtime.h>

namespace loops {
namespace memory {

enum memory_space_t { device, host, managed };

template <typename type_t>
inline type_t* raw_pointer_cast(thrust::device_ptr<type_t> pointer) {
  return thrust::raw_pointer_cast(pointer);
}

template <typename type_t>
__host__ __device__ inline type_t* raw_pointer_cast(type_t* pointer) {
  return pointer;
}

}  // namespace memory

/// first/second aggregate usable in device code (what `setup<work_oriented>::init()` returns).
template <typename first_t, typename second_t>
struct pair {
  first_t first;
  second_t second;
};

}  // namespace loops
