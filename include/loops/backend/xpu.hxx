/**
 * @file xpu.hxx
 * @brief `loops::xpu` -- the runtime vocabulary every loops header and caller uses
 * (streams, events, memcpy, attributes, occupancy).  This library targets AMD CDNA4
 * (gfx950) only, so the names bind straight to the HIP runtime: there is no second
 * backend and no dispatch layer.
 *
 * Mirrors the name set of the reference's `loops/backend/xpu.hxx` + `backend/hip.hxx:25-143`
 * (callers such as examples/spmv/custom_layout.cu:235 use `loops::xpu::stream_synchronize`).
 */
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#ifndef LOOPS_BACKEND_HIP
#define LOOPS_BACKEND_HIP 1
#endif

#define LOOPS_VERSION_MAJOR 0
#define LOOPS_VERSION_MINOR 2
#define LOOPS_VERSION_PATCH 0

namespace loops {
namespace xpu {

using error_t = hipError_t;
using stream_t = hipStream_t;
using event_t = hipEvent_t;
using device_properties_t = hipDeviceProp_t;
using memcpy_kind_t = hipMemcpyKind;
using device_attribute_t = hipDeviceAttribute_t;

inline constexpr error_t success = hipSuccess;

inline constexpr memcpy_kind_t memcpy_host_to_device = hipMemcpyHostToDevice;
inline constexpr memcpy_kind_t memcpy_device_to_host = hipMemcpyDeviceToHost;
inline constexpr memcpy_kind_t memcpy_device_to_device = hipMemcpyDeviceToDevice;

inline constexpr device_attribute_t attr_multiprocessor_count = hipDeviceAttributeMultiprocessorCount;
inline constexpr device_attribute_t attr_compute_capability_major = hipDeviceAttributeComputeCapabilityMajor;
inline constexpr device_attribute_t attr_compute_capability_minor = hipDeviceAttributeComputeCapabilityMinor;
inline constexpr device_attribute_t attr_max_grid_dim_x = hipDeviceAttributeMaxGridDimX;

inline error_t set_device(int ordinal) { return hipSetDevice(ordinal); }
inline error_t get_device(int* ordinal) { return hipGetDevice(ordinal); }
inline error_t get_device_properties(device_properties_t* p, int ordinal) { return hipGetDeviceProperties(p, ordinal); }
inline error_t device_get_attribute(int* v, device_attribute_t a, int ordinal) { return hipDeviceGetAttribute(v, a, ordinal); }
inline error_t device_synchronize() { return hipDeviceSynchronize(); }

inline error_t malloc(void** ptr, std::size_t bytes) { return hipMalloc(ptr, bytes); }
inline error_t free(void* ptr) { return hipFree(ptr); }
inline error_t memcpy(void* dst, const void* src, std::size_t bytes, memcpy_kind_t kind) { return hipMemcpy(dst, src, bytes, kind); }
inline error_t memcpy_async(void* dst, const void* src, std::size_t bytes, memcpy_kind_t kind, stream_t s) { return hipMemcpyAsync(dst, src, bytes, kind, s); }
inline error_t memset_async(void* dst, int value, std::size_t bytes, stream_t s) { return hipMemsetAsync(dst, value, bytes, s); }

inline error_t stream_synchronize(stream_t stream = 0) { return hipStreamSynchronize(stream); }

inline error_t event_create(event_t* e) { return hipEventCreate(e); }
inline error_t event_destroy(event_t e) { return hipEventDestroy(e); }
inline error_t event_record(event_t e, stream_t s = 0) { return hipEventRecord(e, s); }
inline error_t event_synchronize(event_t e) { return hipEventSynchronize(e); }
inline error_t event_elapsed_time(float* ms, event_t a, event_t b) { return hipEventElapsedTime(ms, a, b); }

inline const char* get_error_string(error_t status) { return hipGetErrorString(status); }
inline error_t get_last_error() { return hipGetLastError(); }

template <typename kernel_t>
inline error_t occupancy_max_active_blocks_per_multiprocessor(int* blocks_per_cu, kernel_t kernel, int block_size,
                                                              std::size_t dynamic_lds_bytes) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, block_size, dynamic_lds_bytes);
}

template <typename func_t>
inline error_t launch_cooperative_kernel(const func_t* kernel, dim3 grid, dim3 block, void** args,
                                         std::size_t lds_bytes, stream_t stream) {
  return hipLaunchCooperativeKernel<func_t>(kernel, grid, block, args, lds_bytes, stream);
}

}  // namespace xpu
}  // namespace loops
