// TESTS-ONLY stand-in for <cuda_runtime.h>: three of the reference's unit-test files include it (they are CUDA-only even
// upstream) for cudaMemcpy / cudaDeviceSynchronize.  Maps exactly those names onto the HIP runtime so the files compile
// unchanged (scripts/build_reference_unittests.sh).  Never on the include path of the product: include/loops is HIP-only.
#pragma once
#include <hip/hip_runtime.h>
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetLastError hipGetLastError
