// Experiment (round 4): does PHASING the x gathers of a tile by column range raise the L2 hit rate?
//
// C2's x (4 MB) exactly fills one XCD's L2, the matrix stream evicts a fifth of it, and every evicted line costs an
// Infinity-Cache round trip.  If all workgroups of the chip gather from the SAME 1/M of x at the same time,
// the working set of a phase is 4 MB / M and stays resident.  This probe is the SpMV without the row reduction:
//   sum_t += vals[i] * table[idx[i]]      (two 64 MB streams + one gather per item, like the tile kernel)
// with a workgroup of 512 threads owning TILES consecutive tiles of 512 x IPT items.  For each tile the gathers are issued in
// M passes: pass p takes the items whose index lies in [p, p + 1) * table / M (execution-masked loads), and waits.
// Workgroups start together and do equal work, so passes line up in time by themselves ("natural lockstep"); LOCK = 1
// additionally picks the first pass from a shared clock (s_memrealtime / period).
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o build/variants/libexp_phased.so tests/perf/exp_phased_gather.hip
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

template <int M, int IPT, bool SYNC>
__global__ void __launch_bounds__(512)
phased_gather_kernel(const float* __restrict__ table, const int* __restrict__ idx, const float* __restrict__ vals,
                     float* __restrict__ out, const long long n, const int table_elems, const int lock_period) {
  using i4 = int __attribute__((ext_vector_type(4)));
  using f4 = float __attribute__((ext_vector_type(4)));
  constexpr int V = IPT / 4;
  const int t = threadIdx.x;
  float sum = 0.f;
  const long long tile_items = 512ll * IPT;
  const int shift = 31 - __builtin_clz(static_cast<unsigned int>(table_elems) / M);  // (table / M is a power of two here)
  for (long long base = static_cast<long long>(blockIdx.x) * tile_items; base + tile_items <= n;
       base += static_cast<long long>(gridDim.x) * tile_items) {
    i4 c[V];
    f4 v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const long long i = base + (static_cast<long long>(k) * 512 + t) * 4;
      c[k] = *reinterpret_cast<const i4*>(idx + i);
      v[k] = *reinterpret_cast<const f4*>(vals + i);
    }
    float x[IPT];
#pragma unroll
    for (int e = 0; e < IPT; ++e) x[e] = 0.f;
    int first = 0;
    if (lock_period > 0) first = static_cast<int>((__builtin_amdgcn_s_memrealtime() / static_cast<unsigned long long>(lock_period)) % M);
#pragma unroll
    for (int pp = 0; pp < M; ++pp) {
      const unsigned int p = static_cast<unsigned int>((pp + first) % M);
#pragma unroll
      for (int k = 0; k < V; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int col = static_cast<unsigned int>(c[k][e]);
          if (M == 1 || (col >> shift) == p) x[k * 4 + e] = table[col];
        }
      }
      if (M > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SYNC) __syncthreads();
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum = __builtin_fmaf(v[k][e], x[k * 4 + e], sum);
  }
  out[static_cast<size_t>(blockIdx.x) * 512 + t] = sum;
}

#define CASE(MM, II, SS) \
  if (m == MM && ipt == II && sync == SS) { \
    hipLaunchKernelGGL((phased_gather_kernel<MM, II, SS != 0>), dim3(blocks), dim3(512), 0, static_cast<hipStream_t>(stream), table, idx, vals, out, \
                       n, table_elems, lock_period); \
    return static_cast<int>(hipGetLastError()); \
  }

extern "C" int exp_phased_gather(const float* table, const int* idx, const float* vals, float* out, long long n, int table_elems, int m,
                                 int ipt, int sync, int blocks, int lock_period, void* stream) {
  CASE(1, 8, 0) CASE(2, 8, 0) CASE(4, 8, 0) CASE(8, 8, 0) CASE(2, 8, 1) CASE(4, 8, 1) CASE(8, 8, 1)
  CASE(1, 16, 0) CASE(2, 16, 0) CASE(4, 16, 0) CASE(8, 16, 0) CASE(2, 16, 1) CASE(4, 16, 1) CASE(8, 16, 1)
  CASE(1, 32, 0) CASE(2, 32, 0) CASE(4, 32, 0) CASE(8, 32, 0) CASE(2, 32, 1) CASE(4, 32, 1) CASE(8, 32, 1)
  return -1;
}
