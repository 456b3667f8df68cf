// loops_probes.hip -- libloops_probes.so: measurement code only (see loops_probes.h).  Built next to the product
// library by loops_amd/_lib.py; never linked into, loaded by, or required by libloops_amd.so.
#include "loops_probes.h"

#include <cstdint>

#include <hip/hip_runtime.h>

#include <loops/kernels/launch.hxx>
#include <loops/util/math.hxx>

#include "probes.hxx"

using namespace loops;
using kernels::coord_t;

namespace {

constexpr int E_BADARG = -1, E_CONFIG = -3;
inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

constexpr int TPB = 512, IPT = 8;

namespace pol = kernels::detail::policy;

struct policy_entry {
  const char* name;
};

// policy id -> (engine NT parameter); keep the two tables in step
constexpr int kPolicies[] = {
    0,                                               // 0 plain global loads (the product kernel)
    1,                                               // 1 __builtin_nontemporal_load on both streams
    pol::make(0, 0),                                 // 2 buffer loads, no bits
    pol::make(pol::sc0, pol::sc0),                   // 3
    pol::make(pol::nt, pol::nt),                     // 4
    pol::make(pol::sc1, pol::sc1),                   // 5
    pol::make(pol::sc0 | pol::sc1, pol::sc0 | pol::sc1),  // 6
    pol::make(pol::nt | pol::sc1, pol::nt | pol::sc1),    // 7
    pol::make(pol::nt | pol::sc0, pol::nt | pol::sc0),    // 8
    pol::make(19, 19),                               // 9 sc0 sc1 nt
    pol::make(pol::nt, 0),                           // 10
    pol::make(0, pol::nt),                           // 11
    pol::make(0, 0, pol::sc0),                       // 12
    pol::make(0, 0, pol::sc1),                       // 13
    pol::make(0, 0, pol::nt),                        // 14
    pol::make(0, 0, pol::sc0 | pol::sc1),            // 15
    pol::make(pol::nt, pol::nt, pol::sc1),           // 16
    pol::make(pol::nt | pol::sc1, pol::nt | pol::sc1, pol::sc1),  // 17
    pol::windowed(11),                               // 18 LDS window of x: 2048 columns around the tile's middle row
    pol::windowed(12),                               // 19 4096 columns (16 KB: still 4 workgroups per CU)
    pol::windowed(13),                               // 20 8192 columns (32 KB: 3 workgroups per CU)
};
const char* const kPolicyNames[] = {
    "global plain", "global nontemporal", "buffer plain", "stream sc0", "stream nt", "stream sc1", "stream sc0+sc1",
    "stream nt+sc1", "stream nt+sc0", "stream sc0+sc1+nt", "idx nt / val plain", "idx plain / val nt", "gather sc0",
    "gather sc1", "gather nt", "gather sc0+sc1", "stream nt, gather sc1", "stream nt+sc1, gather sc1",
    "x window 2048 in LDS", "x window 4096 in LDS", "x window 8192 in LDS",
};
constexpr int kNumPolicies = sizeof(kPolicies) / sizeof(kPolicies[0]);
static_assert(kNumPolicies == sizeof(kPolicyNames) / sizeof(kPolicyNames[0]), "policy tables out of step");

struct scratch_view {
  coord_t* coords;
  float* carry_val;
  int* carry_row;
  int m;
};

scratch_view carve(void* scratch, int rows, int nnz) {
  const int m = static_cast<int>(math::ceil_div(static_cast<long long>(rows) + nnz, static_cast<long long>(TPB) * IPT));
  auto* coords = static_cast<coord_t*>(scratch);
  auto* carry_val = reinterpret_cast<double*>(coords + (m + 1));
  auto* carry_row = reinterpret_cast<int*>(carry_val + (m + 2));
  return {coords, reinterpret_cast<float*>(carry_val), carry_row, m};
}

// policy::windowed engines: the fused tile kernel with the column count handed to the engine
template <int NTP>
__global__ void __launch_bounds__(TPB)
windowed_tile_kernel(const coord_t* __restrict__ coords, const int rows, const int cols, const int nnz, const int* __restrict__ offsets,
                     const int* __restrict__ indices, const float* __restrict__ values, const float* __restrict__ x,
                     float* __restrict__ y, int* __restrict__ carry_row, float* __restrict__ carry_val) {
  kernels::detail::phase_args phase;
  phase.window_cols = static_cast<unsigned int>(cols);
  kernels::merge_path_spmv_tile_to<TPB, IPT, true, NTP, true, false, true>(
      coords, rows, nnz, kernels::csr_row_end<int>{offsets}, indices, values, x, kernels::plain_store<float>{y}, carry_row,
      carry_val, nullptr, phase);
}

// Ordering experiment (round-4 review, item 8): a PERSISTENT workgroup walks a contiguous share of the merge tiles (the product's
// work_oriented_spmv_fused) and, pipelined, issues a tile's x gathers BEFORE the stream loads of its next tile.  Two register
// sets: `cur` (dead once the tile's products are in LDS) and `next` (in flight during the walk), copied at the end of the tile.
template <bool LATE>
__global__ void __launch_bounds__(TPB)
persistent_pipelined_kernel(const coord_t* __restrict__ coords, const int num_tiles, const int tiles_per_group, const int rows,
                            const int nnz, const int* __restrict__ offsets, const int* __restrict__ indices,
                            const float* __restrict__ values, const float* __restrict__ x, float* __restrict__ y,
                            int* __restrict__ carry_row, float* __restrict__ carry_val) {
  using engine_t = kernels::merge_tile_engine<TPB, IPT, true, 0, true, int, int, float, true>;
  __shared__ typename engine_t::storage_t s_engine;
  const int tid = threadIdx.x;
  const int g = kernels::detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int t_begin = g * tiles_per_group;
  int t_end = t_begin + tiles_per_group;
  t_end = t_end < num_tiles ? t_end : num_tiles;
  if (t_begin >= t_end) return;
  float carry = 0.f;
  int open_row = 0;
  typename engine_t::tile_regs cur, next;
  engine_t::issue_streams(cur, static_cast<int>(coords[t_begin].y) & ~3, static_cast<int>(coords[t_begin + 1].y), indices, values);
  for (int t = t_begin; t < t_end; ++t) {
    const coord_t c0 = coords[t];
    const coord_t c1 = coords[t + 1];
    const int tn = t + 1 < t_end ? t + 1 : t;
    const int next_abase = static_cast<int>(coords[tn].y) & ~3;
    const int next_nz1 = static_cast<int>(coords[tn + 1].y);
    const int row0 = static_cast<int>(c0.x);
    const int nz0 = static_cast<int>(c0.y);
    const int nrows = static_cast<int>(c1.x) - row0;
    const int natoms = static_cast<int>(c1.y) - nz0;
    engine_t::clear_marks(s_engine);
    __syncthreads();
    carry = engine_t::run_to(
        s_engine, static_cast<const int*>(nullptr), row0, nz0, nrows, natoms, nnz, indices, values, x, kernels::plain_store<float>{y},
        carry,
        [&]() {
          for (int i = tid; i < nrows; i += TPB) engine_t::mark_row_end(s_engine, i, offsets[row0 + i + 1], nz0);
        },
        kernels::detail::phase_args{}, typename engine_t::template tile_pipe<LATE>{cur, next, next_abase, next_nz1});
    open_row = row0 + nrows;
    // (the copy stays HERE, behind the walk: an opaque use keeps the compiler from waiting for the next tile's streams any earlier)
#pragma unroll
    for (int k = 0; k < engine_t::KV; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(next.col[k][j]), "+v"(next.val[k][j]));
        cur.col[k][j] = next.col[k][j];
        cur.val[k][j] = next.val[k][j];
      }
    }
  }
  if (tid == 0) {
    carry_row[g] = open_row;
    carry_val[g] = carry;
  }
}

// The persistent form with the phased gathers of the headline kernel (policy::phased(8 / 16 / 32)): does a share of a few tiles per workgroup
// help there as it helps the plain kernel?
template <int PHASES>
__global__ void __launch_bounds__(TPB)
persistent_phased_kernel(const coord_t* __restrict__ coords, const int num_tiles, const int tiles_per_group, const int rows,
                         const int nnz, const int* __restrict__ offsets, const int* __restrict__ indices,
                         const float* __restrict__ values, const float* __restrict__ x, float* __restrict__ y,
                         int* __restrict__ carry_row, float* __restrict__ carry_val, const kernels::detail::phase_args phase) {
  using engine_t = kernels::merge_tile_engine<TPB, IPT, true, pol::phased(PHASES), true, int, int, float, true>;
  __shared__ typename engine_t::storage_t s_engine;
  const int tid = threadIdx.x;
  const int g = kernels::detail::xcd_contiguous(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
  const int t_begin = g * tiles_per_group;
  int t_end = t_begin + tiles_per_group;
  t_end = t_end < num_tiles ? t_end : num_tiles;
  float carry = 0.f;
  int open_row = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const coord_t c0 = coords[t];
    const coord_t c1 = coords[t + 1];
    const int row0 = static_cast<int>(c0.x);
    const int nz0 = static_cast<int>(c0.y);
    const int nrows = static_cast<int>(c1.x) - row0;
    const int natoms = static_cast<int>(c1.y) - nz0;
    engine_t::clear_marks(s_engine);
    __syncthreads();
    carry = engine_t::run_to(
        s_engine, static_cast<const int*>(nullptr), row0, nz0, nrows, natoms, nnz, indices, values, x, kernels::plain_store<float>{y},
        carry,
        [&]() {
          for (int i = tid; i < nrows; i += TPB) engine_t::mark_row_end(s_engine, i, offsets[row0 + i + 1], nz0);
        },
        phase);
    open_row = row0 + nrows;
  }
  if (tid == 0 && t_begin < t_end) {
    carry_row[g] = open_row;
    carry_val[g] = carry;
  }
}

template <int P>
void launch_policy(const scratch_view& v, int rows, int cols, int nnz, const int* off, const int* idx, const float* val,
                   const float* x, float* y, hipStream_t stream) {
  if constexpr (pol::window(kPolicies[P]) > 0) {
    hipLaunchKernelGGL((windowed_tile_kernel<kPolicies[P]>), dim3(v.m), dim3(TPB), 0, stream, v.coords, rows, cols, nnz, off, idx,
                       val, x, y, v.carry_row, v.carry_val);
    return;
  }
  hipLaunchKernelGGL((kernels::merge_path_spmv_fused<TPB, IPT, true, kPolicies[P], true, int, int, float, true>), dim3(v.m),
                     dim3(TPB), 0, stream, v.coords, rows, nnz, off, idx, val, x, y, v.carry_row, v.carry_val);
}

template <int... Ps>
bool dispatch(int policy, std::integer_sequence<int, Ps...>, const scratch_view& v, int rows, int cols, int nnz, const int* off,
              const int* idx, const float* val, const float* x, float* y, hipStream_t stream) {
  return ((policy == Ps ? (launch_policy<Ps>(v, rows, cols, nnz, off, idx, val, x, y, stream), true) : false) || ...);
}

// Tile-shape experiment: shapes the product does not ship (a 512 x 16 tile halves the number of resident workgroups and
// of concurrently streamed lines per XCD; 1024 x 8 synchronises 16 wavefronts per workgroup barrier).
template <int T, int I>
int run_shape(int stages, int rows, int nnz, const int* off, const int* idx, const float* val, const float* x, float* y,
              void* scratch, hipStream_t st) {
  const int m = static_cast<int>(math::ceil_div(static_cast<long long>(rows) + nnz, static_cast<long long>(T) * I));
  if (m < 2) return E_CONFIG;
  auto* coords = static_cast<coord_t*>(scratch);
  auto* carry_val = reinterpret_cast<float*>(coords + (m + 1));
  auto* carry_row = reinterpret_cast<int*>(reinterpret_cast<double*>(coords + (m + 1)) + (m + 2));
  if (stages & 4) {
    const int err = kernels::launch_merge_path_coordinates(st, off, rows, nnz, T * I, m, coords);
    if (err) return err;
  }
  if (stages & 1)
    hipLaunchKernelGGL((kernels::merge_path_spmv_fused<T, I, true, 0, true, int, int, float, true>), dim3(m), dim3(T), 0, st, coords,
                       rows, nnz, off, idx, val, x, y, carry_row, carry_val);
  if (stages & 2)
    hipLaunchKernelGGL(kernels::merge_path_spmv_fixup<float>, dim3(math::ceil_div(m, 256)), dim3(256), 0, st, carry_row, carry_val, m,
                       rows, y);
  return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" {

/* shape: 0 = 512 x 8 (the product's), 1 = 512 x 16, 2 = 1024 x 8, 3 = 1024 x 4, 4 = 256 x 8; scratch sized for 256 x 4 tiles
 * (loops_probe_merge_path_scratch_bytes() * 4 is enough for every shape here) */
int loops_probe_merge_path_shape_f32(int shape, int stages, int rows, int cols, int nnz, const int* offsets,
                                     const int* indices, const float* values, const float* x, float* y, void* scratch,
                                     void* stream) {
  (void)cols;
  if (!offsets || !indices || !values || !x || !y || !scratch || rows <= 0 || nnz <= 0) return E_BADARG;
  if ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) return E_BADARG;
  hipStream_t st = as_stream(stream);
  switch (shape) {
    case 0: return run_shape<512, 8>(stages, rows, nnz, offsets, indices, values, x, y, scratch, st);
    case 1: return run_shape<512, 16>(stages, rows, nnz, offsets, indices, values, x, y, scratch, st);
    case 2: return run_shape<1024, 8>(stages, rows, nnz, offsets, indices, values, x, y, scratch, st);
    case 3: return run_shape<1024, 4>(stages, rows, nnz, offsets, indices, values, x, y, scratch, st);
    case 4: return run_shape<256, 8>(stages, rows, nnz, offsets, indices, values, x, y, scratch, st);
    default: return E_CONFIG;
  }
}

int loops_mixed_gather_f32(const float* table, const int* idx, float* out, size_t n, int scalar_per_64, void* stream) {
  if (!table || !idx || !out) return E_BADARG;
  const int rc = kernels::launch_mixed_gather(as_stream(stream), table, idx, out, n, scalar_per_64);
  return rc == -1 ? E_CONFIG : rc;
}

int loops_stream_read_prefetch_f32(const float* src, float* sink, size_t n, int distance, int line_words, int waves_per_cu,
                                   void* stream) {
  if (!src || !sink) return E_BADARG;
  const int rc = kernels::launch_stream_read_prefetch(as_stream(stream), src, sink, n, distance, line_words, waves_per_cu);
  return rc == -1 ? E_CONFIG : rc;
}

int loops_stream_copy_f32(const float* src, float* dst, size_t n, void* stream) {
  if (!src || !dst) return E_BADARG;
  return kernels::launch_stream_copy(as_stream(stream), src, dst, n);
}

int loops_stream_copy_tuned_f32(const float* src, float* dst, size_t n, int unroll, int flags, int blocks, void* stream) {
  if (!src || !dst || blocks <= 0) return E_BADARG;
  const int rc = kernels::launch_stream_copy_tuned(as_stream(stream), src, dst, n, unroll, flags, blocks);
  return rc == -1 ? E_BADARG : rc;
}

int loops_gather_f32(const float* table, const int* idx, float* out, size_t n, int mode, void* stream) {
  if (!table || !idx || !out) return E_BADARG;
  return kernels::launch_gather(as_stream(stream), table, idx, out, n, mode);
}

int loops_address_rate_f32(const float* table, int table_words, int reps, int pattern, int blocks, float* out,
                           void* stream) {
  if (!table || !out || table_words <= 0 || (table_words & (table_words - 1)) || reps < 0 || blocks <= 0) return E_BADARG;
  return kernels::launch_address_rate(as_stream(stream), table, table_words, reps, pattern, blocks, out);
}

int loops_lds_update_rate_f32(int mode, int pattern, int reps, int blocks, float* out, void* stream) {
  if (!out || reps < 0 || blocks <= 0) return E_BADARG;
  const int rc = kernels::launch_lds_update(as_stream(stream), mode, pattern, reps, blocks, out);
  return rc == -1 ? E_BADARG : rc;
}

int loops_row_gather_f32(const float* table, const int* idx, size_t count, int row_floats, int blocks, float* out,
                         void* stream) {
  if (!table || !idx || !out || blocks <= 0) return E_BADARG;
  const int rc = kernels::launch_row_gather(as_stream(stream), table, idx, count, row_floats, blocks, out);
  return rc == -1 ? E_CONFIG : rc;
}

size_t loops_probe_merge_path_scratch_bytes(int rows, int nnz) {
  const size_t m = static_cast<size_t>(math::ceil_div(static_cast<long long>(rows) + nnz, static_cast<long long>(TPB) * IPT));
  return (m + 1) * sizeof(coord_t) + (m + 2) * (sizeof(double) + sizeof(int)) + 64;
}

int loops_probe_persistent_f32(int pipelined, int groups, int stages, int rows, int cols, int nnz, const int* offsets,
                               const int* indices, const float* values, const float* x, float* y, void* scratch, void* stream) {
  if (!offsets || !indices || !values || !x || !y || !scratch || rows <= 0 || nnz < 8 || ((pipelined == 1 || pipelined == 2) && (nnz & 3)) || groups <= 0) return E_BADARG;
  if ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) return E_BADARG;
  hipStream_t st = as_stream(stream);
  const scratch_view v = carve(scratch, rows, nnz);
  if (v.m < 2 || groups > v.m) return E_CONFIG;
  if (stages & 4) {
    const int err = kernels::launch_merge_path_coordinates(st, offsets, rows, nnz, TPB * IPT, v.m, v.coords);
    if (err) return err;
  }
  const int per = math::ceil_div(v.m, groups);
  const int grid = math::ceil_div(v.m, per);
  if (stages & 1) {
    if (pipelined == 3) {
      const kernels::phased_config cfg = kernels::phased_config_for(cols, 4);   // 8 / 16 / 32 parts by the size of x
      auto k = cfg.parts == 8 ? persistent_phased_kernel<8> : cfg.parts == 16 ? persistent_phased_kernel<16> : persistent_phased_kernel<32>;
      hipLaunchKernelGGL(k, dim3(grid), dim3(TPB), 0, st, v.coords, v.m, per, rows, nnz, offsets, indices, values, x, y, v.carry_row,
                         v.carry_val, cfg.args);
    }
    else if (pipelined == 2)
      hipLaunchKernelGGL(persistent_pipelined_kernel<true>, dim3(grid), dim3(TPB), 0, st, v.coords, v.m, per, rows, nnz, offsets,
                         indices, values, x, y, v.carry_row, v.carry_val);
    else if (pipelined)
      hipLaunchKernelGGL(persistent_pipelined_kernel<false>, dim3(grid), dim3(TPB), 0, st, v.coords, v.m, per, rows, nnz, offsets, indices,
                         values, x, y, v.carry_row, v.carry_val);
    else
      hipLaunchKernelGGL((kernels::work_oriented_spmv_fused<TPB, IPT, true, 0, true, int, int, float, true>), dim3(grid), dim3(TPB), 0,
                         st, v.coords, v.m, per, rows, nnz, offsets, indices, values, x, y, v.carry_row, v.carry_val);
  }
  if (stages & 2)
    hipLaunchKernelGGL(kernels::merge_path_spmv_fixup<float>, dim3(math::ceil_div(grid, 256)), dim3(256), 0, st, v.carry_row,
                       v.carry_val, grid, rows, y);
  return static_cast<int>(hipGetLastError());
}

int loops_probe_policy_count(void) { return kNumPolicies; }
const char* loops_probe_policy_name(int policy) {
  return policy >= 0 && policy < kNumPolicies ? kPolicyNames[policy] : nullptr;
}

int loops_probe_merge_path_f32(int policy, int stages, int rows, int cols, int nnz, const int* offsets,
                               const int* indices, const float* values, const float* x, float* y, void* scratch,
                               void* stream) {
  if (!offsets || !indices || !values || !x || !y || !scratch || rows <= 0 || nnz <= 0) return E_BADARG;
  if (policy < 0 || policy >= kNumPolicies) return E_CONFIG;
  if (pol::window(kPolicies[policy]) > cols) return E_CONFIG;
  if ((reinterpret_cast<std::uintptr_t>(indices) | reinterpret_cast<std::uintptr_t>(values)) & 15u) return E_BADARG;
  hipStream_t st = as_stream(stream);
  const scratch_view v = carve(scratch, rows, nnz);
  if (v.m < 2) return E_CONFIG;  // single-tile matrices take another path in the product
  if (stages & 4) {
    const int err = kernels::launch_merge_path_coordinates(st, offsets, rows, nnz, TPB * IPT, v.m, v.coords);
    if (err) return err;
  }
  if (stages & 1) {
    if (!dispatch(policy, std::make_integer_sequence<int, kNumPolicies>{}, v, rows, cols, nnz, offsets, indices, values, x, y, st))
      return E_CONFIG;
  }
  if (stages & 2)
    hipLaunchKernelGGL(kernels::merge_path_spmv_fixup<float>, dim3(math::ceil_div(v.m, 256)), dim3(256), 0, st, v.carry_row,
                       v.carry_val, v.m, rows, y);
  return static_cast<int>(hipGetLastError());
}

}  // extern "C"
