// loops_c_abi.hip -- extern "C" surface of libloops_amd.so (declared in include/loops_amd.h).
//
// Thin wrappers: every entry point instantiates the C++ templates of include/loops/ for
// index_t = offset_t = int, type_t = float | double and launches them on the caller's
// stream.  No CPU fallbacks live here: if a kernel cannot be launched the hipError_t is
// returned.  Built by __graft_entry__.build() with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude loops_c_abi.hip
#include <loops_amd.h>

#include <algorithm>
#include <cstdint>
#include <limits>
#include <new>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>

#include <loops/schedule.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/launch_box.hxx>
#include <loops/util/math.hxx>
#include <loops/kernels/launch.hxx>
#include <loops/kernels/panel_binned.hxx>
#include <loops/kernels/rowband.hxx>
#include <loops/multi_gpu/partition.hxx>
#include <loops/kernels/coo_spmv.hxx>
#include <loops/kernels/ell_spmv.hxx>
#include <loops/kernels/dia_spmv.hxx>
#include <loops/kernels/csc_spmv.hxx>
#include <loops/kernels/bcsr_spmv.hxx>

using namespace loops;
using kernels::coord_t;

namespace {

constexpr int kSpmvBlock = 256;  // launch_t<T>::block_size on gfx950 (launch_box.hxx:75-77)

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }
inline int last_error() { return static_cast<int>(hipGetLastError()); }

struct tile_shape {
  int tpb, ipt;
};
inline bool shape_of(int cfg, tile_shape* s) {
  switch (cfg) {
    case LOOPS_TILE_256x8: *s = {256, 8}; return true;
    case LOOPS_TILE_128x7: *s = {128, 7}; return true;
    case LOOPS_TILE_4x2: *s = {4, 2}; return true;
    case LOOPS_TILE_256x7: *s = {256, 7}; return true;
    case LOOPS_TILE_512x8: *s = {512, 8}; return true;
    case LOOPS_TILE_256x16: *s = {256, 16}; return true;
    default: return false;
  }
}

inline int check_csr(int rows, int cols, int nnz, const void* off, const void* idx, const void* val, const void* x,
                     const void* y) {
  if (rows < 0 || cols < 0 || nnz < 0) return LOOPS_E_BADARG;
  if (!off || !y || (nnz > 0 && (!idx || !val || !x))) return LOOPS_E_BADARG;
  if (static_cast<long long>(rows) + nnz >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------ plan
struct loops_merge_plan {
  int rows, nnz, cfg, tpb, ipt, num_tiles;
  int capacity;        // merge tiles the allocation can hold (>= num_tiles)
  coord_t* coords;     // M + 1
  double* carry_val;   // M + 2 (8 B slots: float or double)
  int* carry_row;      // M + 2
  void* base;
  mutable void* wide_carry;         // SpMM carry-outs, M x n values, grown on demand
  mutable size_t wide_carry_bytes;
  int* head_flag;      // device word written by merge_path_head_check
  int* head_start;     // M + 1: first nonzero of the row each tile starts in
  unsigned int* scatter_stats;  // kernels::scatter_scratch_words words: what column_scatter_sample / _decide leave for the plan-less entry point
  int self_complete;   // 1: every tile head <= tpb -> one kernel, no carry-outs / fix-up (held plans only)
};

namespace {

int plan_compute(loops_merge_plan* p, const int* offsets, hipStream_t stream) {
  return kernels::launch_merge_path_coordinates(stream, offsets, p->rows, p->nnz, p->tpb * p->ipt, p->num_tiles,
                                                p->coords);
}

// Held plans only (one synchronisation at creation / refresh): can every tile finish its rows by itself?
int plan_classify(loops_merge_plan* p, const int* offsets, hipStream_t stream) {
  p->self_complete = 0;
  if (p->num_tiles <= 1) return 0;
  int err = kernels::launch_merge_path_head_check(stream, p->coords, p->num_tiles, p->rows, offsets, p->tpb, p->head_flag,
                                                  p->head_start);
  if (err) return err;
  int flag = 1;
  hipError_t e = hipMemcpyAsync(&flag, p->head_flag, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return static_cast<int>(e);
  p->self_complete = flag == 0 ? 1 : 0;
  return 0;
}

int plan_alloc(int rows, int nnz, int cfg, loops_merge_plan** out) {
  tile_shape s;
  if (!shape_of(cfg, &s)) return LOOPS_E_CONFIG;
  if (rows < 0 || nnz < 0) return LOOPS_E_BADARG;
  if (static_cast<long long>(rows) + nnz >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
  auto* p = new (std::nothrow) loops_merge_plan();
  if (!p) return static_cast<int>(hipErrorOutOfMemory);
  p->rows = rows; p->nnz = nnz; p->cfg = cfg; p->tpb = s.tpb; p->ipt = s.ipt;
  p->num_tiles = static_cast<int>(math::ceil_div(static_cast<long long>(rows) + nnz, static_cast<long long>(s.tpb) * s.ipt));
  p->capacity = p->num_tiles;
  const size_t m = static_cast<size_t>(p->num_tiles);
  const size_t bytes = (m + 1) * sizeof(coord_t) + (m + 2) * (sizeof(double) + sizeof(int)) + (m + 2) * sizeof(int) + kernels::scatter_scratch_words * sizeof(unsigned int);
  hipError_t e = hipMalloc(&p->base, bytes);
  if (e != hipSuccess) { delete p; return static_cast<int>(e); }
  p->coords = static_cast<coord_t*>(p->base);
  p->carry_val = reinterpret_cast<double*>(p->coords + (m + 1));
  p->carry_row = reinterpret_cast<int*>(p->carry_val + (m + 2));
  p->head_flag = p->carry_row + (m + 2);
  p->head_start = p->head_flag + 1;
  p->scatter_stats = reinterpret_cast<unsigned int*>(p->head_start + (m + 1));
  p->self_complete = 0;
  *out = p;
  return 0;
}

// Lazily grown scratch plans for the plan-less entry points: they rebuild the coordinates on every call, exactly
// like the reference wrapper constructs a preprocess_t per call (merge_path_flat.cuh:111-114), but without a
// hipMalloc per call.  One plan per (host thread, DEVICE, STREAM, tile config): the coordinates and carry-outs of a
// call are read by kernels that are still in flight when the call returns, so two calls may share a buffer only if
// stream order serialises them -- same device, same stream.  At most kScratchSlots plans are kept per thread; the
// least recently used one is released first (hipFree waits for the device, so nothing in flight loses its buffer).
struct scratch_slot {
  int device;
  hipStream_t stream;
  int cfg;
  unsigned long long used;
  loops_merge_plan* plan;
};
constexpr int kScratchSlots = 16;

void plan_release(loops_merge_plan* p) {
  if (!p) return;
  (void)hipFree(p->wide_carry);
  (void)hipFree(p->base);
  delete p;
}

// LOOPS_TILE_AUTO: the shape a held plan should have on THIS structure.  256 x 8 when the plan is self-completing with it
// (no tile starts more than 256 nonzeros inside a row: one kernel, no carry-outs -- band / FEM / short-row matrices, where
// 256 x 8 measured 25 us against 29-35 for the larger tiles); 512 x 8 otherwise (rows longer than a tile: every power-law
// input of profiles/r02_structure_sweep.json, C2 94.4 against 95.9 us).  One or two coordinate pre-passes + one stream
// synchronisation each, at plan creation only.
int plan_create_auto(int rows, int nnz, const int* offsets, hipStream_t stream, loops_merge_plan** out) {
  loops_merge_plan* p = nullptr;
  int err = plan_alloc(rows, nnz, LOOPS_TILE_256x8, &p);
  if (!err) err = plan_compute(p, offsets, stream);
  if (!err) err = plan_classify(p, offsets, stream);
  if (err) { plan_release(p); return err; }
  if (p->self_complete || p->num_tiles <= 1) { *out = p; return 0; }
  plan_release(p);
  p = nullptr;
  err = plan_alloc(rows, nnz, LOOPS_TILE_512x8, &p);
  if (!err) err = plan_compute(p, offsets, stream);
  if (!err) err = plan_classify(p, offsets, stream);
  if (err) { plan_release(p); return err; }
  *out = p;
  return 0;
}

// (namespace scope so that loops_release_scratch() can reach the calling thread's cache; trivially destructible:
// a thread that exits without releasing leaves its device buffers to process teardown)
thread_local scratch_slot scratch_slots[kScratchSlots] = {};
thread_local unsigned long long scratch_tick = 0;

int scratch_release_all() {
  int freed = 0;
  for (scratch_slot& c : scratch_slots) {
    if (c.plan) { plan_release(c.plan); ++freed; }
    c = scratch_slot{};
  }
  return freed;
}

loops_merge_plan* scratch_plan(int rows, int nnz, int cfg, hipStream_t stream, int* err) {
  auto& slots = scratch_slots;
  auto& tick = scratch_tick;
  tile_shape s;
  if (!shape_of(cfg, &s)) { *err = LOOPS_E_CONFIG; return nullptr; }
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) { *err = static_cast<int>(e); return nullptr; }
  scratch_slot* slot = nullptr;
  scratch_slot* victim = &slots[0];
  for (scratch_slot& c : slots) {
    if (c.plan && c.device == device && c.stream == stream && c.cfg == cfg) { slot = &c; break; }
    if (!c.plan) { if (victim->plan) victim = &c; }
    else if (victim->plan && c.used < victim->used) victim = &c;
  }
  if (!slot) {
    plan_release(victim->plan);
    *victim = scratch_slot{device, stream, cfg, 0, nullptr};
    slot = victim;
  }
  slot->used = ++tick;
  loops_merge_plan*& p = slot->plan;
  const int need = static_cast<int>(math::ceil_div(static_cast<long long>(rows) + nnz, static_cast<long long>(s.tpb) * s.ipt));
  if (p && p->capacity < need) { plan_release(p); p = nullptr; }
  if (!p) {
    *err = plan_alloc(rows, nnz, cfg, &p);
    if (*err) return nullptr;
  }
  // re-target the (possibly larger) allocation at this problem; the array bases were laid out
  // for `capacity` tiles and stay put
  p->rows = rows; p->nnz = nnz; p->num_tiles = need;
  return p;
}

// ------------------------------------------------------------------------- fused merge path
// stages: bit 0 = fused tile kernel, bit 1 = fix-up (3 = the whole SpMV)
template <int TPB, int IPT, bool PAD, int NT, typename T, bool MASK = false>  // false: the search-based tuning variants
int launch_fused(const loops_merge_plan* p, int num_tiles, int rows, int nnz, const int* off, const int* idx,
                 const T* val, const T* x, T* y, hipStream_t stream, int stages, bool planned = false) {
  kernels::merge_plan_view view{p->coords, p->carry_row, p->carry_val, num_tiles, p->self_complete != 0, p->head_start};
  return kernels::launch_merge_path_fused<TPB, IPT, PAD, NT, int, int, T, MASK>(stream, view, rows, nnz, off, idx, val, x, y,
                                                                                  stages, planned);
}

template <typename T>
int spmv_merge_path(const loops_merge_plan* p, int variant, int rows, int nnz, const int* off, const int* idx,
                    const T* val, const T* x, T* y, hipStream_t stream, int stages = 3, bool planned = false, int cols = 0) {
  const int m = static_cast<int>(math::ceil_div(static_cast<long long>(rows) + nnz, static_cast<long long>(p->tpb) * p->ipt));
  if (rows != p->rows || nnz != p->nnz) return LOOPS_E_BADARG;
  if (variant == LOOPS_VARIANT_PHASED) {
    // the default kernel with PHASED x gathers (kernels::merge_path_spmv_fused_phased): compiled for the two shapes it pays on
    if (cols <= 0) return LOOPS_E_BADARG;
    kernels::merge_plan_view view{p->coords, p->carry_row, p->carry_val, m, p->self_complete != 0, p->head_start};
    switch (p->cfg) {
      case LOOPS_TILE_512x8:
        return kernels::launch_merge_path_fused_phased<512, 8, int, int, T>(stream, view, rows, cols, nnz, off, idx, val, x, y, stages, planned);
      case LOOPS_TILE_256x16:
        return kernels::launch_merge_path_fused_phased<256, 16, int, int, T>(stream, view, rows, cols, nnz, off, idx, val, x, y, stages, planned);
      default: return LOOPS_E_CONFIG;
    }
  }
  // variant 0 = the default kernel (bit-mask split, padded LDS products, temporal loads).  Tuning aids, all with
  // the per-thread halving search of the first implementation: 4 = otherwise as 0; bit 0 = non-temporal
  // streaming loads, bit 1 = unpadded LDS product array (1, 2, 3)
  const bool nt = variant & 1, nopad = variant & 2, mask = variant == 0;
#define LOOPS_FUSED(TPB, IPT)                                                                                    \
  if (mask) return launch_fused<TPB, IPT, true, false, T, true>(p, m, rows, nnz, off, idx, val, x, y, stream, stages, planned); \
  if (!nopad && !nt) return launch_fused<TPB, IPT, true, false, T>(p, m, rows, nnz, off, idx, val, x, y, stream, stages); \
  if (!nopad && nt) return launch_fused<TPB, IPT, true, true, T>(p, m, rows, nnz, off, idx, val, x, y, stream, stages);   \
  if (nopad && !nt) return launch_fused<TPB, IPT, false, false, T>(p, m, rows, nnz, off, idx, val, x, y, stream, stages); \
  return launch_fused<TPB, IPT, false, true, T>(p, m, rows, nnz, off, idx, val, x, y, stream, stages);
  switch (p->cfg) {
    case LOOPS_TILE_256x8: { LOOPS_FUSED(256, 8) }
    case LOOPS_TILE_128x7: { LOOPS_FUSED(128, 7) }
    case LOOPS_TILE_256x7: { LOOPS_FUSED(256, 7) }
    case LOOPS_TILE_512x8: { LOOPS_FUSED(512, 8) }
    case LOOPS_TILE_256x16: { LOOPS_FUSED(256, 16) }
    case LOOPS_TILE_4x2: {
      // tiny tiles are for parity fixtures only: 64-thread workgroups, 4 "real" lanes would waste
      // the wavefront, so the fused kernel is not built for it; use the schedule-API kernel.
      return LOOPS_E_CONFIG;
    }
    default: return LOOPS_E_CONFIG;
  }
#undef LOOPS_FUSED
}

// ------------------------------------------------------------------- schedule-API launchers
template <std::size_t TPB, std::size_t IPT>
int launch_merge_dump(bool use_plan, int rows, int nnz, const int* off, unsigned* ts, int* owner, int* arow,
                      int* visits, hipStream_t stream) {
  using pre_t = schedule::merge_path::preprocess_t<TPB, IPT, int, int, std::size_t, std::size_t>;
  using layout_t = layout::csr<int, int>;
  pre_t meta(layout_t(off, rows, nnz), stream, use_plan ? pre_t::prepass_always : pre_t::prepass_never);
  const std::size_t m = meta.merge_tiles();
  if (m == 0) return 0;
  launch::non_cooperative(stream, kernels::merge_path_flat_dump<TPB, IPT, pre_t, int>, dim3(static_cast<unsigned>(m)),
                          dim3(TPB), meta, std::size_t(rows), std::size_t(nnz), const_cast<int*>(off), ts, owner, arow,
                          visits);
  (void)hipStreamSynchronize(stream);
  return last_error();
}

template <typename T>
int spmv_schedule_api(int schedule, int cfg, int rows, int cols, int nnz, const int* off, const int* idx,
                      const T* val, const T* x, T* y, hipStream_t stream) {
  if (rows == 0) return 0;
  const std::size_t R = rows, C = cols, N = nnz;
  switch (schedule) {
    case LOOPS_THREAD_MAPPED: return kernels::launch_thread_mapped(stream, R, C, N, off, idx, val, x, y, /*reference shape*/ true);
    case LOOPS_ORIGINAL: return kernels::launch_original(stream, R, C, N, off, idx, val, x, y);
    case LOOPS_GROUP_MAPPED: return kernels::launch_group_mapped_atomic(stream, R, C, N, off, idx, val, x, y);
    case LOOPS_WORK_ORIENTED: return kernels::launch_work_oriented_atomic(stream, R, C, N, off, idx, val, x, y);
    case LOOPS_FLAT_PARTITIONED: return kernels::launch_flat_partitioned<8>(stream, R, N, off, idx, val, x, y, /*reference shape*/ 1);
    case LOOPS_MERGE_PATH_FLAT: {
      switch (cfg) {
        case LOOPS_TILE_256x8: return kernels::launch_merge_path_atomic<256, 8>(stream, R, C, N, off, idx, val, x, y);
        case LOOPS_TILE_128x7: return kernels::launch_merge_path_atomic<128, 7>(stream, R, C, N, off, idx, val, x, y);
        case LOOPS_TILE_4x2: return kernels::launch_merge_path_atomic<4, 2>(stream, R, C, N, off, idx, val, x, y);
        case LOOPS_TILE_256x7: return kernels::launch_merge_path_atomic<256, 7>(stream, R, C, N, off, idx, val, x, y);
        case LOOPS_TILE_512x8: return kernels::launch_merge_path_atomic<512, 8>(stream, R, C, N, off, idx, val, x, y);
        case LOOPS_TILE_256x16: return kernels::launch_merge_path_atomic<256, 16>(stream, R, C, N, off, idx, val, x, y);
        default: return LOOPS_E_CONFIG;
      }
    }
    default: return LOOPS_E_BADARG;
  }
}

template <typename T>
int spmv_tuned(int schedule, int rows, int cols, int nnz, const int* off, const int* idx, const T* val, const T* x,
               T* y, hipStream_t stream) {
  int err = check_csr(rows, cols, nnz, off, idx, val, x, y);
  if (err) return err;
  if (rows == 0) return 0;
  switch (schedule) {
    case LOOPS_MERGE_PATH_FLAT: {
      // 512 x 8 merge tiles: the measured best shape of this kernel on MI355X (see launch_box.hxx)
      loops_merge_plan* p = scratch_plan(rows, nnz, LOOPS_TILE_512x8, stream, &err);
      if (!p) return err;
      if (p->num_tiles > 1) err = plan_compute(p, off, stream);  // a single-tile kernel derives its own coordinates
      if (err) return err;
      // large matrices over an x of 6 MB or more: a sample of the columns (two small kernels, ~5 us) decides ON THE DEVICE whether
      // this product gathers in phases (kernels::merge_path_spmv_fused_auto) -- the call stays asynchronous
      if (kernels::columns_worth_sampling(static_cast<long long>(nnz), static_cast<long long>(cols), static_cast<int>(sizeof(T)), /*timed_path=*/true)) {
        kernels::merge_plan_view view{p->coords, p->carry_row, p->carry_val, p->num_tiles};
        return kernels::launch_merge_path_fused_auto<512, 8, int, int, T>(stream, view, rows, cols, nnz, off, idx, val, x, y, p->scatter_stats);
      }
      return spmv_merge_path<T>(p, 0, rows, nnz, off, idx, val, x, y, stream);
    }
    case LOOPS_THREAD_MAPPED:  // the schedule as given (a thread owns whole rows), the row's atoms 16 / 4 at a time
      return kernels::launch_thread_mapped(stream, std::size_t(rows), std::size_t(cols), std::size_t(nnz), off, idx, val, x, y);
    case LOOPS_ORIGINAL:
      return spmv_schedule_api<T>(schedule, 0, rows, cols, nnz, off, idx, val, x, y, stream);
    case LOOPS_WORK_ORIENTED: {
      loops_merge_plan* p = scratch_plan(rows, nnz, LOOPS_TILE_DEFAULT, stream, &err);
      if (!p) return err;
      err = plan_compute(p, off, stream);
      if (err) return err;
      kernels::merge_plan_view view{p->coords, p->carry_row, p->carry_val, p->num_tiles};
      return kernels::launch_work_oriented_fused<256, 8, true>(stream, view, rows, nnz, off, idx, val, x, y);
    }
    case LOOPS_GROUP_MAPPED:
#ifdef LOOPS_GROUP_MAPPED_SEARCH  // A/B aid (never defined in a product build): the search-based split
      return kernels::launch_group_mapped_fused<256, 8, true, int, int, T, false>(stream, rows, nnz, off, idx, val, x, y);
#else
      return kernels::launch_group_mapped_fused<256, 8, true>(stream, rows, nnz, off, idx, val, x, y);
#endif
    case LOOPS_FLAT_PARTITIONED: {
      // atomic kernel accumulates into y: zero it on the stream first
      hipError_t e = hipMemsetAsync(y, 0, sizeof(T) * static_cast<size_t>(rows), stream);
      if (e != hipSuccess) return static_cast<int>(e);
      return kernels::launch_flat_partitioned<8>(stream, std::size_t(rows), std::size_t(nnz), off, idx, val, x, y);
    }
    default: return LOOPS_E_BADARG;
  }
}

// ------------------------------------------------------------------------------------- SpMM
template <typename T>
int spmm_merge_path(const loops_merge_plan* p, int rows, int cols, int nnz, const int* off, const int* idx, const T* val,
                    const T* B, int n, T* C, hipStream_t stream) {
  if (rows != p->rows || nnz != p->nnz) return LOOPS_E_BADARG;
  if (p->tpb != 256 || p->ipt != 8) return LOOPS_E_CONFIG;
  if (static_cast<unsigned long long>(cols) * static_cast<unsigned long long>(n) >= (1ull << 32)) return LOOPS_E_RANGE;
  const size_t need = static_cast<size_t>(p->num_tiles) * static_cast<size_t>(n) * sizeof(T);
  if (need > p->wide_carry_bytes) {
    (void)hipFree(p->wide_carry);
    p->wide_carry = nullptr;
    p->wide_carry_bytes = 0;
    hipError_t e = hipMalloc(&p->wide_carry, need);
    if (e != hipSuccess) return static_cast<int>(e);
    p->wide_carry_bytes = need;
  }
  kernels::merge_plan_view view{p->coords, p->carry_row, p->carry_val, p->num_tiles};
  return kernels::launch_merge_path_spmm<256, 8>(stream, view, static_cast<T*>(p->wide_carry), rows, cols, nnz, off, idx, val,
                                                 B, n, static_cast<size_t>(n), C, static_cast<size_t>(n));
}

template <typename T>
int spmm_tuned(int schedule, int rows, int cols, int nnz, const int* off, const int* idx, const T* val, const T* B,
               int n, T* C, hipStream_t stream) {
  if (n < 0 || rows < 0) return LOOPS_E_BADARG;
  if (n == 0 || rows == 0) return 0;  // C is empty: nothing to check, nothing to write
  int err = check_csr(rows, cols, nnz, off, idx, val, B, C);
  if (err) return err;
  switch (schedule) {
    case LOOPS_MERGE_PATH_FLAT: {
      loops_merge_plan* p = scratch_plan(rows, nnz, LOOPS_TILE_DEFAULT, stream, &err);
      if (!p) return err;
      err = plan_compute(p, off, stream);
      if (!err) err = spmm_merge_path<T>(p, rows, cols, nnz, off, idx, val, B, n, C, stream);
      return err;
    }
    case LOOPS_THREAD_MAPPED:
      return kernels::launch_thread_mapped_spmm(stream, rows, off, idx, val, B, n, static_cast<size_t>(n), C,
                                                static_cast<size_t>(n));
    default: return LOOPS_E_CONFIG;
  }
}

// work_oriented over a held plan (256 x 8 tiles): the persistent kernel + its fix-up, no coordinate pre-pass per call
template <typename T>
int spmv_work_oriented_planned(const loops_merge_plan* plan, int rows, int cols, int nnz, const int* off, const int* idx,
                                      const T* val, const T* x, T* y, hipStream_t stream) {
  if (!plan) return LOOPS_E_BADARG;
  int err = check_csr(rows, cols, nnz, off, idx, val, x, y);
  if (err) return err;
  if (rows != plan->rows || nnz != plan->nnz) return LOOPS_E_BADARG;
  if (plan->tpb != 256 || plan->ipt != 8) return LOOPS_E_CONFIG;
  if (rows == 0) return 0;
  kernels::merge_plan_view view{plan->coords, plan->carry_row, plan->carry_val, plan->num_tiles};
  return kernels::launch_work_oriented_fused<256, 8, true>(stream, view, rows, nnz, off, idx, val, x, y);
}

// ------------------------------------------------------------------------ other formats
template <typename T>
int spmv_bcsr(int R, int C, int mode, int rows, int num_block_rows, int num_blocks, const int* block_offsets,
              const int* block_cols, const T* block_values, const T* x_padded, T* y, hipStream_t s) {
  if (!block_offsets || !y || rows < 0 || num_block_rows < 0 || num_blocks < 0) return LOOPS_E_BADARG;
  if (num_blocks > 0 && (!block_cols || !block_values || !x_padded)) return LOOPS_E_BADARG;
  if (num_block_rows == 0) return 0;
  if (mode == 3)  // tuned: the MFMA kernel where it exists (4 x 4 fp32), the coalesced lane-group kernel otherwise
    mode = (R == 4 && C == 4 && std::is_same<T, float>::value) ? 1 : 2;
  if (mode == 2 || mode >= 100000) {
    // coalesced lane-group kernels (bcsr_vector_mapped_spmv / bcsr_block_mapped_spmv).  2: automatic shape;
    // tuning aid 100000 + 100 h + u: h in {1, 4, 16} blocks of a block-row per step, u in {1, 2, 4} steps in flight
    int h = 0, u = 0;
    if (mode >= 100000) { h = (mode - 100000) / 100; u = (mode - 100000) % 100; }
    if (mode >= 100000 && ((h != 1 && h != 4 && h != 16) || (u != 1 && u != 2 && u != 4))) return LOOPS_E_BADARG;
#define LOOPS_BCSR_COALESCED(RR, CC)                                                                                        \
    if (R == RR && C == CC)                                                                                                \
      return kernels::launch_bcsr_coalesced<RR, CC, T>(s, rows, num_block_rows, num_blocks, block_offsets, block_cols,     \
                                                      block_values, x_padded, y, h, u);
    LOOPS_BCSR_COALESCED(2, 2) LOOPS_BCSR_COALESCED(3, 3) LOOPS_BCSR_COALESCED(4, 4) LOOPS_BCSR_COALESCED(8, 8)
#undef LOOPS_BCSR_COALESCED
    return LOOPS_E_CONFIG;
  }
  if (mode == 1 || (mode > 10 && mode < 20) || mode >= 100) {
    // MFMA path.  1: automatic shape; tuning aids: 1u = one block per block-row per step, u steps in
    // flight; 100 + 10 h + u = h blocks of a block-row per step (1, 2, 4, 8, 16), u steps in flight;
    // + 1000 g = g consecutive groups of block-rows per wavefront (1 .. 64; 0 = automatic)
    if constexpr (!std::is_same<T, float>::value) {
      return LOOPS_E_CONFIG;  // the MFMA kernel is fp32 (v_mfma_f32_4x4x1); fp64 blocks take the register path (mode 0)
    } else {
      if (R != 4 || C != 4) return LOOPS_E_CONFIG;
      int h = 0, u = 0, g = 0;
      if (mode >= 1000) { g = mode / 1000; mode %= 1000; if (mode < 100) return LOOPS_E_BADARG; }
      if (mode > 10 && mode < 20) { h = 1; u = mode - 10; }
      if (mode >= 100) { h = (mode - 100) / 10; u = (mode - 100) % 10; }
      if (mode != 1 && ((h != 1 && h != 2 && h != 4 && h != 8 && h != 16) || (u != 1 && u != 2 && u != 4 && u != 8)))
        return LOOPS_E_BADARG;
      if (g > 64) return LOOPS_E_BADARG;
      return kernels::launch_bcsr4x4_mfma(s, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values,
                                          x_padded, y, u, h, g);
    }
  }
  if (mode != 0) return LOOPS_E_BADARG;
  if (R == 2 && C == 2) return kernels::launch_bcsr_thread_mapped<2, 2>(s, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values, x_padded, y);
  if (R == 3 && C == 3) return kernels::launch_bcsr_thread_mapped<3, 3>(s, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values, x_padded, y);
  if (R == 4 && C == 4) return kernels::launch_bcsr_thread_mapped<4, 4>(s, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values, x_padded, y);
  if (R == 8 && C == 8) return kernels::launch_bcsr_thread_mapped<8, 8>(s, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values, x_padded, y);
  return LOOPS_E_CONFIG;
}

template <typename T>
int spmv_coo(int mode, int rows, int cols, int nnz, const int* row_indices, const int* col_indices, const T* values,
             const T* x, T* y, hipStream_t s) {
  if (rows < 0 || cols < 0 || nnz < 0 || !y || (nnz > 0 && (!row_indices || !col_indices || !values || !x)))
    return LOOPS_E_BADARG;
  if (rows == 0) return 0;
  if (mode == 0)  // reference shape: the caller zero-fills y (coo_thread_mapped.cuh:95-97 convention)
    return kernels::launch_coo_atom(s, static_cast<size_t>(nnz), row_indices, col_indices, values, x, y);
  if (mode != 1) return LOOPS_E_BADARG;
  hipError_t e = hipMemsetAsync(y, 0, sizeof(T) * static_cast<size_t>(rows), s);
  if (e != hipSuccess) return static_cast<int>(e);
  return kernels::launch_coo_runs(s, static_cast<size_t>(nnz), row_indices, col_indices, values, x, y);
}

template <typename T>
int spmv_ell(int mode, int rows, int cols, int pitch, const int* indices, const T* values, const T* x, T* y,
             hipStream_t s) {
  if (rows < 0 || cols < 0 || pitch < 0 || !y || (rows > 0 && pitch > 0 && (!indices || !values || !x))) return LOOPS_E_BADARG;
  if (rows == 0) return 0;
  if (mode == 0) return kernels::launch_ell_thread(s, static_cast<size_t>(rows), static_cast<size_t>(pitch), indices, values, x, y);
  if (mode == 1) return kernels::launch_ell_row_split(s, static_cast<size_t>(rows), static_cast<size_t>(pitch), indices, values, x, y);
  if (mode != 2) return LOOPS_E_BADARG;
  // merge_path_flat over the ELL cells on the fused engine (ell_merge_path.cuh:76-125): coordinates from the row-end
  // functor, tile kernel + fix-up; 512 x 8 (fp32) / 512 x 4 (fp64) tiles, the merge_path launch box
  if (static_cast<long long>(rows) * pitch + rows >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
  constexpr int IPT = sizeof(T) > 4 ? 4 : 8;
  int err = 0;
  // scratch sized by merge tiles of 512 * IPT items: the 512 x 8 slot for fp32, the 256 x 8 slot (= 512 x 4 items) for fp64
  loops_merge_plan* p = scratch_plan(rows, rows * pitch, sizeof(T) > 4 ? LOOPS_TILE_256x8 : LOOPS_TILE_512x8, s, &err);
  if (!p) return err;
  const int m = static_cast<int>(math::ceil_div(static_cast<long long>(rows) * pitch + rows, 512ll * IPT));
  if (m > p->capacity) return LOOPS_E_RANGE;  // (cannot happen: the slot was sized for exactly this tile count)
  err = kernels::launch_merge_path_coordinates_ell(s, rows, pitch, 512 * IPT, m, p->coords);
  if (err) return err;
  kernels::merge_plan_view view{p->coords, p->carry_row, p->carry_val, m};
  return kernels::launch_ell_merge_path_fused<512, IPT>(s, view, rows, pitch, indices, values, x, y);
}

template <typename T>
int spmv_csc(int mode, int rows, int cols, int nnz, const int* col_offsets, const int* row_indices, const T* values,
             const T* x, T* y, hipStream_t s) {
  if (rows < 0 || cols < 0 || nnz < 0 || !y || !col_offsets || (nnz > 0 && (!row_indices || !values || !x)))
    return LOOPS_E_BADARG;
  if (rows == 0) return 0;
  if (mode == 0)  // reference shape: the caller zero-fills y
    return kernels::launch_csc_column(s, static_cast<size_t>(cols), col_offsets, row_indices, values, x, y);
  if (mode != 1) return LOOPS_E_BADARG;
  hipError_t e = hipMemsetAsync(y, 0, sizeof(T) * static_cast<size_t>(rows), s);
  if (e != hipSuccess) return static_cast<int>(e);
  return kernels::launch_csc_nonzero_split(s, cols, nnz, col_offsets, row_indices, values, x, y);
}

template <typename T>
int spmv_dia(int mode, int rows, int cols, int num_diagonals, size_t stride, const int* diag_offsets, const T* values,
             const T* x, T* y, hipStream_t s) {
  if (rows < 0 || cols < 0 || num_diagonals < 0 || !y || stride < static_cast<size_t>(rows)) return LOOPS_E_BADARG;
  if (num_diagonals > 0 && rows > 0 && (!diag_offsets || !values || !x)) return LOOPS_E_BADARG;
  if (rows == 0) return 0;
  if (mode == 0) return kernels::launch_dia_thread(s, rows, cols, stride, num_diagonals, diag_offsets, values, x, y);
  if (mode == 1) return kernels::launch_dia_row4(s, rows, cols, stride, num_diagonals, diag_offsets, values, x, y);
  return LOOPS_E_BADARG;
}

}  // namespace

namespace {
template <typename T>
int fanout_peers(int num_peers, T* const* h_peer_y, kernels::peer_fanout<T>* peers) {
  if (num_peers < 0 || num_peers > kernels::max_peers || (num_peers > 0 && !h_peer_y)) return LOOPS_E_BADARG;
  *peers = kernels::peer_fanout<T>{};
  peers->count = num_peers;
  for (int p = 0; p < num_peers; ++p) {
    if (!h_peer_y[p]) return LOOPS_E_BADARG;
    peers->base[p] = h_peer_y[p];
  }
  return 0;
}

template <typename T>
int merge_path_fanout(const loops_merge_plan* plan, int rows, int cols, int nnz, const int* offsets, const int* indices,
                             const T* values, const T* x, T* y, int num_peers, T* const* h_peer_y, hipStream_t stream) {
  if (!plan) return LOOPS_E_BADARG;
  kernels::peer_fanout<T> peers;
  int err = fanout_peers<T>(num_peers, h_peer_y, &peers);
  if (!err) err = check_csr(rows, cols, nnz, offsets, indices, values, x, y);
  if (err) return err;
  if (rows != plan->rows || nnz != plan->nnz) return LOOPS_E_BADARG;
  if (plan->tpb != 512 || plan->ipt != 8) return LOOPS_E_CONFIG;  // compiled for the merge_path launch box only
  if (rows == 0) return 0;
  kernels::merge_plan_view view{plan->coords, plan->carry_row, plan->carry_val, plan->num_tiles};
  constexpr int IPT = sizeof(T) > 4 ? 4 : 8;  // (an fp64 plan of shape 512 x 8 is walked as 512 x 4 + 512 x 4: not compiled)
  if constexpr (sizeof(T) > 4) return LOOPS_E_CONFIG;
  else return kernels::launch_merge_path_fused_fanout<512, IPT>(stream, view, rows, nnz, offsets, indices, values, x, y, peers);
}

}  // namespace

// ------------------------------------------------------------------------------------ panel-binned layout
static_assert(kernels::panel_e_badarg == LOOPS_E_BADARG && kernels::panel_e_range == LOOPS_E_RANGE, "panel_binned_create speaks the C ABI's error codes");
struct loops_panel_plan : kernels::panel_binned_storage {};  // (the builder and the owned arrays: include/loops/kernels/panel_binned.hxx)

namespace {

void panel_free(loops_panel_plan* p) { delete p; }

template <typename T>
kernels::panel_binned_view<T> panel_view(const loops_panel_plan* p) { return p->view<T>(); }

template <typename T>
int panel_create(int rows, int cols, int nnz, const int* offsets, const int* indices, const T* values, hipStream_t st,
                 loops_panel_plan** out, int subband_rows = 0, int panel_cols = 0, int compact = -1) {
  if (!out) return LOOPS_E_BADARG;
  auto* p = new (std::nothrow) loops_panel_plan();
  if (!p) return static_cast<int>(hipErrorOutOfMemory);
  const int err = kernels::panel_binned_create<int, int, T>(st, rows, cols, nnz, offsets, indices, values, subband_rows, panel_cols, compact, *p);
  if (err) { delete p; return err; }  // (panel_e_badarg / panel_e_range are LOOPS_E_BADARG / LOOPS_E_RANGE)
  *out = p;
  return 0;
}

template <typename T>
int panel_spmv(const loops_panel_plan* p, int stages, const T* x, T* y, hipStream_t st) {
  if (!p || p->vbytes != static_cast<int>(sizeof(T))) return LOOPS_E_BADARG;
  if (p->rows == 0) return 0;
  if (!y || (p->nnz > 0 && !x)) return LOOPS_E_BADARG;
  return kernels::launch_panel_binned<T>(st, panel_view<T>(p), x, y, stages);
}

template <typename T>
int panel_refresh(loops_panel_plan* p, const T* values, hipStream_t st) {
  if (!p || p->vbytes != static_cast<int>(sizeof(T)) || (p->nnz > 0 && !values)) return LOOPS_E_BADARG;
  if (p->rows == 0 || p->padded == 0) return 0;
  hipLaunchKernelGGL((kernels::panel::refresh_values<T>), dim3(math::ceil_div(p->padded, 256)), dim3(256), 0, st, p->perm, values,
                     p->padded, static_cast<T*>(p->val));
  return static_cast<int>(hipGetLastError());
}

}  // namespace

// ------------------------------------------------------------------------------------ row-band layout
#include "abi_rowband.inc"

// ------------------------------------------------------------------ SpMV plan: tile shape AND layout chosen at plan time
// What a caller that performs many products with one matrix should hold (loops_spmv_plan_*): the merge-path plan of the
// unmodified CSR in the best tile shape, or -- if the caller allows a copy and it is measurably faster -- a re-ordered copy of
// the matrix: row-band (y accumulators in LDS, coalescing gathers: x of a few MB, column locality) or panel-binned (x panels
// in LDS: x far beyond the L2).
struct loops_spmv_plan {
  int rows, cols, nnz, vbytes, flags;
  int layout;                   // LOOPS_LAYOUT_CSR / LOOPS_LAYOUT_ROW_BAND / LOOPS_LAYOUT_PANEL_BINNED
  loops_merge_plan* merge;      // held for LOOPS_LAYOUT_CSR
  loops_rowband_plan* band;     // held for LOOPS_LAYOUT_ROW_BAND
  loops_panel_plan* panel;      // held for LOOPS_LAYOUT_PANEL_BINNED
  int merge_variant;            // LOOPS_LAYOUT_CSR: 0 = the default kernel, LOOPS_VARIANT_PHASED = phased x gathers
  float ms_phased;              // measured ms per product of the best phased candidate; -1 = not timed
  float ms[4];                  // measured ms per product: CSR 256 x 8, CSR 512 x 8, row-band, panel-binned; -1 = not timed
};

namespace {

void spmv_plan_free(loops_spmv_plan* p) {
  if (!p) return;
  plan_release(p->merge);
  rowband_free(p->band);
  panel_free(p->panel);
  delete p;
}

template <typename fn_t>
int time_ms(hipStream_t st, int repeats, float* ms, fn_t&& run) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return static_cast<int>(hipGetLastError());
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return static_cast<int>(hipGetLastError()); }
  int err = 0;
  for (int it = 0; !err && it < 2; ++it) err = run();
  if (!err) {
    (void)hipEventRecord(e0, st);
    for (int it = 0; !err && it < repeats; ++it) err = run();
    (void)hipEventRecord(e1, st);
    if (!err) err = static_cast<int>(hipEventSynchronize(e1));
    if (!err) err = static_cast<int>(hipEventElapsedTime(ms, e0, e1));
    *ms /= static_cast<float>(repeats);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return err;
}

template <typename T>
int spmv_plan_create(int rows, int cols, int nnz, const int* off, const int* idx, const T* val, int flags, int repeats,
                     hipStream_t st, loops_spmv_plan** out) {
  if (!out || !off || rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && (!idx || !val))) return LOOPS_E_BADARG;
  if (static_cast<long long>(rows) + nnz >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
  if (repeats < 1) repeats = 10;
  auto* p = new (std::nothrow) loops_spmv_plan();
  if (!p) return static_cast<int>(hipErrorOutOfMemory);
  p->rows = rows; p->cols = cols; p->nnz = nnz; p->vbytes = static_cast<int>(sizeof(T)); p->flags = flags;
  p->layout = LOOPS_LAYOUT_CSR;
  p->merge_variant = 0;
  p->ms_phased = -1.f;
  p->ms[0] = p->ms[1] = p->ms[2] = p->ms[3] = -1.f;
  const bool measure = (flags & LOOPS_PLAN_MEASURE) != 0 && rows > 0 && nnz > 0;
  const bool may_copy = (flags & LOOPS_PLAN_ALLOW_COPY) != 0 && rows > 0 && nnz > 0;
  const long long x_bytes = static_cast<long long>(cols) * static_cast<long long>(sizeof(T));
  int err = 0;
  if (!measure) {
    // structural choice only: tile by the self-completing test; a copy when the matrix is a candidate for one -- panel-binned
    // when x exceeds 1.5 per-XCD L2s (the fastest layout on every such input with scattered columns), row-band (4-byte values)
    // for an x between 2 and 6 MB under rows of >= 8 nonzeros (C2: 37 against 83 us).  Structure cannot see column LOCALITY
    // (a narrow band runs faster from the CSR as it is, a wide one from the row-band copy whatever the size of x): callers who
    // cannot rule it out should pass LOOPS_PLAN_MEASURE.
    err = plan_create_auto(rows, nnz, off, st, &p->merge);
    if (!err && may_copy && x_bytes >= (2ll << 20)) {
      if (x_bytes > (6ll << 20)) {
        err = panel_create<T>(rows, cols, nnz, off, idx, val, st, &p->panel);
        if (!err) p->layout = LOOPS_LAYOUT_PANEL_BINNED;
      } else if (sizeof(T) == 4 && nnz / (rows > 0 ? rows : 1) >= 8) {
        if constexpr (sizeof(T) == 4) err = rowband_create_plan<T>(rows, cols, nnz, off, idx, val, 0, 0, st, &p->band);
        if (!err) p->layout = LOOPS_LAYOUT_ROW_BAND;
      }
      if (!err && p->layout != LOOPS_LAYOUT_CSR) { plan_release(p->merge); p->merge = nullptr; }
      else if (err == LOOPS_E_RANGE || err == LOOPS_E_CONFIG) err = 0;  // the copy does not fit 32-bit positions: stay on the CSR
      else if (err == static_cast<int>(hipErrorOutOfMemory)) {            // no memory for the OPTIONAL copy: stay on the CSR, as
        (void)hipGetLastError();                                          // the measured path does (clear the sticky error)
        err = 0;
      }
    }
    // still on the CSR: columns that look scattered over an x of 3 MB or more (kernels::columns_look_scattered: 65 536 sampled
    // pairs, one small kernel) get 512 x 8 tiles with phased x gathers -- the guess the plan-less C++ wrapper goes by
    if (!err && p->layout == LOOPS_LAYOUT_CSR && rows > 0 && nnz > 0 &&
        kernels::columns_worth_sampling(static_cast<long long>(nnz), static_cast<long long>(cols), static_cast<int>(sizeof(T)))) {
      unsigned int* scratch = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&scratch), kernels::scatter_scratch_words * sizeof(unsigned int)) == hipSuccess) {
        if (kernels::columns_look_scattered(st, idx, static_cast<long long>(nnz), static_cast<long long>(cols), static_cast<int>(sizeof(T)),
                                            scratch)) {
          loops_merge_plan* m = nullptr;
          int perr = plan_alloc(rows, nnz, LOOPS_TILE_512x8, &m);
          if (!perr) perr = plan_compute(m, off, st);
          if (!perr) perr = plan_classify(m, off, st);
          if (!perr && m->num_tiles > 1) {
            plan_release(p->merge);
            p->merge = m;
            p->merge_variant = LOOPS_VARIANT_PHASED;
          } else {
            plan_release(m);
          }
        }
        (void)hipFree(scratch);
      } else {
        (void)hipGetLastError();
      }
    }
    if (err) { spmv_plan_free(p); return err; }
    *out = p;
    return 0;
  }
  // measured choice: the candidates are timed on this device with this matrix (x values do not matter to the time)
  T *x = nullptr, *y = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&x), sizeof(T) * static_cast<size_t>(cols > 0 ? cols : 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&y), sizeof(T) * static_cast<size_t>(rows));
  if (e == hipSuccess) e = hipMemsetAsync(x, 0, sizeof(T) * static_cast<size_t>(cols > 0 ? cols : 1), st);
  err = static_cast<int>(e);
  loops_merge_plan* best = nullptr;
  float best_ms = 0.f;
  const int shapes[2] = {LOOPS_TILE_256x8, LOOPS_TILE_512x8};
  for (int i = 0; !err && i < 2; ++i) {
    loops_merge_plan* m = nullptr;
    err = plan_alloc(rows, nnz, shapes[i], &m);
    if (!err) err = plan_compute(m, off, st);
    if (!err) err = plan_classify(m, off, st);
    float ms = 0.f;
    if (!err) err = time_ms(st, repeats, &ms, [&]() { return spmv_merge_path<T>(m, 0, rows, nnz, off, idx, val, x, y, st, 3, true); });
    if (err) { plan_release(m); break; }
    p->ms[i] = ms;
    // 512 x 8 must be measurably (> 1 %) faster to displace 256 x 8 and vice versa: ties go to the structural choice
    const bool better = !best || ms < 0.99f * best_ms;
    if (better) { plan_release(best); best = m; best_ms = ms; }
    else plan_release(m);
  }
  p->merge = best;
  // The same CSR with PHASED x gathers (kernels::merge_path_spmv_fused_phased: no copy, same bits): a candidate where it can
  // pay at all -- more than one tile and an x of at least a quarter of one XCD's L2 -- adopted when it is measurably
  // (> 2 %) faster than the best plain shape (C2, x = 4 MB: 1.14 x; scattered columns over 8-64 MB: 1.3-1.9 x).
  if (!err && best && best->num_tiles > 1 && x_bytes >= (1ll << 20)) {
    const int pshapes[2] = {LOOPS_TILE_512x8, LOOPS_TILE_256x16};
    for (int i = 0; !err && i < 2; ++i) {
      loops_merge_plan* m = nullptr;
      err = plan_alloc(rows, nnz, pshapes[i], &m);
      if (!err) err = plan_compute(m, off, st);
      if (!err) err = plan_classify(m, off, st);
      float ms = 0.f;
      if (!err)
        err = time_ms(st, repeats, &ms, [&]() { return spmv_merge_path<T>(m, LOOPS_VARIANT_PHASED, rows, nnz, off, idx, val, x, y, st, 3, true, cols); });
      if (err) { plan_release(m); break; }
      if (ms > 0.f && (p->ms_phased < 0.f || ms < p->ms_phased)) p->ms_phased = ms;
      if (ms > 0.f && ms < 0.98f * best_ms) {
        plan_release(p->merge);
        p->merge = best = m;
        best_ms = ms;
        p->merge_variant = LOOPS_VARIANT_PHASED;
      } else {
        plan_release(m);
      }
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (!err && may_copy && x_bytes >= (1ll << 20)) {
      // second candidate: the row-band copy (4-byte values; y accumulators in LDS, column-sorted gathers), its kernel shape
      // tuned by measurement too; adopted when >= 5 % faster than the best CSR shape (the copy doubles the matrix's footprint)
      loops_rowband_plan* rb = nullptr;
      int rerr = rowband_create_plan<T>(rows, cols, nnz, off, idx, val, 0, 0, st, &rb);
      if (!rerr) {
        float ms2[2] = {-1.f, -1.f};
        rerr = kernels::rowband_tune<T>(st, *rb, repeats, ms2);
        const float ms = rb->waves == 16 ? ms2[1] : ms2[0];
        if (!rerr) p->ms[2] = ms;
        if (!rerr && ms > 0.f && ms < 0.95f * best_ms) {
          p->band = rb;
          p->layout = LOOPS_LAYOUT_ROW_BAND;
          plan_release(p->merge);
          p->merge = nullptr;
          rb = nullptr;
        }
      }
      rowband_free(rb);
      if (rerr && rerr != LOOPS_E_RANGE && rerr != LOOPS_E_CONFIG && rerr != static_cast<int>(hipErrorOutOfMemory)) err = rerr;
      if (rerr == static_cast<int>(hipErrorOutOfMemory)) (void)hipGetLastError();  // no room for the copy: stay on the CSR
    }
  }
  if (!err && may_copy && x_bytes >= (2ll << 20)) {
    // third candidate: the panel-binned copy (x panels in LDS, no gather leaves the CU): adopted like the row-band copy
    loops_panel_plan* pp = nullptr;
    int perr = panel_create<T>(rows, cols, nnz, off, idx, val, st, &pp);
    if (!perr) {
      float ms = 0.f;
      perr = time_ms(st, repeats, &ms, [&]() { return panel_spmv<T>(pp, 3, x, y, st); });
      if (!perr) p->ms[3] = ms;
      const float incumbent = p->band ? p->ms[2] : best_ms;
      if (!perr && ms < 0.95f * incumbent && ms < 0.95f * best_ms) {
        p->panel = pp;
        p->layout = LOOPS_LAYOUT_PANEL_BINNED;
        plan_release(p->merge);
        p->merge = nullptr;
        rowband_free(p->band);
        p->band = nullptr;
        pp = nullptr;
      }
    }
    panel_free(pp);
    if (perr && perr != LOOPS_E_RANGE && perr != LOOPS_E_CONFIG && perr != static_cast<int>(hipErrorOutOfMemory)) err = perr;
    if (perr == static_cast<int>(hipErrorOutOfMemory)) (void)hipGetLastError();
  }
  if (!err) err = static_cast<int>(hipStreamSynchronize(st));
  (void)hipFree(x);
  (void)hipFree(y);
  if (err) { spmv_plan_free(p); return err; }
  *out = p;
  return 0;
}

template <typename T>
int spmv_planned(const loops_spmv_plan* p, const int* off, const int* idx, const T* val, const T* x, T* y, hipStream_t st) {
  if (!p || p->vbytes != static_cast<int>(sizeof(T))) return LOOPS_E_BADARG;
  if (p->rows == 0) return 0;
  if (!y || (p->nnz > 0 && !x)) return LOOPS_E_BADARG;
  if (p->layout == LOOPS_LAYOUT_ROW_BAND) {
    if constexpr (sizeof(T) == 4) return rowband_spmv<T>(p->band, 3, x, y, st);
    else return LOOPS_E_CONFIG;
  }
  if (p->layout == LOOPS_LAYOUT_PANEL_BINNED) return panel_spmv<T>(p->panel, 3, x, y, st);
  int err = check_csr(p->rows, p->cols, p->nnz, off, idx, val, x, y);
  if (err) return err;
  return spmv_merge_path<T>(p->merge, p->merge_variant, p->rows, p->nnz, off, idx, val, x, y, st, 3, /*planned=*/true, p->cols);
}

}  // namespace

// ------------------------------------------------------------------ CSC plan: transpose the storage once, then a SpMV plan
struct loops_csc_plan {
  int rows, cols, nnz, vbytes;
  int *off, *idx, *perm;   // the matrix as CSR (owned) and, per CSR position, the CSC position its value comes from
  void* val;
  loops_spmv_plan* inner;
};

namespace {

void csc_plan_free(loops_csc_plan* p) {
  if (!p) return;
  spmv_plan_free(p->inner);
  (void)hipFree(p->off); (void)hipFree(p->idx); (void)hipFree(p->perm); (void)hipFree(p->val);
  delete p;
}

/// `col_off` != nullptr: CSC (column offsets + row indices); else COO triplets (row_idx, col_idx), any order.
template <typename T>
int csc_plan_create(int rows, int cols, int nnz, const int* col_off, const int* row_idx, const T* val, int flags, int repeats,
                    hipStream_t st, loops_csc_plan** out, const int* col_idx = nullptr) {
  if (!out || (!col_off && nnz > 0 && !col_idx) || rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && (!row_idx || !val))) return LOOPS_E_BADARG;
  if (static_cast<long long>(rows) + nnz >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
  auto* p = new (std::nothrow) loops_csc_plan();
  if (!p) return static_cast<int>(hipErrorOutOfMemory);
  p->rows = rows; p->cols = cols; p->nnz = nnz; p->vbytes = static_cast<int>(sizeof(T));
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->off), sizeof(int) * (static_cast<size_t>(rows) + 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->idx), sizeof(int) * n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->perm), sizeof(int) * n);
  if (e == hipSuccess) e = hipMalloc(&p->val, sizeof(T) * n);
  unsigned long long *keys_in = nullptr, *keys_out = nullptr;
  int* pos = nullptr;
  void* cub_temp = nullptr;
  size_t sort_bytes = 0, scan_bytes = 0;
  int end_bit = 33;
  while (end_bit < 64 && (static_cast<unsigned long long>(rows > 0 ? rows - 1 : 0) >> (end_bit - 32)) != 0) ++end_bit;
  if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_in, keys_out, pos, p->perm, nnz, 0, end_bit, st);
  if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, p->off, p->off, rows + 1, st);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&keys_in), sizeof(unsigned long long) * n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&keys_out), sizeof(unsigned long long) * n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&pos), sizeof(int) * n);
  size_t cub_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  if (e == hipSuccess) e = hipMalloc(&cub_temp, cub_bytes > 0 ? cub_bytes : 16);
  int* bad = nullptr;   // set by the key kernels when an index lies outside the matrix
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&bad), sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(int), st);
  if (e == hipSuccess) e = hipMemsetAsync(p->off, 0, sizeof(int) * (static_cast<size_t>(rows) + 1), st);
  if (e == hipSuccess && nnz > 0) {
    const dim3 grid(math::ceil_div(nnz, 256)), block(256);
    if (col_off) hipLaunchKernelGGL((kernels::csc_transpose_keys<int, int>), grid, block, 0, st, rows, cols, nnz, col_off, row_idx, keys_in, pos, p->off, bad);
    else hipLaunchKernelGGL((kernels::coo_transpose_keys<int>), grid, block, 0, st, rows, cols, nnz, row_idx, col_idx, keys_in, pos, p->off, bad);
    size_t bytes = cub_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(cub_temp, bytes, keys_in, keys_out, pos, p->perm, nnz, 0, end_bit, st);
    if (e == hipSuccess) hipLaunchKernelGGL((kernels::csc_transpose_finish<int, T>), grid, block, 0, st, nnz, keys_out, p->perm, val, p->idx, static_cast<T*>(p->val));
  }
  if (e == hipSuccess) {
    size_t bytes = cub_bytes;
    e = hipcub::DeviceScan::InclusiveSum(cub_temp, bytes, p->off, p->off, rows + 1, st);
  }
  if (e == hipSuccess) e = hipGetLastError();
  int h_bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(pos); (void)hipFree(cub_temp); (void)hipFree(bad);
  int err = static_cast<int>(e);
  if (!err && h_bad) err = LOOPS_E_BADARG;  // a row (or COO column) index outside the matrix
  if (!err) err = spmv_plan_create<T>(rows, cols, nnz, p->off, p->idx, static_cast<const T*>(p->val), flags, repeats, st, &p->inner);
  if (err) { csc_plan_free(p); return err; }
  *out = p;
  return 0;
}

template <typename T>
int csc_plan_spmv(const loops_csc_plan* p, const T* x, T* y, hipStream_t st) {
  if (!p || p->vbytes != static_cast<int>(sizeof(T))) return LOOPS_E_BADARG;
  return spmv_planned<T>(p->inner, p->off, p->idx, static_cast<const T*>(p->val), x, y, st);
}

template <typename T>
int csc_plan_refresh(loops_csc_plan* p, const T* values, hipStream_t st) {
  if (!p || p->vbytes != static_cast<int>(sizeof(T)) || (p->nnz > 0 && !values)) return LOOPS_E_BADARG;
  if (p->nnz == 0) return 0;
  hipLaunchKernelGGL((kernels::gather_values<T>), dim3(math::ceil_div(p->nnz, 256)), dim3(256), 0, st, p->nnz, p->perm, values, static_cast<T*>(p->val));
  int err = static_cast<int>(hipGetLastError());
  if (!err && p->inner->panel) err = panel_refresh<T>(p->inner->panel, static_cast<const T*>(p->val), st);
  else if (!err && p->inner->band) {
    if constexpr (sizeof(T) == 4) err = rowband_refresh<T>(p->inner->band, static_cast<const T*>(p->val), st);
  }
  return err;
}

}  // namespace

// =============================================================================== extern "C"
extern "C" {

const char* loops_version(void) { return "0.2.0-mi355x"; }

int loops_release_scratch(void) { return scratch_release_all(); }

int loops_device_compute_units(int* out) {
  if (!out) return LOOPS_E_BADARG;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return static_cast<int>(e);
  return static_cast<int>(hipDeviceGetAttribute(out, hipDeviceAttributeMultiprocessorCount, dev));
}

int loops_merge_plan_create(int rows, int nnz, const int* offsets, int tile_config, void* stream,
                            loops_merge_plan_t** out) {
  if (!out || !offsets) return LOOPS_E_BADARG;
  if (tile_config == LOOPS_TILE_AUTO) {
    if (rows < 0 || nnz < 0) return LOOPS_E_BADARG;
    if (static_cast<long long>(rows) + nnz >= (1ll << 31) - 4096) return LOOPS_E_RANGE;
    return plan_create_auto(rows, nnz, offsets, as_stream(stream), out);
  }
  loops_merge_plan* p = nullptr;
  int err = plan_alloc(rows, nnz, tile_config, &p);
  if (err) return err;
  err = plan_compute(p, offsets, as_stream(stream));
  if (!err) err = plan_classify(p, offsets, as_stream(stream));
  if (err) { (void)hipFree(p->base); delete p; return err; }
  *out = p;
  return 0;
}

int loops_merge_plan_destroy(loops_merge_plan_t* plan) {
  if (!plan) return 0;
  (void)hipFree(plan->wide_carry);
  hipError_t e = hipFree(plan->base);
  delete plan;
  return static_cast<int>(e);
}

int loops_merge_plan_refresh(loops_merge_plan_t* plan, const int* offsets, void* stream) {
  if (!plan || !offsets) return LOOPS_E_BADARG;
  int err = plan_compute(plan, offsets, as_stream(stream));
  if (!err) err = plan_classify(plan, offsets, as_stream(stream));
  return err;
}

int loops_merge_plan_self_complete(const loops_merge_plan_t* plan) { return plan ? plan->self_complete : LOOPS_E_BADARG; }

int loops_merge_plan_num_tiles(const loops_merge_plan_t* plan) { return plan ? plan->num_tiles : LOOPS_E_BADARG; }

int loops_merge_plan_coords(const loops_merge_plan_t* plan, unsigned* h_coords) {
  if (!plan || !h_coords) return LOOPS_E_BADARG;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return static_cast<int>(e);
  return static_cast<int>(hipMemcpy(h_coords, plan->coords, sizeof(coord_t) * (static_cast<size_t>(plan->num_tiles) + 1),
                                    hipMemcpyDeviceToHost));
}

int loops_spmv_csr_f32(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const float* values, const float* x, float* y, void* stream) {
  return spmv_tuned<float>(schedule, rows, cols, nnz, offsets, indices, values, x, y, as_stream(stream));
}
int loops_spmv_csr_f64(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const double* values, const double* x, double* y, void* stream) {
  return spmv_tuned<double>(schedule, rows, cols, nnz, offsets, indices, values, x, y, as_stream(stream));
}

int loops_spmv_merge_path_f32(const loops_merge_plan_t* plan, int variant, int rows, int cols, int nnz,
                              const int* offsets, const int* indices, const float* values, const float* x, float* y,
                              void* stream) {
  if (!plan) return LOOPS_E_BADARG;
  int err = check_csr(rows, cols, nnz, offsets, indices, values, x, y);
  if (err) return err;
  return spmv_merge_path<float>(plan, variant, rows, nnz, offsets, indices, values, x, y, as_stream(stream), 3, false, cols);
}
int loops_spmv_merge_path_f64(const loops_merge_plan_t* plan, int variant, int rows, int cols, int nnz,
                              const int* offsets, const int* indices, const double* values, const double* x,
                              double* y, void* stream) {
  if (!plan) return LOOPS_E_BADARG;
  int err = check_csr(rows, cols, nnz, offsets, indices, values, x, y);
  if (err) return err;
  return spmv_merge_path<double>(plan, variant, rows, nnz, offsets, indices, values, x, y, as_stream(stream), 3, false, cols);
}

int loops_enable_peer_access(int peer_device) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return static_cast<int>(e);
  if (peer_device == dev) return 0;
  int can = 0;
  e = hipDeviceCanAccessPeer(&can, dev, peer_device);
  if (e != hipSuccess) return static_cast<int>(e);
  if (!can) return static_cast<int>(hipErrorPeerAccessUnsupported);
  e = hipDeviceEnablePeerAccess(peer_device, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 0; }
  return static_cast<int>(e);
}

int loops_spmv_merge_path_fanout_f32(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                                     const int* indices, const float* values, const float* x, float* y, int num_peers,
                                     float* const* h_peer_y, void* stream) {
  return merge_path_fanout<float>(plan, rows, cols, nnz, offsets, indices, values, x, y, num_peers, h_peer_y, as_stream(stream));
}

int loops_spmv_work_oriented_f32(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                                 const int* indices, const float* values, const float* x, float* y, void* stream) {
  return spmv_work_oriented_planned<float>(plan, rows, cols, nnz, offsets, indices, values, x, y, as_stream(stream));
}
int loops_spmv_work_oriented_f64(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                                 const int* indices, const double* values, const double* x, double* y, void* stream) {
  return spmv_work_oriented_planned<double>(plan, rows, cols, nnz, offsets, indices, values, x, y, as_stream(stream));
}

int loops_spmv_merge_path_stage_f32(const loops_merge_plan_t* plan, int variant, int stage, int rows, int cols,
                                    int nnz, const int* offsets, const int* indices, const float* values,
                                    const float* x, float* y, void* stream) {
  if (!plan || stage < 0 || stage > 1) return LOOPS_E_BADARG;
  int err = check_csr(rows, cols, nnz, offsets, indices, values, x, y);
  if (err) return err;
  return spmv_merge_path<float>(plan, variant, rows, nnz, offsets, indices, values, x, y, as_stream(stream),
                                1 << stage, false, cols);
}

int loops_spmv_csr_schedule_api_f32(int schedule, int tile_config, int rows, int cols, int nnz, const int* offsets,
                                    const int* indices, const float* values, const float* x, float* y, void* stream) {
  int err = check_csr(rows, cols, nnz, offsets, indices, values, x, y);
  if (err) return err;
  return spmv_schedule_api<float>(schedule, tile_config, rows, cols, nnz, offsets, indices, values, x, y,
                                  as_stream(stream));
}

int loops_schedule_dump_merge_path(int tile_config, int use_plan, int rows, int nnz, const int* offsets,
                                   unsigned* thread_start, int* atom_owner, int* atom_row, int* atom_visits,
                                   void* stream) {
  if (!offsets || !thread_start || rows < 0 || nnz < 0) return LOOPS_E_BADARG;
  hipStream_t s = as_stream(stream);
  switch (tile_config) {
    case LOOPS_TILE_256x8: return launch_merge_dump<256, 8>(use_plan, rows, nnz, offsets, thread_start, atom_owner, atom_row, atom_visits, s);
    case LOOPS_TILE_128x7: return launch_merge_dump<128, 7>(use_plan, rows, nnz, offsets, thread_start, atom_owner, atom_row, atom_visits, s);
    case LOOPS_TILE_4x2: return launch_merge_dump<4, 2>(use_plan, rows, nnz, offsets, thread_start, atom_owner, atom_row, atom_visits, s);
    case LOOPS_TILE_256x7: return launch_merge_dump<256, 7>(use_plan, rows, nnz, offsets, thread_start, atom_owner, atom_row, atom_visits, s);
    case LOOPS_TILE_512x8: return launch_merge_dump<512, 8>(use_plan, rows, nnz, offsets, thread_start, atom_owner, atom_row, atom_visits, s);
    case LOOPS_TILE_256x16: return launch_merge_dump<256, 16>(use_plan, rows, nnz, offsets, thread_start, atom_owner, atom_row, atom_visits, s);
    default: return LOOPS_E_CONFIG;
  }
}

int loops_schedule_dump_work_oriented(int grid_blocks, int rows, int nnz, const int* offsets, int* thread_map,
                                      int* atom_owner, int* atom_row, int* atom_visits, void* stream) {
  if (!offsets || !thread_map || grid_blocks <= 0 || rows < 0 || nnz < 0) return LOOPS_E_BADARG;
  launch::non_cooperative(as_stream(stream), kernels::work_oriented_dump<kSpmvBlock, int>, dim3(grid_blocks),
                          dim3(kSpmvBlock), std::size_t(rows), std::size_t(nnz), const_cast<int*>(offsets), thread_map,
                          atom_owner, atom_row, atom_visits);
  return last_error();
}

int loops_schedule_dump_group_mapped(int group_size, int rows, int nnz, const int* offsets, int* atom_owner,
                                     int* atom_row, int* atom_visits, void* stream) {
  if (!offsets || rows < 0 || nnz < 0) return LOOPS_E_BADARG;
  if (rows == 0) return 0;
  const dim3 grid(math::ceil_div(rows, kSpmvBlock)), block(kSpmvBlock);
  if (group_size == 256)
    launch::non_cooperative(as_stream(stream), kernels::group_mapped_dump<kSpmvBlock, 256, int>, grid, block,
                            std::size_t(rows), std::size_t(nnz), const_cast<int*>(offsets), atom_owner, atom_row,
                            atom_visits);
  else if (group_size == 64)
    launch::non_cooperative(as_stream(stream), kernels::group_mapped_dump<kSpmvBlock, 64, int>, grid, block,
                            std::size_t(rows), std::size_t(nnz), const_cast<int*>(offsets), atom_owner, atom_row,
                            atom_visits);
  else if (group_size == 16)
    launch::non_cooperative(as_stream(stream), kernels::group_mapped_dump<kSpmvBlock, 16, int>, grid, block,
                            std::size_t(rows), std::size_t(nnz), const_cast<int*>(offsets), atom_owner, atom_row,
                            atom_visits);
  else
    return LOOPS_E_CONFIG;
  return last_error();
}

int loops_work_oriented_grid(int* out_blocks) {
  if (!out_blocks) return LOOPS_E_BADARG;
  auto kernel = kernels::work_oriented_atomic_spmv<kSpmvBlock, int, int, float>;
  *out_blocks = static_cast<int>(launch_box::occupancy_grid(kernel, kSpmvBlock));
  return 0;
}

int loops_spmv_bcsr_f32(int R, int C, int mode, int rows, int num_block_rows, int num_blocks, const int* block_offsets,
                        const int* block_cols, const float* block_values, const float* x_padded, float* y,
                        void* stream) {
  return spmv_bcsr<float>(R, C, mode, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values, x_padded, y,
                          as_stream(stream));
}
int loops_spmv_bcsr_f64(int R, int C, int mode, int rows, int num_block_rows, int num_blocks, const int* block_offsets,
                        const int* block_cols, const double* block_values, const double* x_padded, double* y,
                        void* stream) {
  return spmv_bcsr<double>(R, C, mode, rows, num_block_rows, num_blocks, block_offsets, block_cols, block_values, x_padded,
                           y, as_stream(stream));
}

int loops_spmm_csr_f32(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const float* values, const float* B, int n, float* C, void* stream) {
  return spmm_tuned<float>(schedule, rows, cols, nnz, offsets, indices, values, B, n, C, as_stream(stream));
}
int loops_spmm_csr_f64(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const double* values, const double* B, int n, double* C, void* stream) {
  return spmm_tuned<double>(schedule, rows, cols, nnz, offsets, indices, values, B, n, C, as_stream(stream));
}
int loops_spmm_merge_path_f32(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                              const int* indices, const float* values, const float* B, int n, float* C,
                              void* stream) {
  if (!plan || n < 0 || rows < 0) return LOOPS_E_BADARG;
  if (n == 0 || rows == 0) return 0;
  int err = check_csr(rows, cols, nnz, offsets, indices, values, B, C);
  if (err) return err;
  return spmm_merge_path<float>(plan, rows, cols, nnz, offsets, indices, values, B, n, C, as_stream(stream));
}

int loops_spmm_merge_path_f64(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                              const int* indices, const double* values, const double* B, int n, double* C,
                              void* stream) {
  if (!plan || n < 0 || rows < 0) return LOOPS_E_BADARG;
  if (n == 0 || rows == 0) return 0;
  int err = check_csr(rows, cols, nnz, offsets, indices, values, B, C);
  if (err) return err;
  return spmm_merge_path<double>(plan, rows, cols, nnz, offsets, indices, values, B, n, C, as_stream(stream));
}

int loops_spmv_coo_f32(int mode, int rows, int cols, int nnz, const int* row_indices, const int* col_indices,
                       const float* values, const float* x, float* y, void* stream) {
  return spmv_coo<float>(mode, rows, cols, nnz, row_indices, col_indices, values, x, y, as_stream(stream));
}
int loops_spmv_coo_f64(int mode, int rows, int cols, int nnz, const int* row_indices, const int* col_indices,
                       const double* values, const double* x, double* y, void* stream) {
  return spmv_coo<double>(mode, rows, cols, nnz, row_indices, col_indices, values, x, y, as_stream(stream));
}

int loops_spmv_ell_f32(int mode, int rows, int cols, int pitch, const int* indices, const float* values,
                       const float* x, float* y, void* stream) {
  return spmv_ell<float>(mode, rows, cols, pitch, indices, values, x, y, as_stream(stream));
}
int loops_spmv_ell_f64(int mode, int rows, int cols, int pitch, const int* indices, const double* values,
                       const double* x, double* y, void* stream) {
  return spmv_ell<double>(mode, rows, cols, pitch, indices, values, x, y, as_stream(stream));
}

int loops_spmv_dia_f32(int mode, int rows, int cols, int num_diagonals, size_t stride, const int* diag_offsets,
                       const float* values, const float* x, float* y, void* stream) {
  return spmv_dia<float>(mode, rows, cols, num_diagonals, stride, diag_offsets, values, x, y, as_stream(stream));
}
int loops_spmv_dia_f64(int mode, int rows, int cols, int num_diagonals, size_t stride, const int* diag_offsets,
                       const double* values, const double* x, double* y, void* stream) {
  return spmv_dia<double>(mode, rows, cols, num_diagonals, stride, diag_offsets, values, x, y, as_stream(stream));
}

namespace {
// The launch-box autotuner: every compiled tile shape (and, with `phased`, the phased-gather twin of the shapes that have
// one) timed on this matrix; ms[cfg] = plain kernel, ms[6 + cfg] = phased (-1 = not timed).
int autotune_merge_path(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                        const float* x, float* y, int repeats, hipStream_t st, bool phased, int* best_cfg, int* best_variant,
                        float* ms_out /* 12 entries or NULL */) {
  int err = check_csr(rows, cols, nnz, offsets, indices, values, x, y);
  if (err) return err;
  if (repeats < 1) repeats = 5;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return static_cast<int>(hipGetLastError());
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return static_cast<int>(hipGetLastError()); }
  static const int candidates[] = {LOOPS_TILE_256x8, LOOPS_TILE_256x7, LOOPS_TILE_128x7, LOOPS_TILE_512x8, LOOPS_TILE_256x16};
  float best = 0.f;
  *best_cfg = LOOPS_TILE_DEFAULT;
  *best_variant = 0;
  if (ms_out) for (int i = 0; i < 12; ++i) ms_out[i] = -1.f;
  for (int cfg : candidates) {
    loops_merge_plan* p = nullptr;
    err = plan_alloc(rows, nnz, cfg, &p);
    if (!err) err = plan_compute(p, offsets, st);
    if (!err) err = plan_classify(p, offsets, st);  // time what a held plan of this shape would run
    const bool twin = phased && !err && p->num_tiles > 1 && (cfg == LOOPS_TILE_512x8 || cfg == LOOPS_TILE_256x16);
    for (int variant : {0, LOOPS_VARIANT_PHASED}) {
      if (err || (variant != 0 && !twin)) continue;
      for (int it = 0; !err && it < 2; ++it) err = spmv_merge_path<float>(p, variant, rows, nnz, offsets, indices, values, x, y, st, 3, false, cols);
      float ms = 0.f;
      if (!err) {
        (void)hipEventRecord(e0, st);
        for (int it = 0; !err && it < repeats; ++it)
          err = spmv_merge_path<float>(p, variant, rows, nnz, offsets, indices, values, x, y, st, 3, false, cols);
        (void)hipEventRecord(e1, st);
        if (!err) err = static_cast<int>(hipEventSynchronize(e1));
        if (!err) err = static_cast<int>(hipEventElapsedTime(&ms, e0, e1));
        ms /= static_cast<float>(repeats);
      }
      if (err) break;
      if (ms_out) ms_out[(variant ? 6 : 0) + cfg] = ms;
      if (best == 0.f || ms < best) { best = ms; *best_cfg = cfg; *best_variant = variant; }
    }
    plan_release(p);
    if (err) break;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return err;
}
}  // namespace

int loops_autotune_merge_path_f32(int rows, int cols, int nnz, const int* offsets, const int* indices,
                                  const float* values, const float* x, float* y, int repeats, void* stream,
                                  int* best_tile_config, float* ms_per_config /* 6 entries, may be NULL */) {
  if (!best_tile_config) return LOOPS_E_BADARG;
  float ms[12];
  int variant = 0;
  const int err = autotune_merge_path(rows, cols, nnz, offsets, indices, values, x, y, repeats, as_stream(stream), false,
                                      best_tile_config, &variant, ms);
  if (ms_per_config) for (int i = 0; i < 6; ++i) ms_per_config[i] = ms[i];
  return err;
}

int loops_autotune_merge_path_variants_f32(int rows, int cols, int nnz, const int* offsets, const int* indices,
                                           const float* values, const float* x, float* y, int repeats, void* stream,
                                           int* best_tile_config, int* best_variant, float* ms_per_config /* 12 entries, may be NULL */) {
  if (!best_tile_config || !best_variant) return LOOPS_E_BADARG;
  return autotune_merge_path(rows, cols, nnz, offsets, indices, values, x, y, repeats, as_stream(stream), true, best_tile_config,
                             best_variant, ms_per_config);
}

int loops_spmv_csc_f32(int mode, int rows, int cols, int nnz, const int* col_offsets, const int* row_indices,
                       const float* values, const float* x, float* y, void* stream) {
  return spmv_csc<float>(mode, rows, cols, nnz, col_offsets, row_indices, values, x, y, as_stream(stream));
}
int loops_spmv_csc_f64(int mode, int rows, int cols, int nnz, const int* col_offsets, const int* row_indices,
                       const double* values, const double* x, double* y, void* stream) {
  return spmv_csc<double>(mode, rows, cols, nnz, col_offsets, row_indices, values, x, y, as_stream(stream));
}

int loops_spmv_plan_create_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                               int flags, int repeats, void* stream, loops_spmv_plan_t** out) {
  return spmv_plan_create<float>(rows, cols, nnz, offsets, indices, values, flags, repeats, as_stream(stream), out);
}
int loops_spmv_plan_create_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                               int flags, int repeats, void* stream, loops_spmv_plan_t** out) {
  return spmv_plan_create<double>(rows, cols, nnz, offsets, indices, values, flags, repeats, as_stream(stream), out);
}
void loops_spmv_plan_destroy(loops_spmv_plan_t* plan) { spmv_plan_free(plan); }
int loops_spmv_plan_info(const loops_spmv_plan_t* plan, int* layout, int* tile_config, int* num_blocks, float* ms4) {
  if (!plan) return LOOPS_E_BADARG;
  if (layout) *layout = plan->layout;
  if (tile_config) *tile_config = plan->merge ? plan->merge->cfg : LOOPS_TILE_512x8;  // (copies: no merge tiles; a valid id)
  if (num_blocks) *num_blocks = plan->band ? plan->band->B : plan->panel ? plan->panel->P : 0;
  if (ms4) for (int i = 0; i < 4; ++i) ms4[i] = plan->ms[i];
  return 0;
}
int loops_columns_look_scattered(int cols, int nnz, const int* indices, int value_bytes, void* stream, int* scattered) {
  if (!scattered || cols < 0 || nnz < 0 || (nnz > 0 && !indices) || (value_bytes != 4 && value_bytes != 8)) return LOOPS_E_BADARG;
  *scattered = 0;
  if (!kernels::columns_worth_sampling(static_cast<long long>(nnz), static_cast<long long>(cols), value_bytes)) return 0;
  unsigned int* scratch = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&scratch), kernels::scatter_scratch_words * sizeof(unsigned int));
  if (e != hipSuccess) return static_cast<int>(e);
  *scattered = kernels::columns_look_scattered(as_stream(stream), indices, static_cast<long long>(nnz), static_cast<long long>(cols),
                                               value_bytes, scratch) ? 1 : 0;
  (void)hipFree(scratch);
  return last_error();
}
int loops_spmv_plan_variant(const loops_spmv_plan_t* plan, int* variant, float* ms_phased) {
  if (!plan) return LOOPS_E_BADARG;
  if (variant) *variant = plan->layout == LOOPS_LAYOUT_CSR ? plan->merge_variant : 0;
  if (ms_phased) *ms_phased = plan->ms_phased;
  return 0;
}
int loops_spmv_plan_refresh_values_f32(loops_spmv_plan_t* plan, const float* values, void* stream) {
  if (!plan || plan->vbytes != 4) return LOOPS_E_BADARG;
  if (plan->panel) return panel_refresh<float>(plan->panel, values, as_stream(stream));
  return plan->band ? rowband_refresh<float>(plan->band, values, as_stream(stream)) : 0;
}
int loops_spmv_plan_refresh_values_f64(loops_spmv_plan_t* plan, const double* values, void* stream) {
  if (!plan || plan->vbytes != 8) return LOOPS_E_BADARG;
  if (plan->panel) return panel_refresh<double>(plan->panel, values, as_stream(stream));
  return 0;
}
int loops_spmv_planned_f32(const loops_spmv_plan_t* plan, const int* offsets, const int* indices, const float* values,
                           const float* x, float* y, void* stream) {
  return spmv_planned<float>(plan, offsets, indices, values, x, y, as_stream(stream));
}
int loops_spmv_planned_f64(const loops_spmv_plan_t* plan, const int* offsets, const int* indices, const double* values,
                           const double* x, double* y, void* stream) {
  return spmv_planned<double>(plan, offsets, indices, values, x, y, as_stream(stream));
}

int loops_panel_plan_create_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                                int panel_columns, int subband_rows, void* stream, loops_panel_plan_t** out) {
  return panel_create<float>(rows, cols, nnz, offsets, indices, values, as_stream(stream), out, subband_rows, panel_columns);
}
int loops_panel_plan_create_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                                int panel_columns, int subband_rows, void* stream, loops_panel_plan_t** out) {
  return panel_create<double>(rows, cols, nnz, offsets, indices, values, as_stream(stream), out, subband_rows, panel_columns);
}
int loops_panel_plan_create_layout_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                                       int panel_columns, int subband_rows, int compact, void* stream, loops_panel_plan_t** out) {
  return panel_create<float>(rows, cols, nnz, offsets, indices, values, as_stream(stream), out, subband_rows, panel_columns, compact);
}
int loops_panel_plan_create_layout_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                                       int panel_columns, int subband_rows, int compact, void* stream, loops_panel_plan_t** out) {
  return panel_create<double>(rows, cols, nnz, offsets, indices, values, as_stream(stream), out, subband_rows, panel_columns, compact);
}
int loops_panel_plan_layout(const loops_panel_plan_t* plan, long long* info4) {
  if (!plan || !info4) return LOOPS_E_BADARG;
  info4[0] = plan->compact; info4[1] = plan->runs; info4[2] = plan->padded_b; info4[3] = plan->awin;
  return 0;
}
void loops_panel_plan_destroy(loops_panel_plan_t* plan) { panel_free(plan); }
int loops_panel_plan_info(const loops_panel_plan_t* plan, int* info7) {
  if (!plan || !info7) return LOOPS_E_BADARG;
  const int v[7] = {plan->W, plan->Hw, plan->P, plan->S, plan->padded, plan->num_chunks, plan->vbytes};
  for (int i = 0; i < 7; ++i) info7[i] = v[i];
  return 0;
}
int loops_panel_plan_arrays(const loops_panel_plan_t* plan, void* values, unsigned short* col16, int* dst4, unsigned short* row16,
                            int* perm, int* subband_start) {
  if (!plan) return LOOPS_E_BADARG;
  if (plan->rows == 0) return 0;
  const size_t n = static_cast<size_t>(plan->padded);
  hipError_t e = hipDeviceSynchronize();
  auto copy = [&](void* dst, const void* src, size_t bytes) {
    if (e == hipSuccess && dst && bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
  };
  copy(values, plan->val, static_cast<size_t>(plan->vbytes) * n);
  copy(col16, plan->col16, sizeof(unsigned short) * n);
  copy(row16, plan->row16, sizeof(unsigned short) * static_cast<size_t>(plan->padded_b));
  copy(perm, plan->perm, sizeof(int) * n);
  copy(dst4, plan->dst4, sizeof(int) * (n / 4));
  copy(subband_start, plan->bstart, sizeof(int) * (static_cast<size_t>(plan->S) + 1));
  return static_cast<int>(e);
}
int loops_panel_plan_windows(const loops_panel_plan_t* plan, int* window_start, int* windows, int* segment_start) {
  if (!plan || !window_start) return LOOPS_E_BADARG;
  if (plan->rows == 0) return 0;
  hipError_t e = hipDeviceSynchronize();
  const size_t S = static_cast<size_t>(plan->S);
  if (e == hipSuccess) e = hipMemcpy(window_start, plan->wstart, sizeof(int) * (S + 1), hipMemcpyDeviceToHost);
  if (e == hipSuccess && windows && window_start[S] > 0)
    e = hipMemcpy(windows, plan->wins, sizeof(int) * 2 * static_cast<size_t>(window_start[S]), hipMemcpyDeviceToHost);
  if (e == hipSuccess && segment_start)
    e = hipMemcpy(segment_start, plan->segb, sizeof(int) * (S * static_cast<size_t>(plan->P) + 1), hipMemcpyDeviceToHost);
  return static_cast<int>(e);
}
int loops_panel_plan_refresh_values_f32(loops_panel_plan_t* plan, const float* values, void* stream) {
  return panel_refresh<float>(plan, values, as_stream(stream));
}
int loops_panel_plan_refresh_values_f64(loops_panel_plan_t* plan, const double* values, void* stream) {
  return panel_refresh<double>(plan, values, as_stream(stream));
}
int loops_spmv_panel_f32(const loops_panel_plan_t* plan, const float* x, float* y, void* stream) {
  return panel_spmv<float>(plan, 3, x, y, as_stream(stream));
}
int loops_spmv_panel_f64(const loops_panel_plan_t* plan, const double* x, double* y, void* stream) {
  return panel_spmv<double>(plan, 3, x, y, as_stream(stream));
}
int loops_spmv_panel_stage_f32(const loops_panel_plan_t* plan, int stage, const float* x, float* y, void* stream) {
  if (stage < 0 || stage > 1) return LOOPS_E_BADARG;
  return panel_spmv<float>(plan, 1 << stage, x, y, as_stream(stream));
}
int loops_spmv_panel_stage_f64(const loops_panel_plan_t* plan, int stage, const double* x, double* y, void* stream) {
  if (stage < 0 || stage > 1) return LOOPS_E_BADARG;
  return panel_spmv<double>(plan, 1 << stage, x, y, as_stream(stream));
}
int loops_spmv_panel_fanout_f32(const loops_panel_plan_t* plan, const float* x, float* y, int num_peers, float* const* h_peer_y,
                                void* stream) {
  if (!plan || plan->vbytes != 4 || !y || (plan->nnz > 0 && !x)) return LOOPS_E_BADARG;
  kernels::peer_fanout<float> peers;
  int err = fanout_peers<float>(num_peers, h_peer_y, &peers);
  if (err) return err;
  if (plan->rows == 0) return 0;
  return kernels::launch_panel_binned_fanout<float>(as_stream(stream), panel_view<float>(plan), x, y, peers);
}
int loops_spmv_panel_fanout_f64(const loops_panel_plan_t* plan, const double* x, double* y, int num_peers, double* const* h_peer_y,
                                void* stream) {
  if (!plan || plan->vbytes != 8 || !y || (plan->nnz > 0 && !x)) return LOOPS_E_BADARG;
  kernels::peer_fanout<double> peers;
  int err = fanout_peers<double>(num_peers, h_peer_y, &peers);
  if (err) return err;
  if (plan->rows == 0) return 0;
  return kernels::launch_panel_binned_fanout<double>(as_stream(stream), panel_view<double>(plan), x, y, peers);
}


int loops_csc_plan_create_f32(int rows, int cols, int nnz, const int* col_offsets, const int* row_indices, const float* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out) {
  return csc_plan_create<float>(rows, cols, nnz, col_offsets, row_indices, values, flags, repeats, as_stream(stream), out);
}
int loops_csc_plan_create_f64(int rows, int cols, int nnz, const int* col_offsets, const int* row_indices, const double* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out) {
  return csc_plan_create<double>(rows, cols, nnz, col_offsets, row_indices, values, flags, repeats, as_stream(stream), out);
}
int loops_coo_plan_create_f32(int rows, int cols, int nnz, const int* row_indices, const int* col_indices, const float* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out) {
  return csc_plan_create<float>(rows, cols, nnz, nullptr, row_indices, values, flags, repeats, as_stream(stream), out, col_indices);
}
int loops_coo_plan_create_f64(int rows, int cols, int nnz, const int* row_indices, const int* col_indices, const double* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out) {
  return csc_plan_create<double>(rows, cols, nnz, nullptr, row_indices, values, flags, repeats, as_stream(stream), out, col_indices);
}
void loops_csc_plan_destroy(loops_csc_plan_t* plan) { csc_plan_free(plan); }
int loops_csc_plan_info(const loops_csc_plan_t* plan, int* layout, int* tile_config, int* num_blocks, float* ms4) {
  if (!plan) return LOOPS_E_BADARG;
  return loops_spmv_plan_info(plan->inner, layout, tile_config, num_blocks, ms4);
}
int loops_csc_plan_refresh_values_f32(loops_csc_plan_t* plan, const float* values, void* stream) {
  return csc_plan_refresh<float>(plan, values, as_stream(stream));
}
int loops_csc_plan_refresh_values_f64(loops_csc_plan_t* plan, const double* values, void* stream) {
  return csc_plan_refresh<double>(plan, values, as_stream(stream));
}
int loops_spmv_csc_planned_f32(const loops_csc_plan_t* plan, const float* x, float* y, void* stream) {
  return csc_plan_spmv<float>(plan, x, y, as_stream(stream));
}
int loops_spmv_csc_planned_f64(const loops_csc_plan_t* plan, const double* x, double* y, void* stream) {
  return csc_plan_spmv<double>(plan, x, y, as_stream(stream));
}

}  // extern "C"

// ------------------------------------------------------------------------------------ multi-GPU: partition + RCCL allgatherv
// The RCCL entry points are resolved at run time (no link-time dependency, and -- in a Python process -- the RCCL instance
// PyTorch has already loaded is the one used): first the process's global symbols, then a library that is already mapped
// under the name librccl.so / librccl.so.1 (RTLD_NOLOAD), then the library path.
namespace {
struct rccl_api {
  using result_t = int;  // ncclResult_t
  result_t (*GetUniqueId)(void*) = nullptr;
  // ncclUniqueId is a 128-byte struct passed BY VALUE: the same calling convention as this stand-in
  struct unique_id { char internal[128]; };
  result_t (*CommInitRank)(void**, int, unique_id, int) = nullptr;
  result_t (*CommDestroy)(void*) = nullptr;
  result_t (*GroupStart)() = nullptr;
  result_t (*GroupEnd)() = nullptr;
  result_t (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  result_t (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(result_t) = nullptr;
  bool ok = false;
};

const rccl_api& rccl() {
  static const rccl_api api = [] {
    rccl_api a;
    void* handles[4] = {RTLD_DEFAULT, nullptr, nullptr, nullptr};
    if (!dlsym(RTLD_DEFAULT, "ncclSend")) {
      int n = 0;
      for (const char* name : {"librccl.so", "librccl.so.1"})
        if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) { handles[n++] = h; break; }
      if (n == 0)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
          if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) { handles[n++] = h; break; }
      if (n == 0) return a;
    }
    void* h = handles[0];
    auto sym = [&](const char* name) { return dlsym(h, name); };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv;
    return a;
  }();
  return api;
}

// ncclDataType_t (rccl.h:466-467): ncclFloat32 = 7, ncclFloat64 = 8
template <typename T> constexpr int nccl_dtype() { return sizeof(T) == 4 ? 7 : 8; }

// The same group of sends / receives as multi_gpu::allgatherv (include/loops/multi_gpu/allgatherv.hxx), through the resolved
// entry points.
template <typename T>
int allgatherv_rt(void* comm, int rank, int world, T* y_full, const long long* bounds, hipStream_t st) {
  if (world <= 1) return 0;
  if (!comm || !y_full || !bounds || rank < 0 || rank >= world) return LOOPS_E_BADARG;
  const rccl_api& r = rccl();
  if (!r.ok) return LOOPS_E_CONFIG;
  const size_t mine = static_cast<size_t>(bounds[rank + 1] - bounds[rank]);
  int rc = r.GroupStart();
  for (int peer = 0; peer < world && rc == 0; ++peer) {
    if (peer == rank) continue;
    const size_t theirs = static_cast<size_t>(bounds[peer + 1] - bounds[peer]);
    if (mine) rc = r.Send(y_full + bounds[rank], mine, nccl_dtype<T>(), peer, comm, st);
    if (theirs && rc == 0) rc = r.Recv(y_full + bounds[peer], theirs, nccl_dtype<T>(), peer, comm, st);
  }
  const int e = r.GroupEnd();
  return rc ? rc : e;
}
}  // namespace

extern "C" {
int loops_row_ranges(int rows, const int* offsets, int parts, long long* bounds) {
  if (rows < 0 || !offsets || parts < 1 || !bounds) return LOOPS_E_BADARG;
  if (static_cast<long long>(rows) + offsets[rows] >= (1ll << 31)) return LOOPS_E_RANGE;
  return multi_gpu::row_ranges(offsets, static_cast<size_t>(rows), parts, bounds) ? 0 : LOOPS_E_BADARG;
}
int loops_comm_unique_id(void* id128) {
  if (!id128) return LOOPS_E_BADARG;
  const rccl_api& r = rccl();
  return r.ok ? r.GetUniqueId(id128) : LOOPS_E_CONFIG;
}
int loops_comm_init(int world, int rank, const void* id128, void** comm) {
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return LOOPS_E_BADARG;
  const rccl_api& r = rccl();
  if (!r.ok) return LOOPS_E_CONFIG;
  rccl_api::unique_id id;
  __builtin_memcpy(&id, id128, sizeof(id));
  return r.CommInitRank(comm, world, id, rank);
}
int loops_comm_destroy(void* comm) {
  if (!comm) return 0;
  const rccl_api& r = rccl();
  return r.ok ? r.CommDestroy(comm) : LOOPS_E_CONFIG;
}
const char* loops_comm_error_string(int code) {
  if (code == LOOPS_E_BADARG) return "LOOPS_E_BADARG";
  if (code == LOOPS_E_RANGE) return "LOOPS_E_RANGE";
  if (code == LOOPS_E_CONFIG) return "LOOPS_E_CONFIG: no RCCL entry points in this process or on the library path";
  const rccl_api& r = rccl();
  return r.ok && r.GetErrorString ? r.GetErrorString(code) : "unknown";
}
int loops_allgatherv_f32(void* comm, int rank, int world, float* y_full, const long long* bounds, void* stream) {
  return allgatherv_rt<float>(comm, rank, world, y_full, bounds, as_stream(stream));
}
int loops_allgatherv_f64(void* comm, int rank, int world, double* y_full, const long long* bounds, void* stream) {
  return allgatherv_rt<double>(comm, rank, world, y_full, bounds, as_stream(stream));
}
}  // extern "C"
