/*
 * loops_probes.h -- C ABI of libloops_probes.so: MEASUREMENT code only (calibration kernels and
 * experimental instantiations of the product kernels).  Nothing here is part of the drop-in
 * boundary (include/loops_amd.h); only bench.py, scripts/ and tests/perf/ load this library.
 *
 * Conventions as in loops_amd.h: device pointers, `stream` = hipStream_t as void*, asynchronous,
 * 0 on success / hipError_t / negative LOOPS_E_*.
 */
#ifndef LOOPS_PROBES_H_
#define LOOPS_PROBES_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Streaming copy dst[i] = src[i] (16 B per lane) -- the achievable HBM rate the roofline fraction is
 * also quoted against.  dst == src selects a READ-ONLY stream (per-lane sums, nothing
 * written): the achievable read rate. */
int loops_stream_copy_f32(const float* src, float* dst, size_t n, void* stream);
/* Copy-rate tuning: `unroll` (1, 2, 4, 8) 16-byte vectors per lane in flight; flags bit 0 non-temporal loads, bit 1
 * non-temporal stores, bit 2 one contiguous chunk per workgroup instead of grid-stride steps; `blocks` workgroups of 256. */
int loops_stream_copy_tuned_f32(const float* src, float* dst, size_t n, int unroll, int flags, int blocks, void* stream);
/* out[i] = table[idx[i]] with `scalar_per_64` (0, 4, 8, 16, 24, 32, 64) of every 64 gathers of a wavefront issued through the
 * scalar memory path (v_readlane -> s_load_dword -> select) and the rest as a divergent vector load: does the scalar path
 * add gather throughput on top of the vector L1's outstanding-read capacity? */
int loops_mixed_gather_f32(const float* table, const int* idx, float* out, size_t n, int scalar_per_64, void* stream);
/* Read-only stream (16 B per lane, a contiguous chunk per wavefront) with SCALAR prefetch `distance` KB-steps ahead: one
 * wave-uniform (SMEM) dword load per `line_words` words (32 = per 128-byte line, 16 = per 64 bytes); distance 0 = none.
 * `waves_per_cu` resident wavefronts per CU (grid = 256 CUs x that).  Does the scalar path add memory-level parallelism
 * on top of the vector L1's outstanding-read slots?  distance in {0,1,2,4,8,16} x line_words 32, {2,4,8} x 16. */
int loops_stream_read_prefetch_f32(const float* src, float* sink, size_t n, int distance, int line_words, int waves_per_cu,
                                   void* stream);
/* out[i] = table[idx[i]] -- the L2 / Infinity-Cache gather rate that bounds x reads.
 * mode: 0 plain loads, 1 non-temporal, 2 agent-scope (sc1: bypass the CU's L1), 3 system-scope,
 * 4 plain gather with non-temporal index / output streams. */
int loops_gather_f32(const float* table, const int* idx, float* out, size_t n, int mode, void* stream);
/* `blocks` x 256 lanes each issue `reps` 4-byte loads from a power-of-two table (pattern 0
 * consecutive, 1 hashed, 2 broadcast): the address rate of the CU's vector-memory path. */
int loops_address_rate_f32(const float* table, int table_words, int reps, int pattern, int blocks, float* out,
                           void* stream);
/* LDS update rate: `blocks` x 256 lanes each apply `reps` updates to a 4096-word LDS table.  mode 0 ds_add_f32, 1 ds_add_u32,
 * 2 plain read-add-write (not atomic), 3 ds_add_rtn_f32, 4 compare-and-swap loop; 10 / 11 / 12: ds_add_f64 / read-add-write /
 * compare-and-swap loop on 8-byte words; pattern 0 consecutive words (conflict-free), 1 hashed, 2 all lanes
 * one word, 3 adjacent lane pairs share a word.  `out`: blocks * 256 floats.  What bounds panel_reduce's small windows. */
int loops_lds_update_rate_f32(int mode, int pattern, int reps, int blocks, float* out, void* stream);
/* Row-gather probe (the SpMM's B access pattern in isolation): sub-groups of row_floats / 4 lanes
 * read `count` rows of row_floats floats (8..256, power of two) of `table` selected by `idx`, 16 B
 * per lane, 8 rows in flight; `out` needs blocks * 256 floats. */
int loops_row_gather_f32(const float* table, const int* idx, size_t count, int row_floats, int blocks, float* out,
                         void* stream);

/* Cache-policy experiment: the product's fused merge_path_flat kernel (merge_path_spmv_fused<512, 8>,
 * bit-mask split, 16-byte aligned arrays required) instantiated with explicit gfx950 cache-policy bits on
 * its col_idx stream / values stream / x gather.  `policy` indexes the table returned by
 * loops_probe_policy_name(); stages: bit 0 = tile kernel, bit 1 = fix-up, bit 2 = build the coordinates
 * into `scratch` first (needed once per matrix).  `scratch`: loops_probe_merge_path_scratch_bytes() bytes. */
size_t loops_probe_merge_path_scratch_bytes(int rows, int nnz);
/* Ordering experiment: the merge tiles walked by `groups` PERSISTENT workgroups (contiguous shares, the product's
 * work_oriented_spmv_fused<512, 8>); pipelined = 1: each tile's x gathers are issued BEFORE the stream loads of the workgroup's
 * next tile (two register sets); 2: the next tile's streams leave once the gathers have returned (in flight during the walk); 3: plain order with the
 * phased gathers of the headline kernel (8 parts).  stages / scratch as loops_probe_merge_path_f32; nnz % 4 == 0; bit-equal results. */
int loops_probe_persistent_f32(int pipelined, int groups, int stages, int rows, int cols, int nnz, const int* offsets,
                               const int* indices, const float* values, const float* x, float* y, void* scratch, void* stream);
int loops_probe_policy_count(void);
const char* loops_probe_policy_name(int policy);
int loops_probe_merge_path_f32(int policy, int stages, int rows, int cols, int nnz, const int* offsets,
                               const int* indices, const float* values, const float* x, float* y, void* scratch,
                               void* stream);

/* Tile-shape experiment: the same kernel with merge-tile shapes the product does not ship.  shape: 0 = 512 x 8 (the
 * product's), 1 = 512 x 16, 2 = 1024 x 8, 3 = 1024 x 4, 4 = 256 x 8.  `scratch`: 4 x loops_probe_merge_path_scratch_bytes(). */
int loops_probe_merge_path_shape_f32(int shape, int stages, int rows, int cols, int nnz, const int* offsets,
                                     const int* indices, const float* values, const float* x, float* y, void* scratch,
                                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOOPS_PROBES_H_ */
