/*
 * loops_amd.h -- C ABI of libloops_amd.so: the drop-in boundary of the MI355X-native
 * load-balanced CSR SpMV path (gunrock/loops hot path), for hosts that cannot consume the
 * C++ header API in include/loops/ directly (ctypes, cgo, JNI, N-API, ...).
 *
 * The reference has no C ABI: its boundary is the header-only C++ template API
 * (SURVEY 8b).  Every entry point below is a thin wrapper over the corresponding C++
 * template in include/loops/ instantiated for index_t = offset_t = int and
 * type_t = float | double -- the instantiation all reference examples use
 * (examples/spmv/merge_path.cu:18-20).  The reference interface each one replaces is cited
 * as file:line under the reference tree.
 *
 * Conventions: all array pointers are DEVICE pointers unless the name starts with `h_`;
 * `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 * on `stream` unless stated otherwise; the return value is 0 on success, a hipError_t (> 0)
 * from the runtime, or a negative LOOPS_E_* argument error.  No entry point falls back to a
 * CPU implementation.
 */
#ifndef LOOPS_AMD_H_
#define LOOPS_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOOPS_E_BADARG (-1)   /* null pointer / negative size / unknown enum value        */
#define LOOPS_E_RANGE (-2)    /* rows + nnz does not fit the int search arithmetic (2^31) */
#define LOOPS_E_CONFIG (-3)   /* tile configuration not compiled into the library         */

/* schedule::algorithms_t (include/loops/schedule.hxx:26-32) + the two non-schedule kernels */
enum loops_schedule {
  LOOPS_MERGE_PATH_FLAT = 0,
  LOOPS_WORK_ORIENTED = 1,
  LOOPS_THREAD_MAPPED = 2,
  LOOPS_GROUP_MAPPED = 3,
  LOOPS_ORIGINAL = 4,        /* algorithms/spmv/original.cuh:26-46          */
  LOOPS_FLAT_PARTITIONED = 5 /* algorithms/spmv/flat_partitioned.cuh:46-110 */
};

/* Compiled (threads-per-block, items-per-thread) merge-tile shapes.  LOOPS_TILE_DEFAULT is
 * the gfx950 launch box (algorithms/spmv/launch_box.hxx:75-77 in the reference: 256 x 8). */
enum loops_tile_config {
  LOOPS_TILE_256x8 = 0,
  LOOPS_TILE_128x7 = 1, /* the reference's NVIDIA / fallback box, launch_box.hxx:63-90 */
  LOOPS_TILE_4x2 = 2,   /* tiny: lets small fixtures span many merge tiles             */
  LOOPS_TILE_256x7 = 3,
  LOOPS_TILE_512x8 = 4,
  LOOPS_TILE_256x16 = 5, /* 4096-item tiles: twice the bytes in flight per lane */
  LOOPS_TILE_DEFAULT = 0,
  /* loops_merge_plan_create only: the shape is picked from the structure -- 256 x 8 when the plan is self-completing with it
   * (no row needs a carry-out: one kernel per product), 512 x 8 otherwise (rows longer than a merge tile) */
  LOOPS_TILE_AUTO = -1
};

/* ---- library / device ------------------------------------------------------------------ */
const char* loops_version(void);                 /* "0.2.0-mi355x" (CMakeLists.txt:41: 0.2.0) */
int loops_device_compute_units(int* out);        /* util/device.hxx:93-108 multi_processor_count */
/* Frees the CALLING THREAD's cached scratch of the plan-less entry points (see "Concurrency" below); returns the number
 * of buffers released (>= 0).  The cache is per host thread and is not released when a thread exits: a thread that
 * rotates through many streams, or is about to end, calls this.  hipFree synchronises the device. */
int loops_release_scratch(void);

/* ---- merge-path plan ---------------------------------------------------------------------
 * Replaces schedule::merge_path::preprocess_t (schedule/merge_path_flat.hxx:99-172) and its
 * pre-pass kernel generate_search_coordinates (:45-76).  A plan holds the M + 1 per-workgroup
 * start coordinates for (offsets, rows, nnz, tile shape) and the carry-out scratch of the
 * fused kernel; it depends on the sparsity structure only and may be reused for any number
 * of SpMVs with the same offsets array.
 * ONE PRODUCT IN FLIGHT PER PLAN: a plan owns a single set of carry-out buffers that every product through it writes,
 * so two products that use the same plan must be ordered (same stream, or an event between them).  The same holds for
 * layout plans (loops_rowband_plan_t: partial vectors; loops_panel_plan_t: products scratch).  One plan per stream for concurrent products. */
typedef struct loops_merge_plan loops_merge_plan_t;

int loops_merge_plan_create(int rows, int nnz, const int* offsets, int tile_config, void* stream,
                            loops_merge_plan_t** out);
int loops_merge_plan_destroy(loops_merge_plan_t* plan);
/* Recompute the coordinates on `stream` (what constructing a new preprocess_t does). */
int loops_merge_plan_refresh(loops_merge_plan_t* plan, const int* offsets, void* stream);
int loops_merge_plan_num_tiles(const loops_merge_plan_t* plan);
/* 1 if no merge tile starts more than TPB nonzeros inside a row (checked when the plan is created or
 * refreshed, one stream synchronisation): the planned merge_path_flat SpMV then runs as ONE kernel -- every
 * tile re-reads the short head of its first row itself, nothing is carried between tiles, no fix-up launch.
 * 0 otherwise (some row is long: tile kernel + carry-out fix-up, as in every plan-less call). */
int loops_merge_plan_self_complete(const loops_merge_plan_t* plan);
/* Synchronous copy of the 2 * (M + 1) unsigned coordinates {x0, y0, x1, y1, ...} to the host. */
int loops_merge_plan_coords(const loops_merge_plan_t* plan, unsigned* h_coords);

/* ---- CSR SpMV, tuned paths: y = A * x  (y need not be initialised) --------------------------
 * Replaces algorithms::spmv::{merge_path_flat, work_oriented, thread_mapped, group_mapped,
 * original, flat_partitioned}(csr, x, y, stream)
 *   (algorithms/spmv/merge_path_flat.cuh:97-139, work_oriented.cuh:103-121,
 *    thread_mapped.cuh:70-91, group_mapped.cuh:72-105, original.cuh:58-75,
 *    flat_partitioned.cuh:70-110).
 * Unlike the reference wrappers these do NOT block on the stream.
 * LOOPS_THREAD_MAPPED keeps the schedule (a thread owns whole rows) and takes a row's nonzeros 16 / 4 at a time -- same
 * bits as the plain loop, 5-12 x its speed; LOOPS_ORIGINAL is the plain loop.
 * Concurrency: the plan-less entry points (this one, loops_spmm_csr_*) keep their merge-path scratch
 * (coordinates, carry-outs) in one lazily grown buffer per (host thread, device, stream, tile shape): calls on
 * the same stream reuse it in stream order, calls on different streams or devices never share it, so products
 * issued from one thread on several streams may overlap.  (At most 16 such buffers are cached per thread; the
 * least recently used is released first.  Growing or evicting a buffer calls hipFree, which waits for the DEVICE:
 * an otherwise asynchronous call then blocks once -- steady-state calls on up to 16 (stream, tile shape) pairs per
 * thread never do; entries of destroyed streams stay until evicted or loops_release_scratch() is called.  HIP-graph capture of a
 * plan-less call works on a stream the same call has run on before -- the scratch exists then; the first call on a stream allocates.)  A held plan (loops_merge_plan_t, loops_rowband_plan_t, loops_panel_plan_t) owns ONE set of
 * scratch buffers and is passed as const only because its coordinates are read-only: it serves one product at a
 * time -- do not run the same plan on two streams or from two threads concurrently; create one plan per stream. */
/* (LOOPS_MERGE_PATH_FLAT: from an x of 3 MB on (8-byte values: 6 MB) and 2^20 nonzeros the call samples the columns on the device --
 * two small kernels, ~5 us, on the FIRST call on a matrix: the stream's scratch remembers (indices pointer, nnz, columns) -- and its
 * tile kernel gathers in phases when they are scattered, loops_columns_look_scattered's rule evaluated on the device: the call stays
 * asynchronous; |x| = 4 / 8 / 16 MB, scattered columns: 1.12 / 1.5 / 1.7 x (C2: 106 -> 94 us per call).  A matrix edited in place under
 * the same pointer keeps its sample: the gather ORDER only, never the result.
 * LOOPS_WORK_ORIENTED: workgroups walk shares of 1-4 merge tiles (one long share per resident workgroup kept them in step);
 * above the same thresholds the share is one tile and the launch is LOOPS_MERGE_PATH_FLAT's, device-side choice included.
 * LOOPS_GROUP_MAPPED: groups of more than 24 tiles are shared out (kernels/group_mapped_spmv.hxx); from an x of 6 MB on the same
 * device-side choice of gather order.) */
int loops_spmv_csr_f32(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const float* values, const float* x, float* y, void* stream);
int loops_spmv_csr_f64(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const double* values, const double* x, double* y, void* stream);

/* merge_path_flat with a prebuilt plan: exactly the region the reference times
 * (merge_path_flat.cuh:121-136 starts its timer after the pre-pass).  `variant` selects a
 * compiled code variant of the fused kernel: 0 = default (bit-mask split of the tile among the threads);
 * tuning aids with the per-thread halving search instead: 4 (otherwise as 0), 1 = non-temporal streaming
 * loads, 2 = unpadded LDS product array, 3 = both.
 * LOOPS_VARIANT_PHASED = the default kernel with PHASED x gathers (plans of shape LOOPS_TILE_512x8 / LOOPS_TILE_256x16,
 * LOOPS_E_CONFIG otherwise): a tile's gathers leave in passes by column range and the pass a workgroup starts with is read
 * off a chip-wide clock, so that the workgroups sharing an L2 gather from the same eighth of x at the same time (8 parts up
 * to an x of 6 MB, 16 up to 24 MB, 32 beyond).  Same bits as variant 0; faster where the columns are scattered over an x of one
 * XCD's L2 or more (C2, 4 MB: 94.8 -> 82.9 us; 8 / 16 / 32 / 64 MB: 1.65 / 1.9 / 1.45 / 1.3 x), slower by 2-10 % where the
 * gathers hit anyway -- pick it by measurement (loops_autotune_merge_path_variants_f32, LOOPS_PLAN_MEASURE).
 * No reference counterpart (merge_path_flat.cuh:66-82 leaves the gather order to the hardware). */
#define LOOPS_VARIANT_PHASED 8
int loops_spmv_merge_path_f32(const loops_merge_plan_t* plan, int variant, int rows, int cols, int nnz,
                              const int* offsets, const int* indices, const float* values, const float* x,
                              float* y, void* stream);
int loops_spmv_merge_path_f64(const loops_merge_plan_t* plan, int variant, int rows, int cols, int nnz,
                              const int* offsets, const int* indices, const double* values, const double* x,
                              double* y, void* stream);

/* work_oriented with a prebuilt plan (tile config LOOPS_TILE_256x8 only, LOOPS_E_CONFIG otherwise): the persistent kernel of
 * algorithms::spmv::work_oriented (algorithms/spmv/work_oriented.cuh:33-121) walks an even share of the plan's merge tiles
 * per workgroup; no coordinate pre-pass per call.  Over an x of 3 MB or more (8-byte values: 6 MB) and 2^20 nonzeros the share is ONE
 * tile and the launch is merge_path_flat's over the plan's tiles, gather order decided on the device from a sample of the columns
 * the plan remembers (first call): C2's rows over x = 4 / 8 / 16 MB 99 / 178 / 243 -> 90 / 108 / 151 us. */
int loops_spmv_work_oriented_f32(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                                 const int* indices, const float* values, const float* x, float* y, void* stream);
int loops_spmv_work_oriented_f64(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                                 const int* indices, const double* values, const double* x, double* y, void* stream);

/* ---- multi-GPU: allgatherv(y) fused into the SpMV epilogue ----------------------------------------------
 * No reference counterpart (the reference is single-GPU).  A rank of a row-range sharded SpMV owns rows
 * [row_begin, row_end) of y; instead of exchanging its slice afterwards, the kernels that FINISH rows of y (the fused
 * merge-tile kernel and its fix-up; for row-band and panel-binned shards the kernels that store rows of y) also store every finished value to
 * the same element of up to 7 peer vectors over xGMI.  h_peer_y: HOST array of num_peers device-accessible pointers,
 * h_peer_y[p] = where THIS shard's y[0] lives in peer p's full-length vector (a hipIpcOpenMemHandle / peer-access
 * mapping; loops_enable_peer_access(peer device) first).  The peers' copies are complete when the launches have
 * completed on `stream`; the ranks still need one synchronisation point per step (any barrier) before reading.
 * The plan of loops_spmv_merge_path_fanout_f32 must have tile shape LOOPS_TILE_512x8 (LOOPS_E_CONFIG otherwise). */
int loops_enable_peer_access(int peer_device);
int loops_spmv_merge_path_fanout_f32(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                                     const int* indices, const float* values, const float* x, float* y, int num_peers,
                                     float* const* h_peer_y, void* stream);

/* The two kernels of loops_spmv_merge_path_f32 one at a time, so a harness can bracket each with
 * its own hipEvents: stage 0 = fused merge-tile kernel (writes y and the carry-outs),
 * stage 1 = carry-out fix-up.  Running stage 0 then stage 1 == loops_spmv_merge_path_f32. */
int loops_spmv_merge_path_stage_f32(const loops_merge_plan_t* plan, int variant, int stage, int rows, int cols,
                                    int nnz, const int* offsets, const int* indices, const float* values,
                                    const float* x, float* y, void* stream);

/* ---- CSR SpMV through the public schedule API (reference-shaped kernels) --------------------
 * The kernels a user of schedule::setup<> would write (and the reference ships): one global
 * atomicAdd per nonzero for merge_path_flat / group_mapped / flat_partitioned, store+atomic mix
 * for work_oriented.  y MUST be zero-filled by the caller for every schedule except
 * thread_mapped / original (flat_partitioned.cuh:95-97).  `tile_config` applies to
 * merge_path_flat only. */
int loops_spmv_csr_schedule_api_f32(int schedule, int tile_config, int rows, int cols, int nnz,
                                    const int* offsets, const int* indices, const float* values,
                                    const float* x, float* y, void* stream);

/* ---- schedule introspection (parity hooks) ---------------------------------------------------
 * Run a schedule and record what it hands out; outputs are device arrays.
 *   thread_start : 2 * M * TPB unsigned, the per-thread LOCAL start coordinates
 *                  (schedule/merge_path_flat.hxx:328-330)
 *   thread_map   : 4 * grid * 256 ints {st.tile, st.atom, en.tile, en.atom}
 *                  (schedule/work_oriented.hxx:93-114)
 *   atom_owner / atom_row / atom_visits : nnz ints (global thread id that visits the atom, the
 *                  tile it is attributed to, visit count -- visits must be zero-filled). */
int loops_schedule_dump_merge_path(int tile_config, int use_plan, int rows, int nnz, const int* offsets,
                                   unsigned* thread_start, int* atom_owner, int* atom_row, int* atom_visits,
                                   void* stream);
int loops_schedule_dump_work_oriented(int grid_blocks, int rows, int nnz, const int* offsets, int* thread_map,
                                      int* atom_owner, int* atom_row, int* atom_visits, void* stream);
int loops_schedule_dump_group_mapped(int group_size /* 64 or 256 */, int rows, int nnz, const int* offsets,
                                     int* atom_owner, int* atom_row, int* atom_visits, void* stream);
/* Grid algorithms::spmv::work_oriented launches: occupancy x CUs (util/launch_box.hxx:228-239). */
int loops_work_oriented_grid(int* out_blocks);

/* ---- BCSR SpMV (R x C dense blocks), thread-per-block-row and the MFMA path -------------------
 * Replaces algorithms::spmv::bcsr_thread_mapped<R, C> (algorithms/spmv/bcsr_thread_mapped.cuh:91-123).
 * x must be padded to num_block_cols * C; rows of y >= `rows` are not written.
 * Block shapes compiled into the library: 2x2, 3x3, 4x4, 8x8 (fp32 and fp64); anything else returns LOOPS_E_CONFIG (the
 * header API instantiates any <R, C>).
 * mode 3: the tuned kernel of the shape (what algorithms::spmv::bcsr_thread_mapped<R, C> launches): mode 2 for every shape but
 * 4x4 fp32; there mode 1 where the block-row lengths are even and mode 4 where they are skewed (longest block-row beyond
 * max(64, blocks / 14 000) blocks, or a lockstep walk of groups of 4 block-rows touching more than 1.2 x the blocks).  The class
 * of a matrix is found by a probe the FIRST mode-3 call on it launches (remembered per host thread by block_offsets pointer and
 * sizes, up to 8 matrices; reported through mapped host memory, no synchronisation): that call, and any until the report has
 * arrived, run mode 4, which is safe on any lengths.  A matrix edited in place keeps its class (the product stays right).
 * mode 2: coalesced lane-group kernel, any shape, fp32 / fp64 -- a slot of lanes reads whole lines of
 * consecutive blocks of one block-row, 16 bytes per lane (36-byte 3x3 blocks: lane per block), block inner product on the
 * VALU, log2 cross-lane reduce; tuning aid 100000 + 100 h + u = h in {1,4,16} blocks of a block-row per step, u in {1,2,4}
 * steps in flight.
 * mode 4 (4x4 fp32 only): the MERGE-PATH form (include/loops/kernels/bcsr_merge_path.hxx) -- equal tiles of (block-row ends,
 * blocks), block-row sums in LDS, MFMA block products, 4-wide carry-outs + a fix-up launch; no reference counterpart.  For BCSR
 * with skewed block-row lengths (64 block-rows of 16 384 blocks among 2^17 of 8: 39 us, against 1.8 ms for modes 0-2, which give
 * a block-row to one owner as the reference does); ~20 % slower than mode 1 where the lengths are uniform (C4: 81.5 against
 * 66-70 us).  Uses a per-stream scratch block (loops_release_scratch frees it).
 * mode 0: register accumulation, thread per block-row (the reference's kernel shape); mode 1: MFMA 4x4x1 block inner product
 * (4x4 only), kernel shape picked from the mean blocks per block-row.  Tuning aids: mode 1u = one block
 * of a block-row per step with u in {1,2,4,8} steps in flight; mode 100 + 10 h + u = h in {1,2,4,8,16}
 * consecutive blocks of a block-row per step (h * 64 contiguous bytes per request); + 1000 g = a wavefront
 * pipelines through g in {1..64} consecutive groups of 16 / h block-rows (default: automatic). */
int loops_spmv_bcsr_f32(int R, int C, int mode, int rows, int num_block_rows, int num_blocks,
                        const int* block_offsets, const int* block_cols, const float* block_values,
                        const float* x_padded, float* y, void* stream);
/* fp64 blocks (the reference builds every example as .f32 and .f64, examples/spmv/CMakeLists.txt:29-50): modes 0, 2, 3 --
 * the MFMA kernel is fp32 (v_mfma_f32_4x4x1); any MFMA mode returns LOOPS_E_CONFIG. */
int loops_spmv_bcsr_f64(int R, int C, int mode, int rows, int num_block_rows, int num_blocks,
                        const int* block_offsets, const int* block_cols, const double* block_values,
                        const double* x_padded, double* y, void* stream);

/* The class of a BCSR's block-row lengths by mode 3's rule, computed now (one small kernel + a 4-byte copy; SYNCHRONISES the
 * stream): *out_class = 1 even (mode 3 -> mode 1), 2 skewed (mode 3 -> mode 4).  For callers that pick the mode themselves. */
int loops_bcsr_row_length_class(int num_block_rows, int num_blocks, const int* block_offsets, int* out_class, void* stream);

/* ---- Block-band plan for 4 x 4 fp32 BCSR (include/loops/kernels/bcsr_band.hxx) ------------------------------------------------
 * A held plan for algorithms::spmv::bcsr_thread_mapped<4, 4> (algorithms/spmv/bcsr_thread_mapped.cuh:36-123) products over ONE
 * matrix: a re-ordered COPY of the blocks (container/bcsr.hxx:13-17 cells, row-major) -- bands of `band_block_rows` block-rows
 * (0 = automatic; else a power of two in [16, 4096]), the blocks of a band sorted by block column so that the 16-byte x
 * gathers of a wavefront share 128-byte lines; the band's 4 HB row sums live in LDS as fp64 words (ds_add_f64), the block
 * inner product runs on v_mfma_f32_4x4x1.  Bands are cut into about `target_chunks` workgroups (0 = automatic: the CU count
 * when there are fewer bands); cut bands go through fp32 partial vectors and a second kernel.  y needs no zero-fill, rows of
 * y >= `rows` are not written, x must be padded to 4 * num_block_cols.  SUMMATION ORDER: the fp64 LDS atomics of different
 * wavefronts arrive in no fixed order -- results are bit-identical to bcsr_thread_mapped (and between runs) whenever the
 * fp64 sums of a row's fp32 block products are exact (the exactly-summable inputs of the tests; in general the last bit of a
 * row may differ from run to run).  Callers that need run-to-run determinism on arbitrary values use loops_spmv_bcsr_f32.
 * Returns LOOPS_E_BADARG for a block column outside [0, num_block_cols), LOOPS_E_RANGE when row code + block column do not
 * fit one 32-bit word at the band height asked for.
 * loops_bcsr_band_plan_info: info10 = {HB, bands, column bits, steps of 16 blocks, chunks, partial vectors, cut bands,
 * wavefronts per workgroup, steps per batch, non-temporal streams}.  loops_bcsr_band_plan_arrays: HOST copies (any pointer may
 * be NULL): values [steps * 256], words [steps * 16] = (row code << column bits) | block column -- row code = the block-row inside
 * the band; HB for padding; HB + 1 + 16 hub + (slot mod 16) for a block of the band's hub number `hub` (a block-row holding >= 1 / 32
 * of its band's blocks, at least 64: its blocks are spread over 16 replicated accumulator groups) --, perm [steps * 16] = BCSR
 * position of the slot's block or -1, chunks [n * 4] = {band, first step, end step, partial slot or -1}, multi [m * 3] = {band,
 * first partial slot, chunks}, hubs [bands * 33] = per band the number of hubs (<= 32), then their block-rows inside the band.  _set_chunks re-cuts the bands of a built plan;
 * _tune times every compiled kernel shape (ms12, may be NULL: (waves 8 | 16) x (steps 1 | 2 | 4) x (plain | non-temporal), in ms)
 * and keeps the fastest; _set_shape sets one.  One plan per stream for concurrent products (the plan owns its partial vectors). */
typedef struct loops_bcsr_band_plan loops_bcsr_band_plan_t;
int loops_bcsr_band_plan_create_f32(int rows, int num_block_rows, int num_block_cols, int num_blocks, const int* block_offsets,
                                    const int* block_cols, const float* block_values, int band_block_rows, int target_chunks,
                                    void* stream, loops_bcsr_band_plan_t** out);
void loops_bcsr_band_plan_destroy(loops_bcsr_band_plan_t* plan);
int loops_bcsr_band_plan_info(const loops_bcsr_band_plan_t* plan, int* info10);
int loops_bcsr_band_plan_arrays(const loops_bcsr_band_plan_t* plan, float* values, unsigned int* words, int* perm, int* chunks, int* multi,
                                unsigned short* hubs);
int loops_bcsr_band_plan_set_chunks(loops_bcsr_band_plan_t* plan, int target_chunks);
int loops_bcsr_band_plan_tune(loops_bcsr_band_plan_t* plan, int repeats, float* ms12, void* stream);
int loops_bcsr_band_plan_set_shape(loops_bcsr_band_plan_t* plan, int waves, int unroll, int nt);
int loops_bcsr_band_plan_refresh_values_f32(loops_bcsr_band_plan_t* plan, const float* block_values, void* stream);
int loops_spmv_bcsr_band_f32(const loops_bcsr_band_plan_t* plan, const float* x_padded, float* y, void* stream);
/* stage 0: the accumulate kernel only (partial vectors of cut bands are left in the plan), stage 1: the combine kernel only. */
int loops_spmv_bcsr_band_stage_f32(const loops_bcsr_band_plan_t* plan, int stage, const float* x_padded, float* y, void* stream);

/* ---- CSR SpMM  C[rows x n] = A[rows x cols] * B[cols x n], dense row-major B and C ------------
 * Replaces algorithms::spmm::thread_mapped (algorithms/spmm/thread_mapped.cuh:28-90; caller:
 * examples/spmm/thread_mapped.cu:30-41).  schedule LOOPS_MERGE_PATH_FLAT = the tuned kernel
 * (merge tiles x 64-column slabs of B, coalesced B rows, no atomics, C needs no zero-fill);
 * LOOPS_THREAD_MAPPED = the reference-shaped kernel (thread per row, columns outer).  Leading
 * dimensions are n (packed), as matrix_t stores them (container/matrix.cuh:28-36). */
int loops_spmm_csr_f32(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const float* values, const float* B, int n, float* C, void* stream);
int loops_spmm_csr_f64(int schedule, int rows, int cols, int nnz, const int* offsets, const int* indices,
                       const double* values, const double* B, int n, double* C, void* stream);
/* Same with a held plan (tile config LOOPS_TILE_256x8 only): no coordinate pre-pass per call. */
int loops_spmm_merge_path_f32(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                              const int* indices, const float* values, const float* B, int n, float* C,
                              void* stream);
int loops_spmm_merge_path_f64(const loops_merge_plan_t* plan, int rows, int cols, int nnz, const int* offsets,
                              const int* indices, const double* values, const double* B, int n, double* C,
                              void* stream);

/* ---- panel-binned layout: SpMV without a gather, for x far larger than the per-XCD L2 ------------------------------
 * No reference counterpart.  The plan holds a re-ordered COPY of the matrix (include/loops/kernels/panel_binned.hxx): nonzeros
 * sorted by (panel of W = 64 KB (or 128 KB) / sizeof(T) consecutive columns, sub-band of Hw consecutive rows) -- value + 16-bit column
 * inside the panel + one int per 4 items saying where their products go -- and, in (sub-band, panel) order, the 16-bit row
 * inside the sub-band + a products scratch.  y = A x runs as two streaming kernels: products with the x panel held in LDS
 * (the x value of a nonzero is an LDS read, not a memory gather), stored in 16-byte groups so that every sub-band's products
 * form one contiguous run; then one workgroup per sub-band adds its run into LDS-resident accumulators (one set per
 * wavefront, combined in wavefront order; its work list -- windows of <= 256 items -- is built with the plan) and stores its
 * rows of y.  17 bytes of HBM traffic per nonzero (4-byte values) and no scattered read; y needs no zero-fill; reproducible (no floating-point atomics on global memory; LDS adds in program order).
 * Creation is synchronous (device radix sort + scans, O(nnz)); LOOPS_E_RANGE when panels x sub-bands exceed 2^26 or
 * nnz + padding reaches 2^31.  info7 = {W, Hw, panels, sub-bands, items incl. padding, kernel-A chunks, sizeof(T)}.
 * loops_panel_plan_arrays: HOST copies (any pointer may be NULL) for inspection and tests: values / col16 / perm [padded]
 * and dst4 [padded / 4] in (panel, sub-band) order, row16 [padded] in (sub-band, panel) order, subband_start [sub-bands + 1].
 * The fan-out entries store the finished rows of y to the peers as well (see loops_spmv_merge_path_fanout_f32).
 * One product in flight per plan. */
typedef struct loops_panel_plan loops_panel_plan_t;
/* subband_rows: 0 = automatic (the power of two that brings a (panel, sub-band) segment to ~192 nonzeros, within 256 rows ..
 * 16 KB of accumulators per wavefront, at least 512 sub-bands when the matrix has the rows for it), or an explicit power of
 * two in [64, 4096] for 4-byte values, [64, 2048] for 8-byte values (LOOPS_E_BADARG otherwise). */
/* panel_columns: 0 = automatic (128 KB of x per panel whenever the matrix spans at least four 64 KB panels, 64 KB otherwise), or
 * explicitly 65536 / sizeof(T) or 131072 / sizeof(T) (LOOPS_E_BADARG otherwise). */
int loops_panel_plan_create_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                                int panel_columns, int subband_rows, void* stream, loops_panel_plan_t** out);
int loops_panel_plan_create_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                                int panel_columns, int subband_rows, void* stream, loops_panel_plan_t** out);
/* The same with the B order chosen explicitly.  compact: -1 = automatic (what loops_panel_plan_create_* does), 0 = one slot
 * per nonzero, 1 = COMPACT: kernel A sums runs of equal (row, panel) -- cut so that none crosses a 256-item wavefront step --
 * before they leave the CU, and the B order holds one slot per run (matrices with column locality: a row's 20-30 products
 * become 1-2 slots; adopted automatically when the runs are at most 0.7 of the nonzeros).  Then col16 carries the run-end
 * flag in bit 15, dst4 the slot of the group's first run end and in bit 31 "the group holds padding", and row16 / the
 * products scratch have loops_panel_plan_layout's info4[2] slots.
 * loops_panel_plan_layout: info4 = {compact, runs, B-order slots incl. padding, items per wavefront step of kernel A (the
 * window no run crosses)}. */
int loops_panel_plan_create_layout_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                                       int panel_columns, int subband_rows, int compact, void* stream, loops_panel_plan_t** out);
int loops_panel_plan_create_layout_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                                       int panel_columns, int subband_rows, int compact, void* stream, loops_panel_plan_t** out);
int loops_panel_plan_layout(const loops_panel_plan_t* plan, long long* info4);
void loops_panel_plan_destroy(loops_panel_plan_t* plan);
int loops_panel_plan_info(const loops_panel_plan_t* plan, int* info7);
/* Row blocks (round 5): with automatic parameters a matrix of 2^28 nonzeros or more is held as independent copies over contiguous row
 * blocks of ~2^26 nonzeros, run back to back, so that kernel B reads a block's products while they are still in the 256 MB Infinity
 * Cache (C5 on one GPU: 2.10 -> 1.91 ms; smaller inputs are best as one copy and stay one).  count = the number of blocks (1 = one
 * copy); row_bounds (optional, count + 1 entries) = their first rows.  For a blocked plan the info calls describe the whole (sums
 * over the blocks, W / Hw of the first) and loops_panel_plan_arrays / _windows return LOOPS_E_CONFIG. */
int loops_panel_plan_row_blocks(const loops_panel_plan_t* plan, int* count, int* row_bounds);
int loops_panel_plan_arrays(const loops_panel_plan_t* plan, void* values, unsigned short* col16, int* dst4, unsigned short* row16,
                            int* perm, int* subband_start);
/* Kernel B's work list (synchronous copies): window_start[subbands + 1]; windows[2 * window_start[subbands]] = {first item,
 * items | packed << 16} in (sub-band, panel) order (NULL: only the starts, to size the second call); segment_start
 * [subbands * panels + 1] (may be NULL).  A window holds at most 256 items: a piece of ONE segment (rows sorted), or -- packed
 * -- consecutive segments / segment tails of at most 64 (f64: 128) items each. */
int loops_panel_plan_windows(const loops_panel_plan_t* plan, int* window_start, int* windows, int* segment_start);
int loops_panel_plan_refresh_values_f32(loops_panel_plan_t* plan, const float* values, void* stream);
int loops_panel_plan_refresh_values_f64(loops_panel_plan_t* plan, const double* values, void* stream);
int loops_spmv_panel_f32(const loops_panel_plan_t* plan, const float* x, float* y, void* stream);
int loops_spmv_panel_f64(const loops_panel_plan_t* plan, const double* x, double* y, void* stream);
/* one kernel at a time for timing: stage 0 = products, 1 = sub-band reduce */
int loops_spmv_panel_stage_f32(const loops_panel_plan_t* plan, int stage, const float* x, float* y, void* stream);
int loops_spmv_panel_stage_f64(const loops_panel_plan_t* plan, int stage, const double* x, double* y, void* stream);
int loops_spmv_panel_fanout_f32(const loops_panel_plan_t* plan, const float* x, float* y, int num_peers, float* const* h_peer_y,
                                void* stream);
int loops_spmv_panel_fanout_f64(const loops_panel_plan_t* plan, const double* x, double* y, int num_peers, double* const* h_peer_y,
                                void* stream);

/* ---- row-band layout: y accumulators of a band of rows in LDS, x read through column-sorted (coalescing) gathers ----------
 * No reference counterpart (the reference's merge_path_flat.cuh:71-82 issues one global atomic per nonzero; its CSR kernels one
 * scattered x gather per nonzero).  The plan holds a re-ordered COPY of the matrix (include/loops/kernels/rowband.hxx): nonzeros
 * sorted by (band of H consecutive rows, column, CSR order), every band padded to steps of 256 slots; per slot the value, a 16-bit
 * row code and the one-byte column DELTA to the previous slot -- 7 bytes per nonzero with 4-byte values -- plus, per group of 64
 * slots, the absolute column of its first slot; gaps of more than 255 columns are bridged by padding slots.  Row code: the row
 * inside the band; H = padding; above H: one of 16 replicated accumulators of a HUB row (a row holding >= 1/128 of its band's
 * nonzeros, at most 32 per band), so that the lanes of one instruction do not meet in one LDS word.
 * y = A x: one workgroup per chunk of a band adds its products into fp64 words of LDS (ds_add_f64) and stores the band's rows
 * -- straight to y, or, where a band was cut into several chunks, as fp32 partial vectors that a second small kernel adds in
 * chunk order.  Products are fp32 (one rounding each), all sums fp64, y is rounded once (twice where a band was cut): bit-equal to
 * the CSR kernels on exactly summable inputs, within 1e-6 of the f64-accumulated product otherwise (util/reference.hxx:146-166);
 * no global atomics; y needs no zero-fill.  4-byte values only.
 * band_rows: 0 = automatic, else a power of two in [64, 16384]; target_chunks: 0 = automatic, else about how many workgroups
 * the first kernel is cut into (LOOPS_E_BADARG otherwise).  LOOPS_E_RANGE when the padded layout may reach 2^31 slots (nnz +
 * bands x (cols / 255 + 256)).  Creation is synchronous (device radix sort, O(nnz)).
 * info8 = {H, bands, padding slots that bridge column gaps, steps of 256 slots incl. padding, chunks, partial vectors, bands cut into several chunks,
 * wavefronts per workgroup of the first kernel}.  loops_rowband_plan_arrays: HOST copies (any pointer may be NULL): values / row16 / delta8 / perm [steps * 256]
 * (perm = CSR position of the slot, -1 = padding), stepbase [4 * steps], chunks [4 * chunks] = {band, first step, end step, partial slot or -1},
 * multi [3 * cut bands] = {band, first partial slot, chunks}, hubs [bands * 33] = per band the number of hubs, then their rows
 * inside the band.  loops_rowband_plan_set_chunks re-cuts the bands of a built plan (tuning; synchronous).
 * loops_rowband_plan_tune times the product with 8 and 16 wavefronts per workgroup of the first kernel (`repeats` launches each,
 * <= 0: 10; ms2, may be NULL: the two times in ms) and keeps the faster; loops_rowband_plan_set_waves sets it (8 or 16;
 * a new plan runs 8). */
typedef struct loops_rowband_plan loops_rowband_plan_t;
int loops_rowband_plan_create_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                                  int band_rows, int target_chunks, void* stream, loops_rowband_plan_t** out);
/* 8-byte values: the same record per slot (3 bytes of row code + column delta), fp64 products, fp64 LDS sums, fp64 partial vectors. */
int loops_rowband_plan_create_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                                  int band_rows, int target_chunks, void* stream, loops_rowband_plan_t** out);
void loops_rowband_plan_destroy(loops_rowband_plan_t* plan);
int loops_rowband_plan_info(const loops_rowband_plan_t* plan, int* info8);
int loops_rowband_plan_arrays(const loops_rowband_plan_t* plan, void* values, unsigned short* row16, unsigned char* delta8, int* perm,
                              int* stepbase, int* chunks, int* multi, unsigned short* hubs);
int loops_rowband_plan_set_chunks(loops_rowband_plan_t* plan, int target_chunks);
int loops_rowband_plan_tune(loops_rowband_plan_t* plan, int repeats, float* ms2, void* stream);
int loops_rowband_plan_set_waves(loops_rowband_plan_t* plan, int waves);
int loops_rowband_plan_refresh_values_f32(loops_rowband_plan_t* plan, const float* values, void* stream);
int loops_rowband_plan_refresh_values_f64(loops_rowband_plan_t* plan, const double* values, void* stream);
int loops_spmv_rowband_f32(const loops_rowband_plan_t* plan, const float* x, float* y, void* stream);
int loops_spmv_rowband_f64(const loops_rowband_plan_t* plan, const double* x, double* y, void* stream);
int loops_spmv_rowband_stage_f64(const loops_rowband_plan_t* plan, int stage, const double* x, double* y, void* stream);
/* one kernel at a time for timing: stage 0 = accumulate, 1 = combine */
int loops_spmv_rowband_stage_f32(const loops_rowband_plan_t* plan, int stage, const float* x, float* y, void* stream);
int loops_spmv_rowband_fanout_f32(const loops_rowband_plan_t* plan, const float* x, float* y, int num_peers, float* const* h_peer_y,
                                  void* stream);

/* ---- SpMV plan: tile shape AND layout chosen at plan time ------------------------------------------------
 * What an iterative caller holds for one matrix.  The reference fixes both at compile time (launch_box.hxx:56-90) and always
 * runs the CSR as given; here the plan decides per matrix, once:
 *   flags & LOOPS_PLAN_MEASURE     time the candidates on the device at creation (256 x 8 and 512 x 8 merge tiles over the
 *                                  unmodified CSR; `repeats` launches each, <= 0: 10) instead of choosing by structure alone;
 *   flags & LOOPS_PLAN_ALLOW_COPY  the plan may keep a re-ordered COPY of the matrix (about another nnz * (4 + sizeof(T)) bytes + tables):
 *                                  the row-band copy ("row-band layout" above; not under LOOPS_PLAN_DETERMINISTIC) or the panel-binned copy
 *                                  ("panel-binned layout" above).  With MEASURE both are built and timed (x of at least 1 MB /
 *                                  2 MB) and one is adopted only if >= 5 % faster than the best CSR shape (and than the other);
 *                                  without MEASURE a copy is taken by size alone: panel-binned when cols * sizeof(T) > 6 MB,
 *                                  row-band (mean row >= 8 nonzeros) from 2 MB.  (Structure does not show
 *                                  column locality -- a narrow band is faster from the CSR as given, a wide one from the
 *                                  row-band copy whatever the size of x: MEASURE finds out.)
 *                                  A plan that stays on the CSR takes 512 x 8 (from 16 parts of x on: 256 x 16) tiles with phased x gathers (LOOPS_VARIANT_PHASED) when
 *                                  the columns LOOK scattered over an x of 3 MB or more (loops_columns_look_scattered).
 * Without ALLOW_COPY the product always runs on the caller's arrays.  Creation is synchronous.  One product in flight per plan.
 * loops_spmv_planned_*: y = A x; offsets / indices / values are the arrays the plan was created from (ignored -- may be NULL
 * -- when the plan holds a copy; after changing the VALUES of the matrix call loops_spmv_plan_refresh_values_* first).
 * loops_spmv_plan_info: layout (LOOPS_LAYOUT_*), tile config, number of bands / panels (0 for CSR), and ms4[4] = measured
 * ms per product of {CSR 256 x 8, CSR 512 x 8, row-band, panel-binned}, -1 where not timed.  Any output pointer may be NULL. */
#define LOOPS_PLAN_MEASURE 1
#define LOOPS_PLAN_ALLOW_COPY 2
/* Only layouts whose summation order is fixed (the CSR kernels, the panel-binned copy): the row-band copy adds a row's products
 * with fp64 LDS atomics that arrive in no fixed order -- exact, hence reproducible, whenever the fp64 sum of a row's products is
 * exact (fp32 products spanning < 53 - 24 - log2 n binary orders of magnitude per row; the exactly summable inputs of the
 * tests), otherwise the last bit of a row may differ from run to run and between devices. */
#define LOOPS_PLAN_DETERMINISTIC 4
#define LOOPS_LAYOUT_CSR 0
/* (1 was the column-blocked layout, retired in round 5: never returned) */
#define LOOPS_LAYOUT_PANEL_BINNED 2
#define LOOPS_LAYOUT_ROW_BAND 3
typedef struct loops_spmv_plan loops_spmv_plan_t;
int loops_spmv_plan_create_f32(int rows, int cols, int nnz, const int* offsets, const int* indices, const float* values,
                               int flags, int repeats, void* stream, loops_spmv_plan_t** out);
int loops_spmv_plan_create_f64(int rows, int cols, int nnz, const int* offsets, const int* indices, const double* values,
                               int flags, int repeats, void* stream, loops_spmv_plan_t** out);
void loops_spmv_plan_destroy(loops_spmv_plan_t* plan);
int loops_spmv_plan_info(const loops_spmv_plan_t* plan, int* layout, int* tile_config, int* num_blocks, float* ms4);
/* LOOPS_LAYOUT_CSR plans: the kernel variant the plan runs (0 or LOOPS_VARIANT_PHASED -- with LOOPS_PLAN_MEASURE the phased
 * twins of 512 x 8 and 256 x 16 are candidates when x is at least 1 MB, adopted when > 2 % faster than the best plain shape) and the best phased candidate's measured ms per product (-1 = not timed).  Either pointer may be NULL. */
int loops_spmv_plan_variant(const loops_spmv_plan_t* plan, int* variant, float* ms_phased);
int loops_spmv_plan_refresh_values_f32(loops_spmv_plan_t* plan, const float* values, void* stream);
int loops_spmv_plan_refresh_values_f64(loops_spmv_plan_t* plan, const double* values, void* stream);
int loops_spmv_planned_f32(const loops_spmv_plan_t* plan, const int* offsets, const int* indices, const float* values,
                           const float* x, float* y, void* stream);
int loops_spmv_planned_f64(const loops_spmv_plan_t* plan, const int* offsets, const int* indices, const double* values,
                           const double* x, double* y, void* stream);

/* ---- COO SpMV ------------------------------------------------------------------------------------
 * Replaces algorithms::spmv::coo_thread_mapped (algorithms/spmv/coo_thread_mapped.cuh:37-100).
 * mode 0: the reference shape, one atomicAdd per nonzero, y zero-filled by the CALLER;
 * mode 1: tuned -- 8 consecutive triplets per lane (16-byte loads), one atomicAdd per run of equal row
 * indices, y zero-filled here; any triplet order is correct, row-sorted order is the fast case. */
int loops_spmv_coo_f32(int mode, int rows, int cols, int nnz, const int* row_indices, const int* col_indices,
                       const float* values, const float* x, float* y, void* stream);
int loops_spmv_coo_f64(int mode, int rows, int cols, int nnz, const int* row_indices, const int* col_indices,
                       const double* values, const double* x, double* y, void* stream);

/* ---- ELL SpMV ------------------------------------------------------------------------------------
 * Replaces algorithms::spmv::ell_thread_mapped (algorithms/spmv/ell_thread_mapped.cuh:36-85).  ELL as the
 * reference stores it: ROW-major rows x pitch arrays, padding = negative column index
 * (container/ell.hxx:31-55).  mode 0: lane per row (reference shape); mode 1: tuned -- a row is read by
 * G lanes with 16-byte loads (contiguous runs) and reduced across lanes; mode 2: the merge_path_flat schedule over
 * the ELL cells on the fused merge-tile engine -- replaces algorithms::spmv::ell_merge_path
 * (algorithms/spmv/ell_merge_path.cuh:32-125: merge path of (row ends (r + 1) * pitch, rows * pitch cells), one
 * atomicAdd per cell there; here no atomics, row ends from a functor, padding cells contribute 0).  y is overwritten
 * in every mode.  rows * pitch + rows must stay below 2^31 for mode 2 (LOOPS_E_RANGE). */
int loops_spmv_ell_f32(int mode, int rows, int cols, int pitch, const int* indices, const float* values,
                       const float* x, float* y, void* stream);
int loops_spmv_ell_f64(int mode, int rows, int cols, int pitch, const int* indices, const double* values,
                       const double* x, double* y, void* stream);

/* ---- DIA SpMV ------------------------------------------------------------------------------------
 * Replaces algorithms::spmv::dia_thread_mapped (algorithms/spmv/dia_thread_mapped.cuh:36-110).  DIA as the reference
 * stores it (container/dia.hxx:69-230): diag_offsets[d] = (col - row) of stored diagonal d, values column-major,
 * values[d * stride + r], stride >= rows.  mode 0: lane per row (reference shape); mode 1: tuned -- a lane owns four
 * consecutive rows (16-byte loads when stride % 4 == 0), several diagonals in flight.  y is overwritten. */
int loops_spmv_dia_f32(int mode, int rows, int cols, int num_diagonals, size_t stride, const int* diag_offsets,
                       const float* values, const float* x, float* y, void* stream);
int loops_spmv_dia_f64(int mode, int rows, int cols, int num_diagonals, size_t stride, const int* diag_offsets,
                       const double* values, const double* x, double* y, void* stream);

/* ---- launch-box autotuner -----------------------------------------------------------
 * The reference picks (threads per block, items per thread) per architecture from a compile-time
 * table of "analytical" values (algorithms/spmv/launch_box.hxx:56-90).  This times the planned
 * merge_path_flat SpMV of THIS matrix with every compiled tile shape (2 warm-up + `repeats` timed
 * launches each, hipEvents on `stream`) and returns the fastest loops_tile_config, to be passed to
 * loops_merge_plan_create.  ms_per_config (optional) receives the mean time per config index, -1 for
 * shapes that are not timed.  y is overwritten with A x.  Synchronous. */
int loops_autotune_merge_path_f32(int rows, int cols, int nnz, const int* offsets, const int* indices,
                                  const float* values, const float* x, float* y, int repeats, void* stream,
                                  int* best_tile_config, float* ms_per_config);
/* The same over tile shapes AND kernel variants: additionally times the phased-gather twin (LOOPS_VARIANT_PHASED) of the
 * shapes that have one (plans of more than one tile).  best_variant = 0 or LOOPS_VARIANT_PHASED, to be passed to
 * loops_spmv_merge_path_*; ms_per_config (optional, 12 entries): [cfg] = plain, [6 + cfg] = phased, -1 where not timed. */
/* A STRUCTURAL guess at the same question, for callers that cannot measure: *scattered = 1 when x (cols x value_bytes) is at
 * least 3 MB (6 MB for 8-byte values), the matrix holds at least 2^20 nonzeros, of 16 384 sampled pairs of nonzeros one merge tile
 * apart fewer than 2 / parts share a part of x (parts = 8 / 16 / 32 by the size of x; uniformly random columns: 1 / parts; hub columns
 * at neighbouring ids: 2-3 / parts; bands, host blocks, dense hub rows: most)
 * and fewer than a quarter of the adjacent pairs share a 128-byte line of x (runs of consecutive columns gather cheaply).  What the
 * plan-less C++ wrapper algorithms::spmv::merge_path_flat(csr, x, y) consults in its untimed set-up.  Synchronous. */
int loops_columns_look_scattered(int cols, int nnz, const int* indices, int value_bytes, void* stream, int* scattered);
int loops_autotune_merge_path_variants_f32(int rows, int cols, int nnz, const int* offsets, const int* indices,
                                           const float* values, const float* x, float* y, int repeats, void* stream,
                                           int* best_tile_config, int* best_variant, float* ms_per_config);

/* ---- CSC SpMV ------------------------------------------------------------------------------------
 * Replaces algorithms::spmv::csc_thread_mapped (algorithms/spmv/csc_thread_mapped.cuh:36-95).
 * mode 0: lane per column (reference shape), y zero-filled by the CALLER; mode 1: tuned -- the nonzeros
 * are split evenly over the lanes (8 consecutive ones each, 16-byte loads, column found by a search over
 * the column offsets), one atomicAdd per nonzero, y zero-filled here; from 2^20 nonzeros on the BINNED product instead
 * (include/loops/kernels/csc_spmv.hxx: products counted and scattered into bins of 4 096 rows, fp64 LDS sums -- no atomic per nonzero:
 * C2 0.26 against 1.03 ms, a hub row of 2^19 nonzeros 0.21 against 6.6 ms; sums exact on exactly summable inputs, within an ulp
 * of the fp64 sum otherwise; uses a per-stream scratch block of ~3 x 4 bytes per nonzero that loops_release_scratch frees). */
int loops_spmv_csc_f32(int mode, int rows, int cols, int nnz, const int* col_offsets, const int* row_indices,
                       const float* values, const float* x, float* y, void* stream);
int loops_spmv_csc_f64(int mode, int rows, int cols, int nnz, const int* col_offsets, const int* row_indices,
                       const double* values, const double* x, double* y, void* stream);

/* ---- CSC plan: the storage transposed once -------------------------------------------------------------------------------
 * A CSC product scatters into y: one global atomic per nonzero, ~16 G/s on this chip -- 1.04 ms on the C2 matrix whatever the
 * kernel, ten times the CSR product of the same matrix.  A caller that multiplies more than once holds this plan instead: it
 * transposes the storage to CSR on the device once (one 64-bit radix sort of (row, column) keys; rows of y then have one
 * writer and need no zero-fill) and keeps a SpMV plan (above; same `flags` / `repeats`) over that copy.
 * loops_csc_plan_info = loops_spmv_plan_info of the inner plan.  After changing the VALUES (same structure) call
 * loops_csc_plan_refresh_values_* with the CSC values array.  One product in flight per plan.  No reference counterpart
 * (algorithms/spmv/csc_thread_mapped.cuh:36-95 is the scatter). */
typedef struct loops_csc_plan loops_csc_plan_t;
int loops_csc_plan_create_f32(int rows, int cols, int nnz, const int* col_offsets, const int* row_indices, const float* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out);
int loops_csc_plan_create_f64(int rows, int cols, int nnz, const int* col_offsets, const int* row_indices, const double* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out);
/* The same handle from COO triplets in ANY order (duplicates are added up in their original order): the tuned COO kernel
 * (loops_spmv_coo_*) needs row-sorted triplets to be fast and atomics otherwise; this sorts once.  Everything else as above. */
int loops_coo_plan_create_f32(int rows, int cols, int nnz, const int* row_indices, const int* col_indices, const float* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out);
int loops_coo_plan_create_f64(int rows, int cols, int nnz, const int* row_indices, const int* col_indices, const double* values,
                              int flags, int repeats, void* stream, loops_csc_plan_t** out);
void loops_csc_plan_destroy(loops_csc_plan_t* plan);
int loops_csc_plan_info(const loops_csc_plan_t* plan, int* layout, int* tile_config, int* num_blocks, float* ms4);
int loops_csc_plan_refresh_values_f32(loops_csc_plan_t* plan, const float* values, void* stream);
int loops_csc_plan_refresh_values_f64(loops_csc_plan_t* plan, const double* values, void* stream);
int loops_spmv_csc_planned_f32(const loops_csc_plan_t* plan, const float* x, float* y, void* stream);
int loops_spmv_csc_planned_f64(const loops_csc_plan_t* plan, const double* x, double* y, void* stream);

/* ---- multi-GPU: row-range partition + allgatherv over RCCL (include/loops/multi_gpu/) --------------------------------------
 * The reference is single-GPU; this is the leg BASELINE.json's C5 adds.  y = A x is independent per row: the CSR is cut into
 * `parts` contiguous row ranges balanced by rows + nonzeros (the merge-path diagonal split the kernels use per workgroup,
 * reference include/loops/util/search.hxx:35-60, applied per GPU), x is replicated, and the slices of y are exchanged by one
 * group of RCCL point-to-point operations (every xGMI link once, all at the same time).
 * loops_row_ranges: HOST function over HOST offsets (rows + 1 entries); bounds has parts + 1 entries; LOOPS_E_RANGE when
 * rows + nnz reaches 2^31.
 * loops_comm_*: a communicator of this library's own (ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy of the RCCL the
 * process has loaded -- resolved at run time, no link-time dependency): rank 0 obtains the 128-byte id, the caller ships it to
 * the other ranks (any transport), every rank calls loops_comm_init.  A caller that already owns an ncclComm_t passes that
 * instead (same RCCL instance).  LOOPS_E_CONFIG: no RCCL entry points found in the process or on the library path; other
 * non-zero codes are ncclResult_t values (loops_comm_error_string).
 * loops_allgatherv_*: on entry rank r has written y_full[bounds[r] .. bounds[r + 1]); after the call (stream-ordered) y_full
 * is complete on every rank.  `bounds` is a HOST array.  Unmeasured on multi-GPU hardware so far. */
int loops_row_ranges(int rows, const int* offsets, int parts, long long* bounds);
int loops_comm_unique_id(void* id128);
int loops_comm_init(int world, int rank, const void* id128, void** comm);
int loops_comm_destroy(void* comm);
const char* loops_comm_error_string(int code);
int loops_allgatherv_f32(void* comm, int rank, int world, float* y_full, const long long* bounds, void* stream);
int loops_allgatherv_f64(void* comm, int rank, int world, double* y_full, const long long* bounds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOOPS_AMD_H_ */
