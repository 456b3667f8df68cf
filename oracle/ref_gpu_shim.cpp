// oracle/ref_gpu_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The reference's OWN HIP device path (its -DLOOPS_BACKEND=HIP switch), compiled for gfx950
// from the headers where they lie under /root/reference/include, behind a tiny extern "C"
// surface taking host arrays.  Built in the dev container into
// oracle/_ref/libloops_ref_gpu.so and shipped prebuilt to the GPU box.  Purposes:
//   * pin the oracle's restatement of the DEVICE-only schedule code (merge-path per-block /
//     per-thread coordinates, work_oriented thread maps) against the reference's real device
//     code executed on the MI355X (tests/test_ref_gpu_pin.py);
//   * y parity of our kernels vs the reference's kernels on the same GPU;
//   * a "reference-as-shipped on MI355X" timing next to ours (bench.py --ref-gpu), using the
//     reference's own util::timer_t bracketing (algorithms/spmv/merge_path_flat.cuh:121-136).
//
// Reference entry points used (file:line under /root/reference):
//   algorithms::spmv::merge_path_flat   include/loops/algorithms/spmv/merge_path_flat.cuh:97-139
//   algorithms::spmv::thread_mapped     include/loops/algorithms/spmv/thread_mapped.cuh:70-91
//   algorithms::spmv::work_oriented     include/loops/algorithms/spmv/work_oriented.cuh:103-121
//   schedule::setup<merge_path_flat>    include/loops/schedule/merge_path_flat.hxx:193-390
//   schedule::setup<work_oriented>      include/loops/schedule/work_oriented.hxx:45-190
//   merge_path::preprocess_t            include/loops/schedule/merge_path_flat.hxx:99-172
//   algorithms::spmm::thread_mapped     include/loops/algorithms/spmm/thread_mapped.cuh:68-94
//   algorithms::spmv::coo_/csc_/ell_thread_mapped, bcsr_thread_mapped<4,4>
//                                       include/loops/algorithms/spmv/{coo,csc,ell,dia,bcsr}_thread_mapped.cuh, ell_merge_path.cuh
// (group_mapped is excluded from the reference's HIP build: schedule.hxx:69-74.)
#include <loops/schedule.hxx>
#include <loops/container/formats.hxx>
#include <loops/container/vector.hxx>
#include <loops/util/launch.hxx>
#include <loops/util/timer.hxx>
#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/algorithms/spmv/thread_mapped.cuh>
#include <loops/algorithms/spmv/work_oriented.cuh>
#include <loops/algorithms/spmm/thread_mapped.cuh>
#include <loops/algorithms/spmv/coo_thread_mapped.cuh>
#include <loops/algorithms/spmv/csc_thread_mapped.cuh>
#include <loops/algorithms/spmv/ell_thread_mapped.cuh>
#include <loops/algorithms/spmv/ell_merge_path.cuh>
#include <loops/algorithms/spmv/dia_thread_mapped.cuh>
#include <loops/algorithms/spmv/bcsr_thread_mapped.cuh>

#include <algorithm>

using namespace loops;
using dev_csr = csr_t<int, int, float>;
using host_csr = csr_t<int, int, float, memory_space_t::host>;

static host_csr make_host(long rows, long cols, long nnz, const int* off, const int* idx, const float* val) {
  host_csr c(rows, cols, nnz);
  std::copy(off, off + rows + 1, c.offsets.begin());
  std::copy(idx, idx + nnz, c.indices.begin());
  std::copy(val, val + nnz, c.values.begin());
  return c;
}

// ---- dump kernels: run the reference's schedule objects and record what they hand out ----
template <std::size_t TPB, std::size_t IPT, typename meta_t>
__global__ void __launch_bounds__(int(TPB))
dump_merge_path(meta_t meta, std::size_t rows, std::size_t nnz, int* offsets,
                unsigned* thread_start, int* atom_owner, int* atom_row, int* atom_visits) {
  using setup_t = schedule::setup<schedule::algorithms_t::merge_path_flat, TPB, IPT, int, int,
                                  std::size_t, std::size_t>;
  using storage_t = typename setup_t::storage_t;
  __shared__ storage_t temporary_storage;
  setup_t config(meta, temporary_storage, offsets, rows, nnz);
  auto map = config.init();
  if (!config.is_valid_accessor(map)) return;
  const std::size_t tile = blockIdx.x * gridDim.y + blockIdx.y;
  const std::size_t gt = tile * TPB + threadIdx.x;
  thread_start[2 * gt] = map.x;
  thread_start[2 * gt + 1] = map.y;
  for (auto item : config.virtual_idx()) {
    auto nz = config.atom_idx(item, map);
    auto row = config.tile_idx(map);
    if (config.atoms_counting_it[map.y] < temporary_storage.tile_end_offset[map.x]) {
      atom_owner[nz] = int(gt);
      atom_row[nz] = row;
      atomicAdd(&atom_visits[nz], 1);
      map.y++;
    } else {
      map.x++;
    }
  }
}

template <std::size_t TPB>
__global__ void __launch_bounds__(TPB)
dump_work_oriented(std::size_t rows, std::size_t nnz, int* offsets, int* thread_map,
                   int* atom_owner, int* atom_row, int* atom_visits) {
  using setup_t = schedule::setup<schedule::algorithms_t::work_oriented, TPB, 1, int, int,
                                  std::size_t, std::size_t>;
  setup_t config(offsets, rows, nnz);
  auto map = config.init();
  const int g = threadIdx.x + blockIdx.x * blockDim.x;
  thread_map[4 * g + 0] = map.first.first;
  thread_map[4 * g + 1] = map.first.second;
  thread_map[4 * g + 2] = map.second.first;
  thread_map[4 * g + 3] = map.second.second;
  for (auto row : config.tiles(map)) {
    for (auto nz : config.atoms(row, map)) {
      atom_owner[nz] = g; atom_row[nz] = row; atomicAdd(&atom_visits[nz], 1);
    }
  }
  for (auto row : config.remainder_tiles(map)) {
    for (auto nz : config.remainder_atoms(map)) {
      atom_owner[nz] = g; atom_row[nz] = row; atomicAdd(&atom_visits[nz], 1);
    }
  }
}

template <std::size_t TPB, std::size_t IPT>
static int dump_mp(long rows, long nnz, const int* h_off, unsigned* h_thread_start,
                   int* h_owner, int* h_row, int* h_visits) {
  using pre_t = schedule::merge_path::preprocess_t<TPB, IPT, int, int, std::size_t, std::size_t>;
  thrust::device_vector<int> off(h_off, h_off + rows + 1);
  const long M = math::ceil_div(rows + nnz, long(TPB * IPT));
  thrust::device_vector<unsigned> ts(2 * M * TPB, 0u);
  thrust::device_vector<int> owner(std::max(nnz, 1L), -1), arow(std::max(nnz, 1L), -1), vis(std::max(nnz, 1L), 0);
  pre_t meta(off.data().get(), rows, nnz);
  launch::non_cooperative(0, dump_merge_path<TPB, IPT, pre_t>, dim3(M, 1, 1), dim3(TPB), meta,
                          std::size_t(rows), std::size_t(nnz), off.data().get(), ts.data().get(),
                          owner.data().get(), arow.data().get(), vis.data().get());
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  thrust::copy(ts.begin(), ts.end(), h_thread_start);
  thrust::copy(owner.begin(), owner.begin() + nnz, h_owner);
  thrust::copy(arow.begin(), arow.begin() + nnz, h_row);
  thrust::copy(vis.begin(), vis.begin() + nnz, h_visits);
  return 0;
}

// The reference's bcsr_thread_mapped<R, R> on this GPU, from ready block arrays (its host-side
// CSR -> BCSR builder is not what is being timed).  x_padded has num_block_cols * R entries.
template <std::size_t R, typename T>
static int ref_bcsr(long rows, long cols, long num_block_rows, long num_block_cols, long num_blocks,
                    const int* block_offsets, const int* block_cols, const T* block_values, const T* x_padded, T* y,
                    int iters, float* ms) {
  try {
    bcsr_t<R, R, int, int, T> b;
    b.rows = rows; b.cols = cols; b.nnzs = num_blocks * R * R;
    b.num_block_rows = num_block_rows; b.num_block_cols = num_block_cols; b.num_blocks = num_blocks;
    b.block_offsets = vector_t<int>(block_offsets, block_offsets + num_block_rows + 1);
    b.block_col_indices = vector_t<int>(block_cols, block_cols + num_blocks);
    b.values = vector_t<T>(block_values, block_values + num_blocks * R * R);
    vector_t<T> dx(x_padded, x_padded + num_block_cols * R);
    vector_t<T> dy(rows);
    float best = 1e30f;
    for (int it = 0; it < iters; ++it) {
      auto timer = algorithms::spmv::bcsr_thread_mapped(b, dx, dy);
      best = std::min(best, timer.milliseconds());
    }
    if (ms) *ms = best;
    thrust::copy(dy.begin(), dy.end(), y);
    return 0;
  } catch (...) { return 1; }
}

extern "C" {

// kind: 0 thread_mapped, 1 work_oriented, 2 merge_path_flat.  y is zero-filled first, as the
// reference's callers do (examples/spmv/merge_path.cu:29-30).  *ms = kernel time of the LAST of
// `iters` runs as the reference measures it (merge_path) or a hipEvent pair around the wrapper.
int refgpu_spmv_f32(int kind, long rows, long cols, long nnz, const int* off, const int* idx,
                    const float* val, const float* x, float* y, int iters, float* ms) {
  try {
    host_csr h = make_host(rows, cols, nnz, off, idx, val);
    dev_csr csr(h);
    vector_t<float> dx(x, x + cols);
    vector_t<float> dy(rows);
    float best = 1e30f;
    for (int it = 0; it < iters; ++it) {
      thrust::fill(dy.begin(), dy.end(), 0.0f);
      hipDeviceSynchronize();
      float t = 0;
      if (kind == 2) {
        auto timer = algorithms::spmv::merge_path_flat(csr, dx, dy);
        t = timer.milliseconds();
      } else {
        util::timer_t timer;
        timer.start();
        if (kind == 0) algorithms::spmv::thread_mapped(csr, dx, dy);
        else algorithms::spmv::work_oriented(csr, dx, dy);
        timer.stop();
        t = timer.milliseconds();
      }
      best = std::min(best, t);
    }
    if (ms) *ms = best;
    thrust::copy(dy.begin(), dy.end(), y);
    return 0;
  } catch (...) { return 1; }
}

// The reference's SpMM (C = A * B, dense row-major B [cols x n], C [rows x n]) on this GPU,
// bracketed with util::timer_t the way examples/spmm/thread_mapped.cu:36-41 does.
int refgpu_spmm_f32(long rows, long cols, long nnz, const int* off, const int* idx, const float* val,
                    const float* B, long n, float* Cm, int iters, float* ms) {
  try {
    host_csr h = make_host(rows, cols, nnz, off, idx, val);
    dev_csr csr(h);
    matrix_t<float> dB(cols, n), dC(rows, n);
    thrust::copy(B, B + cols * n, dB.m_data.begin());
    float best = 1e30f;
    for (int it = 0; it < iters; ++it) {
      util::timer_t timer;
      timer.start();
      algorithms::spmm::thread_mapped(csr, dB, dC);
      timer.stop();
      best = std::min(best, timer.milliseconds());
    }
    if (ms) *ms = best;
    thrust::copy(dC.m_data.begin(), dC.m_data.end(), Cm);
    return 0;
  } catch (...) { return 1; }
}

// The reference's kernels for the other sparse formats on this GPU.  The container is built from the
// CSR with the reference's OWN converting constructors (coo.hxx:88, csc.hxx:105, ell.hxx:114), on the
// device.  format: 0 = COO, 1 = CSC, 2 = ELL, 3 = DIA (dia.hxx:117, dia_thread_mapped.cuh:66-110), 4 = ELL through
// ell_merge_path (ell_merge_path.cuh:76-125).  y is zero-filled before every run (their contract).
int refgpu_format_spmv_f32(int format, long rows, long cols, long nnz, const int* off, const int* idx,
                           const float* val, const float* x, float* y, int iters, float* ms) {
  try {
    host_csr h = make_host(rows, cols, nnz, off, idx, val);
    dev_csr csr(h);
    vector_t<float> dx(x, x + cols);
    vector_t<float> dy(rows);
    float best = 1e30f;
    auto run = [&](auto&& call) {
      for (int it = 0; it < iters; ++it) {
        thrust::fill(dy.begin(), dy.end(), 0.0f);
        hipDeviceSynchronize();
        util::timer_t timer;
        timer.start();
        call();
        timer.stop();
        best = std::min(best, timer.milliseconds());
      }
    };
    if (format == 0) {
      coo_t<int, float> coo(csr);
      run([&] { algorithms::spmv::coo_thread_mapped(coo, dx, dy); });
    } else if (format == 1) {
      csc_t<int, int, float> csc(csr);
      run([&] { algorithms::spmv::csc_thread_mapped(csc, dx, dy); });
    } else if (format == 2) {
      ell_t<int, float> ell(csr);
      run([&] { algorithms::spmv::ell_thread_mapped(ell, dx, dy); });
    } else if (format == 3) {
      dia_t<int, int, float> dia(csr);
      run([&] { algorithms::spmv::dia_thread_mapped(dia, dx, dy); });
    } else if (format == 4) {
      ell_t<int, float> ell(csr);
      run([&] { algorithms::spmv::ell_merge_path(ell, dx, dy); });
    } else {
      return 2;
    }
    if (ms) *ms = best;
    thrust::copy(dy.begin(), dy.end(), y);
    return 0;
  } catch (...) { return 1; }
}

int refgpu_bcsr4x4_spmv_f32(long rows, long cols, long num_block_rows, long num_block_cols, long num_blocks,
                            const int* block_offsets, const int* block_cols, const float* block_values,
                            const float* x_padded, float* y, int iters, float* ms) {
  return ref_bcsr<4, float>(rows, cols, num_block_rows, num_block_cols, num_blocks, block_offsets, block_cols, block_values,
                            x_padded, y, iters, ms);
}

// Any of the square block shapes this repository compiles (2, 3, 4, 8), fp32 (`f64` = 0) or fp64 (`f64` = 1: the value
// pointers are double*).  Returns 2 for a shape that is not instantiated here.
int refgpu_bcsr_spmv(int R, int f64, long rows, long cols, long num_block_rows, long num_block_cols, long num_blocks,
                     const int* block_offsets, const int* block_cols, const void* block_values, const void* x_padded,
                     void* y, int iters, float* ms) {
#define REF_BCSR(RR)                                                                                                     \
  if (R == RR)                                                                                                           \
    return f64 ? ref_bcsr<RR, double>(rows, cols, num_block_rows, num_block_cols, num_blocks, block_offsets, block_cols, \
                                      static_cast<const double*>(block_values), static_cast<const double*>(x_padded),   \
                                      static_cast<double*>(y), iters, ms)                                               \
               : ref_bcsr<RR, float>(rows, cols, num_block_rows, num_block_cols, num_blocks, block_offsets, block_cols,  \
                                     static_cast<const float*>(block_values), static_cast<const float*>(x_padded),      \
                                     static_cast<float*>(y), iters, ms);
  REF_BCSR(2) REF_BCSR(3) REF_BCSR(4) REF_BCSR(8)
#undef REF_BCSR
  return 2;
}

// Merge-path assignment as handed out by the reference's device schedule.
// cfg: 0 -> (256, 8) [the gfx950 launch box], 1 -> (128, 7), 2 -> (4, 2) [tiny, for small fixtures].
int refgpu_merge_path_dump(int cfg, long rows, long nnz, const int* off, unsigned* thread_start,
                           int* atom_owner, int* atom_row, int* atom_visits) {
  try {
    if (cfg == 0) return dump_mp<256, 8>(rows, nnz, off, thread_start, atom_owner, atom_row, atom_visits);
    if (cfg == 1) return dump_mp<128, 7>(rows, nnz, off, thread_start, atom_owner, atom_row, atom_visits);
    if (cfg == 2) return dump_mp<4, 2>(rows, nnz, off, thread_start, atom_owner, atom_row, atom_visits);
    return 2;
  } catch (...) { return 1; }
}

// The reference's generate_search_coordinates pre-pass kernel, launched directly
// (schedule/merge_path_flat.hxx:45-76), TPB x IPT = 256 x 8.
int refgpu_merge_path_coords(long rows, long nnz, const int* h_off, unsigned* h_coords /* 2*(M+1) */) {
  try {
    constexpr std::size_t TPB = 256, IPT = 8;
    using layout_t = layout::csr<int, int>;
    thrust::device_vector<int> off(h_off, h_off + rows + 1);
    const std::size_t M = math::ceil_div(std::size_t(rows + nnz), TPB * IPT);
    thrust::device_vector<schedule::coord_t> coords(M + 1);
    layout_t lay(off.data().get(), rows, nnz);
    auto kernel = schedule::merge_path::generate_search_coordinates<TPB, IPT, layout_t, std::size_t, std::size_t>;
    launch::non_cooperative(0, kernel, dim3(math::ceil_div(M + 1, TPB)), dim3(TPB), lay,
                            std::size_t(rows), std::size_t(nnz), M, coords.data().get());
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    thrust::host_vector<schedule::coord_t> hc(coords);
    for (std::size_t i = 0; i <= M; ++i) { h_coords[2 * i] = hc[i].x; h_coords[2 * i + 1] = hc[i].y; }
    return 0;
  } catch (...) { return 1; }
}

int refgpu_work_oriented_dump(long rows, long nnz, const int* h_off, int grid, int* thread_map,
                              int* atom_owner, int* atom_row, int* atom_visits) {
  try {
    constexpr std::size_t TPB = 256;
    thrust::device_vector<int> off(h_off, h_off + rows + 1);
    thrust::device_vector<int> tm(4L * grid * TPB, 0);
    thrust::device_vector<int> owner(std::max(nnz, 1L), -1), arow(std::max(nnz, 1L), -1), vis(std::max(nnz, 1L), 0);
    launch::non_cooperative(0, dump_work_oriented<TPB>, dim3(grid), dim3(TPB), std::size_t(rows),
                            std::size_t(nnz), off.data().get(), tm.data().get(), owner.data().get(),
                            arow.data().get(), vis.data().get());
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    thrust::copy(tm.begin(), tm.end(), thread_map);
    thrust::copy(owner.begin(), owner.begin() + nnz, atom_owner);
    thrust::copy(arow.begin(), arow.begin() + nnz, atom_row);
    thrust::copy(vis.begin(), vis.begin() + nnz, atom_visits);
    return 0;
  } catch (...) { return 1; }
}

// Grid the reference would pick for work_oriented on this device (util/launch_box.hxx:228-239).
int refgpu_work_oriented_grid() {
  auto kernel = algorithms::spmv::__work_oriented<256, int, int, float>;
  return (int)launch_box::occupancy_grid(kernel, 256);
}

}  // extern "C"
