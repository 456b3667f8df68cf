// Device-side tests of the C++ header API (include/loops) -- the counterpart of the reference's
// SpMV battery (unittests/test_spmv_{csr,coo,csc,dia,ell,bcsr,partitioned}.cu + test_spmv_battery.hxx,
// restated, not copied): every `algorithms::spmv::*` wrapper, on a battery of small matrices, in
// f32 and f64, against `reference::spmv` on the host; plus plan reuse, device-side COO->CSR,
// duplicate removal, the device generator and SpMM.  Built in the dev container by
// tests/test_cpp_device_api.py (hipcc), run on the GPU box.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include <loops/container/formats.hxx>
#include <loops/container/matrix.cuh>
#include <loops/util/generate.hxx>
#include <loops/util/reference.hxx>
#include <loops/util/sample.hxx>
#include <loops/algorithms/spmv/original.cuh>
#include <loops/algorithms/spmv/thread_mapped.cuh>
#include <loops/algorithms/spmv/group_mapped.cuh>
#include <loops/algorithms/spmv/work_oriented.cuh>
#include <loops/algorithms/spmv/merge_path_flat.cuh>
#include <loops/algorithms/spmv/flat_partitioned.cuh>
#include <loops/algorithms/spmv/bcsr_thread_mapped.cuh>
#include <loops/algorithms/spmv/bcsr_band.cuh>
#include <loops/algorithms/spmv/bcsr_merge_path.cuh>
#include <loops/algorithms/spmv/coo_thread_mapped.cuh>
#include <loops/algorithms/spmv/csc_thread_mapped.cuh>
#include <loops/algorithms/spmv/dia_thread_mapped.cuh>
#include <loops/algorithms/spmv/ell_thread_mapped.cuh>
#include <loops/algorithms/spmv/ell_merge_path.cuh>
#include <loops/algorithms/spmm/thread_mapped.cuh>
#include <loops/algorithms/spmm/merge_path_flat.cuh>
#include <loops/algorithms/spmv/rowband.cuh>
#include <loops/algorithms/spmv/spmv_plan.cuh>

using namespace loops;
static int g_fail = 0, g_checks = 0;
#define CHECK(...) do { ++g_checks; if (!(__VA_ARGS__)) { ++g_fail; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #__VA_ARGS__); } } while (0)
constexpr auto H = memory_space_t::host;

template <typename T>
using hcsr_t = csr_t<int, int, T, H>;

template <typename T>
static hcsr_t<T> from_dense(const std::vector<std::vector<double>>& d) {
  const std::size_t rows = d.size(), cols = rows ? d[0].size() : 0;
  std::size_t nnz = 0;
  for (auto& r : d) for (double v : r) nnz += v != 0.0;
  hcsr_t<T> m(rows, cols, nnz);
  std::size_t k = 0;
  for (std::size_t r = 0; r < rows; ++r) {
    m.offsets[r] = static_cast<int>(k);
    for (std::size_t c = 0; c < cols; ++c)
      if (d[r][c] != 0.0) { m.indices[k] = static_cast<int>(c); m.values[k] = static_cast<T>(d[r][c]); ++k; }
  }
  m.offsets[rows] = static_cast<int>(k);
  return m;
}

// identity, banded, block-diagonal, skewed (one heavy row), empty rows, random, one long row
// spanning several 2048-item merge tiles, all-empty.
static std::vector<std::vector<std::vector<double>>> battery() {
  std::mt19937 rng(12345);
  std::uniform_real_distribution<double> u(0.5, 1.5);
  std::vector<std::vector<std::vector<double>>> out;
  auto zeros = [](std::size_t r, std::size_t c) { return std::vector<std::vector<double>>(r, std::vector<double>(c, 0.0)); };
  { auto d = zeros(16, 16); for (int i = 0; i < 16; ++i) d[i][i] = 1.0; out.push_back(d); }
  { auto d = zeros(32, 32); for (int i = 0; i < 32; ++i) for (int j = std::max(0, i - 3); j <= std::min(31, i + 4); ++j) d[i][j] = u(rng); out.push_back(d); }
  { auto d = zeros(9, 9); for (int b = 0; b < 3; ++b) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d[3 * b + i][3 * b + j] = u(rng); out.push_back(d); }
  { auto d = zeros(20, 50); for (int j = 0; j < 50; ++j) d[0][j] = u(rng); for (int i = 1; i < 20; ++i) { d[i][(7 * i) % 50] = u(rng); d[i][(13 * i + 5) % 50] = u(rng); } out.push_back(d); }
  { auto d = zeros(20, 12); for (int i = 0; i < 20; ++i) if (i % 4) for (int j = 0; j < 12; ++j) if (u(rng) < 0.8) d[i][j] = u(rng); out.push_back(d); }
  { auto d = zeros(50, 50); for (auto& r : d) for (auto& v : r) if (u(rng) < 0.55) v = u(rng) - 1.0; out.push_back(d); }
  { auto d = zeros(300, 6000); for (int j = 0; j < 6000; ++j) d[7][j] = u(rng); for (int i = 0; i < 300; ++i) d[i][(37 * i) % 6000] = u(rng); for (int i = 100; i < 140; ++i) for (auto& v : d[i]) v = 0.0; out.push_back(d); }
  { out.push_back(zeros(6, 4)); }
  return out;
}

template <typename T>
static void check_y(const char* what, int m, const thrust::device_vector<T>& y, const thrust::host_vector<T>& ref) {
  const std::size_t errors = reference::count_errors(y.data().get(), ref.data(), ref.size());
  ++g_checks;
  if (errors) { ++g_fail; std::printf("FAIL %s on matrix %d: %zu rows differ\n", what, m, errors); }
}

template <typename T>
static void run_battery() {
  int m = 0;
  for (auto& dense : battery()) {
    hcsr_t<T> h = from_dense<T>(dense);
    csr_t<int, int, T> csr(h);
    vector_t<T, H> xh(h.cols);
    generate::random::uniform_distribution(xh.begin(), xh.end(), 1, 10, 42u);
    vector_t<T> x(xh);
    const auto ref = reference::spmv(h, xh);
    {
      vector_t<T> xd(h.cols);  // device generator == host generator
      generate::random::uniform_distribution(xd.begin(), xd.end(), 1, 10, 42u);
      vector_t<T, H> back(xd);
      bool same = true;
      for (std::size_t i = 0; i < back.size(); ++i) same = same && back[i] == xh[i];
      CHECK(same);
    }
    auto fresh = [&] { return vector_t<T>(h.rows, T(0)); };
    { auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::original(csr, x, y); check_y("original", m, y, ref); }
    { auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::thread_mapped(csr, x, y); check_y("thread_mapped", m, y, ref); }
    { auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::group_mapped(csr, x, y); check_y("group_mapped", m, y, ref); }
    { auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::work_oriented(csr, x, y); check_y("work_oriented", m, y, ref); }
    { auto y = vector_t<T>(h.rows, T(7)); auto t = algorithms::spmv::merge_path_flat(csr, x, y); CHECK(t.milliseconds() >= 0.f); check_y("merge_path_flat", m, y, ref); }
    { auto y = fresh(); algorithms::spmv::flat_partitioned(csr, x, y); check_y("flat_partitioned<8>", m, y, ref); }
    { auto y = fresh(); algorithms::spmv::flat_partitioned<4>(csr, x, y); check_y("flat_partitioned<4>", m, y, ref); }
    if (h.rows && h.nnzs) {
      { coo_t<int, T> coo(csr); auto y = fresh(); algorithms::spmv::coo_run_mapped(coo, x, y); check_y("coo_run_mapped", m, y, ref); }
      { coo_t<int, T> coo(csr); auto y = fresh(); algorithms::spmv::coo_thread_mapped_schedule_api(coo, x, y); check_y("coo_thread_mapped_schedule_api", m, y, ref); }
      { coo_t<int, T> coo(csr); auto y = fresh(); algorithms::spmv::coo_thread_mapped(coo, x, y); check_y("coo_thread_mapped", m, y, ref);
        csr_t<int, int, T> again(coo);  // device-side COO -> CSR (radix sort + lower_bound)
        vector_t<int, H> o2(again.offsets); bool same = true; for (std::size_t i = 0; i <= h.rows; ++i) same = same && o2[i] == h.offsets[i]; CHECK(same); }
      { csc_t<int, int, T> csc(csr); auto y = fresh(); algorithms::spmv::csc_thread_mapped(csc, x, y); check_y("csc_thread_mapped", m, y, ref); }
      { csc_t<int, int, T> csc(csr); auto y = fresh(); algorithms::spmv::csc_nonzero_mapped(csc, x, y); check_y("csc_nonzero_mapped", m, y, ref); }
      { csc_t<int, int, T> csc(csr); auto y = fresh(); algorithms::spmv::csc_thread_mapped_schedule_api(csc, x, y); check_y("csc_thread_mapped_schedule_api", m, y, ref); }
      {  // CSC -> COO -> CSR on the device (coo_t(csc), csr_t(coo)), then the headline schedule: the transposed-once path
        csc_t<int, int, T> csc(csr);
        coo_t<int, T> coo(csc);
        csr_t<int, int, T> back(coo);
        auto y = fresh();
        algorithms::spmv::merge_path_flat(back, x, y);
        check_y("merge_path_flat over csr_t(coo_t(csc))", m, y, ref);
      }
      { ell_t<int, T> ell(csr); auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::ell_row_mapped(ell, x, y); check_y("ell_row_mapped", m, y, ref); }
      { ell_t<int, T> ell(csr); auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::ell_thread_mapped_schedule_api(ell, x, y); check_y("ell_thread_mapped_schedule_api", m, y, ref); }
      { ell_t<int, T> ell(csr); auto y = fresh(); algorithms::spmv::ell_thread_mapped(ell, x, y); check_y("ell_thread_mapped", m, y, ref);
        auto y2 = vector_t<T>(h.rows, T(7)); algorithms::spmv::ell_merge_path(ell, x, y2); check_y("ell_merge_path", m, y2, ref);  // fused engine: y not pre-zeroed
        auto y3 = fresh(); algorithms::spmv::ell_merge_path_atomic(ell, x, y3); check_y("ell_merge_path_atomic", m, y3, ref); }
      if (h.cols <= 64) { dia_t<int, int, T> dia(csr); auto y = fresh(); algorithms::spmv::dia_thread_mapped(dia, x, y); check_y("dia_thread_mapped", m, y, ref);
        auto y2 = vector_t<T>(h.rows, T(7)); algorithms::spmv::dia_row_mapped(dia, x, y2); check_y("dia_row_mapped", m, y2, ref);
        auto y3 = vector_t<T>(h.rows, T(7)); algorithms::spmv::dia_thread_mapped_schedule_api(dia, x, y3); check_y("dia_thread_mapped_schedule_api", m, y3, ref); }
      auto bcsr_case = [&](auto b, const char* what) {
        vector_t<T, H> xp(b.num_block_cols * b.kBlockCols, T(0));
        for (std::size_t i = 0; i < h.cols; ++i) xp[i] = xh[i];
        vector_t<T> xpd(xp);
        auto y = fresh();
        algorithms::spmv::bcsr_thread_mapped(b, xpd, y);
        check_y(what, m, y, ref);
        if constexpr (decltype(b)::kBlockRows == 4 && decltype(b)::kBlockCols == 4 && sizeof(T) == 4) {
          // 4 x 4 fp32: the first call looked at the block-row lengths and remembered their class in the matrix object; both
          // kernels the wrapper chooses from, by forcing the class
          CHECK(b.num_blocks == 0 || b.row_length_class == kernels::bcsr_rows_even || b.row_length_class == kernels::bcsr_rows_skewed);
          for (int cls : {kernels::bcsr_rows_even, kernels::bcsr_rows_skewed}) {
            b.row_length_class = cls;
            auto y2 = fresh();
            algorithms::spmv::bcsr_thread_mapped(b, xpd, y2);
            check_y(cls == kernels::bcsr_rows_skewed ? "bcsr<4,4> on merge-path tiles" : "bcsr<4,4> MFMA", m, y2, ref);
          }
        }
      };
      bcsr_case(bcsr_t<2, 2, int, int, T>(csr), "bcsr<2,2>");
      bcsr_case(bcsr_t<3, 3, int, int, T>(csr), "bcsr<3,3>");
      bcsr_case(bcsr_t<4, 4, int, int, T>(csr), "bcsr<4,4> (MFMA for f32)");
      if constexpr (sizeof(T) == 4) {  // the held block-band plan for 4 x 4 fp32 blocks: automatic and smallest bands, bands cut into chunks
        bcsr_t<4, 4, int, int, T> b4(csr);
        vector_t<T, H> xp(b4.num_block_cols * 4, T(0));
        for (std::size_t i = 0; i < h.cols; ++i) xp[i] = xh[i];
        vector_t<T> xpd(xp);
        { auto y = vector_t<T>(h.rows, T(7)); algorithms::spmv::bcsr_merge_path(b4, xpd, y); check_y("bcsr_merge_path", m, y, ref); }
        for (int cfg : {0, 1, 2}) {
          algorithms::spmv::bcsr_band_t<int, int> plan(b4, cfg == 0 ? 0 : 16, cfg == 2 ? 5 : 0);
          if (cfg == 2) plan.arrays.waves = 8, plan.arrays.unroll = 4;
          auto y = vector_t<T>(h.rows, T(7));  // (y needs no zero-fill)
          plan.spmv(xpd, y);
          check_y("bcsr_band_t", m, y, ref);
        }
      }
    }
    // plan reuse: one preprocess_t, several x (what an iterative solver does)
    if (h.rows) {
      using plan_t = algorithms::spmv::merge_path_plan_t<int, int, T>;
      plan_t plan(typename plan_t::layout_t(csr.offsets.data().get(), static_cast<int>(csr.rows), static_cast<int>(csr.nnzs)), 0, plan_t::prepass_always);
      for (unsigned seed : {7u, 8u, 9u}) {
        generate::random::uniform_distribution(xh.begin(), xh.end(), 1, 10, seed);
        vector_t<T> x2(xh);
        auto y = vector_t<T>(h.rows, T(-1));
        algorithms::spmv::merge_path_flat_async(plan, csr, x2, y);
        (void)xpu::stream_synchronize(0);
        check_y("merge_path_flat_async(plan)", m, y, reference::spmv(h, xh));
        if (seed == 8u) plan.classify();  // third round: the single-kernel path where the rows are short
      }
      // the multi-GPU epilogue fan-out (two stand-in "peers" on this device): y and both peer copies must equal the plain result
      {
        vector_t<T> x2(xh), y = vector_t<T>(h.rows, T(-1)), p0 = vector_t<T>(h.rows, T(-2)), p1 = vector_t<T>(h.rows, T(-3));
        kernels::peer_fanout<T> peers{};
        peers.count = 2;
        peers.base[0] = p0.data().get();
        peers.base[1] = p1.data().get();
        algorithms::spmv::merge_path_flat_fanout_async(plan, csr, x2, y, peers);
        (void)xpu::stream_synchronize(0);
        auto want = reference::spmv(h, xh);
        check_y("merge_path_flat_fanout_async y", m, y, want);
        check_y("merge_path_flat_fanout_async peer 0", m, p0, want);
        check_y("merge_path_flat_fanout_async peer 1", m, p1, want);
      }
      // work_oriented over a held plan (its launch box: 256 x 8 / 256 x 4)
      {
        using wplan_t = algorithms::spmv::work_oriented_plan_t<int, int, T>;
        wplan_t wplan(typename wplan_t::layout_t(csr.offsets.data().get(), static_cast<int>(csr.rows), static_cast<int>(csr.nnzs)), 0, wplan_t::prepass_always);
        vector_t<T> x2(xh), y = vector_t<T>(h.rows, T(-1));
        algorithms::spmv::work_oriented_async(wplan, csr, x2, y);
        (void)xpu::stream_synchronize(0);
        check_y("work_oriented_async(plan)", m, y, reference::spmv(h, xh));
      }
    }
    ++m;
  }
}

static void misc() {
  auto s = sample::csr<>();  // device
  vector_t<float> x(4, 1.0f), y(4);
  algorithms::spmv::merge_path_flat(s, x, y);
  vector_t<float, H> yh(y);
  CHECK(yh[0] == 0.f && yh[1] == 13.f && yh[2] == 3.f && yh[3] == 6.f);
  coo_t<int, float> dup(3, 3, 4);  // device-side duplicate removal
  { coo_t<int, float, H> hd(3, 3, 4); const int r[] = {1, 0, 1, 0}, c[] = {2, 0, 2, 0};
    for (int i = 0; i < 4; ++i) { hd.row_indices[i] = r[i]; hd.col_indices[i] = c[i]; hd.values[i] = float(i + 1); } dup = coo_t<int, float>(hd); }
  dup.remove_duplicates();
  CHECK(dup.nnzs == 2);
  csr_t<int, int, float> rnd;
  generate::random::csr(64, 64, 0.1f, rnd);
  CHECK(rnd.rows == 64 && rnd.nnzs > 0 && rnd.nnzs <= 409);
  // SpMM vs host
  hcsr_t<float> h = from_dense<float>(battery()[1]);
  csr_t<int, int, float> csr(h);
  matrix_t<float> B(h.cols, 10), C(h.rows, 10);
  generate::random::uniform_distribution(B.m_data.begin(), B.m_data.end(), 1, 10, 3u);
  algorithms::spmm::thread_mapped(csr, B, C);
  vector_t<float, H> Bh(B.m_data), Ch(C.m_data);
  bool ok = true;
  for (std::size_t r = 0; r < h.rows; ++r)
    for (int j = 0; j < 10; ++j) {
      float sum = 0;
      for (int k = h.offsets[r]; k < h.offsets[r + 1]; ++k) sum += h.values[k] * Bh[h.indices[k] * 10 + j];
      ok = ok && std::fabs(sum - Ch[r * 10 + j]) <= 1e-3f + 1e-4f * std::fabs(sum);
    }
  CHECK(ok);
  // row-band layout: same y as the plain merge_path_flat, automatic and smallest bands, bands cut into chunks, 8 and 16 wavefronts
  for (auto& dense : battery())
    for (int cfg : {0, 1, 2}) {
      hcsr_t<float> hf = from_dense<float>(dense);
      csr_t<int, int, float> a(hf);
      vector_t<float> xb(hf.cols), y0(hf.rows), y1(hf.rows, -1.f);
      generate::random::uniform_distribution(xb.begin(), xb.end(), 1, 10, 9u);
      algorithms::spmv::merge_path_flat(a, xb, y0);
      algorithms::spmv::rowband_t<int, int, float> banded(a, cfg == 0 ? 0 : 64, cfg == 2 ? 5 : 0);
      if (cfg == 2) banded.arrays.waves = 16;
      banded.spmv(xb, y1);
      vector_t<float, H> h0(y0), h1(y1);
      bool same = true;
      for (std::size_t i = 0; i < h0.size(); ++i) same = same && std::fabs(h0[i] - h1[i]) <= 1e-3f + 1e-5f * std::fabs(h0[i]);
      CHECK(same);
      // the peer fan-out: y and the stand-in peer equal the plain result bit for bit
      vector_t<float> y2(hf.rows, -1.f), peer(hf.rows, -2.f);
      kernels::peer_fanout<float> peers{};
      peers.count = 1;
      peers.base[0] = peer.data().get();
      banded.spmv_fanout_async(xb, y2, peers);
      (void)xpu::stream_synchronize(0);
      vector_t<float, H> h2(y2), hp(peer);
      bool equal = true;
      for (std::size_t i = 0; i < h1.size(); ++i) equal = equal && h2[i] == h1[i] && hp[i] == h1[i];
      CHECK(equal);
    }
  // panel-binned layout: same y as the plain merge_path_flat (f32 bit-exact on these integer x / positive values is not
  // guaranteed -- real values -- so within the usual bound), automatic and smallest sub-bands, with the peer fan-out
  for (auto& dense : battery())
    for (int hw : {0, 64}) {
      hcsr_t<float> hf = from_dense<float>(dense);
      csr_t<int, int, float> a(hf);
      vector_t<float> xb(hf.cols), y0(hf.rows), y1(hf.rows, -1.f), y2(hf.rows, -1.f), peer(hf.rows, -2.f);
      generate::random::uniform_distribution(xb.begin(), xb.end(), 1, 10, 9u);
      algorithms::spmv::merge_path_flat(a, xb, y0);
      algorithms::spmv::panel_binned_t<int, int, float> pb(a, hw);
      pb.spmv(xb, y1);
      kernels::peer_fanout<float> peers{};
      peers.count = 1;
      peers.base[0] = peer.data().get();
      pb.spmv_fanout_async(xb, y2, peers);
      (void)xpu::stream_synchronize(0);
      vector_t<float, H> h0(y0), h1(y1), h2(y2), hp(peer);
      bool same = true, equal = true;
      for (std::size_t i = 0; i < h0.size(); ++i) {
        same = same && std::fabs(h0[i] - h1[i]) <= 1e-3f + 1e-5f * std::fabs(h0[i]);
        equal = equal && h2[i] == h1[i] && hp[i] == h1[i];
      }
      CHECK(same);
      CHECK(equal);
    }
  // spmv_plan_t: tile shape + layout chosen per matrix (structural and measured, copy allowed or not): same y as the plain
  // merge_path_flat on every battery matrix, in both precisions
  for (auto& dense : battery())
    for (int mode = 0; mode < 4; ++mode) {
      const bool allow_copy = mode & 1, measure = mode & 2;
      hcsr_t<float> hf = from_dense<float>(dense);
      csr_t<int, int, float> a(hf);
      vector_t<float> xb(hf.cols), y0(hf.rows), y1(hf.rows, -1.f);
      generate::random::uniform_distribution(xb.begin(), xb.end(), 1, 10, 9u);
      algorithms::spmv::merge_path_flat(a, xb, y0);
      algorithms::spmv::spmv_plan_t<int, int, float> plan(a, allow_copy, measure, 2);
      CHECK(allow_copy || plan.layout == algorithms::spmv::spmv_plan_t<int, int, float>::csr_layout);
      plan.spmv(a, xb, y1);
      vector_t<float, H> h0(y0), h1(y1);
      bool same = true;
      for (std::size_t i = 0; i < h0.size(); ++i) same = same && std::fabs(h0[i] - h1[i]) <= 1e-3f + 1e-5f * std::fabs(h0[i]);
      CHECK(same);
      hcsr_t<double> hd = from_dense<double>(dense);
      csr_t<int, int, double> ad(hd);
      vector_t<double> xd(hd.cols), z0(hd.rows), z1(hd.rows, -1.0);
      generate::random::uniform_distribution(xd.begin(), xd.end(), 1, 10, 9u);
      algorithms::spmv::merge_path_flat(ad, xd, z0);
      algorithms::spmv::spmv_plan_t<int, int, double> pland(ad, allow_copy, measure, 2);
      pland.spmv(ad, xd, z1);
      vector_t<double, H> g0(z0), g1(z1);
      same = true;
      for (std::size_t i = 0; i < g0.size(); ++i) same = same && std::fabs(g0[i] - g1[i]) <= 1e-9 + 1e-12 * std::fabs(g0[i]);
      CHECK(same);
    }
  // csc_thread_mapped from 2^20 nonzeros on: the binned product (kernels::launch_csc_binned) instead of one atomic per nonzero; exactly
  // summable values, so the result is bit for bit the CSR product of the same matrix.  One hub row (a bin shared by several workgroups).
  {
    const std::size_t rows = 150001, cols = 40000, per = 7, hub = 30000;
    std::size_t nnz = rows * per + (hub - per);
    hcsr_t<float> h(rows, cols, nnz);
    std::size_t k = 0;
    for (std::size_t r = 0; r < rows; ++r) {
      h.offsets[r] = static_cast<int>(k);
      const std::size_t deg = r == 777 ? hub : per;
      for (std::size_t j = 0; j < deg; ++j, ++k) {
        h.indices[k] = static_cast<int>(deg == hub ? j : (r * 37 + j * 5003) % cols);  // (distinct within a row)
        h.values[k] = static_cast<float>((static_cast<int>((r + 3 * j) % 17) - 8)) / 8.0f;
      }
    }
    h.offsets[rows] = static_cast<int>(k);
    CHECK(k == nnz && nnz >= (std::size_t(1) << 20));
    vector_t<float, H> hx(cols);
    for (std::size_t c = 0; c < cols; ++c) hx[c] = static_cast<float>(1 + c % 9);
    csr_t<int, int, float> a(h);
    csc_t<int, int, float> c(a);
    vector_t<float> x(hx), y0(rows, -1.f), y1(rows, 0.f);
    algorithms::spmv::merge_path_flat(a, x, y0);
    algorithms::spmv::csc_thread_mapped(c, x, y1);
    vector_t<float, H> g0(y0), g1(y1);
    bool same = true;
    for (std::size_t i = 0; i < rows; ++i) same = same && g0[i] == g1[i];
    CHECK(same);
  }
  // A column index outside the matrix is the caller's mistake, not a missing resource: the row-band copy's build reports it and
  // spmv_plan_t lets it through in the structural AND the measured path (error::bad_argument_t; the C ABI returns LOOPS_E_BADARG
  // for the same input) instead of quietly staying on the CSR.
  {
    const std::size_t rows = 2048, cols = 700000, per = 12;  // x = 2.8 MB, mean row 12: the structural rule takes the row-band copy
    hcsr_t<float> h(rows, cols, rows * per);
    for (std::size_t r = 0; r <= rows; ++r) h.offsets[r] = static_cast<int>(r * per);
    for (std::size_t r = 0; r < rows; ++r)
      for (std::size_t k = 0; k < per; ++k) {
        h.indices[r * per + k] = static_cast<int>((r * 131 + k * 50021) % cols);
        h.values[r * per + k] = 1.0f;
      }
    for (int measure = 0; measure < 2; ++measure) {
      hcsr_t<float> bad = h;
      bad.indices[777] = static_cast<int>(cols);  // one past the last column
      csr_t<int, int, float> good_dev(h), bad_dev(bad);
      bool threw = false, other = false;
      try {
        algorithms::spmv::spmv_plan_t<int, int, float> plan(bad_dev, /*allow_copy=*/true, measure != 0, 2);
      } catch (const error::bad_argument_t&) {
        threw = true;
      } catch (const std::exception&) {
        other = true;
      }
      CHECK(threw && !other);
      algorithms::spmv::spmv_plan_t<int, int, float> plan(good_dev, /*allow_copy=*/true, measure != 0, 2);  // (the same matrix without the mistake builds)
      CHECK(measure != 0 || plan.layout == algorithms::spmv::spmv_plan_t<int, int, float>::row_band_layout);
    }
  }
  // how the phased-gather kernels are configured (host logic, kernels::phased_config_for): parts by |x|, the shift that maps
  // every column below `cols` to a part below `parts`, at most 16 parts for 8-byte values
  for (long long cols : {1ll, 2ll, 9ll, 1000ll, 1ll << 20, (1ll << 20) + 1, 1ll << 21, 1ll << 23, (1ll << 24) - 3, (1ll << 31) - 1}) {
    for (int bytes : {4, 8}) {
      const kernels::phased_config c = kernels::phased_config_for(cols, bytes);
      const double mb = static_cast<double>(cols) * bytes / (1024.0 * 1024.0);
      CHECK(c.parts == (bytes == 8 ? (mb <= 12.0 ? 8 : 16) : mb <= 6.0 ? 8 : mb <= 24.0 ? 16 : 32));
      CHECK(((cols - 1) >> c.args.shift) < c.parts);                          // the last column lands in an existing part
      CHECK(c.args.shift == 0 || ((cols - 1) >> (c.args.shift - 1)) >= c.parts);  // ... and no smaller shift would do
      CHECK(c.args.inv_ticks > 0);
    }
  }
  // phased x gathers (merge_path_flat_phased_async_with; spmv_plan_t's candidate for long rows over an x of at least 1 MB): the
  // plain kernel's bits on real values, for a power-of-two and an odd column count, f32 and f64
  for (std::size_t cols : {std::size_t(1) << 19, std::size_t(300007)}) {
    const std::size_t rows = 1 << 12;
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> u(0.5, 1.5);
    std::vector<int> off(rows + 1, 0), idx;
    std::vector<double> val;
    for (std::size_t r = 0; r < rows; ++r) {
      const std::size_t deg = r % 97 == 0 ? 9000 : 3 + r % 29;      // rows of 9 000 nonzeros span several 4 096-item tiles
      const std::size_t step = cols / deg;
      for (std::size_t k = 0; k < deg; ++k) { idx.push_back(static_cast<int>(k * step + rng() % step)); val.push_back(u(rng)); }
      off[r + 1] = static_cast<int>(idx.size());
    }
    hcsr_t<float> hf(rows, cols, idx.size());
    hcsr_t<double> hd(rows, cols, idx.size());
    for (std::size_t r = 0; r <= rows; ++r) hf.offsets[r] = hd.offsets[r] = off[r];
    for (std::size_t k = 0; k < idx.size(); ++k) { hf.indices[k] = hd.indices[k] = idx[k]; hf.values[k] = static_cast<float>(val[k]); hd.values[k] = val[k]; }
    csr_t<int, int, float> a(hf);
    csr_t<int, int, double> ad(hd);
    vector_t<float, H> hx(cols);
    vector_t<double, H> hxd(cols);
    for (std::size_t c = 0; c < cols; ++c) { hxd[c] = u(rng); hx[c] = static_cast<float>(hxd[c]); }
    vector_t<float> x(hx), y0(rows, -1.f), y1(rows, -2.f), y2(rows, -3.f);
    vector_t<double> xd(hxd), z0(rows, -1.0), z1(rows, -2.0);
    using plan_t = algorithms::spmv::merge_path_plan_of_t<512, 8, int, int>;
    const plan_t::layout_t lay(a.offsets.data().get(), static_cast<int>(rows), static_cast<int>(idx.size()));
    plan_t plan(lay, 0, plan_t::prepass_always);
    plan.classify(0);
    CHECK(!plan.self_complete() && plan.merge_tiles() > 1);
    algorithms::spmv::merge_path_flat_async_with<512, 8>(plan, a, x, y0);
    algorithms::spmv::merge_path_flat_phased_async_with<512, 8>(plan, a, x, y1);
    algorithms::spmv::merge_path_flat_async_with<512, 8>(plan, ad, xd, z0);
    algorithms::spmv::merge_path_flat_phased_async_with<512, 8>(plan, ad, xd, z1);
    algorithms::spmv::spmv_plan_t<int, int, float> sp(a, /*allow_copy=*/false, /*measure=*/true, 3);
    CHECK(sp.layout == algorithms::spmv::spmv_plan_t<int, int, float>::csr_layout && sp.ms_phased > 0.f);
    sp.spmv(a, x, y2);
    (void)xpu::stream_synchronize(0);
    vector_t<float, H> h0(y0), h1(y1), h2(y2);
    vector_t<double, H> g0(z0), g1(z1);
    const auto href = reference::spmv(hf, hx);
    bool equal = true, equal_d = true, close = true, plan_ok = true;
    for (std::size_t i = 0; i < rows; ++i) {
      equal = equal && h0[i] == h1[i];
      equal_d = equal_d && g0[i] == g1[i];
      plan_ok = plan_ok && std::fabs(h2[i] - h0[i]) <= 1e-5f * std::fabs(h0[i]);
      close = close && std::fabs(h1[i] - href[i]) <= 1e-4f * std::fabs(href[i]);
    }
    CHECK(equal);
    CHECK(equal_d);
    CHECK(plan_ok);
    CHECK(close);
  }
  // the plan-less merge_path_flat(csr, x, y) guesses from the structure whether the phased gathers pay (kernels::
  // columns_look_scattered): yes on uniformly scattered columns over an x of 4 MB, no on a band of the same size -- and gives the
  // plain held plan's result either way, fp32 and fp64
  for (int local = 0; local < 2; ++local) {
    const std::size_t rows = 1 << 17, cols = 1 << 20, deg = 10;   // 1.3 M nonzeros, x = 4 MB (fp32) / 8 MB (fp64)
    std::mt19937_64 rng(7 + local);
    std::uniform_real_distribution<double> u(0.5, 1.5);
    hcsr_t<float> hf(rows, cols, rows * deg);
    hcsr_t<double> hd(rows, cols, rows * deg);
    for (std::size_t r = 0; r <= rows; ++r) hf.offsets[r] = hd.offsets[r] = static_cast<int>(r * deg);
    for (std::size_t r = 0; r < rows; ++r) {
      const std::size_t step = local ? 8 : cols / deg;            // local: 10 columns inside a window of 80 at the diagonal
      const std::size_t base = local ? (r * (cols / rows)) % (cols - deg * step) : 0;
      for (std::size_t k = 0; k < deg; ++k) {
        const int c = static_cast<int>(base + k * step + rng() % step);
        const double v = u(rng);
        hf.indices[r * deg + k] = hd.indices[r * deg + k] = c;
        hf.values[r * deg + k] = static_cast<float>(v);
        hd.values[r * deg + k] = v;
      }
    }
    csr_t<int, int, float> a(hf);
    csr_t<int, int, double> ad(hd);
    vector_t<unsigned int> scratch(kernels::scatter_scratch_words);
    const bool scattered = kernels::columns_look_scattered(0, a.indices.data().get(), static_cast<long long>(a.nnzs),
                                                           static_cast<long long>(a.cols), 4, scratch.data().get());
    CHECK(scattered == (local == 0));
    vector_t<float, H> hx(cols);
    vector_t<double, H> hxd(cols);
    for (std::size_t c = 0; c < cols; ++c) { hxd[c] = u(rng); hx[c] = static_cast<float>(hxd[c]); }
    vector_t<float> x(hx), y0(rows, -1.f), y1(rows, -2.f);
    vector_t<double> xd(hxd), z0(rows, -1.0), z1(rows, -2.0);
    using small_t = algorithms::spmv::merge_path_small_plan_t<int, int, float>;
    const small_t::layout_t lay(a.offsets.data().get(), static_cast<int>(rows), static_cast<int>(rows * deg));
    small_t plan(lay, 0, small_t::prepass_always);
    plan.classify(0);
    algorithms::spmv::merge_path_flat_async_with<algorithms::spmv::launch_t<float>::block_size, algorithms::spmv::launch_t<float>::items_per_thread>(plan, a, x, y0);
    algorithms::spmv::merge_path_flat(a, x, y1);
    using small_d = algorithms::spmv::merge_path_small_plan_t<int, int, double>;
    small_d pland(small_d::layout_t(ad.offsets.data().get(), static_cast<int>(rows), static_cast<int>(rows * deg)), 0, small_d::prepass_always);
    pland.classify(0);
    algorithms::spmv::merge_path_flat_async_with<algorithms::spmv::launch_t<double>::block_size, algorithms::spmv::launch_t<double>::items_per_thread>(pland, ad, xd, z0);
    algorithms::spmv::merge_path_flat(ad, xd, z1);
    (void)xpu::stream_synchronize(0);
    vector_t<float, H> h0(y0), h1(y1);
    vector_t<double, H> g0(z0), g1(z1);
    bool close = true, close_d = true;  // (another tile shape is another summation order: equal up to rounding, not bit for bit)
    for (std::size_t i = 0; i < rows; ++i) {
      close = close && std::fabs(h0[i] - h1[i]) <= 2e-6f * std::fabs(h0[i]);
      close_d = close_d && std::fabs(g0[i] - g1[i]) <= 1e-14 * std::fabs(g0[i]);
    }
    CHECK(close);
    CHECK(close_d);
  }
  // the drop-in spmm::thread_mapped (merge-path SpMM since round 4) == the reference-shaped per-thread loop kept behind
  // thread_mapped_schedule_api == spmm::merge_path_flat (every battery matrix, several widths of B, f32 + f64)
  for (auto& dense : battery())
    for (int n : {1, 4, 10, 32, 70}) {
      hcsr_t<float> hf = from_dense<float>(dense);
      csr_t<int, int, float> a(hf);
      matrix_t<float> Bn(hf.cols, n), C0(hf.rows, n), C1(hf.rows, n);
      generate::random::uniform_distribution(Bn.m_data.begin(), Bn.m_data.end(), 1, 10, 5u);
      algorithms::spmm::thread_mapped_schedule_api(a, Bn, C0);
      {
        matrix_t<float> C2(hf.rows, n);
        algorithms::spmm::thread_mapped(a, Bn, C2);
        vector_t<float, H> c0(C0.m_data), c2(C2.m_data);
        bool same = true;
        for (std::size_t i = 0; i < c0.size(); ++i) same = same && std::fabs(c0[i] - c2[i]) <= 1e-3f + 1e-5f * std::fabs(c0[i]);
        CHECK(same);
      }
      auto timer = algorithms::spmm::merge_path_flat(a, Bn, C1);
      CHECK(timer.milliseconds() >= 0.f);
      vector_t<float, H> c0(C0.m_data), c1(C1.m_data);
      bool same = true;
      for (std::size_t i = 0; i < c0.size(); ++i) same = same && std::fabs(c0[i] - c1[i]) <= 1e-3f + 1e-5f * std::fabs(c0[i]);
      CHECK(same);
      hcsr_t<double> hd = from_dense<double>(dense);
      csr_t<int, int, double> ad(hd);
      matrix_t<double> Bd(hd.cols, n), D0(hd.rows, n), D1(hd.rows, n);
      generate::random::uniform_distribution(Bd.m_data.begin(), Bd.m_data.end(), 1, 10, 5u);
      algorithms::spmm::thread_mapped_schedule_api(ad, Bd, D0);
      algorithms::spmm::thread_mapped(ad, Bd, D1);
      vector_t<double, H> d0(D0.m_data), d1(D1.m_data);
      same = true;
      for (std::size_t i = 0; i < d0.size(); ++i) same = same && std::fabs(d0[i] - d1[i]) <= 1e-9 + 1e-12 * std::fabs(d0[i]);
      CHECK(same);
    }
}

int main() {
  run_battery<float>();
  run_battery<double>();
  misc();
  std::printf("%d checks, %d failures\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
