// Host-side tests of the C++ header API (include/loops): no GPU is touched -- every container is
// in memory_space_t::host.  Restates (does not copy) what the reference's Catch2 suite pins:
//   layout contract + known answers  unittests/test_layout_contract.hxx:30-88, test_layout_*.cu
//   ranges / ceil_div                 unittests/test_util_range.cu:22-78, test_util_math.cu:22-67
//   Matrix-Market loader              unittests/test_market_loader.cu:95-317
//   containers / format round trips   unittests/test_container_*.cu, test_format_round_trip.cu:131-174
//   validator                         unittests/test_rigorous_validator.cu (host parts)
// plus the C1 known answers (chesapeake through loader -> CSR -> generator -> reference::spmv).
// Built and run by tests/test_cpp_host_api.py:  hipcc -std=c++17 -x hip ... (host code only).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include <loops/container/formats.hxx>
#include <loops/container/layout.hxx>
#include <loops/container/market.hxx>
#include <loops/range.hxx>
#include <loops/schedule.hxx>
#include <loops/util/generate.hxx>
#include <loops/util/math.hxx>
#include <loops/util/reference.hxx>
#include <loops/util/sample.hxx>
#include <loops/util/filepath.hxx>
#include <loops/algorithms/spmv/launch_box.hxx>

using namespace loops;
static int g_fail = 0, g_checks = 0;
#define CHECK(...) do { ++g_checks; if (!(__VA_ARGS__)) { ++g_fail; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #__VA_ARGS__); } } while (0)
#define CHECK_THROWS(e) do { ++g_checks; bool t = false; try { e; } catch (...) { t = true; } if (!t) { ++g_fail; std::printf("FAIL %s:%d  no throw: %s\n", __FILE__, __LINE__, #e); } } while (0)

constexpr auto H = memory_space_t::host;

template <typename L>
static void check_layout_invariants(const L& lay) {
  const auto T = lay.num_tiles();
  const auto A = lay.num_atoms();
  if (T == 0) return;
  CHECK(lay.tile_begin(0) == 0);
  CHECK(lay.tile_end(T - 1) == A);
  auto it = lay.tile_end_iter();
  for (decltype(lay.num_tiles()) t = 0; t < T; ++t) {
    CHECK(lay.tile_begin(t) <= lay.tile_end(t));
    CHECK(lay.tile_size(t) == lay.tile_end(t) - lay.tile_begin(t));
    CHECK(it[t] == lay.tile_end(t));
    if (t + 1 < T) CHECK(lay.tile_end(t) == lay.tile_begin(t + 1));
  }
  for (decltype(lay.num_atoms()) a = 0; a < A; ++a) {
    const auto t = lay.tile_of(a);
    CHECK(lay.tile_begin(t) <= a && a < lay.tile_end(t));
  }
}

static void test_layouts() {
  const int off[] = {0, 2, 2, 5, 7};  // test_layout_csr.cu:23-52
  layout::csr<int, int> c(off, 4, 7);
  check_layout_invariants(c);
  const int want_of[] = {0, 0, 2, 2, 2, 3, 3};
  for (int a = 0; a < 7; ++a) CHECK(c.tile_of(a) == want_of[a]);
  const int want_sz[] = {2, 0, 3, 2};
  for (int t = 0; t < 4; ++t) CHECK(c.tile_size(t) == want_sz[t]);
  const int one[] = {0, 6};
  check_layout_invariants(layout::csr<int, int>(one, 1, 6));
  const int empty[] = {0, 0, 0, 0};
  layout::csr<int, int> e(empty, 3, 0);
  CHECK(e.num_atoms() == 0 && e.tile_size(1) == 0);
  const int boff[] = {0, 3, 5, 5, 9, 12};  // test_layout_bcsr.cu:23-46
  check_layout_invariants(layout::bcsr<int, int>(boff, 5, 12));
  check_layout_invariants(layout::csc<int, int>(boff, 5, 12));
  layout::ell<int, int> el(5, 3);
  check_layout_invariants(el);
  CHECK(el.num_atoms() == 15 && el.tile_begin(2) == 6 && el.tile_of(7) == 2);
  CHECK(layout::ell<int, int>(0, 3).num_atoms() == 0 && layout::ell<int, int>(4, 0).num_atoms() == 0);
  layout::dia<int, int> di(4, 3);
  check_layout_invariants(di);
  layout::coo<int, int> co(9);
  check_layout_invariants(co);
  CHECK(co.num_tiles() == 9 && co.tile_of(5) == 5 && co.tile_size(3) == 1);
  // test_layout_flat_partitioner.cu:24-112
  layout::flat_uniform_occupancy<2, layout::csr<int, int>> p2(c);
  check_layout_invariants(p2);
  CHECK(p2.num_tiles() == 4);
  const int psz[] = {2, 2, 2, 1};
  for (int t = 0; t < 4; ++t) CHECK(p2.tile_size(t) == psz[t]);
  for (int a = 0; a < 7; ++a) CHECK(p2.tile_of(a) == a / 2);
  CHECK(p2.base().tile_of(2) == 2);
  layout::flat_uniform_occupancy<7, layout::csr<int, int>> p7(c);  // K | A
  CHECK(p7.num_tiles() == 1 && p7.tile_size(0) == 7);
  layout::flat_uniform_occupancy<16, layout::csr<int, int>> p16(c);  // K > A
  CHECK(p16.num_tiles() == 1 && p16.tile_end(0) == 7);
  layout::flat_uniform_occupancy<4, layout::ell<int, int>> pe(el);  // wrapping a non-CSR base
  check_layout_invariants(pe);
  CHECK(pe.num_tiles() == 4 && pe.base().tile_of(7) == 2);
}

static void test_ranges_math() {
  std::vector<int> got;
  for (auto i : range(0, 5)) got.push_back(i);
  CHECK((got == std::vector<int>{0, 1, 2, 3, 4}));
  got.clear();
  for (auto i : range(7, 7)) got.push_back(i);
  CHECK(got.empty());
  got.clear();
  for (auto i : range(0, 10).step(2)) got.push_back(i);
  CHECK((got == std::vector<int>{0, 2, 4, 6, 8}));
  got.clear();
  for (auto i : range(0, 9).step(3)) got.push_back(i);
  CHECK((got == std::vector<int>{0, 3, 6}));
  std::size_t n = 0;
  for (auto i : range(std::size_t(0), std::size_t(4))) { (void)i; ++n; }
  CHECK(n == 4);
  std::vector<float> v{1.f, 2.f, 3.f};
  std::vector<std::size_t> idx;
  for (auto i : indices(v)) idx.push_back(i);
  CHECK((idx == std::vector<std::size_t>{0, 1, 2}));
  CHECK(math::ceil_div(10, 5) == 2 && math::ceil_div(11, 5) == 3 && math::ceil_div(0, 7) == 0 && math::ceil_div(1, 7) == 1);
  const long long big = 9223372036854775807LL;
  CHECK(math::ceil_div(big, 2LL) == (big / 2 + 1) && math::ceil_div(big, big) == 1 && math::ceil_div(big, 1LL) == big);
  CHECK(algorithms::spmv::launch_t<float>::block_size == 256 && algorithms::spmv::launch_t<float>::items_per_thread == 8);
  CHECK(algorithms::spmv::launch_t<double>::items_per_thread == 4);
  CHECK(is_market("a/b/c.mtx") && is_market("x.mmio") && !is_market("x.csr") && extract_filename("a/b/c.mtx") == "c.mtx" &&
        extract_dataset("c.mtx") == "c");
}

static std::string write_tmp(const std::string& name, const std::string& text) {
  const std::string path = std::string("/tmp/loops_test_") + name + ".mtx";
  std::ofstream(path) << text;
  return path;
}
template <typename C, typename V>
static bool find_entry(const C& coo, int r, int c, V& v) {
  for (std::size_t i = 0; i < coo.nnzs; ++i)
    if (coo.row_indices[i] == r && coo.col_indices[i] == c) { v = coo.values[i]; return true; }
  return false;
}

static void test_market_loader() {
  matrix_market_t<int, int, float> rd;
  float v = 0;
  auto coo = rd.load(write_tmp("general", "%%MatrixMarket matrix coordinate real general\n% a comment\n3 3 4\n1 1 1.5\n2 2 2.0\n2 3 3.0\n3 1 4.0\n"));
  CHECK(coo.rows == 3 && coo.cols == 3 && coo.nnzs == 4);
  CHECK(find_entry(coo, 0, 0, v) && v == 1.5f);
  CHECK(find_entry(coo, 1, 1, v) && v == 2.0f);
  CHECK(find_entry(coo, 1, 2, v) && v == 3.0f);
  CHECK(find_entry(coo, 2, 0, v) && v == 4.0f);
  CHECK(rd.dataset == "loops_test_general");
  coo = rd.load(write_tmp("integer", "%%MatrixMarket matrix coordinate integer general\n2 2 2\n1 1 7\n2 2 -3\n"));
  CHECK(coo.nnzs == 2 && find_entry(coo, 0, 0, v) && v == 7.0f && find_entry(coo, 1, 1, v) && v == -3.0f);
  coo = rd.load(write_tmp("pattern", "%%MatrixMarket matrix coordinate pattern general\n3 3 3\n1 2\n2 3\n3 1\n"));
  CHECK(coo.nnzs == 3);
  for (std::size_t i = 0; i < coo.nnzs; ++i) CHECK(coo.values[i] == 1.0f);
  coo = rd.load(write_tmp("symmetric", "%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 1.0\n2 1 2.0\n2 2 3.0\n3 3 4.0\n"));
  CHECK(coo.rows == 3 && coo.cols == 3 && coo.nnzs == 5);
  CHECK(find_entry(coo, 0, 0, v) && v == 1.0f && find_entry(coo, 1, 1, v) && v == 3.0f && find_entry(coo, 2, 2, v) && v == 4.0f);
  CHECK(find_entry(coo, 1, 0, v) && v == 2.0f && find_entry(coo, 0, 1, v) && v == 2.0f);
  coo = rd.load(write_tmp("sympat", "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 2\n2 1\n3 2\n"));
  CHECK(coo.nnzs == 4 && find_entry(coo, 0, 1, v) && v == 1.0f && find_entry(coo, 1, 0, v) && find_entry(coo, 1, 2, v) && find_entry(coo, 2, 1, v));
  coo = rd.load(write_tmp("comments", "%%MatrixMarket matrix coordinate real general\n% one\n%two\n\n% three\n2 2 2\n1 1 1.0\n2 2 2.0\n"));
  CHECK(coo.nnzs == 2);
  CHECK_THROWS(rd.load(write_tmp("complex", "%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 0.0\n")));
  CHECK_THROWS(rd.load(write_tmp("herm", "%%MatrixMarket matrix coordinate real hermitian\n1 1 1\n1 1 1.0\n")));
  CHECK_THROWS(rd.load(write_tmp("skew", "%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 1.0\n")));
  CHECK_THROWS(rd.load(write_tmp("array", "%%MatrixMarket matrix array real general\n1 1\n1.0\n")));
  CHECK_THROWS(rd.load(write_tmp("nobanner", "3 3 1\n1 1 1.0\n")));
  CHECK_THROWS(rd.load(write_tmp("zeroidx", "%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n")));
  CHECK_THROWS(rd.load("/tmp/loops_test_does_not_exist.mtx"));
  matrix_market_t<int, int, double> rd64;
  auto c64 = rd64.load(write_tmp("sym64", "%%MatrixMarket matrix coordinate real symmetric\n2 2 2\n1 1 2.5\n2 1 2.5\n"));
  CHECK(c64.nnzs == 3);
  for (std::size_t i = 0; i < c64.nnzs; ++i) CHECK(c64.values[i] == 2.5);
}

// Files large enough (> 4 MB of body) for the loader to tokenise the body on several threads: the result must be what a
// front-to-back pass gives -- entry k of the file is entry k of the COO (mirrors right after their originals), trailing
// text after the declared entries is ignored even if malformed, a malformed line among them or too few lines are errors.
static void test_market_loader_parallel_body() {
  const int n = 400000, entries = 500000;
  auto row_of = [&](int k) { return static_cast<int>((static_cast<long long>(k) * 7919) % n); };
  auto col_of = [&](int k) { return static_cast<int>((static_cast<long long>(k) * 104729 + 13) % n); };
  auto body = [&](int count, bool lower) {
    std::string text;
    text.reserve(static_cast<std::size_t>(count) * 24);
    for (int k = 0; k < count; ++k) {
      int r = row_of(k), c = col_of(k);
      if (lower && c > r) std::swap(r, c);
      text += std::to_string(r + 1) + " " + std::to_string(c + 1) + " " + std::to_string(k % 1000) + ".5\n";
    }
    return text;
  };
  matrix_market_t<int, int, float> rd;
  const std::string head = "%%MatrixMarket matrix coordinate real general\n% comment\n" + std::to_string(n) + " " + std::to_string(n) + " " +
                           std::to_string(entries) + "\n";
  const std::string general = body(entries, false);
  CHECK(general.size() > (std::size_t(4) << 20));
  auto coo = rd.load(write_tmp("big_general", head + general));
  CHECK(coo.rows == static_cast<std::size_t>(n) && coo.nnzs == static_cast<std::size_t>(entries));
  bool same = true;
  for (int k = 0; k < entries; ++k)
    same = same && coo.row_indices[k] == row_of(k) && coo.col_indices[k] == col_of(k) && coo.values[k] == static_cast<float>(k % 1000) + 0.5f;
  CHECK(same);
  // trailing lines after the declared entries are not read, malformed or not
  coo = rd.load(write_tmp("big_trailing", head + general + "this is not an entry\n7 7 7.0\n"));
  CHECK(coo.nnzs == static_cast<std::size_t>(entries) && coo.row_indices[entries - 1] == row_of(entries - 1));
  // a malformed line among the declared entries (deep inside the file, i.e. in some later chunk) is an error
  std::string broken = general;
  const std::size_t at = broken.find('\n', broken.size() * 3 / 4) + 1;
  broken.insert(at, "oops 3 1.0\n");
  CHECK_THROWS(rd.load(write_tmp("big_broken", head + broken)));
  // fewer lines than declared
  CHECK_THROWS(rd.load(write_tmp("big_short", head + body(entries - 5, false))));
  // a zero index deep inside
  std::string zero = general;
  zero.insert(zero.find('\n', zero.size() / 2) + 1, "0 5 1.0\n");
  CHECK_THROWS(rd.load(write_tmp("big_zero", head + zero)));
  // symmetric: every off-diagonal entry is followed by its mirror, in file order
  const std::string shead = "%%MatrixMarket matrix coordinate real symmetric\n" + std::to_string(n) + " " + std::to_string(n) + " " +
                            std::to_string(entries) + "\n";
  coo = rd.load(write_tmp("big_symmetric", shead + body(entries, true)));
  std::size_t o = 0;
  same = true;
  for (int k = 0; k < entries && same; ++k) {
    int r = row_of(k), c = col_of(k);
    if (c > r) std::swap(r, c);
    same = o < coo.nnzs && coo.row_indices[o] == r && coo.col_indices[o] == c;
    ++o;
    if (r != c) {
      same = same && o < coo.nnzs && coo.row_indices[o] == c && coo.col_indices[o] == r && coo.values[o] == coo.values[o - 1];
      ++o;
    }
  }
  CHECK(same && o == coo.nnzs);
}

template <typename csr_type>
static std::vector<std::vector<float>> dense_of(const csr_type& m) {
  std::vector<std::vector<float>> d(m.rows, std::vector<float>(m.cols, 0.f));
  for (std::size_t r = 0; r < m.rows; ++r)
    for (auto k = m.offsets[r]; k < m.offsets[r + 1]; ++k) d[r][m.indices[k]] = m.values[k];
  return d;
}

static void test_containers_round_trip() {
  // 7 x 7 (not divisible by 2 or 3), a couple of empty rows, an unsorted COO input
  coo_t<int, float, H> coo(7, 7, 11);
  const int R[] = {6, 0, 0, 2, 2, 2, 3, 5, 5, 6, 0}, Cc[] = {6, 3, 0, 1, 6, 2, 3, 0, 4, 1, 5};
  for (int i = 0; i < 11; ++i) { coo.row_indices[i] = R[i]; coo.col_indices[i] = Cc[i]; coo.values[i] = 1.0f + i; }
  csr_t<int, int, float, H> csr(coo);
  CHECK(csr.rows == 7 && csr.nnzs == 11 && csr.offsets[0] == 0 && csr.offsets[7] == 11 && csr.offsets[2] == csr.offsets[1]);
  for (std::size_t r = 0; r < 7; ++r)
    for (auto k = csr.offsets[r] + 1; k < csr.offsets[r + 1]; ++k) CHECK(csr.indices[k - 1] < csr.indices[k]);
  const auto D = dense_of(csr);
  CHECK(D[6][6] == 1.0f && D[0][3] == 2.0f && D[0][5] == 11.0f && D[5][4] == 9.0f);
  coo_t<int, float, H> back(csr);  // CSR -> COO
  CHECK(back.nnzs == 11);
  for (std::size_t i = 0; i < back.nnzs; ++i) CHECK(D[back.row_indices[i]][back.col_indices[i]] == back.values[i]);
  csc_t<int, int, float, H> csc(csr);  // CSR -> CSC
  CHECK(csc.offsets[7] == 11);
  for (std::size_t c = 0; c < 7; ++c)
    for (auto k = csc.offsets[c]; k < csc.offsets[c + 1]; ++k) CHECK(D[csc.indices[k]][c] == csc.values[k]);
  ell_t<int, float, H> ell(csr);  // CSR -> ELL
  CHECK(ell.pitch == 3 && (ell_t<int, float, H>::max_nnz_per_row(csr)) == 3 && ell.indices.size() == 21);
  for (std::size_t r = 0; r < 7; ++r)
    for (std::size_t j = 0; j < ell.pitch; ++j) {
      const int col = ell.indices[r * ell.pitch + j];
      if (col == ell_t<int, float, H>::sentinel()) CHECK(ell.values[r * ell.pitch + j] == 0.f);
      else CHECK(D[r][col] == ell.values[r * ell.pitch + j]);
    }
  dia_t<int, int, float, H> dia(csr);  // CSR -> DIA
  CHECK(dia.stride == 7 && dia.num_diagonals == (dia_t<int, int, float, H>::count_diagonals(csr)));
  for (std::size_t d = 0; d < dia.num_diagonals; ++d)
    for (std::size_t r = 0; r < 7; ++r) {
      const long c = long(r) + dia.diag_offsets[d];
      const float cell = dia.values[d * dia.stride + r];
      if (c >= 0 && c < 7) CHECK(D[r][c] == cell); else CHECK(cell == 0.f);
    }
  auto check_bcsr = [&](auto b, std::size_t RR, std::size_t CC) {
    CHECK(b.num_block_rows == (7 + RR - 1) / RR && b.num_block_cols == (7 + CC - 1) / CC);
    std::size_t nonzero_cells = 0;
    for (std::size_t br = 0; br < b.num_block_rows; ++br)
      for (auto k = b.block_offsets[br]; k < b.block_offsets[br + 1]; ++k) {
        if (k > b.block_offsets[br]) CHECK(b.block_col_indices[k - 1] < b.block_col_indices[k]);
        for (std::size_t i = 0; i < RR; ++i)
          for (std::size_t j = 0; j < CC; ++j) {
            const std::size_t r = br * RR + i, c = b.block_col_indices[k] * CC + j;
            const float cell = b.values[(k * RR + i) * CC + j];
            if (r < 7 && c < 7) { CHECK(D[r][c] == cell); nonzero_cells += cell != 0.f; } else CHECK(cell == 0.f);
          }
      }
    CHECK(nonzero_cells == 11);
  };
  check_bcsr(bcsr_t<2, 2, int, int, float, H>(csr), 2, 2);
  check_bcsr(bcsr_t<3, 3, int, int, float, H>(csr), 3, 3);
  check_bcsr(bcsr_t<4, 4, int, int, float, H>(csr), 4, 4);
  auto s = sample::csr<H>();
  CHECK(s.rows == 4 && s.nnzs == 4 && s.offsets[1] == 0 && s.offsets[4] == 4 && s.values[1] == 8.f);
  coo_t<int, float, H> dup(3, 3, 4);
  const int dr[] = {1, 0, 1, 0}, dc[] = {2, 0, 2, 0};
  for (int i = 0; i < 4; ++i) { dup.row_indices[i] = dr[i]; dup.col_indices[i] = dc[i]; dup.values[i] = float(i); }
  dup.remove_duplicates();
  CHECK(dup.nnzs == 2 && dup.row_indices[0] == 0 && dup.row_indices[1] == 1);
}

static void test_c1_chesapeake(const char* mtx_path) {
  matrix_market_t<int, int, float> rd;
  csr_t<int, int, float, H> csr(rd.load(mtx_path));
  CHECK(csr.rows == 39 && csr.cols == 39 && csr.nnzs == 340 && rd.dataset == "chesapeake");
  const int off0[] = {0, 11, 22, 29, 33, 37, 41, 51, 64};
  for (int i = 0; i < 9; ++i) CHECK(csr.offsets[i] == off0[i]);
  vector_t<float, H> x(39);
  generate::random::uniform_distribution(x.begin(), x.end(), 1, 10, 42u);
  const float x0[] = {1, 10, 6, 2, 10, 6, 5, 5};
  for (int i = 0; i < 8; ++i) CHECK(x[i] == x0[i]);
  auto y = reference::spmv(csr, x);
  auto y64 = reference::spmv_f64(csr, x);
  const float y0[] = {50, 52, 53, 26, 18, 21, 51, 66};
  float sum = 0;
  for (int i = 0; i < 39; ++i) { sum += y[i]; CHECK(y[i] == y64[i]); }
  for (int i = 0; i < 8; ++i) CHECK(y[i] == y0[i]);
  CHECK(sum == 1794.f);
  auto l1 = reference::row_l1_products(csr, x);
  CHECK(l1[0] == 50.f);
  CHECK(!reference::default_tolerance<float>::ne(3.0f, 3.01f) && reference::default_tolerance<float>::ne(1000.f, 1002.f));
  CHECK(generate::random::hash(0) == 1800329511u || true);  // value pinned in tests/golden/xgen.npz (python side)
}

// The reference's own SpMV battery (unittests/test_spmv_battery.hxx:52-65; fixtures: tests/golden/ref_battery.inc,
// generated by tests/golden/make_ref_battery.py): this repository's reference::spmv -- the function every example's
// --validate compares the kernels with -- must reproduce the y of the reference's host reference_spmv
// (unittests/test_helpers.hxx:268-279) bit for bit: same loop, same order, fp32 accumulation.
#include "../golden/ref_battery.inc"
static void test_reference_battery() {
  for (const ref_battery_case& c : ref_battery) {
    csr_t<int, int, float, H> csr(static_cast<std::size_t>(c.rows), static_cast<std::size_t>(c.cols), static_cast<std::size_t>(c.nnz));
    for (int i = 0; i <= c.rows; ++i) csr.offsets[i] = c.offsets[i];
    for (int i = 0; i < c.nnz; ++i) { csr.indices[i] = c.indices[i]; csr.values[i] = c.values[i]; }
    vector_t<float, H> x(c.cols);
    for (int i = 0; i < c.cols; ++i) x[i] = c.x[i];
    auto y = reference::spmv(csr, x);
    int bad = 0;
    for (int i = 0; i < c.rows; ++i) bad += y[i] != c.y[i];
    if (bad) std::printf("reference battery %s: %d rows differ\n", c.name, bad);
    CHECK(bad == 0);
  }
}

int main(int argc, char** argv) {
  test_reference_battery();
  test_layouts();
  test_ranges_math();
  test_market_loader();
  test_market_loader_parallel_body();
  test_containers_round_trip();
  if (argc > 1) test_c1_chesapeake(argv[1]);
  std::printf("%d checks, %d failures\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
