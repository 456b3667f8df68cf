// ref_battery.cpp -- CHECKER / fixture generator (test infrastructure, never linked into the product).
//
// Restatement of the matrix factories of the reference's own SpMV battery, which every one of its kernel tests runs
// (unittests/test_spmv_battery.hxx:52-65 `standard_battery`, factories unittests/test_helpers.hxx:55-240, input vector
// :228-240, host reference :268-279).  The reference's headers cannot be compiled here as they are (they include
// catch2 and cuda_runtime.h, both absent from the image, and stand-in headers are not allowed), so the factories are
// restated below from their text, with the same std::mt19937 seeds (7 / 11 / 17 / 23) and the same libstdc++
// distributions.  Pinned by the known answers SURVEY.md App. D.3 recorded from the reference's code itself
// (rows / cols / nnz / y[0] / sum(y) per matrix): tests/test_oracle_pin.py::test_reference_battery_known_answers.
//
//   g++ -O2 -std=c++17 ref_battery.cpp -o _ref/ref_battery && _ref/ref_battery > dump.txt
//
// Output (text): for every matrix  "name|rows|cols|nnz" then offsets, indices, values, x, y -- one line each.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <string>
#include <tuple>
#include <vector>

struct csr_t {
  int rows, cols;
  std::vector<int> offsets, indices;
  std::vector<float> values;
};

// test_helpers.hxx:55-88: sort by (row, col), count, prefix-sum
static csr_t coords_to_csr(int rows, int cols, std::vector<int> ri, std::vector<int> ci, std::vector<float> vs) {
  const std::size_t nnz = vs.size();
  std::vector<std::size_t> perm(nnz);
  for (std::size_t i = 0; i < nnz; ++i) perm[i] = i;
  std::sort(perm.begin(), perm.end(),
            [&](std::size_t a, std::size_t b) { return std::tie(ri[a], ci[a]) < std::tie(ri[b], ci[b]); });
  csr_t m{rows, cols, std::vector<int>(rows + 1, 0), std::vector<int>(nnz), std::vector<float>(nnz)};
  for (std::size_t i = 0; i < nnz; ++i) m.offsets[ri[perm[i]] + 1]++;
  for (int i = 1; i <= rows; ++i) m.offsets[i] += m.offsets[i - 1];
  for (std::size_t i = 0; i < nnz; ++i) {
    m.indices[i] = ci[perm[i]];
    m.values[i] = vs[perm[i]];
  }
  return m;
}

static csr_t identity(int n) {  // :92-101
  std::vector<int> ri(n), ci(n);
  std::vector<float> vs(n, 1.0f);
  for (int i = 0; i < n; ++i) ri[i] = ci[i] = i;
  return coords_to_csr(n, n, ri, ci, vs);
}

static csr_t banded(int n, int lower, int upper) {  // :104-120
  std::vector<int> ri, ci;
  std::vector<float> vs;
  for (int r = 0; r < n; ++r)
    for (int c = std::max(0, r - lower); c <= std::min(n - 1, r + upper); ++c) {
      ri.push_back(r);
      ci.push_back(c);
      vs.push_back(static_cast<float>(r * 1000 + c) * 0.001f + 0.5f);
    }
  return coords_to_csr(n, n, ri, ci, vs);
}

static csr_t block_diag(int num_blocks, int bs) {  // :124-140
  const int n = num_blocks * bs;
  std::vector<int> ri, ci;
  std::vector<float> vs;
  for (int b = 0; b < num_blocks; ++b)
    for (int i = 0; i < bs; ++i)
      for (int j = 0; j < bs; ++j) {
        ri.push_back(b * bs + i);
        ci.push_back(b * bs + j);
        vs.push_back(0.5f + static_cast<float>(b) + static_cast<float>(i * bs + j) * 0.01f);
      }
  return coords_to_csr(n, n, ri, ci, vs);
}

static csr_t skewed(int rows, int cols, int heavy, int light, std::uint64_t seed = 7u) {  // :145-174
  std::mt19937 rng(seed);
  std::uniform_int_distribution<int> col_dist(0, cols - 1);
  std::uniform_real_distribution<float> val_dist(0.5f, 1.5f);
  std::vector<int> ri, ci;
  std::vector<float> vs;
  auto add_row = [&](int r, int nnz) {
    std::vector<int> picks;
    while (static_cast<int>(picks.size()) < nnz) {
      const int c = col_dist(rng);
      if (std::find(picks.begin(), picks.end(), c) == picks.end()) picks.push_back(c);
    }
    for (int c : picks) {
      ri.push_back(r);
      ci.push_back(c);
      vs.push_back(val_dist(rng));
    }
  };
  add_row(0, std::min(heavy, cols));
  for (int r = 1; r < rows; ++r) add_row(r, std::min(light, cols));
  return coords_to_csr(rows, cols, ri, ci, vs);
}

static csr_t sprinkled(int rows, int cols, float density, int empty_every, std::uint64_t seed) {  // :178-200 / :205-225
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> dens_dist(0.0f, 1.0f);
  std::uniform_real_distribution<float> val_dist(0.5f, 1.5f);
  std::vector<int> ri, ci;
  std::vector<float> vs;
  for (int r = 0; r < rows; ++r) {
    if (empty_every > 0 && (r % empty_every) == 0) continue;
    for (int c = 0; c < cols; ++c)
      if (dens_dist(rng) < density) {
        ri.push_back(r);
        ci.push_back(c);
        vs.push_back(val_dist(rng));
      }
  }
  return coords_to_csr(rows, cols, ri, ci, vs);
}

static std::vector<float> input_vector(const csr_t& m, std::uint64_t seed = 23u) {  // :228-240
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> dist(0.5f, 1.5f);
  std::vector<float> x(m.cols);
  for (int i = 0; i < m.cols; ++i) x[i] = dist(rng);
  return x;
}

static std::vector<float> reference_spmv(const csr_t& m, const std::vector<float>& x) {  // :268-279
  std::vector<float> y(m.rows, 0.0f);
  for (int r = 0; r < m.rows; ++r) {
    float s = 0.0f;
    for (int k = m.offsets[r]; k < m.offsets[r + 1]; ++k) s += m.values[k] * x[m.indices[k]];
    y[r] = s;
  }
  return y;
}

template <typename T>
static void line(const std::vector<T>& v, const char* fmt) {
  for (std::size_t i = 0; i < v.size(); ++i) {
    if (i) std::putchar(' ');
    std::printf(fmt, v[i]);
  }
  std::putchar('\n');
}

int main() {
  // test_spmv_battery.hxx:52-65, in its order and with its labels
  std::vector<std::pair<std::string, csr_t>> battery = {
      {"identity-16", identity(16)},
      {"banded(0,0)/diag-16", banded(16, 0, 0)},
      {"banded(1,1)/tridiag-16", banded(16, 1, 1)},
      {"banded(3,4)/asym-32", banded(32, 3, 4)},
      {"block_diag(4,2)", block_diag(4, 2)},
      {"block_diag(3,3)", block_diag(3, 3)},
      {"skewed(20,50,h=16,l=2)", skewed(20, 50, 16, 2)},
      {"empty_rows(20,12,0.3,every-4)", sprinkled(20, 12, 0.3f, 4, 11u)},
      {"random(50,50,0.05)", sprinkled(50, 50, 0.05f, 0, 17u)},
  };
  for (auto& [name, m] : battery) {
    const auto x = input_vector(m);
    const auto y = reference_spmv(m, x);
    std::printf("%s|%d|%d|%zu\n", name.c_str(), m.rows, m.cols, m.values.size());
    line(m.offsets, "%d");
    line(m.indices, "%d");
    line(m.values, "%.9g");
    line(x, "%.9g");
    line(y, "%.9g");
  }
  return 0;
}
