// loops_gen.cpp -- libloops_gen.so: host-side (CPU, OpenMP) builder of the synthetic workloads BASELINE.json names
// (clamped-Zipf row degrees, hashed per-row distinct sorted columns, dyadic values).  Same functions, bit
// for bit, as the numpy specification in loops_amd/generate.py (tests/test_generate.py compares them); this build
// exists because the numpy version needs minutes for the C3 stand-in (194 M nonzeros) and the C5 matrix (537 M).
// Workload generation only: no part of the SpMV path, nothing from the reference, no device code.
//
//   g++ -O3 -fopenmp -shared -fPIC loops_gen.cpp -o libloops_gen.so
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace {

inline std::uint64_t splitmix64(std::uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// generate.py::_hash_cols for window in {none (< 0 here: LOOPS_GEN_UNIFORM), runs (-1), band (> 0)}
constexpr std::int64_t kUniform = INT64_MIN;

// host-blocked columns (window == -4, generate.py::_hash_cols): `hb`, `he` = the block of consecutive ids ("host") the row
// belongs to.  14 of 16 links stay inside the host when it can hold twice the row's degree; the rest go anywhere.
constexpr std::int64_t kHostBlocked = -4;

inline std::int64_t hash_col(std::uint64_t seed, std::int64_t row_abs, std::int64_t k, std::uint64_t attempt,
                             std::int64_t cols, std::int64_t window, std::int64_t deg, std::int64_t hb = 0, std::int64_t he = 0) {
  const std::uint64_t h = splitmix64(splitmix64(seed + static_cast<std::uint64_t>(row_abs)) + static_cast<std::uint64_t>(k) +
                                     (attempt << 40));
  if (window == kUniform) return static_cast<std::int64_t>(h % static_cast<std::uint64_t>(cols));
  if (window == kHostBlocked) {
    const std::int64_t size = he - hb;
    const bool local = ((h >> 60) < 14ull) && size >= 2 * deg;
    if (local) return hb + static_cast<std::int64_t>((h & 0xFFFFFFFFFFFFull) % static_cast<std::uint64_t>(size));
    return static_cast<std::int64_t>(splitmix64(h) % static_cast<std::uint64_t>(cols));
  }
  if (window == -1) {  // runs: consecutive columns from a hashed start
    const std::uint64_t start = splitmix64(seed * 31ull + static_cast<std::uint64_t>(row_abs)) % static_cast<std::uint64_t>(cols);
    return (static_cast<std::int64_t>(start) + k) % cols;
  }
  // band around the diagonal, at least 4x the row's degree wide, never wider than the matrix
  std::int64_t w = std::max<std::int64_t>(window, 4 * deg);
  w = std::min<std::int64_t>(w, cols);
  const std::int64_t off = static_cast<std::int64_t>(h % static_cast<std::uint64_t>(w)) - w / 2;
  std::int64_t c = (row_abs + off) % cols;
  if (c < 0) c += cols;  // numpy's % is non-negative
  return c;
}

}  // namespace

extern "C" {

// sum over i of clip(floor(c * rank[i]), 1, cap)  (generate.py::powerlaw_degrees::total)
long long loops_gen_degree_total(const double* rank, long long n, double c, long long cap) {
  long long total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
  for (long long i = 0; i < n; ++i) {
    double d = std::floor(c * rank[i]);
    d = d < 1.0 ? 1.0 : (d > static_cast<double>(cap) ? static_cast<double>(cap) : d);
    total += static_cast<long long>(d);
  }
  return total;
}

// d[i] = clip(floor(c * rank[i]), 1, cap)
void loops_gen_degrees(const double* rank, long long n, double c, long long cap, long long* d) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < n; ++i) {
    double v = std::floor(c * rank[i]);
    v = v < 1.0 ? 1.0 : (v > static_cast<double>(cap) ? static_cast<double>(cap) : v);
    d[i] = static_cast<long long>(v);
  }
}

// keys[i] = splitmix64(seed ^ i)  (the row permutation's sort keys)
void loops_gen_perm_keys(unsigned long long seed, long long n, unsigned long long* keys) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < n; ++i) keys[i] = splitmix64(seed ^ static_cast<std::uint64_t>(i));
}

// Rows [row_begin, row_begin + nrows) of the hashed matrix (generate.py::csr_from_degrees): offsets[nrows + 1]
// (int64, prefix sums of `degrees`, computed by the caller), per-row distinct columns sorted ascending, values
// k/8 (exact != 0) or U[0.5, 1.5).  window: LLONG_MIN = uniform columns, -1 = runs, > 0 = band.
// Returns 0, or -1 if a row cannot hold `degree` distinct columns.
static int gen_csr_rows(const long long* degrees, const long long* offsets, long long nrows, long long cols,
                        unsigned long long seed, long long row_begin, int exact, long long window, int* indices,
                        float* values, const long long* hosts, long long num_hosts) {
  int status = 0;
#pragma omp parallel
  {
    std::vector<std::pair<std::int64_t, std::int64_t>> item;  // (column, k)
    std::vector<std::int64_t> col;
#pragma omp for schedule(dynamic, 256)
    for (long long r = 0; r < nrows; ++r) {
      const std::int64_t deg = degrees[r];
      const std::int64_t row_abs = r + row_begin;
      if (deg > cols) {
#pragma omp atomic write
        status = -1;
        continue;
      }
      col.resize(static_cast<std::size_t>(deg));
      item.resize(static_cast<std::size_t>(deg));
      std::int64_t hb = 0, he = cols;
      if (window == kHostBlocked) {  // the host of id min(row, cols - 1): last boundary <= id
        const std::int64_t id = row_abs < cols ? row_abs : cols - 1;
        const long long* up = std::upper_bound(hosts, hosts + num_hosts + 1, static_cast<long long>(id));
        hb = *(up - 1);
        he = *up;
      }
      for (std::int64_t k = 0; k < deg; ++k) col[k] = hash_col(seed, row_abs, k, 0, cols, window, deg, hb, he);
      std::uint64_t attempt = 0;
      for (;;) {
        for (std::int64_t k = 0; k < deg; ++k) item[k] = {col[k], k};
        std::sort(item.begin(), item.end());  // by column, ties by k: the stable order of the specification
        bool dup = false;
        for (std::int64_t i = 1; i < deg; ++i) {
          if (item[i].first == item[i - 1].first) {  // every later entry of an equal run is redrawn
            if (!dup) { dup = true; ++attempt; }
            const std::int64_t k = item[i].second;
            col[k] = hash_col(seed, row_abs, k, attempt, cols, window, deg, hb, he);
          }
        }
        if (!dup) break;
      }
      int* out_i = indices + offsets[r];
      float* out_v = values + offsets[r];
      for (std::int64_t i = 0; i < deg; ++i) {
        const std::int64_t c = item[i].first;
        out_i[i] = static_cast<int>(c);
        const std::uint64_t vh = splitmix64(static_cast<std::uint64_t>(row_abs) * 0x100000001B3ull + static_cast<std::uint64_t>(c) +
                                            seed * 7919ull);
        if (exact) out_v[i] = static_cast<float>(((vh >> 33) % 8ull) + 1ull) / 8.0f;
        else out_v[i] = static_cast<float>(0.5 + static_cast<double>(vh >> 40) / static_cast<double>(1 << 24));
      }
    }
  }
  return status;
}

int loops_gen_csr_rows(const long long* degrees, const long long* offsets, long long nrows, long long cols,
                       unsigned long long seed, long long row_begin, int exact, long long window, int* indices,
                       float* values) {
  if (window == kHostBlocked) return -2;  // needs the host table: loops_gen_csr_rows_hosts
  return gen_csr_rows(degrees, offsets, nrows, cols, seed, row_begin, exact, window, indices, values, nullptr, 0);
}

// window = -4 ("host-blocked"): hosts[0 .. num_hosts] = ascending id boundaries, hosts[0] = 0, hosts[num_hosts] = cols
int loops_gen_csr_rows_hosts(const long long* degrees, const long long* offsets, long long nrows, long long cols,
                             unsigned long long seed, long long row_begin, int exact, const long long* hosts,
                             long long num_hosts, int* indices, float* values) {
  if (!hosts || num_hosts < 1 || hosts[0] != 0 || hosts[num_hosts] != cols) return -2;
  return gen_csr_rows(degrees, offsets, nrows, cols, seed, row_begin, exact, kHostBlocked, indices, values, hosts, num_hosts);
}

// x[i] = reference x generator for INT bounds (generate.py::uniform_distribution_int, util/generate.hxx:33-79)
void loops_gen_x_int(long long start, long long n, int lo, int hi, unsigned seed, float* x) {
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < n; ++j) {
    std::uint64_t a = static_cast<std::uint64_t>(start + j) & 0xFFFFFFFFull;
    const std::uint64_t m32 = 0xFFFFFFFFull;
    a = ((a + 0x7ED55D16ull) + (a << 12)) & m32;
    a = ((a ^ 0xC761C23Cull) ^ (a >> 19)) & m32;
    a = ((a + 0x165667B1ull) + (a << 5)) & m32;
    a = ((a + 0xD3A2646Cull) ^ (a << 9)) & m32;
    a = ((a + 0xFD7046C5ull) + (a << 3)) & m32;
    a = ((a ^ 0xB55A4F09ull) ^ (a >> 16)) & m32;
    const std::uint64_t m = 2147483647ull;
    std::uint64_t s = ((a * static_cast<std::uint64_t>(seed)) & m32) % m;
    if (s == 0) s = 1;
    const std::uint64_t u = (48271ull * s) % m;
    const double r = static_cast<double>(u - 1ull) / (1.0 + static_cast<double>(m - 2ull));
    const double v = r * ((static_cast<double>(hi) + 1.0) - static_cast<double>(lo)) + static_cast<double>(lo);
    x[j] = static_cast<float>(static_cast<long long>(v));
  }
}


// R-MAT / Kronecker edges (generate.py::rmat_edges is the specification): edge e draws one hashed u per level,
// u = (splitmix64(splitmix64(seed) + e * 64 + level) >> 11) * 2^-53, and descends into quadrant (0,0) | (0,1) | (1,0) | (1,1)
// for u < a | < a + b | < a + b + c | else (Graph500: a, b, c = 0.57, 0.19, 0.19); level 0 sets the most significant bit.
void loops_gen_rmat_edges(int scale, long long nedges, double a, double b, double c, unsigned long long seed, int* row, int* col) {
  const std::uint64_t base = splitmix64(seed);
  const double ab = a + b, abc = a + b + c;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nedges; ++e) {
    std::uint32_t r = 0, cc = 0;
    for (int level = 0; level < scale; ++level) {
      const std::uint64_t h = splitmix64(base + static_cast<std::uint64_t>(e) * 64ull + static_cast<std::uint64_t>(level));
      const double u = static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0);
      const std::uint32_t rb = u >= ab ? 1u : 0u;
      const std::uint32_t cb = (u >= a && u < ab) || u >= abc ? 1u : 0u;
      r = (r << 1) | rb;
      cc = (cc << 1) | cb;
    }
    row[e] = static_cast<int>(r);
    col[e] = static_cast<int>(cc);
  }
}

}  // extern "C"

#include <omp.h>
extern "C" {
int loops_gen_threads(void) { return omp_get_max_threads(); }
void loops_gen_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
}
