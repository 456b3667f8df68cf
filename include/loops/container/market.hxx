/**
 * @file market.hxx
 * @brief `matrix_market_t<index_t, offset_t, type_t>::load(path)` -> host `coo_t`.
 *
 * Reads Matrix-Market *coordinate* files (real / integer / pattern, general / symmetric).
 * Behaviour follows the reference loader (include/loops/container/market.hxx:100-289, pinned
 * by its unittests/test_market_loader.cu:95-317): 1-based indices become 0-based, `pattern`
 * entries get value 1, every off-diagonal entry of a `symmetric` file is mirrored right after
 * itself, comments / blank lines are skipped; complex, hermitian, skew-symmetric, dense `array`
 * files, a missing banner and zero-based indices are rejected with `error::exception_t`.
 * The file is memory-mapped and tokenised in ONE pass with std::from_chars (the entry count of a
 * symmetric file is only an upper bound -- 2 * header_nnz -- until the pass ends).
 */
#pragma once

#include <cstddef>
#include <limits>
#include <string>
#include <vector>

#include <loops/container/detail/mapped_file.hxx>
#include <loops/container/detail/mtx_parser.hxx>
#include <loops/container/formats.hxx>
#include <loops/error.hxx>
#include <loops/memory.hxx>
#include <loops/util/filepath.hxx>

namespace loops {
using namespace memory;

template <typename index_t, typename offset_t, typename type_t>
struct matrix_market_t {
  std::string filename;  ///< path given to load()
  std::string dataset;   ///< file name without directory and extension
  detail::mm_typecode_t code;

  matrix_market_t() = default;

  coo_t<index_t, type_t, memory_space_t::host> load(std::string _filename) {
    filename = std::move(_filename);
    dataset = extract_dataset(extract_filename(filename));

    detail::mapped_file_t file(filename);
    const char* p = file.data();
    const char* const end = file.end();
    error::throw_if_exception(p == end, "matrix-market: empty file " + filename);

    p = detail::parse_banner(p, end, code);
    error::throw_if_exception(!code.is_matrix, "matrix-market: object must be 'matrix' in " + filename);
    error::throw_if_exception(!code.is_coordinate,
                              "matrix-market: only the coordinate (sparse) format is supported in " + filename);
    error::throw_if_exception(code.is_complex, "matrix-market: complex values not supported in " + filename);
    error::throw_if_exception(code.is_hermitian || code.is_skew,
                              "matrix-market: hermitian / skew-symmetric not supported in " + filename);
    error::throw_if_exception(!(code.is_general || code.is_symmetric),
                              "matrix-market: missing or unrecognized symmetry tag in " + filename);

    p = detail::skip_comments(p, end);
    std::size_t dims[3] = {0, 0, 0};  // rows, cols, stored entries
    for (auto& d : dims) {
      p = detail::skip_blank(p, end);
      const char* q = detail::parse_size_t(p, end, d);
      error::throw_if_exception(q == p, "matrix-market: malformed dimension line in " + filename);
      p = q;
    }
    p = detail::skip_to_eol(p, end);
    const std::size_t imax = static_cast<std::size_t>(std::numeric_limits<index_t>::max());
    error::throw_if_exception(dims[0] >= imax || dims[1] >= imax,
                              "matrix-market: index_t overflow (rows or cols >= INT_MAX) in " + filename);

    std::vector<index_t> I, J;
    std::vector<type_t> V;
    const std::size_t reserve = code.is_symmetric ? 2 * dims[2] : dims[2];
    I.reserve(reserve);
    J.reserve(reserve);
    V.reserve(reserve);
    for (std::size_t k = 0; k < dims[2]; ++k) {
      p = detail::skip_ws(p, end);
      std::size_t r1 = 0, c1 = 0;
      const char* q = detail::parse_size_t(p, end, r1);
      error::throw_if_exception(q == p, "matrix-market: expected row index in body");
      p = detail::skip_blank(q, end);
      q = detail::parse_size_t(p, end, c1);
      error::throw_if_exception(q == p, "matrix-market: expected column index in body");
      p = q;
      double w = 1.0;
      if (!code.is_pattern) {
        p = detail::skip_blank(p, end);
        q = detail::parse_double(p, end, w);
        error::throw_if_exception(q == p, "matrix-market: expected value in body");
        p = q;
      }
      p = detail::skip_to_eol(p, end);
      error::throw_if_exception(r1 == 0 || c1 == 0, "matrix-market: zero-indexed entry (Matrix Market is 1-indexed)");
      const index_t r = static_cast<index_t>(r1 - 1), c = static_cast<index_t>(c1 - 1);
      I.push_back(r);
      J.push_back(c);
      V.push_back(static_cast<type_t>(w));
      if (code.is_symmetric && r != c) {
        I.push_back(c);
        J.push_back(r);
        V.push_back(static_cast<type_t>(w));
      }
    }
    error::throw_if_exception(I.size() >= static_cast<std::size_t>(std::numeric_limits<offset_t>::max()),
                              "matrix-market: offset_t overflow (final nnz exceeds offset_t max) in " + filename);

    coo_t<index_t, type_t, memory_space_t::host> coo(dims[0], dims[1], I.size());
    std::copy(I.begin(), I.end(), coo.row_indices.begin());
    std::copy(J.begin(), J.end(), coo.col_indices.begin());
    std::copy(V.begin(), V.end(), coo.values.begin());
    return coo;
  }
};

}  // namespace loops
