/**
 * @file mtx_parser.hxx
 * @brief Tokenising helpers for the Matrix-Market coordinate reader: locale-free number parsing
 * with std::from_chars over a memory range (no iostreams, no sscanf), banner decoding.
 * (Role of the reference's container/detail/mtx_parser.hxx.)
 */
#pragma once

#include <cctype>
#include <charconv>
#include <cstddef>
#include <string>

#include <loops/error.hxx>

namespace loops {
namespace detail {

inline bool is_blank(char c) noexcept { return c == ' ' || c == '\t'; }
inline bool is_space(char c) noexcept { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

inline const char* skip_blank(const char* p, const char* end) noexcept {
  while (p < end && is_blank(*p)) ++p;
  return p;
}
inline const char* skip_ws(const char* p, const char* end) noexcept {
  while (p < end && is_space(*p)) ++p;
  return p;
}
inline const char* skip_to_eol(const char* p, const char* end) noexcept {
  while (p < end && *p != '\n') ++p;
  return p < end ? p + 1 : p;
}
/// Skip blank lines and '%' comment lines.
inline const char* skip_comments(const char* p, const char* end) noexcept {
  for (p = skip_ws(p, end); p < end && *p == '%'; p = skip_ws(p, end)) p = skip_to_eol(p, end);
  return p;
}
/// Returns p unchanged when no number could be read.
inline const char* parse_size_t(const char* p, const char* end, std::size_t& v) noexcept {
  const auto r = std::from_chars(p, end, v);
  return r.ec == std::errc() ? r.ptr : p;
}
inline const char* parse_double(const char* p, const char* end, double& v) noexcept {
  const char* q = (p < end && *p == '+') ? p + 1 : p;  // from_chars rejects a leading '+'
  const auto r = std::from_chars(q, end, v);
  return r.ec == std::errc() ? r.ptr : p;
}

/// What the `%%MatrixMarket object format field symmetry` banner says.
struct mm_typecode_t {
  bool is_matrix = false;
  bool is_coordinate = false;
  bool is_real = false;
  bool is_integer = false;
  bool is_pattern = false;
  bool is_complex = false;
  bool is_general = false;
  bool is_symmetric = false;
  bool is_skew = false;
  bool is_hermitian = false;
};

inline const char* parse_banner(const char* p, const char* end, mm_typecode_t& tc) {
  static const std::string magic = "%%MatrixMarket";
  error::throw_if_exception(static_cast<std::size_t>(end - p) < magic.size() ||
                                std::string(p, magic.size()) != magic,
                            "matrix-market: missing %%MatrixMarket banner");
  p += magic.size();
  std::string word[4];
  for (auto& w : word) {
    p = skip_blank(p, end);
    while (p < end && !is_space(*p)) w.push_back(static_cast<char>(std::tolower(static_cast<unsigned char>(*p++))));
  }
  tc.is_matrix = word[0] == "matrix";
  tc.is_coordinate = word[1] == "coordinate";
  tc.is_real = word[2] == "real";
  tc.is_integer = word[2] == "integer";
  tc.is_pattern = word[2] == "pattern";
  tc.is_complex = word[2] == "complex";
  tc.is_general = word[3] == "general";
  tc.is_symmetric = word[3] == "symmetric";
  tc.is_skew = word[3] == "skew-symmetric";
  tc.is_hermitian = word[3] == "hermitian";
  return skip_to_eol(p, end);
}

}  // namespace detail
}  // namespace loops
