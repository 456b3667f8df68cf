/** @file filepath.hxx  File-name helpers used by the example drivers
 *  (reference include/loops/util/filepath.hxx:18-35). */
#pragma once

#include <string>

namespace loops {

/// "dir/sub/name.ext" -> "name.ext".
inline std::string extract_filename(std::string path, std::string /*delim*/ = "/") {
  const auto slash = path.find_last_of('/');
  return slash == std::string::npos ? path : path.substr(slash + 1);
}
/// "name.ext" -> "name".
inline std::string extract_dataset(std::string filename) { return filename.substr(0, filename.find_last_of('.')); }

namespace detail {
inline bool ends_with(const std::string& s, const char* suffix) {
  const std::string t(suffix);
  return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0;
}
}  // namespace detail

inline bool is_market(std::string filename) {
  return detail::ends_with(filename, ".mtx") || detail::ends_with(filename, ".mmio");
}
inline bool is_binary_csr(std::string filename) { return detail::ends_with(filename, ".csr"); }

}  // namespace loops
