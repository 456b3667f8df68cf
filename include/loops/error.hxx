/**
 * @file error.hxx
 * @brief `loops::error::exception_t` / `throw_if_exception` (reference: include/loops/error.hxx:22-46).
 */
#pragma once

#include <exception>
#include <string>

#include <loops/backend/xpu.hxx>

namespace loops {
namespace error {

using error_t = xpu::error_t;

struct exception_t : std::exception {
  std::string report;
  explicit exception_t(error_t status, std::string message = "")
      : report(std::string(xpu::get_error_string(status)) + "\t: " + message) {}
  explicit exception_t(std::string message = "") : report(std::move(message)) {}
  const char* what() const noexcept override { return report.c_str(); }
};

/// An argument the caller got wrong (an index outside the matrix, a size outside the documented range) -- as opposed to a
/// resource that ran out: callers that treat a failed OPTIONAL step as "do without" (algorithms::spmv::spmv_plan_t and its
/// re-ordered copies) let this one through, as the C ABI returns LOOPS_E_BADARG for the same input.
struct bad_argument_t : exception_t {
  explicit bad_argument_t(std::string message = "") : exception_t(std::move(message)) {}
};

inline void throw_if_exception(error_t status, std::string message = "") {
  if (status != xpu::success) throw exception_t(status, std::move(message));
}
inline void throw_if_exception(bool is_exception, std::string message = "") {
  if (is_exception) throw exception_t(std::move(message));
}

}  // namespace error
}  // namespace loops
