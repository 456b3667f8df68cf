// TESTS-ONLY stand-in for <catch2/catch_test_macros.hpp> (Catch2 is not in this image): just enough of the macro surface
// the reference's unit tests use -- TEST_CASE, SECTION, CHECK, REQUIRE, CHECK_THROWS, INFO -- to compile those test files
// UNCHANGED, in place, against include/loops (scripts/build_reference_unittests.sh) and run them on the GPU box
// (tests/test_reference_unittests_gpu.py).  Written from Catch2's documented behaviour, not from its sources: a TEST_CASE
// with SECTIONs is re-entered once per leaf section (flat sections; a nested SECTION runs with its parent), CHECK records a
// failure and goes on, REQUIRE ends the test case.  Provides main() (the reference links Catch2WithMain): one test binary
// per test file, exit status = number of failed test cases (capped), a summary line "test cases: N | M failed".
// Lives under tests/stubs/ and is never on the include path of the product.
#pragma once

#include <cstdio>
#include <exception>
#include <string>
#include <vector>

namespace catch_stub {
struct require_failed {};
struct test_case {
  const char* name;
  void (*fn)();
};
inline std::vector<test_case>& registry() {
  static std::vector<test_case> r;
  return r;
}
struct registrar {
  registrar(const char* name, void (*fn)()) { registry().push_back({name, fn}); }
};
struct state_t {
  int failures = 0;        // failed assertions of the current test case
  int assertions = 0;
  int sections_seen = 0;   // top-level SECTIONs met so far in this pass
  int section_target = 0;  // the one this pass runs
  int depth = 0;
  std::string info;
};
inline state_t& state() {
  static state_t s;
  return s;
}
inline void report(const char* kind, const char* expr, const char* file, int line) {
  state_t& s = state();
  ++s.failures;
  std::printf("%s:%d: FAILED: %s( %s )%s%s\n", file, line, kind, expr, s.info.empty() ? "" : "  with ", s.info.c_str());
}
struct section_guard {
  bool run;
  explicit section_guard(bool r) : run(r) { if (run) ++state().depth; }
  ~section_guard() { if (run) --state().depth; }
  explicit operator bool() const { return run; }
};
inline bool enter_section() {
  state_t& s = state();
  if (s.depth > 0) return true;  // nested: runs with its parent
  return s.sections_seen++ == s.section_target;
}
inline int run_all() {
  int failed_cases = 0, total_assertions = 0;
  for (const test_case& t : registry()) {
    state_t& s = state();
    s.failures = 0;
    s.section_target = 0;
    for (;;) {
      s.sections_seen = 0;
      s.depth = 0;
      s.info.clear();
      try {
        t.fn();
      } catch (const require_failed&) {
      } catch (const std::exception& e) {
        ++s.failures;
        std::printf("%s: FAILED: unexpected exception: %s\n", t.name, e.what());
      } catch (...) {
        ++s.failures;
        std::printf("%s: FAILED: unexpected exception\n", t.name);
      }
      if (++s.section_target >= s.sections_seen) break;  // every leaf section has had its pass
    }
    total_assertions += s.assertions;
    s.assertions = 0;
    if (s.failures) {
      ++failed_cases;
      std::printf("test case FAILED: %s\n", t.name);
    }
  }
  std::printf("test cases: %zu | %d failed | assertions: %d\n", registry().size(), failed_cases, total_assertions);
  return failed_cases > 100 ? 100 : failed_cases;
}
}  // namespace catch_stub

#define CATCH_STUB_CAT2(a, b) a##b
#define CATCH_STUB_CAT(a, b) CATCH_STUB_CAT2(a, b)
#define CATCH_STUB_TEST_CASE(fn, ...)                                                                  \
  static void fn();                                                                                    \
  static ::catch_stub::registrar CATCH_STUB_CAT(fn, _reg)(CATCH_STUB_FIRST(__VA_ARGS__, ""), &fn);     \
  static void fn()
#define CATCH_STUB_FIRST(a, ...) a
#define TEST_CASE(...) CATCH_STUB_TEST_CASE(CATCH_STUB_CAT(catch_stub_test_, __LINE__), __VA_ARGS__)
#define SECTION(...) if (::catch_stub::section_guard CATCH_STUB_CAT(catch_stub_section_, __LINE__){::catch_stub::enter_section()})
#define CHECK(...)                                                                                     \
  do {                                                                                                 \
    ++::catch_stub::state().assertions;                                                                \
    if (!(__VA_ARGS__)) ::catch_stub::report("CHECK", #__VA_ARGS__, __FILE__, __LINE__);               \
  } while (0)
#define REQUIRE(...)                                                                                   \
  do {                                                                                                 \
    ++::catch_stub::state().assertions;                                                                \
    if (!(__VA_ARGS__)) {                                                                              \
      ::catch_stub::report("REQUIRE", #__VA_ARGS__, __FILE__, __LINE__);                               \
      throw ::catch_stub::require_failed{};                                                            \
    }                                                                                                  \
  } while (0)
#define CHECK_THROWS(...)                                                                              \
  do {                                                                                                 \
    ++::catch_stub::state().assertions;                                                                \
    bool catch_stub_threw = false;                                                                     \
    try {                                                                                              \
      static_cast<void>(__VA_ARGS__);                                                                  \
    } catch (...) {                                                                                    \
      catch_stub_threw = true;                                                                         \
    }                                                                                                  \
    if (!catch_stub_threw) ::catch_stub::report("CHECK_THROWS", #__VA_ARGS__, __FILE__, __LINE__);     \
  } while (0)
#define INFO(...)                                                                                      \
  do {                                                                                                 \
    ::catch_stub::state().info = ::catch_stub::info_string() << __VA_ARGS__;                           \
  } while (0)

#include <sstream>
namespace catch_stub {
struct info_string {
  std::ostringstream os;
  template <typename T>
  info_string& operator<<(const T& v) {
    os << v;
    return *this;
  }
  operator std::string() const { return os.str(); }
};
}  // namespace catch_stub

int main() { return ::catch_stub::run_all(); }
