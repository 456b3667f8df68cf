/**
 * @file range.hxx
 * @brief Python-style ranges usable in host and device code:
 *   `for (auto i : range(b, e))`, `range(b, e).step(s)`, `range(b)` (unbounded), `indices(c)`.
 *
 * Behavioural contract restated from the reference (include/loops/range.hxx:52-116,181-226;
 * pinned by unittests/test_util_range.cu:22-78): a unit range stops when `current == end`; a
 * stepped range stops as soon as `current >= end` for a positive step (so `range(0, 9).step(3)`
 * yields 0, 3, 6) and as soon as `current < end` for a non-positive one.
 */
#pragma once

#include <cstddef>
#include <initializer_list>
#include <type_traits>

#include <hip/hip_runtime.h>

namespace loops {

/// [begin, end) advanced by `step`; the sentinel comparison is directional (see file comment).
template <typename type_t>
struct strided_span {
  struct iter {
    type_t current;
    type_t step;
    __host__ __device__ type_t operator*() const { return current; }
    __host__ __device__ iter& operator++() { current += step; return *this; }
    __host__ __device__ iter operator++(int) { iter c = *this; current += step; return c; }
    __host__ __device__ bool operator==(iter const& o) const { return step > 0 ? current >= o.current : current < o.current; }
    __host__ __device__ bool operator!=(iter const& o) const { return !(*this == o); }
  };
  type_t begin_, end_, step_;
  __host__ __device__ strided_span(type_t b, type_t e, type_t s) : begin_(b), end_(e), step_(s) {}
  __host__ __device__ iter begin() const { return iter{begin_, step_}; }
  __host__ __device__ iter end() const { return iter{end_, step_}; }
};

template <typename type_t>
struct range_proxy {
  using step_range_proxy = strided_span<type_t>;
  struct iter {
    type_t current;
    __host__ __device__ type_t operator*() const { return current; }
    __host__ __device__ iter& operator++() { ++current; return *this; }
    __host__ __device__ iter operator++(int) { iter c = *this; ++current; return c; }
    __host__ __device__ bool operator==(iter const& o) const { return current == o.current; }
    __host__ __device__ bool operator!=(iter const& o) const { return current != o.current; }
  };
  type_t begin_, end_;
  __host__ __device__ range_proxy(type_t b, type_t e) : begin_(b), end_(e) {}
  __host__ __device__ step_range_proxy step(type_t s) const { return step_range_proxy(begin_, end_, s); }
  __host__ __device__ iter begin() const { return iter{begin_}; }
  __host__ __device__ iter end() const { return iter{end_}; }
};

template <typename type_t>
struct infinite_range_proxy {
  struct iter {
    type_t current;
    type_t step;
    __host__ __device__ type_t operator*() const { return current; }
    __host__ __device__ iter& operator++() { current += step; return *this; }
    __host__ __device__ bool operator==(iter const&) const { return false; }
    __host__ __device__ bool operator!=(iter const&) const { return true; }
  };
  struct step_range_proxy {
    type_t begin_, step_;
    __host__ __device__ iter begin() const { return iter{begin_, step_}; }
    __host__ __device__ iter end() const { return iter{type_t(), step_}; }
  };
  type_t begin_;
  __host__ __device__ explicit infinite_range_proxy(type_t b) : begin_(b) {}
  __host__ __device__ step_range_proxy step(type_t s) const { return step_range_proxy{begin_, s}; }
  __host__ __device__ iter begin() const { return iter{begin_, type_t(1)}; }
  __host__ __device__ iter end() const { return iter{type_t(), type_t(1)}; }
};

template <typename type_t>
__host__ __device__ range_proxy<type_t> range(type_t begin, type_t end) { return {begin, end}; }

template <typename type_t>
__host__ __device__ infinite_range_proxy<type_t> range(type_t begin) { return infinite_range_proxy<type_t>(begin); }

/// Index range of anything with an integral `.size()` (host only: container sizes are host calls).
template <typename C, typename = std::enable_if_t<std::is_integral<decltype(std::declval<C const&>().size())>::value>>
__host__ auto indices(C const& cont) -> range_proxy<decltype(cont.size())> { return {0, cont.size()}; }

template <typename type_t, std::size_t N>
__host__ __device__ range_proxy<std::size_t> indices(type_t (&)[N]) { return {0, N}; }

template <typename type_t>
__host__ __device__ range_proxy<typename std::initializer_list<type_t>::size_type> indices(std::initializer_list<type_t>&& cont) {
  return {0, cont.size()};
}

// ---------------------------------------------------------------------------- launch-shaped spans
// The strided spans kernels iterate with (<loops/stride_ranges.hxx> forwards here).  Semantics of the
// reference's helpers (stride_ranges.hxx:16-62): the span starts at `begin` (+ the caller's global
// thread rank for the grid flavour) and advances by the stated launch dimension.

template <typename T>
using step_range_t = typename range_proxy<T>::step_range_proxy;

namespace detail {
template <typename T, typename S>
__host__ __device__ __forceinline__ step_range_t<T> span_from(T first, T end, S stride) {
  return step_range_t<T>(first, end, static_cast<T>(stride));
}
}  // namespace detail

/// One element per launched thread, then again one grid further on.
template <typename T>
__device__ __forceinline__ step_range_t<T> grid_stride_range(T begin, T end) {
  const unsigned int rank = blockDim.x * blockIdx.x + threadIdx.x;
  return detail::span_from(static_cast<T>(begin + static_cast<T>(rank)), end, gridDim.x * blockDim.x);
}

/// [begin, end) in steps of the workgroup size (the caller adds its own lane offset to `begin`).
template <typename T>
__device__ __forceinline__ step_range_t<T> block_stride_range(T begin, T end) {
  return detail::span_from(begin, end, blockDim.x);
}

/// [begin, end) in steps of `stride`.
template <typename T>
__device__ __forceinline__ step_range_t<T> custom_stride_range(T begin, T end, T stride) {
  return detail::span_from(begin, end, stride);
}

}  // namespace loops
