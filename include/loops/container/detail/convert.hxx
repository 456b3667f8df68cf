/**
 * @file convert.hxx
 * @brief Compressed-offsets <-> expanded-indices conversions and the COO ordering primitive.
 *
 * `indices_to_offsets`: offsets[i] = number of entries with (sorted) index < i -- a vectorised
 * lower_bound (reference container/detail/convert.hxx:70-78).  `offsets_to_indices`: the inverse
 * (expand row offsets into one row id per nonzero; reference :37-58), done as a vectorised
 * upper_bound over the offsets.  `order_by`: sorts a COO triplet by a (major, minor) pair with ONE
 * 64-bit-key radix sort (major << 32 | minor) instead of the reference's comparison sort over a
 * zip iterator (coo.hxx:104-165) -- rocPRIM's radix sort on packed keys is the fast path on CDNA.
 */
#pragma once

#include <cstdint>

#include <thrust/binary_search.h>
#include <thrust/gather.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/sequence.h>
#include <thrust/sort.h>
#include <thrust/transform.h>

#include <loops/memory.hxx>
#include <loops/container/vector.hxx>

namespace loops {
namespace detail {
using namespace memory;

template <typename index_v_t, typename offset_v_t>
void offsets_to_indices(const offset_v_t& offsets, index_v_t& indices) {
  using offset_t = typename offset_v_t::value_type;
  // indices[k] = (first i with offsets[i] > k) - 1
  thrust::upper_bound(offsets.begin(), offsets.end(), thrust::counting_iterator<offset_t>(0),
                      thrust::counting_iterator<offset_t>(static_cast<offset_t>(indices.size())), indices.begin());
  using index_t = typename index_v_t::value_type;
  thrust::transform(indices.begin(), indices.end(), indices.begin(),
                    [] __host__ __device__(index_t v) { return static_cast<index_t>(v - 1); });
}

template <typename index_v_t, typename offset_v_t>
void indices_to_offsets(const index_v_t& indices, offset_v_t& offsets) {
  using offset_t = typename offset_v_t::value_type;
  thrust::lower_bound(indices.begin(), indices.end(), thrust::counting_iterator<offset_t>(0),
                      thrust::counting_iterator<offset_t>(static_cast<offset_t>(offsets.size())), offsets.begin());
}

/// Reorder (major, minor, values) so that (major, minor) is ascending; equal keys keep their
/// relative order (stable).  One radix sort of packed 64-bit keys + three gathers.
template <typename index_t, typename value_t, memory_space_t space>
void order_by(vector_t<index_t, space>& major, vector_t<index_t, space>& minor, vector_t<value_t, space>& values) {
  static_assert(sizeof(index_t) <= 4, "order_by packs (major, minor) into one 64-bit key");
  const std::size_t n = major.size();
  vector_t<std::uint64_t, space> keys(n);
  thrust::transform(major.begin(), major.end(), minor.begin(), keys.begin(),
                    [] __host__ __device__(index_t a, index_t b) {
                      return (static_cast<std::uint64_t>(static_cast<std::uint32_t>(a)) << 32) |
                             static_cast<std::uint64_t>(static_cast<std::uint32_t>(b));
                    });
  vector_t<std::uint32_t, space> perm(n);
  thrust::sequence(perm.begin(), perm.end());
  thrust::stable_sort_by_key(keys.begin(), keys.end(), perm.begin());
  vector_t<value_t, space> sorted_values(n);
  thrust::gather(perm.begin(), perm.end(), values.begin(), sorted_values.begin());
  values.swap(sorted_values);
  thrust::transform(keys.begin(), keys.end(), major.begin(),
                    [] __host__ __device__(std::uint64_t k) { return static_cast<index_t>(k >> 32); });
  thrust::transform(keys.begin(), keys.end(), minor.begin(),
                    [] __host__ __device__(std::uint64_t k) { return static_cast<index_t>(k & 0xFFFFFFFFull); });
}

}  // namespace detail
}  // namespace loops
