/**
 * @file fwd.hxx
 * @brief Declarations of the six sparse containers.  Template parameter lists as in the reference
 * (container/formats.hxx:18-59): index type, [offset type,] value type, memory space; bcsr additionally
 * carries its block shape R x C as leading non-type parameters.
 */
#pragma once

#include <cstddef>

#include <loops/core.hxx>

namespace loops {

// coordinate list and the two padded / diagonal formats: no offset array
template <typename index_t, typename value_t, memory_space_t space>
struct coo_t;
template <typename index_t, typename value_t, memory_space_t space>
struct ell_t;

// compressed formats
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
struct csr_t;
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
struct csc_t;
template <typename index_t, typename offset_t, typename value_t, memory_space_t space>
struct dia_t;
template <std::size_t R, std::size_t C, typename index_t, typename offset_t, typename value_t, memory_space_t space>
struct bcsr_t;

}  // namespace loops
