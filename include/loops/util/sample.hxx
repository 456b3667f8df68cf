/** @file sample.hxx  A 4 x 4 sample CSR (offsets {0,0,2,3,4}: an empty first row) for smoke tests
 *  (reference include/loops/util/sample.hxx:62-100).
 *      [ 0 0 0 0 ]
 *      [ 5 8 0 0 ]
 *      [ 0 0 3 0 ]
 *      [ 0 6 0 0 ]   */
#pragma once

#include <loops/container/formats.hxx>
#include <loops/memory.hxx>

namespace loops {
namespace sample {
using namespace memory;

template <memory_space_t space = memory_space_t::device, typename index_t = int, typename offset_t = int,
          typename value_t = float>
csr_t<index_t, offset_t, value_t, space> csr() {
  csr_t<index_t, offset_t, value_t, memory_space_t::host> m(4, 4, 4);
  const offset_t off[5] = {0, 0, 2, 3, 4};
  const index_t idx[4] = {0, 1, 2, 1};
  const value_t val[4] = {5, 8, 3, 6};
  for (int i = 0; i < 5; ++i) m.offsets[i] = off[i];
  for (int i = 0; i < 4; ++i) {
    m.indices[i] = idx[i];
    m.values[i] = val[i];
  }
  return csr_t<index_t, offset_t, value_t, space>(m);
}

}  // namespace sample
}  // namespace loops
