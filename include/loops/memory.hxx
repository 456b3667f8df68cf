/** @file memory.hxx  `loops::memory` (memory_space_t, raw_pointer_cast) and `loops::pair`: see core.hxx. */
#pragma once
#include <loops/core.hxx>
