/** @file vector.hxx  `vector_t<T, space>`, `host_vector_t`, `device_vector_t`: see core.hxx. */
#pragma once
#include <loops/core.hxx>
