/** @file math.hxx  `loops::math::ceil_div`, `log2_ceil`: see core.hxx. */
#pragma once
#include <loops/core.hxx>
