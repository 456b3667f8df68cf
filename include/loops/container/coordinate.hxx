/** @file coordinate.hxx  `coordinate_t<index_t>`: see core.hxx. */
#pragma once
#include <loops/core.hxx>
