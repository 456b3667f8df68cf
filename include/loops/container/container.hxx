/** @file container.hxx  Everything under loops/container: dense + sparse containers, layout views, loader. */
#pragma once
#include <loops/container/vector.hxx>
#include <loops/container/matrix.cuh>
#include <loops/container/layout.hxx>
#include <loops/container/formats.hxx>
#include <loops/container/market.hxx>
