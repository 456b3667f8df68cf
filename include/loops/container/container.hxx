/** @file container.hxx  Everything under loops/container/ in one include. */
#pragma once
#include <loops/container/formats.hxx>
#include <loops/container/market.hxx>
#include <loops/container/vector.hxx>
#include <loops/container/matrix.cuh>
