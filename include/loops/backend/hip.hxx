/** @file hip.hxx  Kept so `#include <loops/backend/hip.hxx>` resolves; everything lives in xpu.hxx. */
#pragma once
#include <loops/backend/xpu.hxx>
