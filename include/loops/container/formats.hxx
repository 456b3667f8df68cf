/**
 * @file formats.hxx
 * @brief Umbrella header of the sparse containers (coo, csr, csc, ell, bcsr<R, C>, dia).  They refer to
 * each other in their converting constructors, so all six are declared first (fwd.hxx) and defined after.
 */
#pragma once

#include <loops/container/fwd.hxx>

#include <loops/container/coo.hxx>
#include <loops/container/csr.hxx>
#include <loops/container/csc.hxx>
#include <loops/container/ell.hxx>
#include <loops/container/dia.hxx>
#include <loops/container/bcsr.hxx>
