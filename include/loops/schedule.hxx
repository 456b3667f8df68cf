/**
 * @file schedule.hxx
 * @brief Umbrella header of the load-balancing schedules: `schedule::setup<scheme, ...>` (setup.hxx) and
 * its four specialisations.  `group_mapped` is part of the HIP build here (the reference leaves it out
 * of its HIP backend, schedule.hxx:69-74).
 */
#pragma once

#include <loops/schedule/setup.hxx>

#include <loops/schedule/thread_mapped.hxx>
#include <loops/schedule/group_mapped.hxx>
#include <loops/schedule/work_oriented.hxx>
#include <loops/schedule/merge_path_flat.hxx>
