/**
 * @file device.hxx
 * @brief Device selection and cached device facts (reference: include/loops/util/device.hxx:25-131).
 * One process drives one GPU (one rank per MI355X), so the caches are per-ordinal statics.
 */
#pragma once

#include <loops/backend/xpu.hxx>

namespace loops {
namespace device {

typedef int device_id_t;

inline void set(device_id_t ordinal) { (void)xpu::set_device(ordinal); }

inline device_id_t get() {
  device_id_t ordinal = 0;
  (void)xpu::get_device(&ordinal);
  return ordinal;
}

namespace detail {
constexpr int max_devices = 16;
inline int cached_attribute(xpu::device_attribute_t attr, int slot, device_id_t ordinal) {
  static int cache[4][max_devices];
  static bool valid[4][max_devices] = {};
  if (ordinal < 0 || ordinal >= max_devices) {
    int v = 0;
    (void)xpu::device_get_attribute(&v, attr, ordinal);
    return v;
  }
  if (!valid[slot][ordinal]) {
    (void)xpu::device_get_attribute(&cache[slot][ordinal], attr, ordinal);
    valid[slot][ordinal] = true;
  }
  return cache[slot][ordinal];
}
}  // namespace detail

struct properties_t {
  typedef xpu::device_properties_t device_properties_t;
  device_properties_t properties;
  device_id_t ordinal;
  properties_t() : ordinal(device::get()) { (void)xpu::get_device_properties(&properties, ordinal); }
  int multi_processor_count() { return properties.multiProcessorCount; }
};

/// Compute units on the device (256 on MI355X).
inline int multi_processor_count(device_id_t ordinal = device::get()) {
  return detail::cached_attribute(xpu::attr_multiprocessor_count, 0, ordinal);
}

inline int max_grid_dim_x(device_id_t ordinal = device::get()) {
  return detail::cached_attribute(xpu::attr_max_grid_dim_x, 1, ordinal);
}

inline int compute_capability(device_id_t ordinal = device::get()) {
  return detail::cached_attribute(xpu::attr_compute_capability_major, 2, ordinal) * 10 +
         detail::cached_attribute(xpu::attr_compute_capability_minor, 3, ordinal);
}

}  // namespace device
}  // namespace loops
