/**
 * @file timer.hxx
 * @brief `util::timer_t`: a hipEvent pair (reference: include/loops/util/timer.hxx:19-51).
 * Unlike the reference (which always records on stream 0) the events are recorded
 * on the stream the timed work is launched on; the default stays stream 0 so existing callers
 * (`timer.start(); ...; timer.stop();`) behave identically.
 */
#pragma once

#include <loops/backend/xpu.hxx>

namespace loops {
namespace util {

struct timer_t {
  float time = 0.0f;

  explicit timer_t(xpu::stream_t stream = 0) : stream_(stream) {
    (void)xpu::event_create(&start_);
    (void)xpu::event_create(&stop_);
    (void)xpu::event_record(start_, stream_);
  }
  timer_t(const timer_t& rhs) : time(rhs.time), stream_(rhs.stream_) {
    (void)xpu::event_create(&start_);
    (void)xpu::event_create(&stop_);
  }
  timer_t& operator=(const timer_t& rhs) {
    time = rhs.time;
    return *this;
  }
  ~timer_t() {
    (void)xpu::event_destroy(start_);
    (void)xpu::event_destroy(stop_);
  }

  void begin() { (void)xpu::event_record(start_, stream_); }
  void start() { begin(); }

  float end() {
    (void)xpu::event_record(stop_, stream_);
    (void)xpu::event_synchronize(stop_);
    (void)xpu::event_elapsed_time(&time, start_, stop_);
    return milliseconds();
  }
  float stop() { return end(); }

  float seconds() { return time * 1e-3f; }
  float milliseconds() { return time; }

 private:
  xpu::stream_t stream_;
  xpu::event_t start_, stop_;
};

}  // namespace util
}  // namespace loops
