// Wall-clock stopwatch.  Parity: reference include/singa/utils/timer.h
// (Tick() restarts, Elapsed<T>() reports time since the last tick).
#pragma once
#include <chrono>

namespace singa_amd {

class Timer {
 public:
  Timer() { Tick(); }
  void Tick() { last_ = std::chrono::high_resolution_clock::now(); }
  double ElapsedMs() const {
    return std::chrono::duration<double, std::milli>(
               std::chrono::high_resolution_clock::now() - last_)
        .count();
  }
  double ElapsedSec() const { return ElapsedMs() / 1000.0; }

 private:
  std::chrono::high_resolution_clock::time_point last_;
};

}  // namespace singa_amd
