// Wall-clock stopwatch of the C++ API.
// API parity: reference include/singa/utils/timer.h (Tick / Elapsed).
#pragma once
#include "../../../../runtime/timer.h"

namespace singa {

class Timer : public ::singa_amd::Timer {
 public:
  /// milliseconds since the last Tick()
  double Elapsed() const { return ElapsedMs(); }
};

}  // namespace singa
