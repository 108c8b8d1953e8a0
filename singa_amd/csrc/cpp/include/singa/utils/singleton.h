// Process-wide and per-thread singletons.
// API parity: reference include/singa/utils/singleton.h.
#pragma once

template <typename T>
class Singleton {
 public:
  static T *Instance() {
    static T instance;  // C++11 guarantees thread-safe initialisation
    return &instance;
  }
};

template <typename T>
class TSingleton {
 public:
  static T *Instance() {
    static thread_local T instance;
    return &instance;
  }
};
