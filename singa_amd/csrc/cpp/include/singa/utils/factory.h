// String- or integer-keyed object factory with static self-registration.
// API parity: reference include/singa/utils/factory.h (Register / Create,
// Registra helper used by RegisterLayerClass).
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "logging.h"

template <typename T, typename ID = std::string>
class Factory {
 public:
  void Register(const ID &id, const std::function<T *(void)> &creator) {
    Registry()[id] = creator;
  }
  T *Create(const ID &id) {
    auto &reg = Registry();
    auto it = reg.find(id);
    CHECK(it != reg.end()) << "the creation function for " << id
                           << " has not been registered";
    return it->second();
  }
  static std::vector<ID> GetIDs() {
    std::vector<ID> ids;
    for (const auto &kv : Registry()) ids.push_back(kv.first);
    return ids;
  }

 private:
  static std::map<ID, std::function<T *(void)>> &Registry() {
    static std::map<ID, std::function<T *(void)>> reg;
    return reg;
  }
};

template <typename Base, typename Sub, typename ID = std::string>
class Registra {
 public:
  explicit Registra(const ID &id) {
    Factory<Base, ID>().Register(id, [] { return static_cast<Base *>(new Sub); });
  }
};
