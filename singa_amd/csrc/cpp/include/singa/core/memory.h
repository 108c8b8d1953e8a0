// Device memory pools.
//
// HipMemPool is a caching sub-allocator: requests are rounded up to a size
// class (512 B granularity below 1 MiB, 2 MiB granularity above), served from
// per-class free lists and only reach hipMalloc on a miss.  Freed chunks go
// back to their list, never to the driver, until Release() -- with 288 GB of
// HBM per MI355X holding on to memory is the right trade and it keeps
// allocation off the critical path of an iteration (hipMalloc / hipFree
// synchronise the device).  HipDirectPool forwards every call to the driver.
//
// API parity: reference include/singa/core/memory.h:37-88 (DeviceMemPool,
// CnMemPool, CudaMemPool; GetMemUsage).
#pragma once
#include <cstddef>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../model/conf.h"

namespace singa {

class DeviceMemPool {
 public:
  virtual ~DeviceMemPool() {}
  virtual void Malloc(void **ptr, size_t size) = 0;
  virtual void Free(void *ptr) = 0;
  /// (free bytes, total bytes) of the device the pool serves.
  virtual std::pair<size_t, size_t> GetMemUsage() { return {0, 0}; }
  virtual std::pair<size_t, size_t> GetMemUsage(int /*id*/) {
    return GetMemUsage();
  }
};

class HipMemPool : public DeviceMemPool {
 public:
  explicit HipMemPool(const MemPoolConf &conf = MemPoolConf()) : conf_(conf) {}
  ~HipMemPool() override { Release(); }
  void Malloc(void **ptr, size_t size) override;
  void Free(void *ptr) override;
  std::pair<size_t, size_t> GetMemUsage() override;
  std::pair<size_t, size_t> GetMemUsage(int id) override;
  /// Return every cached chunk to the driver.
  void Release();
  /// Bytes handed out and not yet freed / bytes held in free lists.
  size_t bytes_in_use() const { return in_use_; }
  size_t bytes_cached() const { return cached_; }
  size_t driver_calls() const { return driver_calls_; }
  static size_t RoundUp(size_t size);

 private:
  MemPoolConf conf_;
  std::mutex mtx_;
  // (device, class size) -> free chunks
  std::map<std::pair<int, size_t>, std::vector<void *>> free_;
  std::unordered_map<void *, std::pair<int, size_t>> owner_;
  size_t in_use_ = 0, cached_ = 0, driver_calls_ = 0;
};

class HipDirectPool : public DeviceMemPool {
 public:
  void Malloc(void **ptr, size_t size) override;
  void Free(void *ptr) override;
  std::pair<size_t, size_t> GetMemUsage() override;
};

/// Host pool with the same caching policy (used by CppCPU; 64-byte aligned).
class HostMemPool : public DeviceMemPool {
 public:
  ~HostMemPool() override;
  void Malloc(void **ptr, size_t size) override;
  void Free(void *ptr) override;

 private:
  std::mutex mtx_;
  std::map<size_t, std::vector<void *>> free_;
  std::unordered_map<void *, size_t> owner_;
};

}  // namespace singa
