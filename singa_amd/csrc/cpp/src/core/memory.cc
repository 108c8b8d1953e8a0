#include "singa/core/memory.h"

#include <hip/hip_runtime_api.h>

#include <cstdlib>

#include "singa/utils/logging.h"

#define HIP_CHECK(expr)                                                  \
  do {                                                                   \
    hipError_t e_ = (expr);                                              \
    CHECK(e_ == hipSuccess) << "HIP error: " << hipGetErrorString(e_);   \
  } while (0)

namespace singa {

size_t HipMemPool::RoundUp(size_t size) {
  const size_t kSmall = 512, kLarge = size_t(2) << 20;
  if (size == 0) size = 1;
  if (size < (size_t(1) << 20)) return (size + kSmall - 1) / kSmall * kSmall;
  return (size + kLarge - 1) / kLarge * kLarge;
}

void HipMemPool::Malloc(void **ptr, size_t size) {
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  size_t cls = RoundUp(size);
  std::lock_guard<std::mutex> lock(mtx_);
  auto &list = free_[{dev, cls}];
  if (!list.empty()) {
    *ptr = list.back();
    list.pop_back();
    cached_ -= cls;
  } else {
    if (conf_.max_size_mb)
      CHECK_LE(in_use_ + cached_ + cls, conf_.max_size_mb << 20)
          << "memory pool limit exceeded";
    HIP_CHECK(hipMalloc(ptr, cls));
    ++driver_calls_;
    owner_[*ptr] = {dev, cls};
  }
  in_use_ += cls;
}

void HipMemPool::Free(void *ptr) {
  if (ptr == nullptr) return;
  std::lock_guard<std::mutex> lock(mtx_);
  auto it = owner_.find(ptr);
  CHECK(it != owner_.end()) << "pointer does not belong to this pool";
  free_[it->second].push_back(ptr);
  in_use_ -= it->second.second;
  cached_ += it->second.second;
}

void HipMemPool::Release() {
  std::lock_guard<std::mutex> lock(mtx_);
  for (auto &kv : free_) {
    for (void *p : kv.second) {
      owner_.erase(p);
      (void)hipFree(p);  // the runtime may already be shutting down
    }
    kv.second.clear();
  }
  cached_ = 0;
}

std::pair<size_t, size_t> HipMemPool::GetMemUsage() {
  size_t free_b = 0, total = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total));
  // cached chunks are available to this process
  return {free_b + cached_, total};
}

std::pair<size_t, size_t> HipMemPool::GetMemUsage(int id) {
  int cur = 0;
  HIP_CHECK(hipGetDevice(&cur));
  HIP_CHECK(hipSetDevice(id));
  auto r = GetMemUsage();
  HIP_CHECK(hipSetDevice(cur));
  return r;
}

void HipDirectPool::Malloc(void **ptr, size_t size) {
  HIP_CHECK(hipMalloc(ptr, size ? size : 1));
}
void HipDirectPool::Free(void *ptr) {
  if (ptr) HIP_CHECK(hipFree(ptr));
}
std::pair<size_t, size_t> HipDirectPool::GetMemUsage() {
  size_t free_b = 0, total = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total));
  return {free_b, total};
}

HostMemPool::~HostMemPool() {
  for (auto &kv : free_)
    for (void *p : kv.second) std::free(p);
}

void HostMemPool::Malloc(void **ptr, size_t size) {
  size_t cls = HipMemPool::RoundUp(size);
  std::lock_guard<std::mutex> lock(mtx_);
  auto &list = free_[cls];
  if (!list.empty()) {
    *ptr = list.back();
    list.pop_back();
    return;
  }
  void *p = nullptr;
  CHECK_EQ(posix_memalign(&p, 64, cls), 0) << "host allocation failed";
  owner_[p] = cls;
  *ptr = p;
}

void HostMemPool::Free(void *ptr) {
  if (!ptr) return;
  std::lock_guard<std::mutex> lock(mtx_);
  auto it = owner_.find(ptr);
  CHECK(it != owner_.end());
  free_[it->second].push_back(ptr);
}

}  // namespace singa
