#include "singa/core/opencl_device.h"

#include "runtime.h"

namespace singa {

OpenclDevice::OpenclDevice(int id) : Device(id, 1) {
  lang_ = kOpencl;
  host_ = defaultDevice;
  rt_.reset(new ocl::Runtime(id));
  ctx_.device_id = id;
  ctx_.cl = rt_.get();
}

OpenclDevice::~OpenclDevice() {
  // blocks still held by a recorded graph go back through Free(), which
  // needs the runtime: release them while it is alive
  ResetGraph();
  ReleaseCache();
}

void OpenclDevice::ReleaseCache() {
  rt_->Finish();
  for (auto &kv : free_) rt_->Free(kv.second);
  free_.clear();
  cached_ = 0;
}

void OpenclDevice::SetRandSeed(unsigned seed) {
  rt_->seed = seed;
  rt_->offset = 0;
  ctx_.seed = seed;
}

void OpenclDevice::Sync() { rt_->Finish(); }

// Freed blocks are kept and handed out again: releasing shared virtual
// memory needs the queue to be idle (a kernel in flight may still use it),
// while a recycled block is safe at once -- the queue is in order, so whatever
// touches it next runs after whatever touched it last.
void *OpenclDevice::Malloc(size_t size) {
  if (size == 0) return nullptr;
  const size_t want = (size + 255) / 256 * 256;
  void *p = nullptr;
  auto it = free_.lower_bound(want);
  if (it != free_.end() && it->first <= 2 * want) {
    p = it->second;
    cached_ -= it->first;
    sizes_[p] = it->first;
    free_.erase(it);
  } else {
    p = rt_->Alloc(want);
    sizes_[p] = want;
  }
  allocated_ += size;
  requested_[p] = size;
  return p;
}

void OpenclDevice::Free(void *ptr) {
  if (!ptr) return;
  auto it = sizes_.find(ptr);
  CHECK(it != sizes_.end()) << "block was not allocated by this device";
  allocated_ -= requested_[ptr];
  requested_.erase(ptr);
  free_.emplace(it->second, ptr);
  cached_ += it->second;
  sizes_.erase(it);
  if (cached_ > kCacheLimit) ReleaseCache();
}

void OpenclDevice::CopyRaw(void *dst, const void *src, size_t n, bool, bool) {
  // shared virtual memory: one call serves every direction
  rt_->Copy(dst, src, n);
}

std::string OpenclDevice::Description() const { return rt_->description(); }

int OpenclDevice::GetNumDevices() { return ocl::CountDevices(); }

}  // namespace singa
