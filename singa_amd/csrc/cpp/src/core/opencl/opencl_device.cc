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
}

void OpenclDevice::SetRandSeed(unsigned seed) {
  rt_->seed = seed;
  rt_->offset = 0;
  ctx_.seed = seed;
}

void OpenclDevice::Sync() { rt_->Finish(); }

void *OpenclDevice::Malloc(size_t size) {
  if (size == 0) return nullptr;
  void *p = rt_->Alloc(size);
  allocated_ += size;
  sizes_[p] = size;
  return p;
}

void OpenclDevice::Free(void *ptr) {
  if (!ptr) return;
  auto it = sizes_.find(ptr);
  if (it != sizes_.end()) {
    allocated_ -= it->second;
    sizes_.erase(it);
  }
  rt_->Free(ptr);
}

void OpenclDevice::CopyRaw(void *dst, const void *src, size_t n, bool, bool) {
  // shared virtual memory: one call serves every direction
  rt_->Copy(dst, src, n);
}

std::string OpenclDevice::Description() const { return rt_->description(); }

int OpenclDevice::GetNumDevices() { return ocl::CountDevices(); }

}  // namespace singa
