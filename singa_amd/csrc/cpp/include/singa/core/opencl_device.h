// OpenclDevice: the MI355X driven through ROCm's OpenCL runtime instead of HIP.
//
// The device exists for programs written against the reference's OpenCL
// device (include/singa/core/device.h:204-277, src/core/device/
// opencl_device.cc:37-148); the HIP device is the fast path of this library.
// Design points that differ from the reference's ViennaCL-based device:
//   * libOpenCL is loaded at run time (dlopen), so the library builds and
//     loads on machines without an OpenCL runtime; creating the device there
//     fails with a message instead.
//   * memory is coarse-grained shared virtual memory (clSVMAlloc): a block is
//     a plain pointer, as on the other devices, so Tensor offsets, sub-block
//     views and the lazy Block work unchanged, and kernels take pointers
//     (clSetKernelArgSVMPointer) rather than buffer handles.
//   * one in-order queue per device; Sync() is clFinish.  Freed blocks are
//     cached and recycled (releasing SVM needs an idle queue, reuse does not).
//   * kernel sources (src/core/opencl/*.cl) are embedded in the library at
//     build time and compiled once per device on first use.
// Like the reference's, this device computes in float32 (int32 labels) and
// has no graph buffering.  Every layer of the C++ API runs on it.
#pragma once
#include <map>
#include <memory>
#include <string>

#include "device.h"

namespace singa {

namespace ocl { class Runtime; }

class OpenclDevice : public Device {
 public:
  explicit OpenclDevice(int id = 0);
  ~OpenclDevice() override;
  void SetRandSeed(unsigned seed) override;
  void Sync() override;
  size_t GetAllocatedMem() override { return allocated_; }
  void *Malloc(size_t size) override;
  void Free(void *ptr) override;
  void CopyRaw(void *dst, const void *src, size_t n, bool dst_on_dev,
               bool src_on_dev) override;
  /// "<platform> / <device>, <n> compute units, OpenCL C <version>"
  std::string Description() const;
  ocl::Runtime *runtime() const { return rt_.get(); }

  /// Number of GPU devices the OpenCL platform reports (0 without a runtime).
  static int GetNumDevices();

 private:
  void ReleaseCache();
  static constexpr size_t kCacheLimit = size_t(8) << 30;  // bytes kept for reuse

  std::unique_ptr<ocl::Runtime> rt_;
  size_t allocated_ = 0;                  // bytes in use, as requested
  size_t cached_ = 0;                     // bytes in free_
  std::map<void *, size_t> sizes_;        // live block -> its capacity
  std::map<void *, size_t> requested_;    // live block -> bytes asked for
  std::multimap<size_t, void *> free_;    // capacity -> block kept for reuse
};

}  // namespace singa
