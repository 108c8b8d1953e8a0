// Basic types of the C++ API: data / language enums and the memory Block.
//
// A Block is a reference-counted handle to `size` bytes on one device.  The
// bytes are drawn from the device's pool on the first mutable_data() call
// (lazy allocation) and may be dropped with free_data() while the handle stays
// valid, which is what lets a recorded graph release intermediates between
// replays.  API parity: reference include/singa/core/common.h:64-128.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "../../../../runtime/block.h"
#include "../utils/logging.h"

namespace singa {

enum DataType { kFloat32 = 0, kFloat16 = 1, kInt = 2, kChar = 3, kDouble = 4,
                kUChar = 5, kNumDataType = 6 };
enum LangType { kCpp = 0, kCuda = 1, kOpencl = 2, kHip = 1, kNumDeviceType = 4 };

inline size_t SizeOf(DataType t) {
  static const size_t kSize[kNumDataType] = {4, 2, 4, 1, 8, 1};
  return kSize[t];
}

using Shape = std::vector<size_t>;
inline size_t Product(const Shape &shape, int start = 0, size_t len = 0) {
  if (len == 0) len = shape.size();
  if (len == 0) return 0;
  size_t v = 1;
  for (size_t i = start; i < len; ++i) v *= shape[i];
  return v;
}

class Device;

class Block : public ::singa_amd::Block {
 public:
  Block(Device *dev, size_t size, int64_t id)
      : ::singa_amd::Block(size, id), dev_(dev) {}
  ~Block() override;
  /// Pointer for writing; allocates on first use.
  void *mutable_data();
  /// Pointer for reading; the block must have been written before.
  const void *data() const;
  bool initialized() const override { return data_ != nullptr; }
  void free_data() override;
  Device *device() const { return dev_; }

 private:
  Device *dev_;
  void *data_ = nullptr;
};

/// Per-device execution context handed to every operation closure.
struct Context {
  void *stream = nullptr;   // hipStream_t of the compute queue (HIP devices)
  int device_id = -1;
  uint64_t seed = 0;
  int64_t *rng_state = nullptr;  // device int64[2] {seed, offset} (HIP)
  void *cl = nullptr;            // ocl::Runtime of an OpenclDevice, else null
};

}  // namespace singa
