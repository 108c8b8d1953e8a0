// Block: reference-counted handle to a lazily allocated chunk of device or host
// memory.  The storage itself is owned by a backend-specific subclass (for the
// Python runtime it is a torch tensor drawn from the HIP caching allocator),
// the Block only tracks identity, size, the number of tensors sharing it and
// whether memory is currently attached.
//
// Behaviour parity: reference include/singa/core/common.h:64-97 (lazy
// allocation on first mutable_data(), free_data() that keeps the object alive
// so a recorded graph can re-materialise the buffer on the next replay).
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>

namespace singa_amd {

class Block {
 public:
  Block(size_t size, int64_t id) : size_(size), id_(id) {}
  virtual ~Block() = default;

  size_t size() const { return size_; }
  int64_t id() const { return id_; }

  int IncRefCount() { return ++ref_count_; }
  int DecRefCount() { return --ref_count_; }
  int ref_count() const { return ref_count_.load(); }

  /// True once storage has been attached (allocation is lazy).
  virtual bool initialized() const = 0;
  /// Drop the storage but keep the handle; the next write re-allocates.
  virtual void free_data() = 0;

 protected:
  size_t size_;
  int64_t id_;
  std::atomic<int> ref_count_{0};
};

}  // namespace singa_amd
