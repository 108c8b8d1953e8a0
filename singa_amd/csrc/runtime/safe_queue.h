// Thread-safe FIFO and priority queues used by the threaded data loaders and
// the legacy threaded updater.  Parity: reference
// include/singa/utils/safe_queue.h (Push / Pop blocking, TryPop, Size).
#pragma once
#include <condition_variable>
#include <mutex>
#include <queue>
#include <vector>

namespace singa_amd {

template <typename T, class Container = std::queue<T>>
class SafeQueueBase {
 public:
  void Push(const T &v) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push(v);
    }
    cv_.notify_one();
  }
  /// Blocks until an element is available.
  void Pop(T &out) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !q_.empty(); });
    out = Front();
    q_.pop();
  }
  bool TryPop(T &out) {
    std::lock_guard<std::mutex> lk(mu_);
    if (q_.empty()) return false;
    out = Front();
    q_.pop();
    return true;
  }
  size_t Size() const {
    std::lock_guard<std::mutex> lk(mu_);
    return q_.size();
  }

 private:
  template <typename C = Container>
  auto FrontImpl(int) -> decltype(std::declval<C &>().front()) {
    return q_.front();
  }
  template <typename C = Container>
  auto FrontImpl(long) -> decltype(std::declval<C &>().top()) {
    return q_.top();
  }
  T Front() { return FrontImpl(0); }

  Container q_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
};

template <typename T>
using SafeQueue = SafeQueueBase<T, std::queue<T>>;

template <typename T, class Compare = std::less<T>>
using PriorityQueue =
    SafeQueueBase<T, std::priority_queue<T, std::vector<T>, Compare>>;

}  // namespace singa_amd
