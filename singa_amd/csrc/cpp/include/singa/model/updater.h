// Updaters decide where and when the optimizer runs.  `Updater` applies it on
// the spot.  `LocalUpdater` serves N model replicas that train in N threads of
// one process (one replica per GPU): every replica hands in its gradient for a
// parameter, the last one to arrive sums the partial gradients on the
// aggregation device, runs the optimizer once, and all replicas copy the new
// value back.
// API parity: reference include/singa/model/updater.h:61-94,
// src/model/updater/local_updater.cc:24-75.
#pragma once
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "optimizer.h"

namespace singa {

class Updater {
 public:
  explicit Updater(Optimizer *opt) : opt_(opt) {}
  virtual ~Updater() {}
  virtual void Setup(const OptimizerConf &conf) { opt_->Setup(conf); }
  virtual void Register(const std::string &name, const ParamSpec &specs) {
    opt_->Register(name, specs);
  }
  virtual void Apply(int step, const std::string &name, Tensor &grad,
                     Tensor &value) {
    opt_->Apply(step, name, grad, value);
  }
  Optimizer *GetOptimizer() { return opt_; }

 protected:
  Optimizer *opt_;
};

class LocalUpdater : public Updater {
 public:
  LocalUpdater(int total_num, Optimizer *opt,
               std::shared_ptr<Device> dev = defaultDevice)
      : Updater(opt), total_num_(total_num), dev_(dev) {}
  void Register(const std::string &name, const ParamSpec &specs) override;
  void Apply(int step, const std::string &name, Tensor &grad,
             Tensor &value) override;

 private:
  struct Slot {
    std::mutex mtx;
    std::condition_variable cv;
    int arrived = 0;      // gradients handed in this round
    int to_leave = 0;     // replicas that still have to copy the result out
    uint64_t round = 0;
    std::vector<Tensor> partial;
    Tensor sum, value;
    bool value_ready = false;
  };
  int total_num_;
  std::shared_ptr<Device> dev_;
  std::mutex reg_mtx_;
  std::unordered_map<std::string, std::unique_ptr<Slot>> slots_;
};

}  // namespace singa
