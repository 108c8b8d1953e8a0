// Loss functions: Forward returns the per-sample loss and caches what
// Backward needs; Evaluate averages over the batch.  API parity: reference
// include/singa/model/loss.h (Loss, MSE, SoftmaxCrossEntropy).
#pragma once
#include <stack>

#include "../core/tensor.h"
#include "conf.h"

namespace singa {

class Loss {
 public:
  virtual ~Loss() {}
  virtual void Setup(const std::string & /*conf*/) {}
  virtual void ToDevice(std::shared_ptr<Device> /*device*/) {}
  /// Per-sample loss, shape (batch).
  virtual Tensor Forward(int flag, const Tensor &prediction,
                         const Tensor &target) = 0;
  /// Gradient of the (per-sample) loss w.r.t. the prediction.
  virtual Tensor Backward() = 0;
  /// Mean loss of the batch.
  float Evaluate(int flag, const Tensor &prediction, const Tensor &target) {
    Tensor loss = Forward(flag, prediction, target);
    return Sum<float>(loss) / (1.0f * loss.Size());
  }
};

/// 0.5 * ||prediction - target||^2 per sample.
class MSE : public Loss {
 public:
  Tensor Forward(int flag, const Tensor &prediction,
                 const Tensor &target) override;
  Tensor Backward() override;

 private:
  std::stack<Tensor> buf_;
};

/// Softmax over the prediction followed by cross entropy against int labels
/// (shape (batch)) -- Backward is p - onehot(t).
class SoftmaxCrossEntropy : public Loss {
 public:
  Tensor Forward(int flag, const Tensor &prediction,
                 const Tensor &target) override;
  Tensor Backward() override;

 private:
  std::stack<Tensor> buf_;
};

}  // namespace singa
