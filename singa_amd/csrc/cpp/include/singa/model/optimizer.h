// Optimizers of the C++ model API with per-parameter learning-rate / decay
// multipliers, an optional L2 regularizer and an optional L2-norm constraint.
// API parity: reference include/singa/model/optimizer.h:41-255 (Optimizer,
// Constraint, Regularizer, SGD, Nesterov, AdaGrad, RMSProp).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>

#include "../core/tensor.h"
#include "conf.h"
#include "conf_proto.h"

namespace singa {

class Constraint {
 public:
  Constraint() = default;
  Constraint(const std::string &type, float threshold)
      : type_(type), threshold_(threshold) {}
  void Setup(const ConstraintConf &conf) {
    type_ = conf.type;
    threshold_ = conf.threshold;
  }
  /// From a serialized ConstraintConf message (reference optimizer.h:110-118).
  void Setup(const std::string &conf_str) {
    ConstraintConf conf;
    CHECK(ParseFromString(conf_str, &conf)) << "malformed ConstraintConf";
    Setup(conf);
  }
  /// Rescale `grad` so that its L2 norm does not exceed the threshold.
  void Apply(int step, Tensor &grad, Tensor &value);

 private:
  std::string type_ = "l2";
  float threshold_ = 0.f;
};
/// Every type name gives the one constraint class (as in the reference,
/// include/singa/model/optimizer.h:137-139).
inline std::shared_ptr<Constraint> CreateConstraint(std::string /*type*/) {
  return std::make_shared<Constraint>();
}

class Regularizer {
 public:
  Regularizer() = default;
  Regularizer(const std::string &type, float coefficient)
      : type_(type), coefficient_(coefficient) {}
  void Setup(const RegularizerConf &conf) {
    type_ = conf.type;
    coefficient_ = conf.coefficient;
  }
  /// From a serialized RegularizerConf message (reference optimizer.h:146-154).
  void Setup(const std::string &conf_str) {
    RegularizerConf conf;
    CHECK(ParseFromString(conf_str, &conf)) << "malformed RegularizerConf";
    Setup(conf);
  }
  /// grad += coefficient * value
  void Apply(int step, Tensor &grad, Tensor &value, float scale = 1.0f);

 private:
  std::string type_ = "l2";
  float coefficient_ = 0.f;
};
/// Every type name gives the one regularizer class (as in the reference,
/// include/singa/model/optimizer.h:171-173).
inline std::shared_ptr<Regularizer> CreateRegularizer(std::string /*type*/) {
  return std::make_shared<Regularizer>();
}

class Optimizer {
 public:
  Optimizer() = default;
  virtual ~Optimizer();
  virtual void Setup(const OptimizerConf &conf);
  /// From a serialized OptimizerConf message, the form binding languages
  /// pass (reference optimizer.h:44-52).
  void Setup(const std::string &str) {
    OptimizerConf conf;
    CHECK(ParseFromString(str, &conf)) << "malformed OptimizerConf message";
    this->Setup(conf);
  }
  /// Record the multipliers / regularizer / constraint of one parameter.
  virtual void Register(const std::string &name, const ParamSpec &specs);
  virtual void ApplyRegularizerConstraint(int epoch, const std::string &name,
                                          const Tensor &value, Tensor &grad,
                                          int step = -1);
  /// Update `value` in place from `grad` with learning rate `lr`.
  virtual void Apply(int epoch, float lr, const std::string &name, Tensor &grad,
                     Tensor &value, int step = -1) = 0;
  /// Same with lr = generator(epoch) * lr multiplier of the parameter.
  void Apply(int epoch, const std::string &name, Tensor &grad, Tensor &value,
             int step = -1);
  void SetLearningRateGenerator(std::function<float(int)> func) {
    learning_rate_generator_ = func;
  }
  float GetLearningRate(int step) {
    if (learning_rate_generator_) return learning_rate_generator_(step);
    return 0;
  }

 protected:
  std::function<float(int)> learning_rate_generator_;
  std::unordered_map<std::string, float> learning_rate_multplier_;
  std::unordered_map<std::string, float> weight_decay_multplier_;
  std::unordered_map<std::string, Constraint *> constraints_;
  std::unordered_map<std::string, Regularizer *> regularizers_;
  Constraint *constraint_ = nullptr;
  Regularizer *regularizer_ = nullptr;
  OptimizerConf conf_;
};

class SGD : public Optimizer {
 public:
  using Optimizer::Apply;
  void Setup(const OptimizerConf &conf) override;
  using Optimizer::Setup;
  void Apply(int epoch, float lr, const std::string &name, Tensor &grad,
             Tensor &value, int step = -1) override;
  void SetMomentumGenerator(std::function<float(int)> func) {
    momentum_generator_ = func;
  }

 private:
  std::unordered_map<std::string, Tensor> history_gradient_;
  std::function<float(int)> momentum_generator_;
};

class Nesterov : public Optimizer {
 public:
  using Optimizer::Apply;
  void Setup(const OptimizerConf &conf) override;
  using Optimizer::Setup;
  void Apply(int epoch, float lr, const std::string &name, Tensor &grad,
             Tensor &value, int step = -1) override;
  void SetMomentumGenerator(std::function<float(int)> func) {
    momentum_generator_ = func;
  }

 private:
  std::unordered_map<std::string, Tensor> history_gradient_;
  std::function<float(int)> momentum_generator_;
};

class AdaGrad : public Optimizer {
 public:
  using Optimizer::Apply;
  void Setup(const OptimizerConf &conf) override;
  using Optimizer::Setup;
  void Apply(int epoch, float lr, const std::string &name, Tensor &grad,
             Tensor &value, int step = -1) override;

 private:
  std::unordered_map<std::string, Tensor> history_gradient_;
  float delta_ = 1e-8f;
};

class RMSProp : public Optimizer {
 public:
  using Optimizer::Apply;
  void Setup(const OptimizerConf &conf) override;
  using Optimizer::Setup;
  void Apply(int epoch, float lr, const std::string &name, Tensor &grad,
             Tensor &value, int step = -1) override;

 private:
  std::unordered_map<std::string, Tensor> history_gradient_;
  float delta_ = 1e-8f, rho_ = 0.9f;
};

std::shared_ptr<Optimizer> CreateOptimizer(const std::string &type);

}  // namespace singa
