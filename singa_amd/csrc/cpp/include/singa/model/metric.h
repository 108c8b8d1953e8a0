// Evaluation metrics.  API parity: reference include/singa/model/metric.h.
#pragma once
#include "../core/tensor.h"
#include "conf.h"
#include "conf_proto.h"

namespace singa {

class Metric {
 public:
  virtual ~Metric() {}
  virtual void Setup(const MetricConf & /*conf*/) {}
  /// From a serialized MetricConf message (reference metric.h:36-42).
  void Setup(const std::string &conf_str) {
    MetricConf conf;
    CHECK(ParseFromString(conf_str, &conf)) << "malformed MetricConf message";
    this->Setup(conf);
  }
  virtual void ToDevice(std::shared_ptr<Device> /*device*/) {}
  /// Per-sample metric value, shape (batch).
  virtual Tensor Forward(const Tensor &prediction, const Tensor &target) = 0;
  float Evaluate(const Tensor &prediction, const Tensor &target) {
    const Tensor metric = Forward(prediction, target);
    return Sum<float>(metric) / (1.0f * metric.Size());
  }
};

/// 1 when the label is among the top_k scores of the sample, else 0.
class Accuracy : public Metric {
 public:
  Accuracy() = default;
  explicit Accuracy(size_t top_k) : top_k_(top_k) {}
  void Setup(const MetricConf &conf) override { top_k_ = conf.top_k; }
  using Metric::Setup;
  Tensor Forward(const Tensor &prediction, const Tensor &target) override;

 private:
  size_t top_k_ = 1;
};

}  // namespace singa
