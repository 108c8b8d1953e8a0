// Parameter initialisers.  API parity: reference
// include/singa/model/initializer.h (Constant, Uniform, Gaussian, Xavier,
// MSRA; CreateInitializer from a FillerConf).
#pragma once
#include <cmath>
#include <memory>
#include <string>

#include "../core/tensor.h"
#include "../utils/string.h"
#include "conf.h"
#include "conf_proto.h"

namespace singa {

class Initializer {
 public:
  virtual ~Initializer() {}
  virtual void Setup(const FillerConf &conf) {}
  /// From a serialized FillerConf message (reference initializer.h:30-36).
  void Setup(const std::string &conf_str) {
    FillerConf conf;
    CHECK(ParseFromString(conf_str, &conf)) << "malformed FillerConf message";
    this->Setup(conf);
  }
  virtual void Fill(Tensor &t) = 0;
};

namespace init {

class Constant : public Initializer {
 public:
  Constant() = default;
  explicit Constant(float x) : v_(x) {}
  void Setup(const FillerConf &conf) override { v_ = conf.value; }
  using Initializer::Setup;
  void Fill(Tensor &t) override { t.SetValue(v_); }

 private:
  float v_ = 0.f;
};

class Uniform : public Initializer {
 public:
  Uniform() = default;
  Uniform(float low, float high) : min_(low), max_(high) {}
  void Setup(const FillerConf &conf) override { min_ = conf.min; max_ = conf.max; }
  using Initializer::Setup;
  void Fill(Tensor &t) override { singa::Uniform(min_, max_, &t); }

 private:
  float min_ = 0.f, max_ = 1.f;
};

class Gaussian : public Initializer {
 public:
  Gaussian() = default;
  Gaussian(float mean, float std) : mean_(mean), std_(std) {}
  void Setup(const FillerConf &conf) override { mean_ = conf.mean; std_ = conf.std; }
  using Initializer::Setup;
  void Fill(Tensor &t) override { singa::Gaussian(mean_, std_, &t); }

 private:
  float mean_ = 0.f, std_ = 1.f;
};

/// Uniform in +-sqrt(6 / (fan_in + fan_out)) for a (fan_in, fan_out) matrix.
class Xavier : public Initializer {
 public:
  void Fill(Tensor &t) override {
    CHECK_EQ(t.nDim(), 2u);
    float scale = std::sqrt(6.0f / (t.shape(0) + t.shape(1)));
    singa::Uniform(-scale, scale, &t);
  }
};

/// N(0, sqrt(2 / fan_in)) with fan_in = product of all but the first dim.
class MSRA : public Initializer {
 public:
  void Fill(Tensor &t) override {
    CHECK_GE(t.nDim(), 2u);
    float fan_in = static_cast<float>(t.Size() / t.shape(0));
    singa::Gaussian(0.0f, std::sqrt(2.0f / fan_in), &t);
  }
};

}  // namespace init

inline std::shared_ptr<Initializer> CreateInitializer(const FillerConf &conf) {
  std::shared_ptr<Initializer> init;
  std::string type = ToLowerCase(conf.type);
  if (type == "constant") init = std::make_shared<init::Constant>();
  else if (type == "uniform") init = std::make_shared<init::Uniform>();
  else if (type == "gaussian") init = std::make_shared<init::Gaussian>();
  else if (type == "xavier") init = std::make_shared<init::Xavier>();
  else if (type == "msra") init = std::make_shared<init::MSRA>();
  else LOG(FATAL) << "Unknown initialization type: " << conf.type;
  init->Setup(conf);
  return init;
}

}  // namespace singa
