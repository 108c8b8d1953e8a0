// Layer base class and string-keyed factory of the C++ model API.
//
// A layer is configured by Setup(in_sample_shape, conf), then Forward(flag, x)
// caches what Backward(flag, dy) needs; Backward returns dx and the parameter
// gradients in the order of param_values().  Layer classes register under
// "singacpp_<type>" and "singacuda_<type>" (the same implementation serves the
// host and the MI355X: operations dispatch on the tensor's device) plus the
// "cudnn_<type>" names that programs written for the reference request.
//
// API parity: reference include/singa/model/layer.h:37-250.
#pragma once
#include <memory>
#include <stack>
#include <string>
#include <utility>
#include <vector>

#include "../core/tensor.h"
#include "../utils/factory.h"
#include "conf.h"
#include "conf_proto.h"

namespace singa {

class Layer {
 public:
  Layer() = default;
  virtual ~Layer() {}
  virtual const std::string layer_type() const { return "Unknown"; }

  /// Configure from the shape of ONE input sample (no batch dimension).
  virtual void Setup(const Shape &in_sample, const LayerConf &conf) {
    name_ = conf.name;
    param_specs_ = conf.param;
    conf_ = conf;
  }
  virtual void Setup(const std::vector<Shape> &in_samples,
                     const LayerConf &conf) {
    Setup(in_samples.at(0), conf);
  }
  /// Configure from a serialized LayerConf message (protobuf wire format,
  /// what a binding language's SerializeToString produces; reference
  /// layer.h:40-54).  Subclasses that override Setup re-expose these with
  /// `using Layer::Setup;` or are called through a Layer pointer.
  void Setup(const Shape &in_sample, const std::string &proto_str) {
    LayerConf conf;
    CHECK(ParseFromString(proto_str, &conf)) << "malformed LayerConf message";
    this->Setup(in_sample, conf);
  }
  void Setup(const std::vector<Shape> &in_samples,
             const std::string &proto_str) {
    LayerConf conf;
    CHECK(ParseFromString(proto_str, &conf)) << "malformed LayerConf message";
    this->Setup(in_samples, conf);
  }
  /// The configuration the layer was set up with, with the parameter specs
  /// as they are now (reference layer.h:160-186).
  virtual void ToProto(LayerConf *conf) const {
    *conf = conf_;
    conf->name = name_;
    conf->param = param_specs_;
  }
  std::string ToProtoStr() const {
    LayerConf conf;
    ToProto(&conf);
    return SerializeAsString(conf);
  }
  virtual const Shape GetOutputSampleShape() const {
    LOG(FATAL) << "GetOutputSampleShape not implemented for " << layer_type();
    return Shape{};
  }
  virtual const Shape GetOutputSampleShape(int /*k*/) {
    return GetOutputSampleShape();
  }

  virtual const Tensor Forward(int flag, const Tensor &input) {
    LOG(FATAL) << "Forward(Tensor) not implemented for " << layer_type();
    return Tensor();
  }
  virtual const std::vector<Tensor> Forward(int flag,
                                            const std::vector<Tensor> &inputs) {
    std::vector<Tensor> out;
    for (const auto &t : inputs) out.push_back(Forward(flag, t));
    return out;
  }
  /// -> (gradient w.r.t. the input, gradients of param_values())
  virtual const std::pair<Tensor, std::vector<Tensor>> Backward(
      int flag, const Tensor &grad) {
    LOG(FATAL) << "Backward(Tensor) not implemented for " << layer_type();
    return {Tensor(), {}};
  }
  virtual const std::pair<std::vector<Tensor>, std::vector<Tensor>> Backward(
      int flag, const std::vector<Tensor> &grads) {
    std::vector<Tensor> dx, dp;
    for (const auto &g : grads) {
      auto r = Backward(flag, g);
      dx.push_back(r.first);
      for (const auto &p : r.second) dp.push_back(p);
    }
    return {dx, dp};
  }

  virtual void ToDevice(std::shared_ptr<Device> device) {
    for (Tensor *p : param_ptrs()) p->ToDevice(device);
  }
  virtual void AsType(DataType dtype) {
    for (Tensor *p : param_ptrs()) p->ToType(dtype);
  }

  virtual const std::vector<Tensor> param_values() {
    std::vector<Tensor> v;
    for (Tensor *p : param_ptrs()) v.push_back(*p);
    return v;
  }
  const std::vector<ParamSpec> param_specs() const { return param_specs_; }
  const ParamSpec &param_specs(size_t i) const { return param_specs_.at(i); }
  const std::vector<std::string> param_names() const {
    std::vector<std::string> n;
    for (const auto &s : param_specs_) n.push_back(s.name);
    return n;
  }
  const std::string &param_name(size_t i) const { return param_specs_.at(i).name; }
  const std::string name() const { return name_; }

 protected:
  /// Pointers to the layer's parameter tensors, in a fixed order.
  virtual std::vector<Tensor *> param_ptrs() { return {}; }
  /// Give every parameter a ParamSpec (named "<layer>_<suffix>" by default).
  void EnsureParamSpecs(const std::vector<std::string> &suffixes);
  std::string name_;
  std::vector<ParamSpec> param_specs_;
  LayerConf conf_;
};

std::shared_ptr<Layer> CreateLayer(const std::string type);
const std::vector<std::string> GetRegisteredLayers();

#define RegisterLayerClass(Name, SubLayer) \
  static Registra<Layer, SubLayer> Name##SubLayer(#Name);

}  // namespace singa
