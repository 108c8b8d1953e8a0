// Concrete layers of the C++ model API.  One implementation per layer type
// serves CppCPU and HipGPU (the tensor operations dispatch on the device).
// Parity: reference src/model/layer/*.{h,cc} (Activation, BatchNorm, Concat,
// Convolution, Dense, Dropout, Flatten, LRN, Merge, Pooling, PReLU, Slice,
// Softmax, Split and their cudnn_* counterparts).
#pragma once
#include <stack>

#include "layer.h"
#include "nn_ops.h"

namespace singa {

class Dense : public Layer {
 public:
  const std::string layer_type() const override { return "Dense"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return Shape{hdim_}; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;
  const Tensor &weight() const { return weight_; }
  const Tensor &bias() const { return bias_; }
  void set_weight(const Tensor &w) { weight_.ResetLike(w); weight_.CopyData(w); }
  void set_bias(const Tensor &b) { bias_.ResetLike(b); bias_.CopyData(b); }

 protected:
  std::vector<Tensor *> param_ptrs() override {
    return bias_term_ ? std::vector<Tensor *>{&weight_, &bias_}
                      : std::vector<Tensor *>{&weight_};
  }
  size_t vdim_ = 0, hdim_ = 0;
  bool transpose_ = false, bias_term_ = true;
  Tensor weight_, bias_;
  std::stack<Tensor> buf_;
};

class Activation : public Layer {
 public:
  const std::string layer_type() const override { return "Activation"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;
  const std::string Mode() const { return mode_; }

 protected:
  std::string mode_;  // RELU, SIGMOID, TANH
  float neg_slope_ = 0.f;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

class Convolution : public Layer {
 public:
  const std::string layer_type() const override { return "Convolution"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;
  const Tensor &weight() const { return weight_; }
  const Tensor &bias() const { return bias_; }
  void set_weight(const Tensor &w) { weight_.ResetLike(w); weight_.CopyData(w); }
  void set_bias(const Tensor &b) { bias_.ResetLike(b); bias_.CopyData(b); }

 protected:
  std::vector<Tensor *> param_ptrs() override {
    return bias_term_ ? std::vector<Tensor *>{&weight_, &bias_}
                      : std::vector<Tensor *>{&weight_};
  }
  ConvParams p_;
  size_t channels_ = 0, height_ = 0, width_ = 0;
  bool bias_term_ = true;
  Tensor weight_, bias_;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

class Pooling : public Layer {
 public:
  const std::string layer_type() const override { return "Pooling"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;

 protected:
  PoolParams p_;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

class BatchNorm : public Layer {
 public:
  const std::string layer_type() const override { return "BatchNorm"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;
  const Tensor &running_mean() const { return runningMean_; }
  const Tensor &running_variance() const { return runningVariance_; }
  const Tensor &bnScale() const { return bnScale_; }
  const Tensor &bnBias() const { return bnBias_; }
  void set_bnScale(const Tensor &x) { bnScale_.CopyData(x); }
  void set_bnBias(const Tensor &x) { bnBias_.CopyData(x); }
  void set_runningMean(const Tensor &x) { runningMean_.CopyData(x); }
  void set_runningVariance(const Tensor &x) { runningVariance_.CopyData(x); }

 protected:
  // four "parameters" as in the reference: scale, bias and the running
  // statistics (the latter get zero gradients)
  std::vector<Tensor *> param_ptrs() override {
    return {&bnScale_, &bnBias_, &runningMean_, &runningVariance_};
  }
  float factor_ = 0.9f;
  size_t channels_ = 0;
  Tensor bnScale_, bnBias_, runningMean_, runningVariance_;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

class Dropout : public Layer {
 public:
  const std::string layer_type() const override { return "Dropout"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;
  const Tensor &mask() const { return mask_; }

 protected:
  float dropout_ratio_ = 0.5f;
  Tensor mask_;
  Shape out_sample_shape_;
};

class Flatten : public Layer {
 public:
  const std::string layer_type() const override { return "Flatten"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;

 protected:
  int axis_ = 1;
  Shape input_shape_, out_sample_shape_;
};

class LRN : public Layer {
 public:
  const std::string layer_type() const override { return "LRN"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;

 protected:
  /// sum over a window of local_size_ channels, zero padded
  Tensor WindowSum(const Tensor &x) const;
  int local_size_ = 5;
  float alpha_ = 1.f, beta_ = 0.75f, k_ = 1.f;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

class Softmax : public Layer {
 public:
  const std::string layer_type() const override { return "Softmax"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;

 protected:
  int axis_ = 1;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

class PReLU : public Layer {
 public:
  const std::string layer_type() const override { return "PReLU"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const Tensor Forward(int flag, const Tensor &input) override;
  const std::pair<Tensor, std::vector<Tensor>> Backward(int flag,
                                                        const Tensor &grad) override;
  const Tensor &A() const { return a_; }
  void Set_a(const Tensor &a) { a_.ResetLike(a); a_.CopyData(a); }

 protected:
  std::vector<Tensor *> param_ptrs() override { return {&a_}; }
  /// per-element slope tensor matching the input's shape
  Tensor SlopeLike(const Tensor &x) const;
  bool channel_shared_ = false;
  Tensor a_;
  Shape out_sample_shape_;
  std::stack<Tensor> buf_;
};

/// N inputs -> one, concatenated along `axis` (0 = batch).
class Concat : public Layer {
 public:
  const std::string layer_type() const override { return "Concat"; }
  void Setup(const std::vector<Shape> &in_samples, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const std::vector<Tensor> Forward(int flag,
                                    const std::vector<Tensor> &inputs) override;
  const std::pair<std::vector<Tensor>, std::vector<Tensor>> Backward(
      int flag, const std::vector<Tensor> &grads) override;

 protected:
  int axis_ = 0;
  std::vector<size_t> slice_point_;
  Shape out_sample_shape_;
};

/// One input -> N, cut along `axis` at slice_point.
class Slice : public Layer {
 public:
  const std::string layer_type() const override { return "Slice"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override {
    LOG(FATAL) << "use GetOutputSampleShape(k)";
    return Shape{};
  }
  const Shape GetOutputSampleShape(int k) override {
    return out_sample_shapes_.at(k);
  }
  const std::vector<Tensor> Forward(int flag,
                                    const std::vector<Tensor> &inputs) override;
  const std::pair<std::vector<Tensor>, std::vector<Tensor>> Backward(
      int flag, const std::vector<Tensor> &grads) override;

 protected:
  int axis_ = 0;
  std::vector<size_t> slice_point_;
  std::vector<Shape> out_sample_shapes_;
};

/// One input replicated to output_size consumers; gradients are summed.
class Split : public Layer {
 public:
  const std::string layer_type() const override { return "Split"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const std::vector<Tensor> Forward(int flag,
                                    const std::vector<Tensor> &inputs) override;
  const std::pair<std::vector<Tensor>, std::vector<Tensor>> Backward(
      int flag, const std::vector<Tensor> &grads) override;
  size_t output_size() const { return output_size_; }

 protected:
  size_t output_size_ = 2;
  Shape out_sample_shape_;
};

/// N inputs summed element-wise; the gradient is replicated.
class Merge : public Layer {
 public:
  const std::string layer_type() const override { return "Merge"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  void Setup(const std::vector<Shape> &in_samples, const LayerConf &conf) override {
    Setup(in_samples.at(0), conf);
    input_size_ = in_samples.size();
  }
  const Shape GetOutputSampleShape() const override { return out_sample_shape_; }
  const std::vector<Tensor> Forward(int flag,
                                    const std::vector<Tensor> &inputs) override;
  const std::pair<std::vector<Tensor>, std::vector<Tensor>> Backward(
      int flag, const std::vector<Tensor> &grads) override;

 protected:
  size_t input_size_ = 1;
  Shape out_sample_shape_;
};

/// Single-direction recurrent layer (relu / tanh / lstm / gru) unrolled over a
/// sequence.  Forward takes {x_0 .. x_{T-1}, hx [, cx]} with x_t of shape
/// (batch, feature) and hx / cx of shape (batch, hidden) (empty tensors mean
/// zeros) and returns {y_0 .. y_{T-1}, hy [, cy]}; Backward takes the
/// gradients in the same order and returns {dx_0 .. dx_{T-1}, dhx [, dcx]}
/// plus the parameter gradients {dWx, dWh, db [, dbh]}.  Every step is two
/// GEMMs and element-wise gate math, on either device.
/// Parity: reference src/model/layer/{rnn,cudnn_rnn}.{h,cc} (same I/O
/// convention; cuDNN's flat weight blob becomes explicit matrices).
class RNN : public Layer {
 public:
  const std::string layer_type() const override { return "RNN"; }
  void Setup(const Shape &in_sample, const LayerConf &conf) override;
  const Shape GetOutputSampleShape() const override { return Shape{hidden_}; }
  const std::vector<Tensor> Forward(int flag,
                                    const std::vector<Tensor> &inputs) override;
  const std::pair<std::vector<Tensor>, std::vector<Tensor>> Backward(
      int flag, const std::vector<Tensor> &grads) override;
  size_t hidden_size() const { return hidden_; }
  bool has_cell() const { return mode_ == "lstm"; }

 protected:
  std::vector<Tensor *> param_ptrs() override {
    if (mode_ == "gru") return {&wx_, &wh_, &b_, &bh_};
    return {&wx_, &wh_, &b_};
  }
  struct StepCache {
    Tensor x, h_prev, c_prev, gates, c, extra;  // what Backward needs
  };
  std::string mode_ = "lstm";
  size_t input_ = 0, hidden_ = 0, ngates_ = 4;
  Tensor wx_, wh_, b_, bh_;
  std::stack<std::vector<StepCache>> buf_;
};
using CudnnRNN = RNN;

}  // namespace singa
