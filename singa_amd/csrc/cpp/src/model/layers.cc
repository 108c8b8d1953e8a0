#include "singa/model/layers.h"

#include <cmath>

#include "singa/utils/string.h"

namespace singa {

void Layer::EnsureParamSpecs(const std::vector<std::string> &suffixes) {
  for (size_t i = param_specs_.size(); i < suffixes.size(); ++i) {
    ParamSpec s;
    s.name = name_ + "_" + suffixes[i];
    param_specs_.push_back(s);
  }
  for (size_t i = 0; i < suffixes.size(); ++i)
    if (param_specs_[i].name.empty())
      param_specs_[i].name = name_ + "_" + suffixes[i];
}

std::shared_ptr<Layer> CreateLayer(const std::string type) {
  Factory<Layer> factory;
  return std::shared_ptr<Layer>(factory.Create(type));
}

const std::vector<std::string> GetRegisteredLayers() {
  std::vector<std::string> out;
  for (const auto &id : Factory<Layer>::GetIDs()) {
    auto l = CreateLayer(id);
    out.push_back("Register type: " + id + " --> " + l->layer_type());
  }
  return out;
}

static Tensor TakeBuf(std::stack<Tensor> *buf) {
  CHECK(!buf->empty()) << "Backward without a matching training Forward";
  Tensor t = buf->top();
  buf->pop();
  return t;
}

// ---- Dense -----------------------------------------------------------------------
RegisterLayerClass(singacpp_dense, Dense);
RegisterLayerClass(singacuda_dense, Dense);
RegisterLayerClass(singacl_dense, Dense);

void Dense::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  CHECK_EQ(in_sample.size(), 1u) << "Dense takes flat samples";
  vdim_ = in_sample[0];
  hdim_ = conf.dense_conf.num_output;
  transpose_ = conf.dense_conf.transpose;
  bias_term_ = conf.dense_conf.bias_term;
  weight_.Resize(transpose_ ? Shape{hdim_, vdim_} : Shape{vdim_, hdim_});
  EnsureParamSpecs(bias_term_ ? std::vector<std::string>{"weight", "bias"}
                              : std::vector<std::string>{"weight"});
  if (bias_term_) bias_.Resize(Shape{hdim_});
}

const Tensor Dense::Forward(int flag, const Tensor &input) {
  CHECK_EQ(input.nDim(), 2u);
  Tensor out = Mult(input, transpose_ ? weight_.T() : weight_);
  if (bias_term_) AddRow(bias_, &out);
  if (flag & kTrain) buf_.push(input);
  return out;
}

const std::pair<Tensor, std::vector<Tensor>> Dense::Backward(int flag,
                                                             const Tensor &grad) {
  Tensor src = TakeBuf(&buf_);
  std::vector<Tensor> pg;
  Tensor dx = Mult(grad, transpose_ ? weight_ : weight_.T());
  Tensor dw = transpose_ ? Mult(grad.T(), src) : Mult(src.T(), grad);
  pg.push_back(dw);
  if (bias_term_) {
    Tensor db(Shape{hdim_}, grad.device(), grad.data_type());
    SumRows(grad, &db);
    pg.push_back(db);
  }
  return {dx, pg};
}

// ---- Activation -------------------------------------------------------------------
RegisterLayerClass(singa_relu, Activation);
RegisterLayerClass(singa_sigmoid, Activation);
RegisterLayerClass(singa_tanh, Activation);
RegisterLayerClass(singacpp_relu, Activation);
RegisterLayerClass(singacuda_relu, Activation);
RegisterLayerClass(singacl_relu, Activation);
RegisterLayerClass(cudnn_relu, Activation);
RegisterLayerClass(singacpp_sigmoid, Activation);
RegisterLayerClass(singacuda_sigmoid, Activation);
RegisterLayerClass(singacl_sigmoid, Activation);
RegisterLayerClass(cudnn_sigmoid, Activation);
RegisterLayerClass(singacpp_tanh, Activation);
RegisterLayerClass(singacuda_tanh, Activation);
RegisterLayerClass(singacl_tanh, Activation);
RegisterLayerClass(cudnn_tanh, Activation);

void Activation::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  auto pos = conf.type.find_first_of('_');
  CHECK(pos != std::string::npos) << "activation type is <backend>_<mode>";
  std::string m = ToLowerCase(conf.type.substr(pos + 1));
  if (m == "relu") {
    mode_ = "RELU";
    neg_slope_ = conf.relu_conf.negative_slope;
  } else if (m == "sigmoid") {
    mode_ = "SIGMOID";
  } else if (m == "tanh") {
    mode_ = "TANH";
  } else {
    LOG(FATAL) << "Unknown activation type: " << conf.type;
  }
  out_sample_shape_ = in_sample;
}

const Tensor Activation::Forward(int flag, const Tensor &input) {
  Tensor out;
  if (mode_ == "RELU") {
    out = ReLU(input);
    if (neg_slope_ != 0.f) {
      Tensor neg = input - out;  // min(x, 0)
      Axpy(neg_slope_, neg, &out);
    }
    if (flag & kTrain) buf_.push(input);
  } else {
    out = mode_ == "SIGMOID" ? Sigmoid(input) : Tanh(input);
    if (flag & kTrain) buf_.push(out);
  }
  return out;
}

const std::pair<Tensor, std::vector<Tensor>> Activation::Backward(
    int flag, const Tensor &grad) {
  Tensor saved = TakeBuf(&buf_);
  Tensor dx;
  if (mode_ == "RELU") {
    dx = ReLUBackward(grad, saved);
    if (neg_slope_ != 0.f) {
      Tensor rest = grad - dx;  // gradient where x <= 0
      Axpy(neg_slope_, rest, &dx);
    }
  } else if (mode_ == "SIGMOID") {
    dx = grad * saved * (saved * -1.f + 1.f);
  } else {
    dx = grad * (saved * saved * -1.f + 1.f);
  }
  return {dx, {}};
}

// ---- Convolution ------------------------------------------------------------------
RegisterLayerClass(singacpp_convolution, Convolution);
RegisterLayerClass(singacuda_convolution, Convolution);
RegisterLayerClass(singacl_convolution, Convolution);
RegisterLayerClass(cudnn_convolution, Convolution);

void Convolution::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  const ConvolutionConf &c = conf.convolution_conf;
  CHECK_EQ(in_sample.size(), 3u) << "Convolution takes (C,H,W) samples";
  channels_ = in_sample[0];
  height_ = in_sample[1];
  width_ = in_sample[2];
  CHECK_GT(c.kernel_h, 0);
  CHECK_GT(c.kernel_w, 0);
  p_ = ConvParams{c.kernel_h, c.kernel_w, c.stride_h, c.stride_w, c.pad_h,
                  c.pad_w, c.group, c.num_output};
  bias_term_ = c.bias_term;
  size_t oh = ConvOutSize(height_, c.kernel_h, c.stride_h, c.pad_h);
  size_t ow = ConvOutSize(width_, c.kernel_w, c.stride_w, c.pad_w);
  out_sample_shape_ = Shape{size_t(c.num_output), oh, ow};
  weight_.Resize(Shape{size_t(c.num_output), channels_ / c.group,
                       size_t(c.kernel_h), size_t(c.kernel_w)});
  EnsureParamSpecs(bias_term_ ? std::vector<std::string>{"weight", "bias"}
                              : std::vector<std::string>{"weight"});
  if (bias_term_) bias_.Resize(Shape{size_t(c.num_output)});
}

const Tensor Convolution::Forward(int flag, const Tensor &input) {
  CHECK_EQ(input.nDim(), 4u);
  if (flag & kTrain) buf_.push(input);
  return ConvForward(input, weight_, bias_term_ ? &bias_ : nullptr, p_);
}

const std::pair<Tensor, std::vector<Tensor>> Convolution::Backward(
    int flag, const Tensor &grad) {
  Tensor src = TakeBuf(&buf_);
  std::vector<Tensor> pg;
  pg.push_back(ConvBackwardWeight(grad, src, weight_.shape(), p_));
  if (bias_term_) pg.push_back(ConvBackwardBias(grad));
  Tensor dx = ConvBackwardData(grad, weight_, src.shape(), p_);
  return {dx, pg};
}

// ---- Pooling -----------------------------------------------------------------------
RegisterLayerClass(singacpp_pooling, Pooling);
RegisterLayerClass(singacuda_pooling, Pooling);
RegisterLayerClass(singacl_pooling, Pooling);
RegisterLayerClass(cudnn_pooling, Pooling);

void Pooling::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  const PoolingConf &c = conf.pooling_conf;
  CHECK_EQ(in_sample.size(), 3u);
  p_ = PoolParams{c.kernel_h, c.kernel_w, c.stride_h, c.stride_w, c.pad_h,
                  c.pad_w, c.pool == PoolingConf::MAX};
  size_t oh = ConvOutSize(in_sample[1], c.kernel_h, c.stride_h, c.pad_h);
  size_t ow = ConvOutSize(in_sample[2], c.kernel_w, c.stride_w, c.pad_w);
  out_sample_shape_ = Shape{in_sample[0], oh, ow};
}

const Tensor Pooling::Forward(int flag, const Tensor &input) {
  Tensor out = PoolForward(input, p_);
  if (flag & kTrain) {
    buf_.push(input);
    buf_.push(out);
  }
  return out;
}

const std::pair<Tensor, std::vector<Tensor>> Pooling::Backward(
    int flag, const Tensor &grad) {
  Tensor y = TakeBuf(&buf_);
  Tensor x = TakeBuf(&buf_);
  return {PoolBackward(grad, x, y, p_), {}};
}

// ---- BatchNorm ---------------------------------------------------------------------
RegisterLayerClass(singacpp_batchnorm, BatchNorm);
RegisterLayerClass(singacuda_batchnorm, BatchNorm);
RegisterLayerClass(singacl_batchnorm, BatchNorm);
RegisterLayerClass(cudnn_batchnorm, BatchNorm);

void BatchNorm::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  out_sample_shape_ = in_sample;
  factor_ = static_cast<float>(conf.batchnorm_conf.factor);
  channels_ = in_sample[0];
  for (Tensor *t : param_ptrs()) t->Resize(Shape{channels_});
  bnScale_.SetValue(1.f);
  bnBias_.SetValue(0.f);
  runningMean_.SetValue(0.f);
  runningVariance_.SetValue(1.f);
  EnsureParamSpecs({"scale", "bias", "mean", "var"});
}

const Tensor BatchNorm::Forward(int flag, const Tensor &input) {
  if (flag & kTrain) {
    Tensor mean, invstd;
    Tensor y = BatchNormForwardTrain(input, bnScale_, bnBias_, &runningMean_,
                                     &runningVariance_, factor_, &mean,
                                     &invstd);
    buf_.push(input);
    buf_.push(mean);
    buf_.push(invstd);
    return y;
  }
  return BatchNormForwardInfer(input, bnScale_, bnBias_, runningMean_,
                               runningVariance_);
}

const std::pair<Tensor, std::vector<Tensor>> BatchNorm::Backward(
    int flag, const Tensor &grad) {
  Tensor invstd = TakeBuf(&buf_);
  Tensor mean = TakeBuf(&buf_);
  Tensor x = TakeBuf(&buf_);
  Tensor dscale, dbias;
  Tensor dx = BatchNormBackward(grad, x, bnScale_, mean, invstd, &dscale,
                                &dbias);
  // the running statistics are state, not trained: their gradient slots stay
  // empty, and the training loop leaves such parameters to the layer
  return {dx, {dscale, dbias, Tensor(), Tensor()}};
}

// ---- Dropout -----------------------------------------------------------------------
RegisterLayerClass(singacpp_dropout, Dropout);
RegisterLayerClass(singacuda_dropout, Dropout);
RegisterLayerClass(singacl_dropout, Dropout);
RegisterLayerClass(cudnn_dropout, Dropout);

void Dropout::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  dropout_ratio_ = conf.dropout_conf.dropout_ratio;
  CHECK(dropout_ratio_ >= 0.f && dropout_ratio_ < 1.f);
  out_sample_shape_ = in_sample;
}

const Tensor Dropout::Forward(int flag, const Tensor &input) {
  if (!(flag & kTrain)) return input;
  mask_.ResetLike(input);
  mask_.Reshape(input.shape());
  Bernoulli(1.f - dropout_ratio_, &mask_);
  mask_ *= 1.f / (1.f - dropout_ratio_);
  return input * mask_;
}

const std::pair<Tensor, std::vector<Tensor>> Dropout::Backward(
    int flag, const Tensor &grad) {
  if (!(flag & kTrain)) return {grad, {}};
  return {grad * mask_, {}};
}

// ---- Flatten -----------------------------------------------------------------------
RegisterLayerClass(singa_flatten, Flatten);
RegisterLayerClass(singacpp_flatten, Flatten);
RegisterLayerClass(singacuda_flatten, Flatten);
RegisterLayerClass(singacl_flatten, Flatten);

void Flatten::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  axis_ = conf.flatten_conf.axis;
  CHECK_GE(axis_, 1) << "the batch dimension cannot be folded";
  size_t len = 1;
  out_sample_shape_.clear();
  // sample dims before `axis` are kept, the rest are folded into one
  for (int i = 0; i < axis_ - 1 && i < int(in_sample.size()); ++i)
    out_sample_shape_.push_back(in_sample[i]);
  for (size_t i = axis_ - 1; i < in_sample.size(); ++i) len *= in_sample[i];
  out_sample_shape_.push_back(len);
}

const Tensor Flatten::Forward(int flag, const Tensor &input) {
  input_shape_ = input.shape();
  Shape s{input.shape(0)};
  for (size_t d : out_sample_shape_) s.push_back(d);
  return Reshape(input, s);
}

const std::pair<Tensor, std::vector<Tensor>> Flatten::Backward(
    int flag, const Tensor &grad) {
  return {Reshape(grad, input_shape_), {}};
}

// ---- LRN ---------------------------------------------------------------------------
RegisterLayerClass(singacpp_lrn, LRN);
RegisterLayerClass(singacuda_lrn, LRN);
RegisterLayerClass(singacl_lrn, LRN);
RegisterLayerClass(cudnn_lrn, LRN);

void LRN::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  out_sample_shape_ = in_sample;
  local_size_ = conf.lrn_conf.local_size;
  CHECK_EQ(local_size_ % 2, 1) << "LRN only supports odd values for Localvol";
  alpha_ = conf.lrn_conf.alpha;
  beta_ = conf.lrn_conf.beta;
  k_ = conf.lrn_conf.k;
}

Tensor LRN::WindowSum(const Tensor &x) const {
  // (N,C,H,W) viewed as a one-channel image (N,1,C,H*W) and filtered with a
  // column of ones of height local_size: an exact zero-padded window sum
  // over channels that runs on the convolution kernel
  const Shape s = x.shape();
  Tensor img = Reshape(x, Shape{s[0], 1, s[1], s[2] * s[3]});
  Tensor ones(Shape{1, 1, size_t(local_size_), 1}, x.device(), x.data_type());
  ones.SetValue(1.f);
  ConvParams p{local_size_, 1, 1, 1, local_size_ / 2, 0, 1, 1};
  Tensor out = ConvForward(img, ones, nullptr, p);
  return Reshape(out, s);
}

const Tensor LRN::Forward(int flag, const Tensor &input) {
  CHECK_EQ(input.nDim(), 4u);
  Tensor scale = WindowSum(Square(input)) * (alpha_ / local_size_) + k_;
  Tensor y = input * Pow(scale, -beta_);
  if (flag & kTrain) {
    buf_.push(input);
    buf_.push(scale);
    buf_.push(y);
  }
  return y;
}

const std::pair<Tensor, std::vector<Tensor>> LRN::Backward(int flag,
                                                           const Tensor &grad) {
  Tensor y = TakeBuf(&buf_);
  Tensor scale = TakeBuf(&buf_);
  Tensor x = TakeBuf(&buf_);
  // y = x s^-b, s = k + a W(x^2):  dx = dy s^-b - 2ab x W(dy y / s)
  Tensor t = WindowSum(grad * y / scale);
  Tensor dx = grad * Pow(scale, -beta_) -
              x * t * (2.f * alpha_ * beta_ / local_size_);
  return {dx, {}};
}

// ---- Softmax -----------------------------------------------------------------------
RegisterLayerClass(singa_softmax, Softmax);
RegisterLayerClass(singacpp_softmax, Softmax);
RegisterLayerClass(singacuda_softmax, Softmax);
RegisterLayerClass(singacl_softmax, Softmax);
RegisterLayerClass(cudnn_softmax, Softmax);

void Softmax::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  axis_ = conf.softmax_conf.axis;
  CHECK_EQ(axis_, 1) << "softmax runs over the sample dimensions";
  out_sample_shape_ = Shape{Product(in_sample)};
}

const Tensor Softmax::Forward(int flag, const Tensor &input) {
  Tensor in2 = input.nDim() == 2
                   ? input
                   : Reshape(input, Shape{input.shape(0),
                                          input.Size() / input.shape(0)});
  Tensor out = SoftMax(in2);
  if (flag & kTrain) buf_.push(out);
  return out;
}

const std::pair<Tensor, std::vector<Tensor>> Softmax::Backward(
    int flag, const Tensor &grad) {
  Tensor y = TakeBuf(&buf_);
  Tensor g2 = Reshape(grad, y.shape());
  return {SoftMaxBackward(g2, y), {}};
}

// ---- PReLU -------------------------------------------------------------------------
RegisterLayerClass(singa_prelu, PReLU);
RegisterLayerClass(singacpp_prelu, PReLU);
RegisterLayerClass(singacuda_prelu, PReLU);
RegisterLayerClass(singacl_prelu, PReLU);

void PReLU::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  out_sample_shape_ = in_sample;
  channel_shared_ = conf.prelu_conf.channel_shared;
  CHECK_EQ(conf.prelu_conf.format, std::string("NCHW"));
  a_.Resize(Shape{channel_shared_ ? 1 : in_sample[0]});
  a_.SetValue(0.25f);
  EnsureParamSpecs({"a"});
}

Tensor PReLU::SlopeLike(const Tensor &x) const {
  // (N*C, HW) matrix whose row r holds slope a[r % C]
  const size_t N = x.shape(0), C = x.shape(1), HW = x.Size() / (N * C);
  Tensor m(Shape{N * C, HW}, x.device(), x.data_type());
  m.SetValue(1.f);
  Tensor col;
  if (channel_shared_) {
    col = Tensor(Shape{N * C}, x.device(), x.data_type());
    col.SetValue(1.f);
    // scale a vector of ones by the single slope through a (NC,1)x(1,1) GEMM
    Tensor ones = Reshape(col, Shape{N * C, 1});
    col = Reshape(Mult(ones, Reshape(a_, Shape{1, 1})), Shape{N * C});
  } else {
    std::vector<Tensor> rep(N, Reshape(a_, Shape{1, C}));
    col = Reshape(ConcatenateRows(rep), Shape{N * C});
  }
  MultColumn(col, &m);
  return m;
}

const Tensor PReLU::Forward(int flag, const Tensor &input) {
  CHECK_EQ(input.nDim(), 4u);
  Tensor pos = ReLU(input);
  Tensor neg = input - pos;
  Tensor slope = SlopeLike(input);
  Tensor out = pos + Reshape(Reshape(neg, slope.shape()) * slope, input.shape());
  if (flag & kTrain) buf_.push(input);
  return out;
}

const std::pair<Tensor, std::vector<Tensor>> PReLU::Backward(int flag,
                                                             const Tensor &grad) {
  Tensor x = TakeBuf(&buf_);
  const size_t N = x.shape(0), C = x.shape(1);
  Tensor slope = SlopeLike(x);
  Tensor gpos = ReLUBackward(grad, x);
  Tensor gneg = grad - gpos;
  Tensor dx = gpos + Reshape(Reshape(gneg, slope.shape()) * slope, x.shape());
  Tensor neg = x - ReLU(x);
  Tensor prod = Reshape(gneg * neg, slope.shape());   // (N*C, HW)
  Tensor per_row(Shape{N * C}, x.device(), x.data_type());
  SumColumns(prod, &per_row);
  Tensor da(Shape{C}, x.device(), x.data_type());
  SumRows(Reshape(per_row, Shape{N, C}), &da);
  if (channel_shared_) {
    Tensor one(Shape{1}, x.device(), x.data_type());
    SumColumns(Reshape(da, Shape{1, C}), &one);
    da = one;
  }
  return {dx, {da}};
}

// ---- Concat / Slice / Split / Merge ---------------------------------------------------
RegisterLayerClass(singa_concat, Concat);
RegisterLayerClass(singacpp_concat, Concat);
RegisterLayerClass(singacuda_concat, Concat);
RegisterLayerClass(singacl_concat, Concat);

void Concat::Setup(const std::vector<Shape> &in_samples, const LayerConf &conf) {
  Layer::Setup(in_samples.at(0), conf);
  axis_ = conf.concat_conf.axis;  // 0 = batch dimension
  out_sample_shape_ = in_samples[0];
  if (axis_ > 0) {
    size_t total = 0;
    for (const auto &s : in_samples) total += s.at(axis_ - 1);
    out_sample_shape_[axis_ - 1] = total;
  }
}

const std::vector<Tensor> Concat::Forward(int flag,
                                          const std::vector<Tensor> &inputs) {
  slice_point_.clear();
  for (const auto &t : inputs) slice_point_.push_back(t.shape(axis_));
  return {ConcatOn(inputs, axis_)};
}

const std::pair<std::vector<Tensor>, std::vector<Tensor>> Concat::Backward(
    int flag, const std::vector<Tensor> &grads) {
  CHECK_EQ(grads.size(), 1u);
  std::vector<Tensor> dx;
  size_t off = 0;
  for (size_t len : slice_point_) {
    dx.push_back(SliceOn(grads[0], off, off + len, axis_));
    off += len;
  }
  return {dx, {}};
}

RegisterLayerClass(singa_slice, Slice);
RegisterLayerClass(singacpp_slice, Slice);
RegisterLayerClass(singacuda_slice, Slice);
RegisterLayerClass(singacl_slice, Slice);

void Slice::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  axis_ = conf.slice_conf.axis;
  slice_point_.assign(conf.slice_conf.slice_point.begin(),
                      conf.slice_conf.slice_point.end());
  out_sample_shapes_.clear();
  if (axis_ == 0) {
    out_sample_shapes_.assign(slice_point_.size() + 1, in_sample);
  } else {
    size_t prev = 0;
    std::vector<size_t> ends = slice_point_;
    ends.push_back(in_sample.at(axis_ - 1));
    for (size_t e : ends) {
      Shape s = in_sample;
      s[axis_ - 1] = e - prev;
      out_sample_shapes_.push_back(s);
      prev = e;
    }
  }
}

const std::vector<Tensor> Slice::Forward(int flag,
                                         const std::vector<Tensor> &inputs) {
  CHECK_EQ(inputs.size(), 1u);
  std::vector<Tensor> out;
  size_t prev = 0;
  std::vector<size_t> ends = slice_point_;
  ends.push_back(inputs[0].shape(axis_));
  for (size_t e : ends) {
    out.push_back(SliceOn(inputs[0], prev, e, axis_));
    prev = e;
  }
  return out;
}

const std::pair<std::vector<Tensor>, std::vector<Tensor>> Slice::Backward(
    int flag, const std::vector<Tensor> &grads) {
  return {{ConcatOn(grads, axis_)}, {}};
}

RegisterLayerClass(singa_split, Split);
RegisterLayerClass(singacpp_split, Split);
RegisterLayerClass(singacuda_split, Split);
RegisterLayerClass(singacl_split, Split);

void Split::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  output_size_ = conf.split_conf.output_size;
  out_sample_shape_ = in_sample;
}

const std::vector<Tensor> Split::Forward(int flag,
                                         const std::vector<Tensor> &inputs) {
  CHECK_EQ(inputs.size(), 1u);
  return std::vector<Tensor>(output_size_, inputs[0]);
}

const std::pair<std::vector<Tensor>, std::vector<Tensor>> Split::Backward(
    int flag, const std::vector<Tensor> &grads) {
  CHECK_EQ(grads.size(), output_size_);
  Tensor sum = grads[0].Clone();
  for (size_t i = 1; i < grads.size(); ++i) sum += grads[i];
  return {{sum}, {}};
}

RegisterLayerClass(singa_merge, Merge);
RegisterLayerClass(singacpp_merge, Merge);
RegisterLayerClass(singacuda_merge, Merge);
RegisterLayerClass(singacl_merge, Merge);

void Merge::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  out_sample_shape_ = in_sample;
}

const std::vector<Tensor> Merge::Forward(int flag,
                                         const std::vector<Tensor> &inputs) {
  CHECK_GT(inputs.size(), 0u);
  input_size_ = inputs.size();
  Tensor sum = inputs[0].Clone();
  for (size_t i = 1; i < inputs.size(); ++i) sum += inputs[i];
  return {sum};
}

const std::pair<std::vector<Tensor>, std::vector<Tensor>> Merge::Backward(
    int flag, const std::vector<Tensor> &grads) {
  CHECK_EQ(grads.size(), 1u);
  return {std::vector<Tensor>(input_size_, grads[0]), {}};
}

// ---- RNN ---------------------------------------------------------------------------
RegisterLayerClass(singa_rnn, RNN);
RegisterLayerClass(singacpp_rnn, RNN);
RegisterLayerClass(singacuda_rnn, RNN);
RegisterLayerClass(cudnn_rnn, RNN);

void RNN::Setup(const Shape &in_sample, const LayerConf &conf) {
  Layer::Setup(in_sample, conf);
  CHECK_EQ(in_sample.size(), 1u) << "RNN takes (feature) samples per step";
  mode_ = ToLowerCase(conf.rnn_conf.rnn_mode);
  CHECK(mode_ == "relu" || mode_ == "tanh" || mode_ == "lstm" || mode_ == "gru")
      << "unknown rnn_mode " << mode_;
  CHECK_EQ(conf.rnn_conf.direction, std::string("unidirectional"));
  CHECK_EQ(conf.rnn_conf.num_stacks, 1) << "stack RNN layers explicitly";
  input_ = in_sample[0];
  hidden_ = conf.rnn_conf.hidden_size;
  CHECK_GT(hidden_, 0u);
  ngates_ = mode_ == "lstm" ? 4 : (mode_ == "gru" ? 3 : 1);
  wx_.Resize(Shape{input_, ngates_ * hidden_});
  wh_.Resize(Shape{hidden_, ngates_ * hidden_});
  b_.Resize(Shape{ngates_ * hidden_});
  if (mode_ == "gru") bh_.Resize(Shape{ngates_ * hidden_});
  const float k = 1.f / std::sqrt(static_cast<float>(hidden_));
  Uniform(-k, k, &wx_);
  Uniform(-k, k, &wh_);
  b_.SetValue(0.f);
  if (mode_ == "gru") bh_.SetValue(0.f);
  EnsureParamSpecs(mode_ == "gru"
                       ? std::vector<std::string>{"wx", "wh", "b", "bh"}
                       : std::vector<std::string>{"wx", "wh", "b"});
}

static Tensor Gate(const Tensor &z, size_t k, size_t H) {
  return SliceOn(z, k * H, (k + 1) * H, 1);
}
static Tensor OneMinus(const Tensor &t) { return t * -1.f + 1.f; }
static Tensor ZerosLike(const Tensor &ref, const Shape &s) {
  Tensor z(s, ref.device(), ref.data_type());
  z.SetValue(0.f);
  return z;
}

const std::vector<Tensor> RNN::Forward(int flag,
                                       const std::vector<Tensor> &inputs) {
  const size_t extra = has_cell() ? 2 : 1;
  CHECK_GT(inputs.size(), extra);
  const size_t T = inputs.size() - extra;
  const size_t B = inputs[0].shape(0), H = hidden_;
  Tensor h = inputs[T].Size() ? inputs[T] : ZerosLike(inputs[0], Shape{B, H});
  Tensor c;
  if (has_cell())
    c = inputs[T + 1].Size() ? inputs[T + 1] : ZerosLike(inputs[0], Shape{B, H});
  std::vector<StepCache> caches;
  std::vector<Tensor> out;
  for (size_t t = 0; t < T; ++t) {
    const Tensor &x = inputs[t];
    CHECK_EQ(x.shape(1), input_);
    StepCache sc;
    sc.x = x;
    sc.h_prev = h;
    Tensor a = Mult(x, wx_);
    AddRow(b_, &a);
    Tensor u = Mult(h, wh_);
    if (mode_ == "lstm") {
      Tensor z = a + u;
      Tensor i = Sigmoid(Gate(z, 0, H)), f = Sigmoid(Gate(z, 1, H));
      Tensor o = Sigmoid(Gate(z, 2, H)), g = Tanh(Gate(z, 3, H));
      sc.c_prev = c;
      c = f * c + i * g;
      h = o * Tanh(c);
      sc.gates = ConcatOn({i, f, o, g}, 1);
      sc.c = c;
    } else if (mode_ == "gru") {
      AddRow(bh_, &u);
      Tensor r = Sigmoid(Gate(a, 0, H) + Gate(u, 0, H));
      Tensor zg = Sigmoid(Gate(a, 1, H) + Gate(u, 1, H));
      Tensor un = Gate(u, 2, H);
      Tensor n = Tanh(Gate(a, 2, H) + r * un);
      h = OneMinus(zg) * n + zg * h;
      sc.gates = ConcatOn({r, zg, n}, 1);
      sc.extra = un;
    } else {
      Tensor z = a + u;
      h = mode_ == "relu" ? ReLU(z) : Tanh(z);
      sc.gates = h;
    }
    out.push_back(h);
    if (flag & kTrain) caches.push_back(sc);
  }
  out.push_back(h);
  if (has_cell()) out.push_back(c);
  if (flag & kTrain) buf_.push(caches);
  return out;
}

const std::pair<std::vector<Tensor>, std::vector<Tensor>> RNN::Backward(
    int flag, const std::vector<Tensor> &grads) {
  CHECK(!buf_.empty()) << "Backward without a matching training Forward";
  std::vector<StepCache> caches = buf_.top();
  buf_.pop();
  const size_t T = caches.size(), H = hidden_;
  const size_t extra = has_cell() ? 2 : 1;
  CHECK_EQ(grads.size(), T + extra);
  const Tensor &ref = caches[0].x;
  const size_t B = ref.shape(0);
  Tensor dh = grads[T].Size() ? grads[T].Clone() : ZerosLike(ref, Shape{B, H});
  Tensor dc;
  if (has_cell())
    dc = grads[T + 1].Size() ? grads[T + 1].Clone()
                             : ZerosLike(ref, Shape{B, H});
  Tensor dwx = ZerosLike(ref, wx_.shape()), dwh = ZerosLike(ref, wh_.shape());
  Tensor db = ZerosLike(ref, b_.shape()), dbh;
  if (mode_ == "gru") dbh = ZerosLike(ref, bh_.shape());
  std::vector<Tensor> dxs(T);
  Tensor rowsum(Shape{ngates_ * H}, ref.device(), ref.data_type());
  for (int t = static_cast<int>(T) - 1; t >= 0; --t) {
    const StepCache &sc = caches[t];
    if (grads[t].Size()) dh += grads[t];
    Tensor da, du;  // gradients of the input and recurrent pre-activations
    if (mode_ == "lstm") {
      Tensor i = Gate(sc.gates, 0, H), f = Gate(sc.gates, 1, H);
      Tensor o = Gate(sc.gates, 2, H), g = Gate(sc.gates, 3, H);
      Tensor tc = Tanh(sc.c);
      Tensor dct = dc + dh * o * OneMinus(tc * tc);
      Tensor di = dct * g * i * OneMinus(i);
      Tensor df = dct * sc.c_prev * f * OneMinus(f);
      Tensor d_o = dh * tc * o * OneMinus(o);
      Tensor dg = dct * i * OneMinus(g * g);
      dc = dct * f;
      da = ConcatOn({di, df, d_o, dg}, 1);
      du = da;
    } else if (mode_ == "gru") {
      Tensor r = Gate(sc.gates, 0, H), zg = Gate(sc.gates, 1, H);
      Tensor n = Gate(sc.gates, 2, H);
      Tensor dn = dh * OneMinus(zg) * OneMinus(n * n);
      Tensor dz = dh * (sc.h_prev - n) * zg * OneMinus(zg);
      Tensor dr = dn * sc.extra * r * OneMinus(r);
      da = ConcatOn({dr, dz, dn}, 1);
      du = ConcatOn({dr, dz, dn * r}, 1);
    } else {
      Tensor h = sc.gates;
      da = mode_ == "relu" ? ReLUBackward(dh, h) : dh * OneMinus(h * h);
      du = da;
    }
    Mult<float>(1.f, sc.x.T(), da, 1.f, &dwx);
    Mult<float>(1.f, sc.h_prev.T(), du, 1.f, &dwh);
    SumRows(da, &rowsum);
    db += rowsum;
    if (mode_ == "gru") {
      SumRows(du, &rowsum);
      dbh += rowsum;
    }
    dxs[t] = Mult(da, wx_.T());
    Tensor dh_prev = Mult(du, wh_.T());
    if (mode_ == "gru") dh_prev += dh * Gate(sc.gates, 1, H);
    dh = dh_prev;
  }
  dxs.push_back(dh);
  if (has_cell()) dxs.push_back(dc);
  std::vector<Tensor> pg{dwx, dwh, db};
  if (mode_ == "gru") pg.push_back(dbh);
  return {dxs, pg};
}

}  // namespace singa
