// Loss, metric, optimizers, updaters and FeedForwardNet.
#include <algorithm>
#include <cmath>
#include <iostream>
#include <numeric>
#include <random>

#include "singa/model/feed_forward_net.h"
#include "singa/model/initializer.h"
#include "singa/utils/string.h"

namespace singa {

// ---- losses ----------------------------------------------------------------------
Tensor MSE::Forward(int flag, const Tensor &prediction, const Tensor &target) {
  CHECK(buf_.empty()) << "Do not call Forward successively for more than twice."
                      << " The calling pattern is [Forward|Evaluate] Backward";
  Tensor t = prediction - target;
  size_t batchsize = t.nDim() > 1 ? t.shape(0) : 1;
  size_t dim = t.Size() / batchsize;
  t.Reshape(Shape{batchsize, dim});
  if (flag & kTrain) buf_.push(t);
  Tensor loss(Shape{batchsize}, prediction.device(), prediction.data_type());
  SumColumns(Square(t), &loss);
  return loss * 0.5f;
}

Tensor MSE::Backward() {
  Tensor ret = buf_.top();
  buf_.pop();
  return ret * (1.0f / ret.shape(0));
}

Tensor SoftmaxCrossEntropy::Forward(int flag, const Tensor &prediction,
                                    const Tensor &target) {
  CHECK(buf_.empty()) << "Do not call Forward successively for more than twice."
                      << " The calling pattern is [Forward|Evaluate] Backward";
  size_t batchsize = prediction.nDim() > 1 ? prediction.shape(0) : 1;
  size_t dim = prediction.Size() / batchsize;
  Tensor prob = SoftMax(Reshape(prediction, Shape{batchsize, dim}));
  Tensor loss(Shape{batchsize}, prob.device(), prob.data_type());
  ComputeCrossEntropy(prob, target, &loss);
  if (flag & kTrain) {
    buf_.push(prob);
    buf_.push(target);
  }
  return loss;
}

Tensor SoftmaxCrossEntropy::Backward() {
  const Tensor target = buf_.top();
  buf_.pop();
  Tensor prob = buf_.top();
  buf_.pop();
  SoftmaxCrossEntropyBwd(target, &prob);
  return prob;
}

// ---- metric ----------------------------------------------------------------------
Tensor Accuracy::Forward(const Tensor &prediction, const Tensor &target) {
  // ranking top-k on the host keeps the device free of a sort
  Tensor p = prediction.Clone(defaultDevice), t = target.Clone(defaultDevice);
  size_t batchsize = p.shape(0), dim = p.Size() / batchsize;
  CHECK_EQ(t.Size(), batchsize);
  std::vector<float> pv = p.ToVector<float>();
  std::vector<int> tv = t.ToVector<int>();
  std::vector<float> score(batchsize, 0.f);
  for (size_t b = 0; b < batchsize; ++b) {
    const float *row = pv.data() + b * dim;
    float truth = row[tv[b]];
    size_t better = 0;
    for (size_t i = 0; i < dim; ++i)
      if (row[i] > truth) ++better;
    score[b] = better < top_k_ ? 1.f : 0.f;
  }
  Tensor ret(Shape{batchsize});
  ret.CopyDataFromHostPtr(score.data(), batchsize);
  return ret;
}

// ---- optimizers ------------------------------------------------------------------
void Constraint::Apply(int step, Tensor &grad, Tensor &value) {
  if (threshold_ <= 0.f) return;
  float nrm = grad.L2() * grad.Size();  // L2() is normalised by the size
  if (nrm > threshold_) grad *= threshold_ / nrm;
}

void Regularizer::Apply(int step, Tensor &grad, Tensor &value, float scale) {
  if (coefficient_ * scale != 0.f) Axpy(coefficient_ * scale, value, &grad);
}

Optimizer::~Optimizer() {
  for (auto &kv : regularizers_) delete kv.second;
  for (auto &kv : constraints_) delete kv.second;
  delete constraint_;
  delete regularizer_;
}

void Optimizer::Setup(const OptimizerConf &conf) {
  conf_ = conf;
  delete regularizer_;
  delete constraint_;
  regularizer_ = nullptr;
  constraint_ = nullptr;
  if (conf.has_regularizer)
    regularizer_ = new Regularizer(conf.regularizer.type,
                                   conf.regularizer.coefficient);
  if (conf.has_constraint)
    constraint_ = new Constraint(conf.constraint.type,
                                 conf.constraint.threshold);
}

void Optimizer::Register(const std::string &name, const ParamSpec &specs) {
  if (specs.has_constraint) {
    CHECK(constraints_.find(name) == constraints_.end())
        << "Parameter with name = " << name << " has already registered";
    constraints_[name] = new Constraint("l2", specs.constraint_threshold);
  }
  if (specs.has_regularizer) {
    CHECK(regularizers_.find(name) == regularizers_.end())
        << "Parameter with name = " << name << " has already registered";
    regularizers_[name] = new Regularizer("l2", specs.regularizer_coefficient);
  }
  if (specs.decay_mult != 1.f) weight_decay_multplier_[name] = specs.decay_mult;
  if (specs.lr_mult != 1.f) learning_rate_multplier_[name] = specs.lr_mult;
}

void Optimizer::ApplyRegularizerConstraint(int epoch, const std::string &name,
                                           const Tensor &value, Tensor &grad,
                                           int step) {
  Tensor &v = const_cast<Tensor &>(value);
  if (regularizers_.find(name) != regularizers_.end()) {
    regularizers_.at(name)->Apply(epoch, grad, v);
  } else if (regularizer_ != nullptr) {
    float scale = 1.f;
    auto it = weight_decay_multplier_.find(name);
    if (it != weight_decay_multplier_.end()) scale = it->second;
    regularizer_->Apply(epoch, grad, v, scale);
  }
  if (constraints_.find(name) != constraints_.end())
    constraints_.at(name)->Apply(epoch, grad, v);
  else if (constraint_ != nullptr)
    constraint_->Apply(epoch, grad, v);
}

void Optimizer::Apply(int epoch, const std::string &name, Tensor &grad,
                      Tensor &value, int step) {
  float lr = learning_rate_generator_ ? learning_rate_generator_(epoch) : 0.f;
  auto it = learning_rate_multplier_.find(name);
  if (it != learning_rate_multplier_.end()) lr *= it->second;
  Apply(epoch, lr, name, grad, value, step);
}

void SGD::Setup(const OptimizerConf &conf) {
  Optimizer::Setup(conf);
  if (conf.momentum != 0.f) {
    float m = conf.momentum;
    SetMomentumGenerator([m](int) { return m; });
  }
}

void SGD::Apply(int epoch, float lr, const std::string &name, Tensor &grad,
                Tensor &value, int step) {
  if (grad.empty()) return;
  ApplyRegularizerConstraint(epoch, name, value, grad, step);
  if (momentum_generator_) {
    float mom = momentum_generator_(step >= 0 ? step : epoch);
    if (mom != 0.f) {
      if (history_gradient_.find(name) == history_gradient_.end()) {
        Tensor h(value.shape(), value.device(), value.data_type());
        h.SetValue(0.f);
        history_gradient_[name] = h;
      }
      Tensor &history = history_gradient_[name];
      history *= mom;                 // v = mom * v + lr * g
      Axpy(lr, grad, &history);
      value -= history;
      return;
    }
  }
  Axpy(-lr, grad, &value);
}

void Nesterov::Setup(const OptimizerConf &conf) {
  Optimizer::Setup(conf);
  float m = conf.momentum;
  SetMomentumGenerator([m](int) { return m; });
}

void Nesterov::Apply(int epoch, float lr, const std::string &name, Tensor &grad,
                     Tensor &value, int step) {
  if (grad.empty()) return;
  ApplyRegularizerConstraint(epoch, name, value, grad, step);
  CHECK(momentum_generator_) << "Nesterov needs a momentum";
  float mom = momentum_generator_(step >= 0 ? step : epoch);
  if (history_gradient_.find(name) == history_gradient_.end()) {
    Tensor h(value.shape(), value.device(), value.data_type());
    h.SetValue(0.f);
    history_gradient_[name] = h;
  }
  Tensor &history = history_gradient_[name];
  Tensor prev = history.Clone();
  history *= mom;                     // v' = mom * v + lr * g
  Axpy(lr, grad, &history);
  // w -= (1 + mom) * v' - mom * v
  Axpy(-(1.f + mom), history, &value);
  Axpy(mom, prev, &value);
}

void AdaGrad::Setup(const OptimizerConf &conf) {
  Optimizer::Setup(conf);
  delta_ = conf.delta;
}

void AdaGrad::Apply(int epoch, float lr, const std::string &name, Tensor &grad,
                    Tensor &value, int step) {
  if (grad.empty()) return;
  ApplyRegularizerConstraint(epoch, name, value, grad, step);
  if (history_gradient_.find(name) == history_gradient_.end()) {
    Tensor h(value.shape(), value.device(), value.data_type());
    h.SetValue(0.f);
    history_gradient_[name] = h;
  }
  Tensor &history = history_gradient_[name];
  history += Square(grad);
  Tensor upd = grad / Sqrt(history + delta_);
  Axpy(-lr, upd, &value);
}

void RMSProp::Setup(const OptimizerConf &conf) {
  Optimizer::Setup(conf);
  delta_ = conf.delta;
  rho_ = conf.rho;
}

void RMSProp::Apply(int epoch, float lr, const std::string &name, Tensor &grad,
                    Tensor &value, int step) {
  if (grad.empty()) return;
  ApplyRegularizerConstraint(epoch, name, value, grad, step);
  if (history_gradient_.find(name) == history_gradient_.end()) {
    Tensor h(value.shape(), value.device(), value.data_type());
    h.SetValue(0.f);
    history_gradient_[name] = h;
  }
  Tensor &history = history_gradient_[name];
  history *= rho_;
  Axpy(1.f - rho_, Square(grad), &history);
  Tensor upd = grad / Sqrt(history + delta_);
  Axpy(-lr, upd, &value);
}

std::shared_ptr<Optimizer> CreateOptimizer(const std::string &type) {
  std::string t = ToLowerCase(type);
  if (t == "sgd") return std::make_shared<SGD>();
  if (t == "nesterov") return std::make_shared<Nesterov>();
  if (t == "adagrad") return std::make_shared<AdaGrad>();
  if (t == "rmsprop") return std::make_shared<RMSProp>();
  LOG(FATAL) << "Unknown optimizer type: " << type;
  return nullptr;
}

// ---- LocalUpdater -----------------------------------------------------------------
void LocalUpdater::Register(const std::string &name, const ParamSpec &specs) {
  std::lock_guard<std::mutex> lock(reg_mtx_);
  if (slots_.find(name) == slots_.end()) {
    opt_->Register(name, specs);
    slots_[name].reset(new Slot());
  }
}

void LocalUpdater::Apply(int step, const std::string &name, Tensor &grad,
                         Tensor &value) {
  Slot *s;
  {
    std::lock_guard<std::mutex> lock(reg_mtx_);
    auto it = slots_.find(name);
    CHECK(it != slots_.end()) << "parameter " << name << " is not registered";
    s = it->second.get();
  }
  // the copy to the aggregation device is enqueued on the replica's device;
  // wait for it so that the last arriver sums finished data
  Tensor g = grad.Clone(dev_);
  grad.device()->Sync();
  dev_->Sync();

  std::unique_lock<std::mutex> lock(s->mtx);
  // a fast replica may come back for the next step before the slow ones have
  // picked up this step's result
  s->cv.wait(lock, [&] { return s->to_leave == 0; });
  s->partial.push_back(g);
  if (++s->arrived == total_num_) {
    Tensor sum = s->partial[0];
    for (size_t i = 1; i < s->partial.size(); ++i) sum += s->partial[i];
    sum *= 1.0f / total_num_;
    if (s->value.empty()) s->value = value.Clone(dev_);
    opt_->Apply(step, name, sum, s->value);
    dev_->Sync();
    s->partial.clear();
    s->arrived = 0;
    s->to_leave = total_num_;
    ++s->round;
    s->cv.notify_all();
  } else {
    uint64_t my_round = s->round;
    s->cv.wait(lock, [&] { return s->round != my_round; });
  }
  Tensor latest = s->value;
  lock.unlock();
  value.CopyData(latest.Clone(value.device()));
  value.device()->Sync();
  lock.lock();
  if (--s->to_leave == 0) s->cv.notify_all();
}

// ---- FeedForwardNet ---------------------------------------------------------------
std::shared_ptr<Layer> FeedForwardNet::Add(std::shared_ptr<Layer> layer) {
  layers_.push_back(layer);
  confs_.emplace_back();
  in_shapes_.emplace_back();
  return layer;
}

std::shared_ptr<Layer> FeedForwardNet::Add(const LayerConf &conf,
                                           const Shape *sample_shape) {
  return Add(CreateLayer(conf.type), conf, sample_shape);
}

std::shared_ptr<Layer> FeedForwardNet::Add(std::shared_ptr<Layer> layer,
                                           const LayerConf &conf,
                                           const Shape *sample_shape) {
  CHECK(sample_shape != nullptr || !layers_.empty())
      << "the first layer needs the input sample shape";
  Shape in = sample_shape ? *sample_shape
                          : layers_.back()->GetOutputSampleShape();
  layer->Setup(in, conf);
  layers_.push_back(layer);
  confs_.push_back(conf);
  in_shapes_.push_back(in);
  return layer;
}

const std::vector<std::string> FeedForwardNet::GetParamNames() const {
  std::vector<std::string> names;
  for (auto l : layers_)
    for (const auto &n : l->param_names()) names.push_back(n);
  return names;
}
const std::vector<ParamSpec> FeedForwardNet::GetParamSpecs() const {
  std::vector<ParamSpec> specs;
  for (auto l : layers_)
    for (const auto &s : l->param_specs()) specs.push_back(s);
  return specs;
}
const std::vector<Tensor> FeedForwardNet::GetParamValues() const {
  std::vector<Tensor> values;
  for (auto l : layers_)
    for (const auto &v : l->param_values()) values.push_back(v);
  return values;
}

void FeedForwardNet::Compile(bool shuffle, Optimizer *opt, Loss *loss,
                             Metric *metric) {
  Compile(shuffle, true, std::make_shared<Updater>(opt), loss, metric);
}

void FeedForwardNet::Compile(bool shuffle, bool to_register,
                             std::shared_ptr<Updater> updater, Loss *loss,
                             Metric *metric) {
  shuffle_ = shuffle;
  bool train = updater != nullptr && loss != nullptr;
  bool test = metric != nullptr;
  CHECK(train || test) << "Must set updater and loss, or set metric";
  updater_ = updater;
  loss_ = loss;
  metric_ = metric;
  const auto specs = GetParamSpecs();
  auto params = GetParamValues();
  CHECK_EQ(specs.size(), params.size());
  for (size_t k = 0; k < specs.size(); ++k) {
    // BatchNorm statistics and constant-initialised parameters keep what the
    // layer's Setup wrote unless a filler other than the default is given
    if (specs[k].filler.type != "constant" || specs[k].filler.value != 0.f) {
      auto init = CreateInitializer(specs[k].filler);
      init->Fill(params[k]);
    } else if (!params[k].initialized()) {
      params[k].SetValue(0.f);
    }
    if (to_register && updater_) updater_->Register(specs[k].name, specs[k]);
  }
}

const Tensor FeedForwardNet::Forward(int flag, const Tensor &data) {
  Tensor input(data), output;
  for (auto layer : layers_) {
    output = layer->Forward(flag, input);
    input = output;
  }
  return output;
}

const std::vector<Tensor> FeedForwardNet::Backward(int flag, const Tensor &grad) {
  std::vector<Tensor> param_grads;
  Tensor tmp = grad;
  for (int i = static_cast<int>(layers_.size()) - 1; i >= 0; --i) {
    auto ret = layers_.at(i)->Backward(flag, tmp);
    tmp = ret.first;
    // gradients come out last layer first; keep each layer's own order
    for (int k = static_cast<int>(ret.second.size()) - 1; k >= 0; --k)
      param_grads.push_back(ret.second[k]);
  }
  std::reverse(param_grads.begin(), param_grads.end());
  return param_grads;
}

const std::pair<float, float> FeedForwardNet::TrainOnBatch(int epoch,
                                                           const Tensor &x,
                                                           const Tensor &y) {
  int flag = kTrain;
  const Tensor fea = Forward(flag, x);
  float loss = loss_->Evaluate(flag, fea, y);
  float metric = metric_ ? metric_->Evaluate(fea, y) : 0.f;
  const Tensor grad = loss_->Backward();
  auto grads = Backward(kTrain, grad * (1.0f / x.shape(0)));
  auto names = GetParamNames();
  auto values = GetParamValues();
  CHECK_EQ(grads.size(), values.size());
  for (size_t k = 0; k < grads.size(); ++k)
    if (!grads[k].empty())  // e.g. running statistics of BatchNorm
      updater_->Apply(epoch, names[k], grads[k], values[k]);
  return {loss, metric};
}

void FeedForwardNet::Train(size_t batchsize, int nb_epoch, const Tensor &x,
                           const Tensor &y, float val_split) {
  CHECK_EQ(x.shape(0), y.shape(0)) << "Diff num of samples in x and y";
  size_t num_train = static_cast<size_t>(x.shape(0) * (1.f - val_split));
  if (val_split == 0.f) {
    Train(batchsize, nb_epoch, x, y, Tensor(), Tensor());
    return;
  }
  Train(batchsize, nb_epoch, CopyRows(x, 0, num_train),
        CopyRows(y, 0, num_train), CopyRows(x, num_train, x.shape(0)),
        CopyRows(y, num_train, y.shape(0)));
}

void FeedForwardNet::Train(size_t batchsize, int nb_epoch, const Tensor &x,
                           const Tensor &y, const Tensor &val_x,
                           const Tensor &val_y) {
  CHECK_EQ(x.shape(0), y.shape(0)) << "Diff num of samples in x and y";
  const size_t num_batch = x.shape(0) / batchsize;
  std::vector<size_t> order(num_batch);
  std::iota(order.begin(), order.end(), 0);
  std::mt19937 rng(0);
  for (int epoch = 0; epoch < nb_epoch; ++epoch) {
    if (shuffle_) std::shuffle(order.begin(), order.end(), rng);
    float loss = 0.f, metric = 0.f;
    for (size_t b : order) {
      const Tensor bx = CopyRows(x, b * batchsize, (b + 1) * batchsize);
      const Tensor by = CopyRows(y, b * batchsize, (b + 1) * batchsize);
      auto ret = TrainOnBatch(epoch, bx, by);
      loss += ret.first;
      metric += ret.second;
    }
    if (verbose_)
      LOG(INFO) << "Epoch " << epoch << ", training loss = "
                << loss / std::max<size_t>(num_batch, 1)
                << ", accuracy = " << metric / std::max<size_t>(num_batch, 1)
                << ", lr = " << updater_->GetOptimizer()->GetLearningRate(epoch);
    if (val_x.Size() && val_y.Size()) {
      auto ret = Evaluate(val_x, val_y, batchsize);
      if (verbose_)
        LOG(INFO) << "Epoch " << epoch << ", val loss = "
                  << Sum<float>(ret.first) / ret.first.Size()
                  << ", metric = " << Sum<float>(ret.second) / ret.second.Size();
    }
  }
}

std::pair<Tensor, Tensor> FeedForwardNet::EvaluateOnBatch(const Tensor &x,
                                                          const Tensor &y) {
  int flag = kEval;
  const Tensor fea = Forward(flag, x);
  const Tensor l = loss_ ? loss_->Forward(flag, fea, y) : Tensor();
  const Tensor m = metric_ ? metric_->Forward(fea, y) : Tensor();
  return {l, m};
}

std::pair<Tensor, Tensor> FeedForwardNet::Evaluate(const Tensor &x,
                                                   const Tensor &y,
                                                   size_t batchsize) {
  CHECK_EQ(x.shape(0), y.shape(0)) << "Diff num of samples in x and y";
  CHECK_GE(x.shape(0), batchsize);
  std::vector<Tensor> losses, metrics;
  const size_t num_batch = x.shape(0) / batchsize;
  for (size_t b = 0; b < num_batch; ++b) {
    auto ret = EvaluateOnBatch(CopyRows(x, b * batchsize, (b + 1) * batchsize),
                               CopyRows(y, b * batchsize, (b + 1) * batchsize));
    losses.push_back(ret.first.Clone(defaultDevice));
    metrics.push_back(ret.second.Clone(defaultDevice));
  }
  size_t rest = x.shape(0) % batchsize;
  if (rest) {
    auto ret = EvaluateOnBatch(CopyRows(x, num_batch * batchsize, x.shape(0)),
                               CopyRows(y, num_batch * batchsize, x.shape(0)));
    losses.push_back(ret.first.Clone(defaultDevice));
    metrics.push_back(ret.second.Clone(defaultDevice));
  }
  return {ConcatOn(losses, 0), ConcatOn(metrics, 0)};
}

const Tensor FeedForwardNet::Predict(const Tensor &x, size_t batchsize) {
  CHECK_GE(x.shape(0), batchsize);
  std::vector<Tensor> outs;
  const size_t num_batch = x.shape(0) / batchsize;
  for (size_t b = 0; b < num_batch; ++b)
    outs.push_back(PredictOnBatch(CopyRows(x, b * batchsize, (b + 1) * batchsize)));
  if (x.shape(0) % batchsize)
    outs.push_back(PredictOnBatch(CopyRows(x, num_batch * batchsize, x.shape(0))));
  return ConcatOn(outs, 0);
}

FeedForwardNet FeedForwardNet::Clone(std::shared_ptr<Device> device) {
  FeedForwardNet net;
  for (size_t i = 0; i < layers_.size(); ++i) {
    CHECK(!confs_[i].type.empty())
        << "Clone needs layers that were added together with their LayerConf";
    auto layer = net.Add(confs_[i], &in_shapes_[i]);
    auto src = layers_[i]->param_values();
    auto dst = layer->param_values();
    CHECK_EQ(src.size(), dst.size());
    for (size_t k = 0; k < src.size(); ++k)
      if (src[k].initialized()) dst[k].CopyData(src[k]);
    layer->ToDevice(device);
  }
  net.updater_ = updater_;
  net.loss_ = loss_;
  net.metric_ = metric_;
  net.shuffle_ = shuffle_;
  net.verbose_ = verbose_;
  net.device_ = device.get();
  net.dtype_ = dtype_;
  return net;
}

void FeedForwardNet::ToDevice(std::shared_ptr<Device> device) {
  for (auto layer : layers_) layer->ToDevice(device);
  device_ = device.get();
}

void FeedForwardNet::AsType(DataType dtype) {
  for (auto layer : layers_) layer->AsType(dtype);
  dtype_ = dtype;
}

std::thread FeedForwardNet::TrainThread(size_t batchsize, int nb_epoch,
                                        const Tensor &x, const Tensor &y,
                                        const Tensor &val_x,
                                        const Tensor &val_y) {
  return std::thread(
      [=]() { this->Train(batchsize, nb_epoch, x, y, val_x, val_y); });
}

std::thread FeedForwardNet::TrainThread(size_t batchsize, int nb_epoch,
                                        const Tensor &x, const Tensor &y) {
  return std::thread([=]() { this->Train(batchsize, nb_epoch, x, y); });
}

}  // namespace singa
