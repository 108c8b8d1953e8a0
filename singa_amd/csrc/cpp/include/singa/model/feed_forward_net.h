// Sequential network of Layers with a training loop.
// API parity: reference include/singa/model/feed_forward_net.h:31-162.
#pragma once
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "layer.h"
#include "loss.h"
#include "metric.h"
#include "updater.h"

namespace singa {

class FeedForwardNet {
 public:
  FeedForwardNet() = default;
  ~FeedForwardNet() {}

  /// Append a layer that has been Setup already.
  std::shared_ptr<Layer> Add(std::shared_ptr<Layer> layer);
  /// Create (by conf.type), Setup against the previous layer's output sample
  /// shape (or `sample_shape` for the first layer) and append.
  std::shared_ptr<Layer> Add(const LayerConf &conf,
                             const Shape *sample_shape = nullptr);
  std::shared_ptr<Layer> Add(std::shared_ptr<Layer> layer, const LayerConf &conf,
                             const Shape *sample_shape = nullptr);

  /// Set the training components; initialises parameters from their
  /// ParamSpec fillers and registers them with the optimizer.
  void Compile(bool shuffle, Optimizer *opt, Loss *loss, Metric *metric);
  void Compile(bool shuffle, bool to_register,
               std::shared_ptr<Updater> updater, Loss *loss, Metric *metric);

  void Train(size_t batchsize, int nb_epoch, const Tensor &x, const Tensor &y,
             float val_split = 0.0f);
  void Train(size_t batchsize, int nb_epoch, const Tensor &x, const Tensor &y,
             const Tensor &val_x, const Tensor &val_y);
  /// -> (mean loss, mean metric) of the batch
  const std::pair<float, float> TrainOnBatch(int epoch, const Tensor &x,
                                             const Tensor &y);
  std::pair<Tensor, Tensor> Evaluate(const Tensor &x, const Tensor &y,
                                     size_t batchsize = 128);
  std::pair<Tensor, Tensor> EvaluateOnBatch(const Tensor &x, const Tensor &y);
  const Tensor Predict(const Tensor &x, size_t batchsize = 128);
  const Tensor PredictOnBatch(const Tensor &x) { return Forward(kEval, x); }

  const Tensor Forward(int flag, const Tensor &x);
  const std::vector<Tensor> Backward(int flag, const Tensor &grad);

  /// Replica with its own copies of the parameters on `device`.
  FeedForwardNet Clone(std::shared_ptr<Device> device);
  void ToDevice(std::shared_ptr<Device> device);
  void ToHost() { ToDevice(defaultDevice); }
  void AsType(DataType dtype);

  /// Run Train in a new thread (used for multi-replica training).
  std::thread TrainThread(size_t batchsize, int nb_epoch, const Tensor &x,
                          const Tensor &y, const Tensor &val_x,
                          const Tensor &val_y);
  std::thread TrainThread(size_t batchsize, int nb_epoch, const Tensor &x,
                          const Tensor &y);

  const std::vector<std::shared_ptr<Layer>> layers() const { return layers_; }
  const std::vector<std::string> GetParamNames() const;
  const std::vector<ParamSpec> GetParamSpecs() const;
  const std::vector<Tensor> GetParamValues() const;
  void set_verbose(bool v) { verbose_ = v; }

 protected:
  std::vector<std::shared_ptr<Layer>> layers_;
  std::vector<LayerConf> confs_;
  std::vector<Shape> in_shapes_;
  std::shared_ptr<Updater> updater_;
  Loss *loss_ = nullptr;
  Metric *metric_ = nullptr;
  bool shuffle_ = true;
  bool verbose_ = true;
  Device *device_ = nullptr;
  DataType dtype_ = kFloat32;
};

}  // namespace singa
