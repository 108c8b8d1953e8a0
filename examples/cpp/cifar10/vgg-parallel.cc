// VGG-16 with batch normalisation and dropout on CIFAR-10, trained
// data-parallel inside ONE process: R replicas (one per GPU, or host replicas
// with -cpu) run in their own threads and meet in a LocalUpdater that averages
// gradients on the first device.  The training set is centred with its
// per-pixel mean before it is split between the replicas.
//   singa_vgg_parallel [-epoch N] [-lr X] [-replicas R] [-width W]
//                      [-dropout S] [-cpu]
// -width scales the channel counts (64 = the full network) and -dropout scales
// every dropout ratio (1 = as published, 0 = none); a narrow network has too
// little capacity for the published ratios.
// Counterpart of the reference's examples/cpp/cifar10/vgg-parallel.cc.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "cifar10.h"
#include "singa/singa.h"

namespace singa {

static const float kWeightDecay = 0.0005f;

struct VggBuilder {
  FeedForwardNet net;
  bool first = true;
  float drop_scale = 1.f;

  void Add(const LayerConf &c) {
    if (first) {
      Shape s{3, 32, 32};
      net.Add(c, &s);
      first = false;
    } else {
      net.Add(c);
    }
  }

  void ConvBnRelu(const std::string &name, int filters) {
    LayerConf c;
    c.name = name + "_conv";
    c.type = "singacuda_convolution";
    c.convolution_conf.num_output = filters;
    c.convolution_conf.set_kernel(3);
    c.convolution_conf.set_pad(1);
    ParamSpec w, b;
    w.name = c.name + "_weight";
    w.filler.type = "gaussian";
    w.filler.std = std::sqrt(2.f / (filters * 9.f));  // He et al., fan-out
    b.name = c.name + "_bias";
    c.param = {w, b};
    Add(c);
    BatchNorm(name + "_bn");
    Plain(name + "_relu", "singacuda_relu");
  }

  void BatchNorm(const std::string &name) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_batchnorm";
    ParamSpec gamma, beta, mean, var;
    gamma.name = name + "_gamma";
    gamma.filler.type = "uniform";
    beta.name = name + "_beta";
    mean.name = name + "_mean";
    var.name = name + "_var";
    var.filler.value = 1.f;
    c.param = {gamma, beta, mean, var};
    Add(c);
  }

  void Plain(const std::string &name, const std::string &type) {
    LayerConf c;
    c.name = name;
    c.type = type;
    Add(c);
  }

  void Pool(const std::string &name) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_pooling";
    c.pooling_conf.set_kernel(2);
    c.pooling_conf.set_stride(2);
    Add(c);
  }

  void Drop(const std::string &name, float ratio) {
    ratio *= drop_scale;
    if (ratio <= 0.f) return;
    LayerConf c;
    c.name = name;
    c.type = "singacuda_dropout";
    c.dropout_conf.dropout_ratio = ratio;
    Add(c);
  }

  void Dense(const std::string &name, int out) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_dense";
    c.dense_conf.num_output = out;
    ParamSpec w, b;
    w.name = name + "_weight";
    w.decay_mult = kWeightDecay;
    w.filler.type = "gaussian";
    w.filler.std = 0.02f;
    b.name = name + "_bias";
    b.lr_mult = 2.f;
    b.decay_mult = 0.f;
    c.param = {w, b};
    Add(c);
  }
};

// Five stages of 2,2,3,3,3 convolutions; dropout sits between the
// convolutions of a stage, a 2x2 max pool closes it.
static FeedForwardNet CreateNet(int width, float drop_scale) {
  VggBuilder b;
  b.drop_scale = drop_scale;
  const int reps[5] = {2, 2, 3, 3, 3};
  const int mult[5] = {1, 2, 4, 8, 8};
  for (int s = 0; s < 5; ++s) {
    for (int r = 0; r < reps[s]; ++r) {
      std::string tag = std::to_string(s + 1) + "_" + std::to_string(r + 1);
      b.ConvBnRelu("conv" + tag, width * mult[s]);
      if (r + 1 < reps[s]) b.Drop("drop" + tag, s == 0 ? 0.3f : 0.4f);
    }
    b.Pool("pool" + std::to_string(s + 1));
  }
  b.Plain("flat", "singa_flatten");
  b.Drop("flat_drop", 0.5f);
  b.Dense("ip1", width * 8);
  b.BatchNorm("ip1_bn");
  b.Plain("ip1_relu", "singacuda_relu");
  b.Drop("ip1_drop", 0.5f);
  b.Dense("ip2", 10);
  return b.net;
}

int Train(int num_epoch, float lr, int replicas, int width, float drop_scale,
          bool use_cpu, const std::string &data_dir) {
  Cifar10 data(data_dir);
  auto train = data.ReadTrainData();
  auto test = data.ReadTestData();
  {
    // centre both sets with the training set's per-pixel mean
    size_t n = train.first.shape(0), m = test.first.shape(0);
    Tensor flat = Reshape(train.first, Shape{n, train.first.Size() / n});
    Tensor mean = Average(flat, 0);
    SubRow(mean, &flat);
    train.first = Reshape(flat, train.first.shape());
    Tensor tflat = Reshape(test.first, Shape{m, test.first.Size() / m});
    SubRow(mean, &tflat);
    test.first = Reshape(tflat, test.first.shape());
  }
  size_t per = train.first.shape(0) / replicas;

  std::vector<std::shared_ptr<Device>> devs;
  if (!use_cpu && Platform::GetNumGPUs() >= replicas)
    devs = Platform::CreateCudaGPUs(replicas);
  else
    for (int i = 0; i < replicas; ++i) devs.push_back(defaultDevice);

  SGD sgd;
  OptimizerConf conf;
  conf.momentum = 0.9f;
  conf.has_regularizer = true;
  conf.regularizer.coefficient = kWeightDecay;
  sgd.Setup(conf);
  // the step size halves every 30 epochs
  sgd.SetLearningRateGenerator(
      [lr](int epoch) { return lr * std::pow(0.5f, epoch / 30); });
  auto updater = std::make_shared<LocalUpdater>(replicas, &sgd, devs[0]);
  SoftmaxCrossEntropy loss;
  Accuracy acc;

  FeedForwardNet master = CreateNet(width, drop_scale);
  master.Compile(true, true, updater, &loss, &acc);
  std::vector<FeedForwardNet> nets;
  std::vector<SoftmaxCrossEntropy> losses(replicas);
  std::vector<Accuracy> accs(replicas);
  for (int r = 0; r < replicas; ++r) {
    nets.push_back(master.Clone(devs[r]));
    nets[r].Compile(true, false, updater, &losses[r], &accs[r]);
    nets[r].ToDevice(devs[r]);
    nets[r].set_verbose(r == 0);
  }
  std::vector<std::thread> threads;
  for (int r = 0; r < replicas; ++r) {
    Tensor x = CopyRows(train.first, r * per, (r + 1) * per);
    Tensor y = CopyRows(train.second, r * per, (r + 1) * per);
    x.ToDevice(devs[r]);
    y.ToDevice(devs[r]);
    threads.push_back(nets[r].TrainThread(64, num_epoch, x, y));
  }
  for (auto &t : threads) t.join();
  Tensor tx = test.first, ty = test.second;
  tx.ToDevice(devs[0]);
  ty.ToDevice(devs[0]);
  auto ev = nets[0].Evaluate(tx, ty, 64);
  float a = Sum<float>(ev.second) / ev.second.Size();
  std::printf("test accuracy %.3f with %d replicas, width %d\n", a, replicas,
              width);
  return a > 0.2f ? 0 : 1;
}

}  // namespace singa

int main(int argc, char **argv) {
  singa::LogToStderr();
  int pos = singa::ArgPos(argc, argv, "-epoch");
  int nEpoch = pos != -1 ? std::atoi(argv[pos + 1]) : 2;
  pos = singa::ArgPos(argc, argv, "-lr");
  float lr = pos != -1 ? std::atof(argv[pos + 1]) : 0.05f;
  pos = singa::ArgPos(argc, argv, "-replicas");
  int replicas = pos != -1 ? std::atoi(argv[pos + 1]) : 2;
  pos = singa::ArgPos(argc, argv, "-width");
  int width = pos != -1 ? std::atoi(argv[pos + 1]) : 64;
  pos = singa::ArgPos(argc, argv, "-dropout");
  float drop = pos != -1 ? std::atof(argv[pos + 1]) : 1.f;
  pos = singa::ArgPos(argc, argv, "-data");
  std::string data = pos != -1 ? argv[pos + 1] : "cifar-10-batches-bin";
  bool cpu = false;
  for (int i = 1; i < argc; ++i) cpu = cpu || std::string(argv[i]) == "-cpu";
  return singa::Train(nEpoch, lr, replicas, width, drop, cpu, data);
}
