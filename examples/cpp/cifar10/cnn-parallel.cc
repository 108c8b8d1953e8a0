// Data-parallel CIFAR-10 training inside ONE process: N replicas of the
// network (one per GPU, or N host replicas with -cpu) each train on their
// shard in their own thread; a LocalUpdater averages the gradients on an
// aggregation device and hands the updated parameters back.
//   singa_cnn_parallel [-epoch N] [-lr X] [-replicas R] [-cpu]
// Counterpart of the reference's examples/cpp/cifar10/cnn-parallel.cc.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "cifar10.h"
#include "singa/singa.h"

namespace singa {

static LayerConf Conv(std::string name, int filters, float std) {
  LayerConf c;
  c.name = name;
  c.type = "singacuda_convolution";
  c.convolution_conf.num_output = filters;
  c.convolution_conf.set_kernel(3);
  c.convolution_conf.set_pad(1);
  ParamSpec w, b;
  w.name = name + "_weight";
  w.filler.type = "gaussian";
  w.filler.std = std;
  b.name = name + "_bias";
  c.param = {w, b};
  return c;
}

static LayerConf Simple(std::string name, std::string type) {
  LayerConf c;
  c.name = name;
  c.type = type;
  c.pooling_conf.set_kernel(2);
  c.pooling_conf.set_stride(2);
  return c;
}

static FeedForwardNet CreateNet() {
  FeedForwardNet net;
  Shape s{3, 32, 32};
  net.Add(Conv("conv1", 16, 0.05f), &s);
  net.Add(Simple("relu1", "singacuda_relu"));
  net.Add(Simple("pool1", "singacuda_pooling"));
  net.Add(Conv("conv2", 32, 0.05f));
  net.Add(Simple("relu2", "singacuda_relu"));
  net.Add(Simple("pool2", "singacuda_pooling"));
  net.Add(Simple("flat", "singa_flatten"));
  LayerConf ip;
  ip.name = "ip";
  ip.type = "singacuda_dense";
  ip.dense_conf.num_output = 10;
  ParamSpec w, b;
  w.name = "ip_weight";
  w.filler.type = "gaussian";
  w.filler.std = 0.02f;
  b.name = "ip_bias";
  ip.param = {w, b};
  net.Add(ip);
  return net;
}

int Train(int num_epoch, float lr, int replicas, bool use_cpu,
          const std::string &data_dir) {
  Cifar10 data(data_dir);
  auto train = data.ReadTrainData();
  auto test = data.ReadTestData();
  size_t per = train.first.shape(0) / replicas;

  std::vector<std::shared_ptr<Device>> devs;
  if (!use_cpu && Platform::GetNumGPUs() >= replicas)
    devs = Platform::CreateCudaGPUs(replicas);
  else
    for (int i = 0; i < replicas; ++i) devs.push_back(defaultDevice);

  SGD sgd;
  OptimizerConf conf;
  conf.momentum = 0.9f;
  sgd.Setup(conf);
  sgd.SetLearningRateGenerator([lr](int) { return lr; });
  // gradients meet on the first replica's device
  auto updater = std::make_shared<LocalUpdater>(replicas, &sgd, devs[0]);
  SoftmaxCrossEntropy loss;
  Accuracy acc;

  FeedForwardNet master = CreateNet();
  master.Compile(true, true, updater, &loss, &acc);
  std::vector<FeedForwardNet> nets;
  std::vector<SoftmaxCrossEntropy> losses(replicas);
  std::vector<Accuracy> accs(replicas);
  for (int r = 0; r < replicas; ++r) {
    nets.push_back(master.Clone(devs[r]));
    // every replica needs its own loss / metric state
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
  std::printf("test accuracy %.3f with %d replicas\n", a, replicas);
  return a > 0.2f ? 0 : 1;
}

}  // namespace singa

int main(int argc, char **argv) {
  singa::LogToStderr();
  int pos = singa::ArgPos(argc, argv, "-epoch");
  int nEpoch = pos != -1 ? std::atoi(argv[pos + 1]) : 2;
  pos = singa::ArgPos(argc, argv, "-lr");
  float lr = pos != -1 ? std::atof(argv[pos + 1]) : 0.01f;
  pos = singa::ArgPos(argc, argv, "-replicas");
  int replicas = pos != -1 ? std::atoi(argv[pos + 1]) : 2;
  pos = singa::ArgPos(argc, argv, "-data");
  std::string data = pos != -1 ? argv[pos + 1] : "cifar-10-batches-bin";
  bool cpu = false;
  for (int i = 1; i < argc; ++i) cpu = cpu || std::string(argv[i]) == "-cpu";
  return singa::Train(nEpoch, lr, replicas, cpu, data);
}
