// CIFAR-10 convolutional network (3 x conv-pool-relu-lrn, one dense layer)
// trained with the C++ API on one device.
//   singa_cnn [-epoch N] [-lr X] [-data DIR] [-cpu | -opencl] [-batch B]
// Counterpart of the reference's examples/cpp/cifar10/cnn.cc; the
// configuration records are plain structs here.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "cifar10.h"
#include "singa/singa.h"

namespace singa {

const std::string engine = "singacuda";  // layer registry prefix

LayerConf GenConvConf(std::string name, int nb_filter, int kernel, int stride,
                      int pad, float std) {
  LayerConf conf;
  conf.name = name;
  conf.type = engine + "_convolution";
  conf.convolution_conf.num_output = nb_filter;
  conf.convolution_conf.set_kernel(kernel);
  conf.convolution_conf.set_stride(stride);
  conf.convolution_conf.set_pad(pad);
  ParamSpec w;
  w.name = name + "_weight";
  w.filler.type = "gaussian";
  w.filler.std = std;
  ParamSpec b;
  b.name = name + "_bias";
  b.lr_mult = 2.f;
  conf.param = {w, b};
  return conf;
}

LayerConf GenPoolingConf(std::string name, bool max_pool, int kernel,
                         int stride, int pad) {
  LayerConf conf;
  conf.name = name;
  conf.type = engine + "_pooling";
  conf.pooling_conf.set_kernel(kernel);
  conf.pooling_conf.set_stride(stride);
  conf.pooling_conf.set_pad(pad);
  conf.pooling_conf.pool = max_pool ? PoolingConf::MAX : PoolingConf::AVE;
  return conf;
}

LayerConf GenReLUConf(std::string name) {
  LayerConf conf;
  conf.name = name;
  conf.type = engine + "_relu";
  return conf;
}

LayerConf GenDenseConf(std::string name, int num_output, float std, float wd) {
  LayerConf conf;
  conf.name = name;
  conf.type = "singa_dense";
  conf.type = engine + "_dense";
  conf.dense_conf.num_output = num_output;
  ParamSpec w;
  w.name = name + "_weight";
  w.decay_mult = wd;
  w.filler.type = "gaussian";
  w.filler.std = std;
  ParamSpec b;
  b.name = name + "_bias";
  b.lr_mult = 2.f;
  b.decay_mult = 0.f;
  conf.param = {w, b};
  return conf;
}

LayerConf GenLRNConf(std::string name) {
  LayerConf conf;
  conf.name = name;
  conf.type = engine + "_lrn";
  conf.lrn_conf.local_size = 3;
  conf.lrn_conf.alpha = 5e-05f;
  conf.lrn_conf.beta = 0.75f;
  return conf;
}

LayerConf GenFlattenConf(std::string name) {
  LayerConf conf;
  conf.name = name;
  conf.type = "singa_flatten";
  return conf;
}

FeedForwardNet CreateNet() {
  FeedForwardNet net;
  Shape s{3, 32, 32};
  net.Add(GenConvConf("conv1", 32, 5, 1, 2, 0.0001f), &s);
  net.Add(GenPoolingConf("pool1", true, 3, 2, 1));
  net.Add(GenReLUConf("relu1"));
  net.Add(GenLRNConf("lrn1"));
  net.Add(GenConvConf("conv2", 32, 5, 1, 2, 0.01f));
  net.Add(GenReLUConf("relu2"));
  net.Add(GenPoolingConf("pool2", false, 3, 2, 1));
  net.Add(GenLRNConf("lrn2"));
  net.Add(GenConvConf("conv3", 64, 5, 1, 2, 0.01f));
  net.Add(GenReLUConf("relu3"));
  net.Add(GenPoolingConf("pool3", false, 3, 2, 1));
  net.Add(GenFlattenConf("flat"));
  net.Add(GenDenseConf("ip", 10, 0.01f, 250));
  return net;
}

void Train(int num_epoch, float lr, const std::string &data_dir, bool use_cpu,
           size_t batch, bool use_opencl) {
  Cifar10 data(data_dir);
  auto train = data.ReadTrainData();
  auto test = data.ReadTestData();
  Tensor train_x = train.first, train_y = train.second;
  Tensor test_x = test.first, test_y = test.second;
  std::printf("train samples %zu, test samples %zu\n", train_x.shape(0),
              test_x.shape(0));

  auto net = CreateNet();
  SGD sgd;
  OptimizerConf opt_conf;
  opt_conf.momentum = 0.9f;
  opt_conf.has_regularizer = true;
  opt_conf.regularizer.coefficient = 0.004f;
  sgd.Setup(opt_conf);
  sgd.SetLearningRateGenerator([lr](int epoch) {
    if (epoch <= 120) return lr;
    if (epoch <= 130) return lr * 0.1f;
    return lr * 0.01f;
  });
  SoftmaxCrossEntropy loss;
  Accuracy acc;
  net.Compile(true, &sgd, &loss, &acc);

  std::shared_ptr<Device> dev = defaultDevice;
  if (use_opencl) {
    CHECK_GT(OpenclDevice::GetNumDevices(), 0) << "no OpenCL GPU device";
    dev = std::make_shared<OpenclDevice>(0);
  } else if (!use_cpu && Platform::GetNumGPUs() > 0) {
    dev = Platform::CreateCudaGPUs(1)[0];
  }
  std::printf("training on %s\n",
              dev->lang() == kCpp ? "CppCPU"
                                  : (dev->lang() == kOpencl ? "OpenclDevice"
                                                            : "HipGPU"));
  net.ToDevice(dev);
  train_x.ToDevice(dev);
  train_y.ToDevice(dev);
  test_x.ToDevice(dev);
  test_y.ToDevice(dev);
  net.Train(batch, num_epoch, train_x, train_y, test_x, test_y);
}

}  // namespace singa

int main(int argc, char **argv) {
  singa::LogToStderr();
  int pos = singa::ArgPos(argc, argv, "-epoch");
  int nEpoch = pos != -1 ? std::atoi(argv[pos + 1]) : 1;
  pos = singa::ArgPos(argc, argv, "-lr");
  float lr = pos != -1 ? std::atof(argv[pos + 1]) : 0.001f;
  pos = singa::ArgPos(argc, argv, "-data");
  std::string data = pos != -1 ? argv[pos + 1] : "cifar-10-batches-bin";
  pos = singa::ArgPos(argc, argv, "-batch");
  size_t batch = pos != -1 ? std::atoi(argv[pos + 1]) : 100;
  bool cpu = false, opencl = false;
  for (int i = 1; i < argc; ++i) {
    cpu = cpu || std::string(argv[i]) == "-cpu";
    opencl = opencl || std::string(argv[i]) == "-opencl";
  }
  std::printf("Start training\n");
  singa::Train(nEpoch, lr, data, cpu, batch, opencl);
  std::printf("End training\n");
  return 0;
}
