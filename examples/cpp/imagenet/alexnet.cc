// AlexNet on ImageNet record files with the C++ API.  A loader thread decodes
// and augments the next batch (several decoder threads inside) while the
// network trains on the current one; loss/accuracy go to the channels
// "train_perf" and "val_perf"; a snapshot is written after every epoch.
//   singa_alexnet -data bin/ -ntrain 1281167 -ntest 50000 -filesize 1280
//                 [-epoch 90] [-lr 0.01] [-batchsize 256] [-nthreads 12]
//                 [-pfreq 100] [-classes 1000] [-size 256] [-width 1.0] [-cpu]
// With -synthetic the record files are generated first (tiny defaults), which
// makes the program a self-test of the pipeline.  -size/-width/-classes shrink
// the input, the channel counts and the classifier for such runs.
// Counterpart of the reference's examples/cpp/imagenet/alexnet.cc.
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "ilsvrc12.h"
#include "singa/singa.h"

namespace singa {

struct NetBuilder {
  FeedForwardNet net;
  Shape input;
  bool first = true;
  float width = 1.f;

  int Ch(int n) const { return std::max(4, int(std::lround(n * width))); }

  void Add(const LayerConf &c) {
    if (first) net.Add(c, &input);
    else net.Add(c);
    first = false;
  }
  void Conv(const std::string &name, int filters, int kernel, int stride,
            int pad, float std, float bias = 0.f) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_convolution";
    c.convolution_conf.num_output = Ch(filters);
    c.convolution_conf.set_kernel(kernel);
    c.convolution_conf.set_stride(stride);
    c.convolution_conf.set_pad(pad);
    ParamSpec w, b;
    w.name = name + "_weight";
    w.filler.type = "gaussian";
    w.filler.std = std;
    b.name = name + "_bias";
    b.lr_mult = 2.f;
    b.decay_mult = 0.f;
    b.filler.value = bias;
    c.param = {w, b};
    Add(c);
  }
  void Relu(const std::string &name) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_relu";
    Add(c);
  }
  void Pool(const std::string &name) {  // overlapping 3x3 / stride 2 max pool
    LayerConf c;
    c.name = name;
    c.type = "singacuda_pooling";
    c.pooling_conf.set_kernel(3);
    c.pooling_conf.set_stride(2);
    Add(c);
  }
  void Lrn(const std::string &name) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_lrn";
    c.lrn_conf.local_size = 5;
    c.lrn_conf.alpha = 1e-4f;
    c.lrn_conf.beta = 0.75f;
    c.lrn_conf.k = 2.f;
    Add(c);
  }
  void Flatten(const std::string &name) {
    LayerConf c;
    c.name = name;
    c.type = "singa_flatten";
    Add(c);
  }
  void Dense(const std::string &name, int out, float std, float bias) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_dense";
    c.dense_conf.num_output = out;
    ParamSpec w, b;
    w.name = name + "_weight";
    w.filler.type = "gaussian";
    w.filler.std = std;
    b.name = name + "_bias";
    b.lr_mult = 2.f;
    b.decay_mult = 0.f;
    b.filler.value = bias;
    c.param = {w, b};
    Add(c);
  }
  void Drop(const std::string &name, float ratio) {
    LayerConf c;
    c.name = name;
    c.type = "singacuda_dropout";
    c.dropout_conf.dropout_ratio = ratio;
    Add(c);
  }
};

static FeedForwardNet CreateNet(size_t crop, int classes, float width) {
  NetBuilder b;
  b.input = Shape{3, crop, crop};
  b.width = width;
  // narrow self-test networks start from larger weights: with 0.01 nothing
  // reaches the classifier of a network this shallow in a few steps
  float s = width < 1.f ? 0.05f : 0.01f;
  b.Conv("conv1", 96, 11, 4, 0, s);
  b.Relu("relu1");
  b.Pool("pool1");
  b.Lrn("lrn1");
  b.Conv("conv2", 256, 5, 1, 2, s, 1.f);
  b.Relu("relu2");
  b.Pool("pool2");
  b.Lrn("lrn2");
  b.Conv("conv3", 384, 3, 1, 1, s);
  b.Relu("relu3");
  b.Conv("conv4", 384, 3, 1, 1, s, 1.f);
  b.Relu("relu4");
  b.Conv("conv5", 256, 3, 1, 1, s, 1.f);
  b.Relu("relu5");
  b.Pool("pool5");
  b.Flatten("flat");
  b.Dense("ip6", b.Ch(4096), width < 1.f ? 0.05f : 0.005f, 1.f);
  b.Relu("relu6");
  b.Drop("drop6", 0.5f);
  b.Dense("ip7", b.Ch(4096), width < 1.f ? 0.05f : 0.005f, 1.f);
  b.Relu("relu7");
  b.Drop("drop7", 0.5f);
  b.Dense("ip8", classes, s, 0.f);
  return b.net;
}

// Batches of a sequence of record files, one batch ahead of the consumer.
class BatchStream {
 public:
  BatchStream(ILSVRC *data, int flag, std::vector<std::string> files,
              size_t batchsize, int nthreads)
      : data_(data), flag_(flag), files_(std::move(files)),
        batchsize_(batchsize), nthreads_(nthreads),
        x_(Shape{1}), y_(Shape{1}, kInt) {
    Start();
  }
  ~BatchStream() {
    if (th_.joinable()) th_.join();
  }
  /// false when every file is exhausted; short batches are dropped
  bool Next(Tensor *x, Tensor *y, std::shared_ptr<Device> dev) {
    while (true) {
      if (!th_.joinable()) return false;
      timer_.Tick();  // only the time the consumer has to wait counts
      th_.join();
      load_ms += timer_.Elapsed();
      if (n_read_ == batchsize_) break;
      if (n_read_ > 0)
        LOG(WARNING) << "the last " << n_read_ << " images of "
                     << files_[file_] << " do not fill a batch and are skipped";
      if (++file_ == files_.size()) return false;
      Start();
    }
    *x = x_.Clone(dev);
    *y = y_.Clone(dev);
    Start();
    return true;
  }
  double load_ms = 0;

 private:
  void Start() {
    th_ = data_->AsyncLoadData(flag_, files_[file_], batchsize_, &x_, &y_,
                               &n_read_, nthreads_);
  }
  ILSVRC *data_;
  int flag_;
  std::vector<std::string> files_;
  size_t batchsize_, file_ = 0, n_read_ = 0;
  int nthreads_;
  Tensor x_, y_;
  std::thread th_;
  Timer timer_;
};

static void Checkpoint(FeedForwardNet &net, const std::string &prefix) {
  Snapshot snapshot(prefix, Snapshot::kWrite, 200);
  auto names = net.GetParamNames();
  auto values = net.GetParamValues();
  for (size_t k = 0; k < names.size(); ++k)
    snapshot.Write(names[k], values[k].Clone(defaultDevice));
  LOG(INFO) << "snapshot written to " << prefix;
}

struct Options {
  int epochs = 90, nthreads = 12, classes = 1000;
  float lr = 0.01f, width = 1.f;
  size_t batchsize = 256, file_size = 1280, ntrain = 1281167, ntest = 50000;
  size_t pfreq = 100, size = 256;
  bool cpu = false;
  std::string folder = "imagenet_data";
};

int Train(const Options &o) {
  ILSVRC data(o.size, o.size * 227 / 256);
  data.ReadMean(o.folder + "/mean.bin");
  auto net = CreateNet(data.crop_size(), o.classes, o.width);
  std::shared_ptr<Device> dev = defaultDevice;
  if (!o.cpu && Platform::GetNumGPUs() > 0) dev = Platform::CreateCudaGPUs(1)[0];

  SGD sgd;
  OptimizerConf conf;
  conf.momentum = 0.9f;
  conf.has_regularizer = true;
  conf.regularizer.coefficient = 0.0005f;
  sgd.Setup(conf);
  float lr = o.lr;
  sgd.SetLearningRateGenerator(  // a tenth every 20 epochs
      [lr](int epoch) { return lr * std::pow(0.1f, epoch / 20); });
  SoftmaxCrossEntropy loss;
  Accuracy acc;
  net.Compile(true, &sgd, &loss, &acc);
  net.ToDevice(dev);

  Channel *train_ch = GetChannel("train_perf");
  train_ch->EnableDestStderr(true);
  train_ch->EnableDestFile(false);
  Channel *val_ch = GetChannel("val_perf");
  val_ch->EnableDestStderr(true);
  val_ch->EnableDestFile(false);

  std::vector<std::string> train_files;
  size_t nfiles = (o.ntrain + o.file_size - 1) / o.file_size;
  for (size_t f = 1; f <= nfiles; ++f)
    train_files.push_back(ILSVRC::TrainFile(o.folder, f));

  float val_acc = 0.f;
  char line[256];
  for (int epoch = 0; epoch < o.epochs; ++epoch) {
    {
      BatchStream stream(&data, kTrain, train_files, o.batchsize, o.nthreads);
      Tensor x, y;
      float l = 0.f, m = 0.f;
      double train_ms = 0;
      size_t b = 0;
      Timer t;
      while (stream.Next(&x, &y, dev)) {
        t.Tick();
        auto ret = net.TrainOnBatch(epoch, x, y);
        train_ms += t.Elapsed();
        l += ret.first;
        m += ret.second;
        if (++b % o.pfreq == 0) {
          std::snprintf(line, sizeof(line),
                        "Epoch %d, training loss = %f, accuracy = %f, lr = %g, "
                        "waiting for a batch of %zu images = %.1f ms, "
                        "training on it = %.1f ms", epoch, l / b, m / b,
                        lr * std::pow(0.1f, epoch / 20), o.batchsize,
                        stream.load_ms / b, train_ms / b);
          train_ch->Send(line);
          l = m = 0.f;
          train_ms = stream.load_ms = 0;
          b = 0;
        }
      }
    }
    {
      BatchStream stream(&data, kEval, {o.folder + "/test.bin"}, o.batchsize,
                         o.nthreads);
      Tensor x, y;
      float l = 0.f, m = 0.f;
      size_t seen = 0;
      while (seen + o.batchsize <= o.ntest && stream.Next(&x, &y, dev)) {
        auto ret = net.EvaluateOnBatch(x, y);
        l += Sum<float>(ret.first.Clone(defaultDevice));
        m += Sum<float>(ret.second.Clone(defaultDevice));
        seen += o.batchsize;
      }
      val_acc = seen ? m / seen : 0.f;
      std::snprintf(line, sizeof(line),
                    "Epoch %d, val loss = %f, accuracy = %f over %zu images",
                    epoch, seen ? l / seen : 0.f, val_acc, seen);
      val_ch->Send(line);
    }
    Checkpoint(net, o.folder + "/snapshot_epoch" + std::to_string(epoch));
  }
  return val_acc > 1.5f / o.classes ? 0 : 1;
}

}  // namespace singa

int main(int argc, char **argv) {
  singa::LogToStderr();
  singa::Options o;
  auto arg = [&](const char *name, const std::string &dflt) {
    int pos = singa::ArgPos(argc, argv, name);
    return pos != -1 && pos + 1 < argc ? std::string(argv[pos + 1]) : dflt;
  };
  bool synthetic = singa::ArgPos(argc, argv, "-synthetic") != -1;
  if (synthetic) {  // small defaults that finish in seconds on a CPU
    o.epochs = 3; o.batchsize = 32; o.file_size = 128; o.ntrain = 256;
    o.ntest = 64; o.pfreq = 4; o.size = 72; o.width = 0.125f; o.classes = 4;
    o.nthreads = 4; o.folder = "/tmp/singa_alexnet_synthetic";
  }
  o.epochs = std::atoi(arg("-epoch", std::to_string(o.epochs)).c_str());
  o.lr = std::atof(arg("-lr", std::to_string(o.lr)).c_str());
  o.batchsize = std::stoul(arg("-batchsize", std::to_string(o.batchsize)));
  o.file_size = std::stoul(arg("-filesize", std::to_string(o.file_size)));
  o.ntrain = std::stoul(arg("-ntrain", std::to_string(o.ntrain)));
  o.ntest = std::stoul(arg("-ntest", std::to_string(o.ntest)));
  o.pfreq = std::stoul(arg("-pfreq", std::to_string(o.pfreq)));
  o.nthreads = std::atoi(arg("-nthreads", std::to_string(o.nthreads)).c_str());
  o.classes = std::atoi(arg("-classes", std::to_string(o.classes)).c_str());
  o.size = std::stoul(arg("-size", std::to_string(o.size)));
  o.width = std::atof(arg("-width", std::to_string(o.width)).c_str());
  o.folder = arg("-data", o.folder);
  for (int i = 1; i < argc; ++i) o.cpu = o.cpu || std::string(argv[i]) == "-cpu";
  if (synthetic) {
    ::mkdir(o.folder.c_str(), 0755);
    singa::ILSVRC gen(o.size, o.size * 227 / 256);
    gen.CreateSyntheticData(o.folder, o.ntrain, o.ntest, o.file_size, o.classes);
  }
  return singa::Train(o);
}
