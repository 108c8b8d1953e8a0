// Configuration records of the C++ model API.  They carry the fields of the
// reference's protobuf messages (src/proto/model.proto: LayerConf, ParamSpec,
// FillerConf, OptimizerConf, ...) as plain structs, so the C++ API has no
// protobuf dependency; field names are the proto field names.
#pragma once
#include <string>
#include <vector>

namespace singa {

enum Phase { kTrain = 4, kEval = 8 };

struct FillerConf {
  std::string type = "constant";  // constant, uniform, gaussian, xavier, msra
  float value = 0.f;
  float min = 0.f, max = 1.f;
  float mean = 0.f, std = 1.f;
};

struct ParamSpec {
  std::string name;
  float lr_mult = 1.f;
  float decay_mult = 1.f;
  FillerConf filler;
  bool has_regularizer = false, has_constraint = false;
  float regularizer_coefficient = 0.f;   // L2
  float constraint_threshold = 0.f;      // L2 norm clip
};

struct DenseConf {
  int num_output = 0;
  bool bias_term = true;
  bool transpose = false;  // weight stored as (out, in) when true
};

struct ConvolutionConf {
  int num_output = 0;
  bool bias_term = true;
  int kernel_h = 0, kernel_w = 0;
  int stride_h = 1, stride_w = 1;
  int pad_h = 0, pad_w = 0;
  int group = 1;
  void set_kernel(int k) { kernel_h = kernel_w = k; }
  void set_stride(int s) { stride_h = stride_w = s; }
  void set_pad(int p) { pad_h = pad_w = p; }
};

struct PoolingConf {
  enum PoolMethod { MAX = 0, AVE = 1 };
  PoolMethod pool = MAX;
  int kernel_h = 0, kernel_w = 0;
  int stride_h = 1, stride_w = 1;
  int pad_h = 0, pad_w = 0;
  void set_kernel(int k) { kernel_h = kernel_w = k; }
  void set_stride(int s) { stride_h = stride_w = s; }
  void set_pad(int p) { pad_h = pad_w = p; }
};

struct ReLUConf { float negative_slope = 0.f; };
struct DropoutConf { float dropout_ratio = 0.5f; };
struct BatchNormConf { double factor = 0.9; };
struct LRNConf { int local_size = 5; float alpha = 1.f, beta = 0.75f, k = 1.f; };
struct SoftmaxConf { int axis = 1; };
struct FlattenConf { int axis = 1; };
struct ConcatConf { int axis = 1; };
struct SliceConf { int axis = 1; std::vector<int> slice_point; };
struct SplitConf { int output_size = 2; };
struct PReLUConf { bool channel_shared = false; std::string format = "NCHW";
                   FillerConf filler; };
struct RNNConf {
  int hidden_size = 0, num_stacks = 1;
  float dropout = 0.f;
  bool remember_state = false;
  std::string input_mode = "linear", direction = "unidirectional",
              rnn_mode = "relu";  // relu, tanh, lstm, gru
};

struct LayerConf {
  std::string name;
  std::string type;
  std::vector<ParamSpec> param;
  DenseConf dense_conf;
  ConvolutionConf convolution_conf;
  PoolingConf pooling_conf;
  ReLUConf relu_conf;
  DropoutConf dropout_conf;
  BatchNormConf batchnorm_conf;
  LRNConf lrn_conf;
  SoftmaxConf softmax_conf;
  FlattenConf flatten_conf;
  ConcatConf concat_conf;
  SliceConf slice_conf;
  SplitConf split_conf;
  PReLUConf prelu_conf;
  RNNConf rnn_conf;
};

struct RegularizerConf { std::string type = "l2"; float coefficient = 0.f; };
struct ConstraintConf { std::string type = "l2"; float threshold = 0.f; };

struct OptimizerConf {
  std::string type = "sgd";
  float momentum = 0.f;
  float delta = 1e-8f;   // epsilon of AdaGrad / RMSProp
  float rho = 0.95f;     // RMSProp decay
  bool has_regularizer = false, has_constraint = false;
  RegularizerConf regularizer;
  ConstraintConf constraint;
};

struct MetricConf { int top_k = 1; };

}  // namespace singa
