// Configuration records of the C++ model API.  They carry the fields of the
// reference's protobuf messages (src/proto/model.proto: LayerConf, ParamSpec,
// FillerConf, OptimizerConf, ...) as plain structs, so the C++ API has no
// protobuf dependency; field names are the proto field names.
#pragma once
#include <cstddef>
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

// ---- records of the data pipeline (reference src/proto/io.proto) -----------
// `type` names the record format: "jpg2proto" / "proto2jpg" (the reference's
// names) keep a JPEG file inside an ImageRecord message, "jpg" is the JPEG
// file itself with the label in a comment segment.
struct EncoderConf {
  std::string type = "jpg2proto";
  std::string image_dim_order = "HWC";
};
struct DecoderConf {
  std::string type = "proto2jpg";
  std::string image_dim_order = "CHW";
  bool has_label = true;
};
struct TransformerConf {
  bool featurewise_center = false, samplewise_center = false;
  bool featurewise_std_norm = false, samplewise_std_norm = false;
  bool zca_whitening = false;
  int rotation_range = 0;
  // the message's repeated `crop_shape` (height, width)
  int crop_height = 0, crop_width = 0;
  int resize_height = 0, resize_width = 0;
  bool horizontal_mirror = false, vertical_mirror = false;
  std::string image_dim_order = "CHW";
  float rescale = 0.f;
};
/// One stored image: the extents of the tensor it was made from, its labels
/// and the encoded pixels.
struct ImageRecord {
  std::vector<int> shape, label;
  std::string pixel;
};

// ---- device memory pool (reference src/proto/core.proto:52-67) --------------
struct MemPoolConf {
  std::string type = "cnmem";   // kept for files written for the reference
  size_t init_size_mb = 256;    // first slab requested from the driver
  size_t max_size_mb = 0;       // 0 = unlimited
  int flag = 0;
  std::vector<int> device;
};

}  // namespace singa
