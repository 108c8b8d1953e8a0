// Binary wire format and text format of the configuration records, driven by
// one field table per message (the Describe functions below): each table is
// a sequence of calls on a visitor, and four visitors -- wire writer, wire
// reader, text writer, text reader -- give the four codecs.  Field numbers,
// names and defaults are those of the reference's src/proto/model.proto
// (FillerConf :55, OptimizerConf :77, ParamSpec :117, LayerConf :156,
// ConvolutionConf :338, RNNConf :398, DenseConf :603, PoolingConf :658, ...),
// src/proto/io.proto (EncoderConf :23, DecoderConf :28, TransformerConf :34,
// ImageRecord :55) and src/proto/core.proto (MemPoolConf :52);
// the tables list fields in ascending number, the order protobuf writes.
#include "singa/model/conf_proto.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>

namespace singa {
namespace {

using EnumNames = std::vector<std::pair<const char *, int>>;
using AliasSetter = std::function<void(int value, int occurrence)>;
constexpr int kMaxDepth = 32;

// ---------------------------------------------------------------------------
// field tables
// ---------------------------------------------------------------------------
template <class V> void Describe(V &v, FillerConf &c) {
  v.Str(1, "type", c.type, "constant");
  v.F32(2, "value", c.value, 0.f);
  v.F32(3, "min", c.min, 0.f);
  v.F32(4, "max", c.max, 1.f);
  v.F32(5, "mean", c.mean, 0.f);
  v.F32(6, "std", c.std, 1.f);
}
template <class V> void Describe(V &v, RegularizerConf &c) {
  v.Str(1, "type", c.type, "l2");
  v.F32(2, "coefficient", c.coefficient, 0.f);
}
template <class V> void Describe(V &v, ConstraintConf &c) {
  v.Str(1, "type", c.type, "l2");
  v.F32(2, "threshold", c.threshold, 0.f);
}
template <class V> void Describe(V &v, ParamSpec &c) {
  v.Str(1, "name", c.name, "");
  v.F32(3, "lr_mult", c.lr_mult, 1.f);
  v.F32(4, "decay_mult", c.decay_mult, 1.f);
  v.Msg(20, "filler", c.filler);
  // the record keeps the two sub-messages as (present, value) pairs
  ConstraintConf con;
  con.threshold = c.constraint_threshold;
  v.MsgHas(21, "constraint", con, c.has_constraint);
  c.constraint_threshold = con.threshold;
  RegularizerConf reg;
  reg.coefficient = c.regularizer_coefficient;
  v.MsgHas(22, "regularizer", reg, c.has_regularizer);
  c.regularizer_coefficient = reg.coefficient;
}
template <class V> void Describe(V &v, OptimizerConf &c) {
  v.Str(1, "type", c.type, "sgd");
  v.F32(2, "rho", c.rho, 0.95f);
  // the reference applies momentum only when the field is present
  // (src/model/optimizer/sgd.cc: has_momentum()); absent reads as 0 here
  v.F32(5, "momentum", c.momentum, 0.f);
  v.F32(6, "delta", c.delta, 1e-8f);
  v.MsgHas(10, "regularizer", c.regularizer, c.has_regularizer);
  v.MsgHas(11, "constraint", c.constraint, c.has_constraint);
}
template <class V> void Describe(V &v, MetricConf &c) {
  v.U32(1, "top_k", c.top_k, 1);
}
template <class V> void Describe(V &v, ConcatConf &c) {
  v.Alias(1, "concat_dim", [&c](int x, int) { c.axis = x; });
  v.I32(2, "axis", c.axis, 1);
}
template <class V> void Describe(V &v, ConvolutionConf &c) {
  v.U32(1, "num_output", c.num_output, 0);
  v.Bool(2, "bias_term", c.bias_term, true);
  // repeated legacy forms: one value for both extents, a second one for the
  // width
  v.Alias(3, "pad", [&c](int x, int k) {
    if (k == 0) c.pad_h = x;
    c.pad_w = x;
  });
  v.Alias(4, "kernel_size", [&c](int x, int k) {
    if (k == 0) c.kernel_h = x;
    c.kernel_w = x;
  });
  // 5 is the group count of the caffe schema; model.proto leaves the number
  // free
  v.U32(5, "group", c.group, 1);
  v.Alias(6, "stride", [&c](int x, int k) {
    if (k == 0) c.stride_h = x;
    c.stride_w = x;
  });
  v.U32(9, "pad_h", c.pad_h, 0);
  v.U32(10, "pad_w", c.pad_w, 0);
  v.U32(11, "kernel_h", c.kernel_h, 0);
  v.U32(12, "kernel_w", c.kernel_w, 0);
  v.U32(13, "stride_h", c.stride_h, 1);
  v.U32(14, "stride_w", c.stride_w, 1);
}
template <class V> void Describe(V &v, DropoutConf &c) {
  v.F32(1, "dropout_ratio", c.dropout_ratio, 0.5f);
}
template <class V> void Describe(V &v, DenseConf &c) {
  v.U32(1, "num_output", c.num_output, 0);
  v.Bool(2, "bias_term", c.bias_term, true);
  v.Bool(21, "transpose", c.transpose, false);
}
template <class V> void Describe(V &v, LRNConf &c) {
  v.U32(1, "local_size", c.local_size, 5);
  v.F32(2, "alpha", c.alpha, 1.f);
  v.F32(3, "beta", c.beta, 0.75f);
  v.F32(5, "k", c.k, 1.f);
}
template <class V> void Describe(V &v, PoolingConf &c) {
  static const EnumNames kMethods = {{"MAX", 0}, {"AVE", 1}};
  int method = (int)c.pool;
  v.Enum(1, "pool", method, 0, kMethods);
  c.pool = method == 1 ? PoolingConf::AVE : PoolingConf::MAX;
  v.Alias(2, "kernel_size", [&c](int x, int) { c.kernel_h = c.kernel_w = x; });
  v.Alias(3, "stride", [&c](int x, int) { c.stride_h = c.stride_w = x; });
  v.Alias(4, "pad", [&c](int x, int) { c.pad_h = c.pad_w = x; });
  v.U32(5, "kernel_h", c.kernel_h, 0);
  v.U32(6, "kernel_w", c.kernel_w, 0);
  v.U32(7, "stride_h", c.stride_h, 1);
  v.U32(8, "stride_w", c.stride_w, 1);
  v.U32(9, "pad_h", c.pad_h, 0);
  v.U32(10, "pad_w", c.pad_w, 0);
}
template <class V> void Describe(V &v, ReLUConf &c) {
  v.F32(1, "negative_slope", c.negative_slope, 0.f);
}
template <class V> void Describe(V &v, SoftmaxConf &c) {
  v.I32(2, "axis", c.axis, 1);
}
template <class V> void Describe(V &v, SliceConf &c) {
  v.Alias(1, "slice_dim", [&c](int x, int) { c.axis = x; });
  v.RepU32(2, "slice_point", c.slice_point);
  v.I32(3, "axis", c.axis, 1);
}
template <class V> void Describe(V &v, PReLUConf &c) {
  v.Msg(1, "filler", c.filler);
  v.Bool(2, "channel_shared", c.channel_shared, false);
  v.Str(20, "format", c.format, "NCHW");
}
template <class V> void Describe(V &v, FlattenConf &c) {
  v.I32(1, "axis", c.axis, 1);
}
template <class V> void Describe(V &v, RNNConf &c) {
  v.U32(1, "hidden_size", c.hidden_size, 0);
  v.U32(2, "num_stacks", c.num_stacks, 1);
  v.F32(3, "dropout", c.dropout, 0.f);
  v.Bool(4, "remember_state", c.remember_state, false);
  v.Str(7, "input_mode", c.input_mode, "linear");
  v.Str(8, "direction", c.direction, "unidirectional");
  v.Str(9, "rnn_mode", c.rnn_mode, "relu");
}
template <class V> void Describe(V &v, BatchNormConf &c) {
  v.F64(1, "factor", c.factor, 0.9);
}
template <class V> void Describe(V &v, SplitConf &c) {
  v.I32(1, "output_size", c.output_size, 2);
}
template <class V> void Describe(V &v, LayerConf &c) {
  v.Str(1, "name", c.name, "");
  v.Str(2, "type", c.type, "");
  v.RepMsg(6, "param", c.param);
  v.Msg(104, "concat_conf", c.concat_conf);
  v.Msg(106, "convolution_conf", c.convolution_conf);
  v.Msg(108, "dropout_conf", c.dropout_conf);
  v.Msg(117, "dense_conf", c.dense_conf);
  v.Msg(118, "lrn_conf", c.lrn_conf);
  v.Msg(121, "pooling_conf", c.pooling_conf);
  v.Msg(123, "relu_conf", c.relu_conf);
  v.Msg(125, "softmax_conf", c.softmax_conf);
  v.Msg(126, "slice_conf", c.slice_conf);
  v.Msg(131, "prelu_conf", c.prelu_conf);
  v.Msg(135, "flatten_conf", c.flatten_conf);
  v.Msg(140, "rnn_conf", c.rnn_conf);
  v.Msg(202, "batchnorm_conf", c.batchnorm_conf);
  v.Msg(203, "split_conf", c.split_conf);
}

template <class V> void Describe(V &v, EncoderConf &c) {
  v.Str(1, "type", c.type, "jpg2proto");
  v.Str(2, "image_dim_order", c.image_dim_order, "HWC");
}
template <class V> void Describe(V &v, DecoderConf &c) {
  v.Str(1, "type", c.type, "proto2jpg");
  v.Str(2, "image_dim_order", c.image_dim_order, "CHW");
  v.Bool(3, "has_label", c.has_label, true);
}
template <class V> void Describe(V &v, TransformerConf &c) {
  v.Bool(1, "featurewise_center", c.featurewise_center, false);
  v.Bool(2, "samplewise_center", c.samplewise_center, false);
  v.Bool(3, "featurewise_std_norm", c.featurewise_std_norm, false);
  v.Bool(4, "samplewise_std_norm", c.samplewise_std_norm, false);
  v.Bool(5, "zca_whitening", c.zca_whitening, false);
  v.I32(6, "rotation_range", c.rotation_range, 0);
  // crop_shape = [height, width]; the record keeps the two numbers
  std::vector<int> crop;
  if (c.crop_height || c.crop_width) crop = {c.crop_height, c.crop_width};
  const size_t before = crop.size();
  v.PackedU32(7, "crop_shape", crop);
  if (crop.size() > before) {  // a reader appended the values it met
    const size_t n = crop.size() - before;
    if (before == 0) {
      c.crop_height = crop[0];
      c.crop_width = n > 1 ? crop[1] : crop[0];
    } else {  // a second occurrence continues the list
      c.crop_width = crop[before];
    }
  }
  v.I32(8, "resize_height", c.resize_height, 0);
  v.I32(9, "resize_width", c.resize_width, 0);
  v.Bool(10, "horizontal_mirror", c.horizontal_mirror, false);
  v.Bool(11, "vertical_mirror", c.vertical_mirror, false);
  v.Str(12, "image_dim_order", c.image_dim_order, "CHW");
  v.F32(13, "rescale", c.rescale, 0.f);
}
template <class V> void Describe(V &v, ImageRecord &c) {
  v.RepI32(1, "shape", c.shape);
  v.RepI32(2, "label", c.label);
  v.Str(3, "pixel", c.pixel, "");
}
template <class V> void Describe(V &v, MemPoolConf &c) {
  v.Str(1, "type", c.type, "cnmem");
  int init = (int)c.init_size_mb, max = (int)c.max_size_mb;
  v.U32(2, "init_size", init, 256);
  v.U32(3, "max_size", max, 0);
  c.init_size_mb = (size_t)(uint32_t)init;
  c.max_size_mb = (size_t)(uint32_t)max;
  v.U32(11, "flag", c.flag, 0);
  v.RepU32(12, "device", c.device);
}

// ---------------------------------------------------------------------------
// wire format
// ---------------------------------------------------------------------------
enum { WT_VARINT = 0, WT_FIXED64 = 1, WT_LEN = 2, WT_FIXED32 = 5 };

void PutVarint(std::string *out, uint64_t v) {
  while (v >= 0x80) {
    out->push_back((char)(v | 0x80));
    v >>= 7;
  }
  out->push_back((char)v);
}

struct WireWriter {
  std::string out;
  void Tag(int num, int wt) { PutVarint(&out, ((uint64_t)num << 3) | wt); }
  void Bytes(int num, const std::string &b) {
    Tag(num, WT_LEN);
    PutVarint(&out, b.size());
    out += b;
  }
  void Str(int num, const char *, std::string &v, const char *dflt) {
    if (v != dflt) Bytes(num, v);
  }
  void F32(int num, const char *, float &v, float dflt) {
    if (v == dflt) return;
    Tag(num, WT_FIXED32);
    uint32_t bits;
    memcpy(&bits, &v, 4);
    for (int i = 0; i < 4; ++i) out.push_back((char)(bits >> (8 * i)));
  }
  void F64(int num, const char *, double &v, double dflt) {
    if (v == dflt) return;
    Tag(num, WT_FIXED64);
    uint64_t bits;
    memcpy(&bits, &v, 8);
    for (int i = 0; i < 8; ++i) out.push_back((char)(bits >> (8 * i)));
  }
  void Int(int num, int64_t v) {
    Tag(num, WT_VARINT);
    PutVarint(&out, (uint64_t)v);  // negative values: ten bytes, as protobuf
  }
  void I32(int num, const char *, int &v, int dflt) {
    if (v != dflt) Int(num, v);
  }
  void U32(int num, const char *, int &v, int dflt) {
    if (v != dflt) Int(num, (uint32_t)v);
  }
  void Bool(int num, const char *, bool &v, bool dflt) {
    if (v != dflt) Int(num, v ? 1 : 0);
  }
  void Enum(int num, const char *, int &v, int dflt, const EnumNames &) {
    if (v != dflt) Int(num, v);
  }
  void RepU32(int num, const char *, std::vector<int> &v) {
    for (int x : v) Int(num, (uint32_t)x);
  }
  void RepI32(int num, const char *, std::vector<int> &v) {
    for (int x : v) Int(num, x);
  }
  void PackedU32(int num, const char *, std::vector<int> &v) {
    if (v.empty()) return;
    std::string body;
    for (int x : v) PutVarint(&body, (uint32_t)x);
    Bytes(num, body);
  }
  void Alias(int, const char *, const AliasSetter &) {}
  template <class Sub> void Msg(int num, const char *, Sub &sub) {
    WireWriter w;
    Describe(w, sub);
    if (!w.out.empty()) Bytes(num, w.out);
  }
  template <class Sub> void MsgHas(int num, const char *, Sub &sub, bool &has) {
    if (!has) return;
    WireWriter w;
    Describe(w, sub);
    Bytes(num, w.out);
  }
  template <class Sub>
  void RepMsg(int num, const char *, std::vector<Sub> &subs) {
    for (auto &s : subs) {
      WireWriter w;
      Describe(w, s);
      Bytes(num, w.out);
    }
  }
};

bool GetVarint(const uint8_t *&p, const uint8_t *end, uint64_t *v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 70 && p < end; shift += 7) {
    const uint8_t b = *p++;
    if (shift < 64) r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}

struct WireReader {
  int depth = 0;
  bool ok = true;
  // the field being delivered
  int num = 0, wt = 0;
  uint64_t value = 0;           // varint / fixed payload
  const uint8_t *data = nullptr;  // length-delimited payload
  size_t len = 0;
  bool taken = false;
  std::map<int, int> occurrences;

  template <class M> bool Parse(const uint8_t *p, size_t n, M &m) {
    if (depth > kMaxDepth) return false;
    const uint8_t *end = p + n;
    while (p < end) {
      uint64_t tag;
      if (!GetVarint(p, end, &tag)) return false;
      num = (int)(tag >> 3);
      wt = (int)(tag & 7);
      if (num <= 0 || (tag >> 3) > 0x1fffffff) return false;
      switch (wt) {
        case WT_VARINT:
          if (!GetVarint(p, end, &value)) return false;
          break;
        case WT_FIXED64:
          if (end - p < 8) return false;
          value = 0;
          for (int i = 0; i < 8; ++i) value |= (uint64_t)p[i] << (8 * i);
          p += 8;
          break;
        case WT_FIXED32:
          if (end - p < 4) return false;
          value = 0;
          for (int i = 0; i < 4; ++i) value |= (uint64_t)p[i] << (8 * i);
          p += 4;
          break;
        case WT_LEN: {
          uint64_t l;
          if (!GetVarint(p, end, &l) || l > (uint64_t)(end - p)) return false;
          data = p;
          len = (size_t)l;
          p += l;
          break;
        }
        default:
          return false;  // groups are not part of the schema
      }
      taken = false;
      Describe(*this, m);
      if (!ok) return false;
      // a field nobody took is one the records do not carry: skipped
    }
    return true;
  }
  bool Mine(int n, int want) {
    if (taken || n != num) return false;
    taken = true;  // a number belongs to one field whatever its wire type
    return wt == want;  // another wire type: skipped like an unknown field
  }
  void Str(int n, const char *, std::string &v, const char *) {
    if (Mine(n, WT_LEN)) v.assign((const char *)data, len);
  }
  void F32(int n, const char *, float &v, float) {
    if (!Mine(n, WT_FIXED32)) return;
    const uint32_t bits = (uint32_t)value;
    memcpy(&v, &bits, 4);
  }
  void F64(int n, const char *, double &v, double) {
    if (Mine(n, WT_FIXED64)) memcpy(&v, &value, 8);
  }
  void I32(int n, const char *, int &v, int) {
    if (Mine(n, WT_VARINT)) v = (int)(int64_t)value;
  }
  void U32(int n, const char *, int &v, int) {
    if (Mine(n, WT_VARINT)) v = (int)(uint32_t)value;
  }
  void Bool(int n, const char *, bool &v, bool) {
    if (Mine(n, WT_VARINT)) v = value != 0;
  }
  void Enum(int n, const char *, int &v, int, const EnumNames &) {
    if (Mine(n, WT_VARINT)) v = (int)(int64_t)value;
  }
  // repeated scalars arrive one by one or packed into one length-delimited
  // field
  template <class F> void Scalars(int n, F &&each) {
    if (taken || n != num) return;
    taken = true;
    if (wt == WT_VARINT) {
      each((int)(uint32_t)value);
    } else if (wt == WT_LEN) {
      const uint8_t *p = data, *end = data + len;
      while (p < end) {
        uint64_t x;
        if (!GetVarint(p, end, &x)) {
          ok = false;
          return;
        }
        each((int)(uint32_t)x);
      }
    } else {
      ok = false;
    }
  }
  void RepU32(int n, const char *, std::vector<int> &v) {
    Scalars(n, [&v](int x) { v.push_back(x); });
  }
  void RepI32(int n, const char *, std::vector<int> &v) {
    Scalars(n, [&v](int x) { v.push_back(x); });
  }
  void PackedU32(int n, const char *, std::vector<int> &v) {
    Scalars(n, [&v](int x) { v.push_back(x); });
  }
  void Alias(int n, const char *, const AliasSetter &set) {
    Scalars(n, [&](int x) { set(x, occurrences[n]++); });
  }
  template <class Sub> void Msg(int n, const char *, Sub &sub) {
    if (!Mine(n, WT_LEN)) return;
    WireReader r;
    r.depth = depth + 1;
    if (!r.Parse(data, len, sub)) ok = false;
  }
  template <class Sub> void MsgHas(int n, const char *, Sub &sub, bool &has) {
    if (taken || n != num) return;
    has = true;
    Msg(n, nullptr, sub);
  }
  template <class Sub> void RepMsg(int n, const char *, std::vector<Sub> &subs) {
    if (!Mine(n, WT_LEN)) return;
    subs.emplace_back();
    WireReader r;
    r.depth = depth + 1;
    if (!r.Parse(data, len, subs.back())) ok = false;
  }
};

// ---------------------------------------------------------------------------
// text format
// ---------------------------------------------------------------------------
std::string Quote(const std::string &s) {
  std::string q = "\"";
  for (unsigned char ch : s) {
    switch (ch) {
      case '\n': q += "\\n"; break;
      case '\r': q += "\\r"; break;
      case '\t': q += "\\t"; break;
      case '"': q += "\\\""; break;
      case '\'': q += "\\'"; break;
      case '\\': q += "\\\\"; break;
      default:
        if (ch < 0x20 || ch >= 0x7f) {
          char buf[8];
          snprintf(buf, sizeof buf, "\\%03o", ch);
          q += buf;
        } else {
          q.push_back((char)ch);
        }
    }
  }
  return q + "\"";
}

// shortest decimal form that reads back to the same value
std::string Shortest(float v) {
  if (std::isnan(v)) return "nan";
  if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
  char buf[40];
  for (int prec = 6; prec <= 9; ++prec) {
    snprintf(buf, sizeof buf, "%.*g", prec, v);
    if (strtof(buf, nullptr) == v) break;
  }
  return buf;
}
std::string Shortest(double v) {
  if (std::isnan(v)) return "nan";
  if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
  char buf[48];
  for (int prec = 15; prec <= 17; ++prec) {
    snprintf(buf, sizeof buf, "%.*g", prec, v);
    if (strtod(buf, nullptr) == v) break;
  }
  return buf;
}

struct TextWriter {
  std::string out;
  int indent = 0;
  void Line(const char *name, const std::string &value) {
    out.append(2 * indent, ' ');
    out += name;
    out += ": ";
    out += value;
    out += "\n";
  }
  void Str(int, const char *name, std::string &v, const char *dflt) {
    if (v != dflt) Line(name, Quote(v));
  }
  void F32(int, const char *name, float &v, float dflt) {
    if (!(v == dflt)) Line(name, Shortest(v));
  }
  void F64(int, const char *name, double &v, double dflt) {
    if (!(v == dflt)) Line(name, Shortest(v));
  }
  void I32(int, const char *name, int &v, int dflt) {
    if (v != dflt) Line(name, std::to_string(v));
  }
  void U32(int, const char *name, int &v, int dflt) {
    if (v != dflt) Line(name, std::to_string((uint32_t)v));
  }
  void Bool(int, const char *name, bool &v, bool dflt) {
    if (v != dflt) Line(name, v ? "true" : "false");
  }
  void Enum(int, const char *name, int &v, int dflt, const EnumNames &names) {
    if (v == dflt) return;
    for (const auto &n : names)
      if (n.second == v) return Line(name, n.first);
    Line(name, std::to_string(v));
  }
  void RepU32(int, const char *name, std::vector<int> &v) {
    for (int x : v) Line(name, std::to_string((uint32_t)x));
  }
  void RepI32(int, const char *name, std::vector<int> &v) {
    for (int x : v) Line(name, std::to_string(x));
  }
  void PackedU32(int n, const char *name, std::vector<int> &v) {
    RepU32(n, name, v);
  }
  void Alias(int, const char *, const AliasSetter &) {}
  template <class Sub> void Block(const char *name, Sub &sub, bool even_empty) {
    TextWriter w;
    w.indent = indent + 1;
    Describe(w, sub);
    if (w.out.empty() && !even_empty) return;
    out.append(2 * indent, ' ');
    out += name;
    out += " {\n";
    out += w.out;
    out.append(2 * indent, ' ');
    out += "}\n";
  }
  template <class Sub> void Msg(int, const char *name, Sub &sub) {
    Block(name, sub, false);
  }
  template <class Sub> void MsgHas(int, const char *name, Sub &sub, bool &has) {
    if (has) Block(name, sub, true);
  }
  template <class Sub>
  void RepMsg(int, const char *name, std::vector<Sub> &subs) {
    for (auto &s : subs) Block(name, s, true);
  }
};

// Tokens of the text format: words (names, numbers, enum values), quoted
// strings and single punctuation characters; '#' starts a comment.
struct Lexer {
  const std::string &s;
  size_t pos = 0;
  int line = 1;
  enum Kind { END, WORD, STRING, PUNCT } kind = END;
  std::string text;
  std::string error;

  explicit Lexer(const std::string &src) : s(src) { Next(); }

  void Fail(const std::string &what) {
    if (error.empty()) error = "line " + std::to_string(line) + ": " + what;
    kind = END;
  }
  static bool WordChar(char c) {
    return isalnum((unsigned char)c) || c == '_' || c == '.' || c == '-' ||
           c == '+';
  }
  void Next() {
    if (!error.empty()) return;
    for (;;) {
      while (pos < s.size() && isspace((unsigned char)s[pos])) {
        if (s[pos] == '\n') ++line;
        ++pos;
      }
      if (pos < s.size() && s[pos] == '#') {
        while (pos < s.size() && s[pos] != '\n') ++pos;
        continue;
      }
      break;
    }
    text.clear();
    if (pos >= s.size()) {
      kind = END;
      return;
    }
    const char c = s[pos];
    if (c == '"' || c == '\'') {
      kind = STRING;
      ++pos;
      while (pos < s.size() && s[pos] != c) {
        char ch = s[pos++];
        if (ch == '\n') return Fail("line break inside a string");
        if (ch != '\\') {
          text.push_back(ch);
          continue;
        }
        if (pos >= s.size()) break;
        ch = s[pos++];
        switch (ch) {
          case 'n': text.push_back('\n'); break;
          case 'r': text.push_back('\r'); break;
          case 't': text.push_back('\t'); break;
          case 'a': text.push_back('\a'); break;
          case 'b': text.push_back('\b'); break;
          case 'f': text.push_back('\f'); break;
          case 'v': text.push_back('\v'); break;
          case 'x': case 'X': {
            int v = 0, n = 0;
            while (n < 2 && pos < s.size() && isxdigit((unsigned char)s[pos])) {
              const char h = s[pos++];
              v = v * 16 + (isdigit((unsigned char)h) ? h - '0'
                                                      : (tolower(h) - 'a' + 10));
              ++n;
            }
            if (n == 0) return Fail("bad \\x escape");
            text.push_back((char)v);
            break;
          }
          default:
            if (ch >= '0' && ch <= '7') {
              int v = ch - '0', n = 1;
              while (n < 3 && pos < s.size() && s[pos] >= '0' && s[pos] <= '7') {
                v = v * 8 + (s[pos++] - '0');
                ++n;
              }
              text.push_back((char)v);
            } else {
              text.push_back(ch);  // \\ \" \' \?
            }
        }
      }
      if (pos >= s.size()) return Fail("unterminated string");
      ++pos;
      return;
    }
    if (WordChar(c)) {
      kind = WORD;
      while (pos < s.size() && WordChar(s[pos])) text.push_back(s[pos++]);
      return;
    }
    kind = PUNCT;
    text.push_back(c);
    ++pos;
  }
  bool Is(char c) const { return kind == PUNCT && text[0] == c; }
  bool Eat(char c) {
    if (!Is(c)) return false;
    Next();
    return true;
  }
};

struct TextReader {
  Lexer *lex;
  int depth = 0;
  std::string path;  // dotted name of the message being read
  std::vector<std::string> *unknown = nullptr;
  std::string key;
  bool taken = false;
  std::map<std::string, int> occurrences;

  bool ok() const { return lex->error.empty(); }

  // '}' or '>' for a nested message, 0 at the top level
  template <class M> void ParseMessage(M &m, char close) {
    if (depth > kMaxDepth) return lex->Fail("messages nested too deeply");
    while (ok()) {
      if (close ? lex->Is(close) : lex->kind == Lexer::END) break;
      if (lex->kind == Lexer::END) return lex->Fail("missing closing bracket");
      if (lex->Is('[')) {  // [extension.name]: none exist in this schema
        key = "[";
        while (ok() && !lex->Is(']') && lex->kind != Lexer::END) {
          lex->Next();
        }
        lex->Eat(']');
        Skip();
      } else {
        if (lex->kind != Lexer::WORD)
          return lex->Fail("field name expected, got '" + lex->text + "'");
        key = lex->text;
        lex->Next();
        taken = false;
        Describe(*this, m);
        if (!ok()) return;
        if (!taken) Skip();
      }
      if (!lex->Eat(',')) lex->Eat(';');
    }
    if (close && ok()) lex->Next();
  }

  // ---- skipping what the records do not carry ----
  void SkipBlock() {
    int level = 0;
    do {
      if (lex->kind == Lexer::END) return lex->Fail("missing closing bracket");
      if (lex->Is('{') || lex->Is('<') || lex->Is('[')) ++level;
      if (lex->Is('}') || lex->Is('>') || lex->Is(']')) --level;
      lex->Next();
    } while (level > 0 && ok());
  }
  void Skip() {
    if (unknown) unknown->push_back(path.empty() ? key : path + "." + key);
    const bool colon = lex->Eat(':');
    if (lex->Is('{') || lex->Is('<') || lex->Is('[')) return SkipBlock();
    if (!colon) return lex->Fail("':' expected after '" + key + "'");
    if (lex->kind == Lexer::STRING) {
      while (lex->kind == Lexer::STRING) lex->Next();
    } else if (lex->kind == Lexer::WORD) {
      lex->Next();
    } else {
      lex->Fail("value expected after '" + key + "'");
    }
  }

  // ---- scalar values ----
  bool Mine(const char *name) {
    if (taken || key != name) return false;
    taken = true;
    return true;
  }
  bool Colon() {
    if (lex->Eat(':')) return true;
    lex->Fail("':' expected after '" + key + "'");
    return false;
  }
  bool Number(double *v) {
    if (lex->kind != Lexer::WORD) {
      lex->Fail("number expected for '" + key + "'");
      return false;
    }
    std::string w = lex->text;
    std::string low;
    for (char c : w) low.push_back((char)tolower(c));
    const bool neg = !low.empty() && low[0] == '-';
    const std::string mag = neg ? low.substr(1) : low;
    if (mag == "inf" || mag == "infinity") {
      *v = neg ? -INFINITY : INFINITY;
    } else if (mag == "nan") {
      *v = NAN;
    } else {
      if (w.size() > 1 && (w.back() == 'f' || w.back() == 'F') &&
          low.compare(0, 2, "0x") != 0 && low.compare(0, 3, "-0x") != 0)
        w.pop_back();
      char *end = nullptr;
      if (low.compare(0, 2, "0x") == 0 || low.compare(0, 3, "-0x") == 0)
        *v = (double)strtoll(w.c_str(), &end, 16);
      else
        *v = strtod(w.c_str(), &end);
      if (end == w.c_str() || *end) {
        lex->Fail("'" + lex->text + "' is not a number (field '" + key + "')");
        return false;
      }
    }
    lex->Next();
    return true;
  }
  bool Integer(int64_t *v) {
    if (lex->kind != Lexer::WORD) {
      lex->Fail("integer expected for '" + key + "'");
      return false;
    }
    char *end = nullptr;
    *v = strtoll(lex->text.c_str(), &end, 0);
    if (end == lex->text.c_str() || *end) {
      lex->Fail("'" + lex->text + "' is not an integer (field '" + key + "')");
      return false;
    }
    lex->Next();
    return true;
  }
  void Str(int, const char *name, std::string &v, const char *) {
    if (!Mine(name) || !Colon()) return;
    if (lex->kind != Lexer::STRING)
      return lex->Fail("quoted string expected for '" + key + "'");
    v.clear();
    while (lex->kind == Lexer::STRING) {  // adjacent pieces concatenate
      v += lex->text;
      lex->Next();
    }
  }
  void F32(int, const char *name, float &v, float) {
    double d;
    if (Mine(name) && Colon() && Number(&d)) v = (float)d;
  }
  void F64(int, const char *name, double &v, double) {
    double d;
    if (Mine(name) && Colon() && Number(&d)) v = d;
  }
  void I32(int, const char *name, int &v, int) {
    int64_t x;
    if (!Mine(name) || !Colon() || !Integer(&x)) return;
    if (x < INT32_MIN || x > INT32_MAX)
      return lex->Fail("value of '" + key + "' does not fit 32 bits");
    v = (int)x;
  }
  void U32(int, const char *name, int &v, int) {
    int64_t x;
    if (!Mine(name) || !Colon() || !Integer(&x)) return;
    if (x < 0 || x > INT32_MAX)
      return lex->Fail("value of '" + key + "' is out of range");
    v = (int)x;
  }
  void Bool(int, const char *name, bool &v, bool) {
    if (!Mine(name) || !Colon()) return;
    const std::string &w = lex->text;
    if (lex->kind == Lexer::WORD &&
        (w == "true" || w == "True" || w == "t" || w == "1"))
      v = true;
    else if (lex->kind == Lexer::WORD &&
             (w == "false" || w == "False" || w == "f" || w == "0"))
      v = false;
    else
      return lex->Fail("true or false expected for '" + key + "'");
    lex->Next();
  }
  void Enum(int, const char *name, int &v, int, const EnumNames &names) {
    if (!Mine(name) || !Colon()) return;
    if (lex->kind != Lexer::WORD)
      return lex->Fail("enum value expected for '" + key + "'");
    for (const auto &n : names)
      if (lex->text == n.first) {
        v = n.second;
        lex->Next();
        return;
      }
    int64_t x;
    if (Integer(&x)) {
      for (const auto &n : names)
        if (n.second == x) {
          v = (int)x;
          return;
        }
      lex->Fail("unknown value for '" + key + "'");
    }
  }
  // one value, or a bracketed list
  template <class F>
  void Scalars(const char *name, F &&each, int64_t lowest = 0) {
    if (!Mine(name) || !Colon()) return;
    auto one = [&] {
      int64_t x;
      if (!Integer(&x)) return false;
      if (x < lowest || x > INT32_MAX) {
        lex->Fail("value of '" + key + "' is out of range");
        return false;
      }
      each((int)x);
      return true;
    };
    if (!lex->Eat('[')) {
      one();
      return;
    }
    while (ok() && !lex->Is(']')) {
      if (!one()) return;
      if (!lex->Eat(',') && !lex->Is(']'))
        return lex->Fail("',' or ']' expected in the list of '" + key + "'");
    }
    lex->Eat(']');
  }
  void RepU32(int, const char *name, std::vector<int> &v) {
    Scalars(name, [&v](int x) { v.push_back(x); });
  }
  void RepI32(int, const char *name, std::vector<int> &v) {
    Scalars(name, [&v](int x) { v.push_back(x); }, INT32_MIN);
  }
  void PackedU32(int, const char *name, std::vector<int> &v) {
    Scalars(name, [&v](int x) { v.push_back(x); });
  }
  void Alias(int, const char *name, const AliasSetter &set) {
    Scalars(name, [&](int x) { set(x, occurrences[name]++); });
  }
  template <class Sub> void Nested(const char *name, Sub &sub) {
    lex->Eat(':');  // optional before a message
    char close;
    if (lex->Eat('{'))
      close = '}';
    else if (lex->Eat('<'))
      close = '>';
    else
      return lex->Fail("'{' expected after '" + key + "'");
    TextReader r;
    r.lex = lex;
    r.depth = depth + 1;
    r.unknown = unknown;
    r.path = path.empty() ? name : path + "." + name;
    r.ParseMessage(sub, close);
  }
  template <class Sub> void Msg(int, const char *name, Sub &sub) {
    if (Mine(name)) Nested(name, sub);
  }
  template <class Sub> void MsgHas(int, const char *name, Sub &sub, bool &has) {
    if (!Mine(name)) return;
    has = true;
    Nested(name, sub);
  }
  template <class Sub>
  void RepMsg(int, const char *name, std::vector<Sub> &subs) {
    if (!Mine(name)) return;
    subs.emplace_back();
    Nested(name, subs.back());
  }
};

template <class M> bool FromWire(const std::string &wire, M *conf) {
  *conf = M();
  WireReader r;
  return r.Parse((const uint8_t *)wire.data(), wire.size(), *conf);
}
template <class M> std::string ToWire(const M &conf) {
  M copy = conf;
  WireWriter w;
  Describe(w, copy);
  return w.out;
}
template <class M>
bool FromText(const std::string &text, M *conf, std::string *error,
              std::vector<std::string> *unknown) {
  *conf = M();
  Lexer lex(text);
  TextReader r;
  r.lex = &lex;
  r.unknown = unknown;
  r.ParseMessage(*conf, 0);
  if (!lex.error.empty()) {
    if (error) *error = lex.error;
    return false;
  }
  return true;
}
template <class M> std::string ToText(const M &conf) {
  M copy = conf;
  TextWriter w;
  Describe(w, copy);
  return w.out;
}

}  // namespace

#define SINGA_CONF_CODEC_IMPL(Msg)                                          \
  bool ParseFromString(const std::string &wire, Msg *conf) {                \
    return FromWire(wire, conf);                                            \
  }                                                                         \
  std::string SerializeAsString(const Msg &conf) { return ToWire(conf); }   \
  bool ParseFromText(const std::string &text, Msg *conf, std::string *error, \
                     std::vector<std::string> *unknown) {                   \
    return FromText(text, conf, error, unknown);                            \
  }                                                                         \
  std::string DebugString(const Msg &conf) { return ToText(conf); }

SINGA_CONF_CODEC_IMPL(FillerConf)
SINGA_CONF_CODEC_IMPL(RegularizerConf)
SINGA_CONF_CODEC_IMPL(ConstraintConf)
SINGA_CONF_CODEC_IMPL(ParamSpec)
SINGA_CONF_CODEC_IMPL(LayerConf)
SINGA_CONF_CODEC_IMPL(OptimizerConf)
SINGA_CONF_CODEC_IMPL(MetricConf)
SINGA_CONF_CODEC_IMPL(EncoderConf)
SINGA_CONF_CODEC_IMPL(DecoderConf)
SINGA_CONF_CODEC_IMPL(TransformerConf)
SINGA_CONF_CODEC_IMPL(ImageRecord)
SINGA_CONF_CODEC_IMPL(MemPoolConf)
#undef SINGA_CONF_CODEC_IMPL

bool ReadFileToString(const std::string &path, std::string *out) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::ostringstream ss;
  ss << in.rdbuf();
  *out = ss.str();
  return true;
}

bool WriteStringToFile(const std::string &path, const std::string &data) {
  std::ofstream out(path, std::ios::binary | std::ios::trunc);
  if (!out) return false;
  out.write(data.data(), (std::streamsize)data.size());
  return (bool)out;
}

}  // namespace singa
