// Protobuf encodings of the configuration records in conf.h.
//
// The reference configures its C++ model API with protobuf messages
// (src/proto/model.proto) and hands them across language bindings as
// serialized strings: Layer::Setup(shape, proto_str), Optimizer::Setup(str),
// Layer::ToProtoStr() (include/singa/model/layer.h:40-54, 177-186,
// optimizer.h:44-52, 110-118, metric.h:36-42, initializer.h:30-36).  The
// records here are plain structs; this header gives them the two protobuf
// encodings with the field numbers and names of model.proto, without a
// protobuf dependency:
//
//   * the binary wire format  (ParseFromString / SerializeAsString), and
//   * the text format         (ParseFromText / DebugString), the form of the
//     hand-written `.conf` / `.prototxt` files.
//
// Only fields that differ from the message's defaults are written, as
// protobuf does for unset optional fields.  Fields of model.proto that have
// no counterpart in conf.h are skipped when read (the wire format carries
// their length; the text reader skips the value or the braced block and
// reports the names through `unknown`).  The legacy repeated forms
// `kernel_size` / `pad` / `stride` of ConvolutionConf and the single-value
// forms of PoolingConf are accepted on input and set both spatial extents.
#pragma once
#include <string>
#include <vector>

#include "conf.h"

namespace singa {

#define SINGA_CONF_CODEC(Msg)                                              \
  /* Binary wire format.  The record is reset to its defaults first. */    \
  bool ParseFromString(const std::string &wire, Msg *conf);                \
  std::string SerializeAsString(const Msg &conf);                          \
  /* Text format.  `error` receives "line N: what" on failure, `unknown`   \
     the dotted names of fields that were skipped. */                      \
  bool ParseFromText(const std::string &text, Msg *conf,                   \
                     std::string *error = nullptr,                         \
                     std::vector<std::string> *unknown = nullptr);         \
  std::string DebugString(const Msg &conf);

SINGA_CONF_CODEC(FillerConf)
SINGA_CONF_CODEC(RegularizerConf)
SINGA_CONF_CODEC(ConstraintConf)
SINGA_CONF_CODEC(ParamSpec)
SINGA_CONF_CODEC(LayerConf)
SINGA_CONF_CODEC(OptimizerConf)
SINGA_CONF_CODEC(MetricConf)
// src/proto/io.proto and the memory pool record of src/proto/core.proto
SINGA_CONF_CODEC(EncoderConf)
SINGA_CONF_CODEC(DecoderConf)
SINGA_CONF_CODEC(TransformerConf)
SINGA_CONF_CODEC(ImageRecord)
SINGA_CONF_CODEC(MemPoolConf)
#undef SINGA_CONF_CODEC

/// Whole-file helpers: false when the file cannot be read or does not parse.
bool ReadFileToString(const std::string &path, std::string *out);
bool WriteStringToFile(const std::string &path, const std::string &data);

template <typename Msg>
bool ReadProtoFromTextFile(const std::string &path, Msg *conf,
                           std::string *error = nullptr) {
  std::string text;
  if (!ReadFileToString(path, &text)) {
    if (error) *error = "cannot read " + path;
    return false;
  }
  return ParseFromText(text, conf, error);
}
template <typename Msg>
bool ReadProtoFromBinaryFile(const std::string &path, Msg *conf) {
  std::string wire;
  return ReadFileToString(path, &wire) && ParseFromString(wire, conf);
}
template <typename Msg>
bool WriteProtoToTextFile(const Msg &conf, const std::string &path) {
  return WriteStringToFile(path, DebugString(conf));
}
template <typename Msg>
bool WriteProtoToBinaryFile(const Msg &conf, const std::string &path) {
  return WriteStringToFile(path, SerializeAsString(conf));
}

}  // namespace singa
