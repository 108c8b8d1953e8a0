// Checkpoint files "<prefix>.bin" + "<prefix>.desc".
//
// <prefix>.bin is a BinFile whose values are protobuf-encoded TensorProto
// messages (fields: 1 shape uint32 repeated, 2 data_type enum, 3 stride int32
// repeated, 4 float_data packed, 5 double_data packed, 6 int_data packed);
// <prefix>.desc is a text file with one human readable line per parameter.
// The wire encoding is written out by hand below, so no protobuf library is
// needed.
//
// Format parity: reference src/io/snapshot.cc:33-133, src/proto/core.proto:70-79,
// src/core/tensor/tensor.cc:307-417 (float32 / double / int payloads only; the
// reader falls back to "<prefix>.model").
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "binfile.h"
#include "textfile.h"

namespace singa_amd {

constexpr int kSingaVersion = 4200;

// data_type values: 0 float32, 1 float16, 2 int32, 3 char, 4 double
struct SnapshotTensor {
  std::vector<uint32_t> shape;
  std::vector<int32_t> stride;
  int dtype = 0;
  std::string data;  // raw little-endian elements (4 or 8 bytes each)
};

std::string EncodeTensorProto(const SnapshotTensor &t);
SnapshotTensor DecodeTensorProto(const std::string &bytes);

class Snapshot {
 public:
  Snapshot(const std::string &prefix, bool write, int max_param_size_mb = 10);
  ~Snapshot() { Close(); }
  void Write(const std::string &key, const SnapshotTensor &t);
  SnapshotTensor Read(const std::string &key) const;
  std::vector<uint32_t> ReadShape(const std::string &key) const;
  std::vector<std::string> Names() const;
  void Close();

 private:
  std::string prefix_;
  bool write_;
  std::unique_ptr<BinFileWriter> bin_w_;
  std::unique_ptr<TextFileWriter> txt_w_;
  std::map<std::string, SnapshotTensor> params_;
  std::vector<std::string> order_;
};

}  // namespace singa_amd
