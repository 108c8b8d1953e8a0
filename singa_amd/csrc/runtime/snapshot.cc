#include "snapshot.h"

#include <algorithm>
#include <cstring>

namespace singa_amd {
namespace {

void PutVarint(std::string *out, uint64_t v) {
  while (v >= 0x80) {
    out->push_back(static_cast<char>((v & 0x7F) | 0x80));
    v >>= 7;
  }
  out->push_back(static_cast<char>(v));
}

bool GetVarint(const std::string &in, size_t *pos, uint64_t *v) {
  *v = 0;
  int shift = 0;
  while (*pos < in.size() && shift < 64) {
    uint8_t b = static_cast<uint8_t>(in[(*pos)++]);
    *v |= static_cast<uint64_t>(b & 0x7F) << shift;
    if (!(b & 0x80)) return true;
    shift += 7;
  }
  return false;
}

inline void PutTag(std::string *out, int field, int wire) {
  PutVarint(out, static_cast<uint64_t>((field << 3) | wire));
}

}  // namespace

std::string EncodeTensorProto(const SnapshotTensor &t) {
  std::string out;
  for (uint32_t s : t.shape) {
    PutTag(&out, 1, 0);
    PutVarint(&out, s);
  }
  PutTag(&out, 2, 0);
  PutVarint(&out, static_cast<uint64_t>(t.dtype));
  for (int32_t s : t.stride) {
    PutTag(&out, 3, 0);
    // negative int32 values are sign-extended to 64 bits on the wire
    PutVarint(&out, static_cast<uint64_t>(static_cast<int64_t>(s)));
  }
  switch (t.dtype) {
    case 0:  // float32 -> packed fixed32
      PutTag(&out, 4, 2);
      PutVarint(&out, t.data.size());
      out.append(t.data);
      break;
    case 4:  // double -> packed fixed64
      PutTag(&out, 5, 2);
      PutVarint(&out, t.data.size());
      out.append(t.data);
      break;
    case 2: {  // int32 -> packed varints
      std::string payload;
      size_t n = t.data.size() / 4;
      for (size_t i = 0; i < n; ++i) {
        int32_t v;
        std::memcpy(&v, t.data.data() + 4 * i, 4);
        PutVarint(&payload, static_cast<uint64_t>(static_cast<int64_t>(v)));
      }
      PutTag(&out, 6, 2);
      PutVarint(&out, payload.size());
      out.append(payload);
      break;
    }
    default:
      SA_LOG(FATAL) << "Unsupported data type for serialisation: " << t.dtype;
  }
  return out;
}

SnapshotTensor DecodeTensorProto(const std::string &in) {
  SnapshotTensor t;
  size_t pos = 0;
  auto append_scalar = [&](int field, uint64_t v) {
    if (field == 1) t.shape.push_back(static_cast<uint32_t>(v));
    else if (field == 2) t.dtype = static_cast<int>(v);
    else if (field == 3) t.stride.push_back(static_cast<int32_t>(v));
    else if (field == 6) {
      int32_t x = static_cast<int32_t>(v);
      t.data.append(reinterpret_cast<const char *>(&x), 4);
    }
  };
  while (pos < in.size()) {
    uint64_t key;
    SA_CHECK(GetVarint(in, &pos, &key)) << "truncated TensorProto";
    int field = static_cast<int>(key >> 3), wire = static_cast<int>(key & 7);
    if (wire == 0) {
      uint64_t v;
      SA_CHECK(GetVarint(in, &pos, &v)) << "truncated TensorProto";
      append_scalar(field, v);
    } else if (wire == 2) {
      uint64_t len;
      SA_CHECK(GetVarint(in, &pos, &len)) << "truncated TensorProto";
      SA_CHECK_LE(pos + len, in.size()) << "truncated TensorProto";
      if (field == 4 || field == 5) {
        t.data.append(in, pos, len);
      } else if (field == 1 || field == 3 || field == 6) {
        // packed varints
        size_t end = pos + len, p = pos;
        while (p < end) {
          uint64_t v;
          SA_CHECK(GetVarint(in, &p, &v)) << "truncated TensorProto";
          append_scalar(field, v);
        }
      }
      pos += len;
    } else if (wire == 5) {  // non-packed float
      SA_CHECK_LE(pos + 4, in.size());
      if (field == 4) t.data.append(in, pos, 4);
      pos += 4;
    } else if (wire == 1) {  // non-packed double
      SA_CHECK_LE(pos + 8, in.size());
      if (field == 5) t.data.append(in, pos, 8);
      pos += 8;
    } else {
      SA_LOG(FATAL) << "unsupported wire type " << wire;
    }
  }
  return t;
}

Snapshot::Snapshot(const std::string &prefix, bool write, int max_param_size_mb)
    : prefix_(prefix), write_(write) {
  if (write_) {
    bin_w_ = std::make_unique<BinFileWriter>();
    txt_w_ = std::make_unique<TextFileWriter>();
    SA_CHECK(bin_w_->Open(prefix + ".bin", false, max_param_size_mb << 20))
        << "Cannot create " << prefix << ".bin";
    SA_CHECK(txt_w_->Open(prefix + ".desc", false))
        << "Cannot create " << prefix << ".desc";
    txt_w_->Write("", "SINGA VERSION: " + std::to_string(kSingaVersion));
  } else {
    BinFileReader reader;
    if (!reader.Open(prefix + ".bin"))
      SA_CHECK(reader.Open(prefix + ".model"))
          << "Cannot open the checkpoint bin file: " << prefix
          << ".bin or " << prefix << ".model";
    std::string key, val;
    while (reader.Read(&key, &val)) {
      SA_CHECK(params_.count(key) == 0) << "duplicate parameter " << key;
      params_[key] = DecodeTensorProto(val);
      order_.push_back(key);
    }
  }
}

void Snapshot::Write(const std::string &key, const SnapshotTensor &t) {
  SA_CHECK(write_) << "snapshot opened for reading";
  SA_CHECK(params_.count(key) == 0) << "duplicate parameter " << key;
  params_[key] = SnapshotTensor();  // remember the name only
  order_.push_back(key);
  bin_w_->Write(key, EncodeTensorProto(t));
  std::string desc = "parameter name: " + key;
  desc += "\tdata type: " + std::to_string(t.dtype);
  desc += "\tdim: " + std::to_string(t.shape.size());
  desc += "\tshape:";
  for (uint32_t s : t.shape) desc += " " + std::to_string(s);
  txt_w_->Write(key, desc);
}

SnapshotTensor Snapshot::Read(const std::string &key) const {
  SA_CHECK(!write_) << "snapshot opened for writing";
  auto it = params_.find(key);
  SA_CHECK(it != params_.end()) << "no parameter named " << key;
  return it->second;
}

std::vector<uint32_t> Snapshot::ReadShape(const std::string &key) const {
  return Read(key).shape;
}

std::vector<std::string> Snapshot::Names() const {
  std::vector<std::string> names;
  if (write_) return order_;
  // the reference keeps a std::map, i.e. names come back sorted
  for (auto &kv : params_) names.push_back(kv.first);
  return names;
}

void Snapshot::Close() {
  if (bin_w_) bin_w_->Close();
  if (txt_w_) txt_w_->Close();
}

}  // namespace singa_amd
