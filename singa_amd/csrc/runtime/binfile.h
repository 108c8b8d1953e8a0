// Record file of (key, value) tuples.
//
// On-disk layout of one record (little endian, 64-bit lengths):
//   's' 'g' <has_key: 0|1> 0 | [u64 key_len | key bytes] | u64 val_len | val
//
// Format parity: reference src/io/binfile_writer.cc:47-83,
// src/io/binfile_reader.cc:43-134 (same magic word, same field order, records
// with an empty value are rejected by the writer).
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "logging.h"

namespace singa_amd {

class BinFileWriter {
 public:
  ~BinFileWriter() { Close(); }
  bool Open(const std::string &path, bool append = false,
            int capacity = 10485760) {
    SA_CHECK(!f_.is_open()) << "writer already open";
    path_ = path;
    buf_.clear();
    buf_.reserve(capacity);
    capacity_ = static_cast<size_t>(capacity);
    f_.open(path, std::ios::binary | std::ios::out |
                      (append ? std::ios::app : std::ios::trunc));
    return f_.is_open();
  }
  void Close() {
    if (f_.is_open()) {
      Flush();
      f_.close();
    }
  }
  bool Write(const std::string &key, const std::string &value) {
    SA_CHECK(f_.is_open()) << "File not open!";
    if (value.empty()) return false;
    size_t need = 4 + 8 + value.size() + (key.empty() ? 0 : 8 + key.size());
    SA_CHECK_LE(need, capacity_) << "record larger than the buffer capacity";
    if (buf_.size() + need > capacity_) Flush();
    const char magic[4] = {'s', 'g', static_cast<char>(key.empty() ? 0 : 1), 0};
    Append(magic, 4);
    if (!key.empty()) {
      uint64_t n = key.size();
      Append(&n, 8);
      Append(key.data(), key.size());
    }
    uint64_t n = value.size();
    Append(&n, 8);
    Append(value.data(), value.size());
    return true;
  }
  void Flush() {
    if (!buf_.empty() && f_.is_open()) {
      f_.write(buf_.data(), static_cast<std::streamsize>(buf_.size()));
      f_.flush();
      buf_.clear();
    }
  }
  const std::string &path() const { return path_; }

 private:
  void Append(const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    buf_.insert(buf_.end(), c, c + n);
  }
  std::string path_;
  std::ofstream f_;
  std::vector<char> buf_;
  size_t capacity_ = 0;
};

class BinFileReader {
 public:
  ~BinFileReader() { Close(); }
  bool Open(const std::string &path, int /*capacity*/ = 10485760) {
    path_ = path;
    f_.open(path, std::ios::binary | std::ios::in);
    if (!f_.is_open()) return false;
    f_.seekg(0, std::ios::end);
    size_ = static_cast<uint64_t>(f_.tellg());
    f_.seekg(0);
    return true;
  }
  void Close() {
    if (f_.is_open()) f_.close();
  }
  /// Reads the next record; returns false at end of file.
  bool Read(std::string *key, std::string *value) {
    SA_CHECK(f_.is_open()) << "File not open!";
    key->clear();
    value->clear();
    char magic[4];
    f_.read(magic, 4);
    if (f_.gcount() < 4) return false;
    SA_CHECK(magic[0] == 's' && magic[1] == 'g' &&
             (magic[2] == 0 || magic[2] == 1))
        << "File format error: magic word does not match!";
    if (magic[2] == 1 && !ReadField(key)) return false;
    return ReadField(value);
  }
  int Count() {
    std::ifstream fin(path_, std::ios::binary | std::ios::in);
    SA_CHECK(fin.is_open()) << "Cannot open file " << path_;
    int count = 0;
    fin.seekg(0, std::ios::end);
    const uint64_t size = static_cast<uint64_t>(fin.tellg());
    fin.seekg(0);
    uint64_t at = 0;
    // a length field that points beyond the end of the file ends the count
    auto skip_field = [&]() {
      uint64_t len = 0;
      fin.read(reinterpret_cast<char *>(&len), 8);
      if (!fin.good() || len > size - (at + 8)) return false;
      at += 8 + len;
      fin.seekg(static_cast<std::streamoff>(at));
      return fin.good();
    };
    while (at + 4 <= size) {
      char magic[4];
      fin.read(magic, 4);
      if (!fin.good() || magic[0] != 's' || magic[1] != 'g') break;
      at += 4;
      if (magic[2] == 1 && !skip_field()) break;
      if (!skip_field()) break;
      ++count;
    }
    return count;
  }
  void SeekToFirst() {
    SA_CHECK(f_.is_open());
    f_.clear();
    f_.seekg(0);
  }
  const std::string &path() const { return path_; }

 private:
  bool ReadField(std::string *out) {
    uint64_t len = 0;
    f_.read(reinterpret_cast<char *>(&len), 8);
    if (f_.gcount() < 8) return false;
    // a damaged length must not size an allocation: the field has to fit in
    // what is left of the file
    const uint64_t at = static_cast<uint64_t>(f_.tellg());
    if (at > size_ || len > size_ - at) return false;
    out->resize(len);
    f_.read(&(*out)[0], static_cast<std::streamsize>(len));
    return static_cast<uint64_t>(f_.gcount()) == len;
  }
  std::string path_;
  std::ifstream f_;
  uint64_t size_ = 0;
};

}  // namespace singa_amd
