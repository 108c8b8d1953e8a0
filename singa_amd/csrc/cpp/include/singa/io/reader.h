// Record readers of the C++ API, the counterparts of io/writer.h.
// API parity: reference include/singa/io/reader.h (Reader, BinFileReader,
// TextFileReader, LMDBReader: the LMDB file format is parsed by the runtime's
// own code, runtime/lmdbfile.h, no library is needed).
#pragma once
#include <string>

#include "../../../../runtime/binfile.h"
#include "../../../../runtime/lmdbfile.h"
#include "../../../../runtime/textfile.h"

namespace singa {
namespace io {

class Reader {
 public:
  virtual ~Reader() {}
  virtual bool Open(const std::string &path) = 0;
  virtual void Close() = 0;
  /// false at the end of the file
  virtual bool Read(std::string *key, std::string *value) = 0;
  /// number of tuples in the file; the read position is kept
  virtual int Count() = 0;
  virtual void SeekToFirst() = 0;
};

class BinFileReader : public Reader {
 public:
  ~BinFileReader() override { Close(); }
  bool Open(const std::string &path) override { return impl_.Open(path); }
  bool Open(const std::string &path, int capacity) {
    return impl_.Open(path, capacity);
  }
  void Close() override { impl_.Close(); }
  bool Read(std::string *key, std::string *value) override {
    return impl_.Read(key, value);
  }
  int Count() override { return impl_.Count(); }
  void SeekToFirst() override { impl_.SeekToFirst(); }

 private:
  ::singa_amd::BinFileReader impl_;
};

/// The key of a line is its number, counted from 0.
class TextFileReader : public Reader {
 public:
  ~TextFileReader() override { Close(); }
  bool Open(const std::string &path) override { return impl_.Open(path); }
  void Close() override { impl_.Close(); }
  bool Read(std::string *key, std::string *value) override {
    return impl_.Read(key, value);
  }
  int Count() override { return impl_.Count(); }
  void SeekToFirst() override { impl_.SeekToFirst(); }

 private:
  ::singa_amd::TextFileReader impl_;
};

/// Tuples of an LMDB environment in key order.
class LMDBReader : public Reader {
 public:
  ~LMDBReader() override { Close(); }
  bool Open(const std::string &path) override { return impl_.Open(path); }
  void Close() override { impl_.Close(); }
  bool Read(std::string *key, std::string *value) override {
    return impl_.Read(key, value);
  }
  int Count() override { return impl_.Count(); }
  void SeekToFirst() override { impl_.SeekToFirst(); }
  const std::string &path() const { return impl_.path(); }
  /// whether the last Read found a tuple
  bool valid() const { return impl_.valid(); }

 private:
  ::singa_amd::LMDBReader impl_;
};

}  // namespace io
}  // namespace singa
