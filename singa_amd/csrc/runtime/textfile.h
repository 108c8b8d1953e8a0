// Line-oriented record file: Write() appends the value as one line, Read()
// returns (line number, line).  Parity: reference src/io/textfile_writer.cc,
// src/io/textfile_reader.cc.
#pragma once
#include <fstream>
#include <string>

#include "logging.h"

namespace singa_amd {

class TextFileWriter {
 public:
  ~TextFileWriter() { Close(); }
  bool Open(const std::string &path, bool append = false) {
    SA_CHECK(!f_.is_open());
    f_.open(path, append ? std::ios::app : (std::ios::out | std::ios::trunc));
    return f_.is_open();
  }
  void Close() {
    if (f_.is_open()) {
      f_.flush();
      f_.close();
    }
  }
  bool Write(const std::string & /*key*/, const std::string &value) {
    SA_CHECK(f_.is_open()) << "File not open!";
    if (value.empty()) return false;
    f_ << value << '\n';
    return true;
  }
  void Flush() {
    if (f_.is_open()) f_.flush();
  }

 private:
  std::ofstream f_;
};

class TextFileReader {
 public:
  ~TextFileReader() { Close(); }
  bool Open(const std::string &path) {
    path_ = path;
    f_.open(path, std::ios::in);
    return f_.is_open();
  }
  void Close() {
    if (f_.is_open()) f_.close();
  }
  bool Read(std::string *key, std::string *value) {
    SA_CHECK(f_.is_open()) << "File not open!";
    key->clear();
    value->clear();
    if (!std::getline(f_, *value)) return false;
    *key = std::to_string(line_no_++);
    return true;
  }
  int Count() {
    std::ifstream fin(path_);
    SA_CHECK(fin.is_open()) << "Cannot open file " << path_;
    int n = 0;
    std::string line;
    while (std::getline(fin, line))
      if (!line.empty()) ++n;
    return n;
  }
  void SeekToFirst() {
    SA_CHECK(f_.is_open());
    line_no_ = 0;
    f_.clear();
    f_.seekg(0);
  }

 private:
  std::string path_;
  std::ifstream f_;
  int line_no_ = 0;
};

}  // namespace singa_amd
