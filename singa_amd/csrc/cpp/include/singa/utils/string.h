// Small string helpers.  API parity: reference include/singa/utils/string.h
// (ToLowerCase, ArgPos, split, trim, Tokenizer).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "logging.h"

namespace singa {

inline std::string ToLowerCase(const std::string &input) {
  std::string out = input;
  std::transform(out.begin(), out.end(), out.begin(),
                 [](unsigned char c) { return std::tolower(c); });
  return out;
}

/// Index of `str` in argv (checked to have a following value), or -1.
/// "[a, b, c]" (reference include/singa/utils/string.h, VecToStr).
template <typename T>
inline std::string VecToStr(const std::vector<T> &in) {
  std::string out = "[";
  for (size_t i = 0; i < in.size(); ++i)
    out += (i ? ", " : "") + std::to_string(in[i]);
  return out + "]";
}

inline int ArgPos(int argc, char **arglist, const char *str) {
  for (int i = 0; i < argc; i++) {
    if (!std::strcmp(str, arglist[i])) {
      CHECK_LT(i, argc - 1) << "Argument missing for " << str;
      return i;
    }
  }
  return -1;
}

template <typename T>
inline std::vector<T> SplitString(const std::string &s, char delim) {
  std::vector<T> out;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, delim)) {
    if (item.empty()) continue;
    std::stringstream conv(item);
    T v;
    conv >> v;
    out.push_back(v);
  }
  return out;
}

inline std::string Trim(const std::string &s) {
  size_t b = s.find_first_not_of(" \t\r\n");
  if (b == std::string::npos) return "";
  size_t e = s.find_last_not_of(" \t\r\n");
  return s.substr(b, e - b + 1);
}

/// Splits on any character of `sep`; `tok >> str` yields successive tokens.
class Tokenizer {
 public:
  Tokenizer(const std::string &str, const std::string &sep)
      : start_(0), sep_(sep), buf_(str) {}
  Tokenizer &operator>>(std::string &out) {
    CHECK_LT(start_, buf_.length());
    size_t b = buf_.find_first_not_of(sep_, start_);
    if (b == std::string::npos) {
      out.clear();
      start_ = buf_.length();
      return *this;
    }
    size_t e = buf_.find_first_of(sep_, b);
    if (e == std::string::npos) e = buf_.length();
    out = buf_.substr(b, e - b);
    start_ = e;
    size_t nxt = buf_.find_first_not_of(sep_, start_);
    start_ = nxt == std::string::npos ? buf_.length() : nxt;
    return *this;
  }
  bool Valid() const { return start_ < buf_.length(); }

 private:
  size_t start_;
  std::string sep_;
  std::string buf_;
};

}  // namespace singa
