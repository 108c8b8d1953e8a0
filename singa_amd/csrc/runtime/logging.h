// Minimal glog-style logging: LOG(INFO|WARNING|ERROR|FATAL), CHECK* macros,
// stderr / file destinations.  FATAL throws std::runtime_error instead of
// aborting so that the Python process can surface the message.
//
// Behaviour parity: reference include/singa/utils/logging.h,
// src/utils/logging.cc (InitLogging, LogToStderr, SetStderrLogging,
// SetLogDestination; severities INFO=0 .. FATAL=3).
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>

namespace singa_amd {

enum LogSeverity { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };

void InitLogging(const std::string &argv0);
/// Send every message to stderr only.
void LogToStderr();
/// Messages at or above `severity` are copied to stderr in addition to files.
void SetStderrLogging(int severity);
/// Messages of `severity` are appended to `path`.
void SetLogDestination(int severity, const std::string &path);
void LogMessage(int severity, const std::string &msg, const std::string &file,
                int line);

class LogLine {
 public:
  LogLine(int severity, const char *file, int line)
      : severity_(severity), file_(file), line_(line) {}
  ~LogLine() noexcept(false) {
    LogMessage(severity_, ss_.str(), file_, line_);
    if (severity_ == FATAL) throw std::runtime_error(ss_.str());
  }
  std::ostringstream &stream() { return ss_; }

 private:
  int severity_;
  const char *file_;
  int line_;
  std::ostringstream ss_;
};

}  // namespace singa_amd

#define SA_LOG(sev) \
  ::singa_amd::LogLine(::singa_amd::sev, __FILE__, __LINE__).stream()
#define SA_CHECK(cond) \
  if (!(cond)) SA_LOG(FATAL) << "Check failed: " #cond " "
#define SA_CHECK_OP(a, b, op) \
  if (!((a)op(b)))            \
  SA_LOG(FATAL) << "Check failed: " #a " " #op " " #b " (" << (a) << " vs " << (b) << ") "
#define SA_CHECK_EQ(a, b) SA_CHECK_OP(a, b, ==)
#define SA_CHECK_NE(a, b) SA_CHECK_OP(a, b, !=)
#define SA_CHECK_LE(a, b) SA_CHECK_OP(a, b, <=)
#define SA_CHECK_LT(a, b) SA_CHECK_OP(a, b, <)
#define SA_CHECK_GE(a, b) SA_CHECK_OP(a, b, >=)
#define SA_CHECK_GT(a, b) SA_CHECK_OP(a, b, >)
