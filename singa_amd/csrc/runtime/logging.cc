#include "logging.h"

#include <chrono>
#include <cstdio>
#include <ctime>
#include <fstream>
#include <iostream>
#include <mutex>

namespace singa_amd {
namespace {
std::mutex g_mu;
bool g_stderr_only = true;
int g_stderr_level = ERROR;
std::string g_dest[4];
std::string g_prog = "singa_amd";
const char kTag[4] = {'I', 'W', 'E', 'F'};
}  // namespace

void InitLogging(const std::string &argv0) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!argv0.empty()) g_prog = argv0;
}

void LogToStderr() {
  std::lock_guard<std::mutex> lk(g_mu);
  g_stderr_only = true;
}

void SetStderrLogging(int severity) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_stderr_level = severity;
}

void SetLogDestination(int severity, const std::string &path) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (severity < 0 || severity > 3) return;
  g_dest[severity] = path;
  g_stderr_only = false;
}

void LogMessage(int severity, const std::string &msg, const std::string &file,
                int line) {
  if (severity < 0) severity = 0;
  if (severity > 3) severity = 3;
  auto now = std::chrono::system_clock::now();
  std::time_t t = std::chrono::system_clock::to_time_t(now);
  std::tm tm_buf;
  localtime_r(&t, &tm_buf);
  char stamp[32];
  std::strftime(stamp, sizeof(stamp), "%m%d %H:%M:%S", &tm_buf);
  std::string base = file;
  auto pos = base.find_last_of('/');
  if (pos != std::string::npos) base = base.substr(pos + 1);
  std::ostringstream ss;
  ss << kTag[severity] << stamp << ' ' << base << ':' << line << "] " << msg
     << '\n';
  std::lock_guard<std::mutex> lk(g_mu);
  bool to_file = !g_stderr_only && !g_dest[severity].empty();
  if (to_file) {
    std::ofstream f(g_dest[severity], std::ios::app);
    f << ss.str();
  }
  if (!to_file || severity >= g_stderr_level) std::cerr << ss.str();
}

}  // namespace singa_amd
