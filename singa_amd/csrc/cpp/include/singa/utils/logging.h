// glog-style macros of the C++ API, backed by the runtime logger.
// API parity: reference include/singa/utils/logging.h.
#pragma once
#include "../../../../runtime/logging.h"

namespace singa {
using ::singa_amd::InitLogging;
using ::singa_amd::LogToStderr;
using ::singa_amd::SetLogDestination;
using ::singa_amd::SetStderrLogging;
}  // namespace singa

#ifndef LOG
#define LOG(sev) SA_LOG(sev)
#define CHECK(c) SA_CHECK(c)
#define CHECK_EQ(a, b) SA_CHECK_EQ(a, b)
#define CHECK_NE(a, b) SA_CHECK_NE(a, b)
#define CHECK_LE(a, b) SA_CHECK_LE(a, b)
#define CHECK_LT(a, b) SA_CHECK_LT(a, b)
#define CHECK_GE(a, b) SA_CHECK_GE(a, b)
#define CHECK_GT(a, b) SA_CHECK_GT(a, b)
#define CHECK_NOTNULL(p) \
  (((p) == nullptr) ? (SA_LOG(FATAL) << #p " is null", (p)) : (p))
#endif
