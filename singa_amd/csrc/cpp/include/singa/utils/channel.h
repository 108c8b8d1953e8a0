// Named metric sinks of the C++ API (file and / or stderr), the runtime's
// channels under the C++ API's namespace.
// API parity: reference include/singa/utils/channel.h.
#pragma once
#include "../../../../runtime/channel.h"

namespace singa {
using ::singa_amd::Channel;
using ::singa_amd::ChannelManager;
using ::singa_amd::GetChannel;
using ::singa_amd::InitChannel;
using ::singa_amd::SetChannelDirectory;
}  // namespace singa
