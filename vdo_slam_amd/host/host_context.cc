#include "host_context.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace VDO_SLAM {
vdo_ctx* HostContext() {
  static vdo_ctx* ctx = nullptr;
  if (!ctx) {
    const char* dev = std::getenv("VDO_DEVICE");
    if (vdo_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != VDO_OK) {
      ctx = nullptr;
      throw std::runtime_error(std::string("VDO_SLAM: cannot create the HIP context: ") + vdo_last_error());   // (not exit: see Frame.cc)
    }
  }
  return ctx;
}
}  // namespace VDO_SLAM
