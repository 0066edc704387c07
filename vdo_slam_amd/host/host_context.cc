#include "host_context.h"

#include <cstdio>
#include <cstdlib>

namespace VDO_SLAM {
vdo_ctx* HostContext() {
  static vdo_ctx* ctx = nullptr;
  if (!ctx) {
    const char* dev = std::getenv("VDO_DEVICE");
    if (vdo_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != VDO_OK) {
      std::fprintf(stderr, "VDO_SLAM: cannot create the HIP context: %s\n", vdo_last_error());
      std::exit(-1);
    }
  }
  return ctx;
}
}  // namespace VDO_SLAM
