// Process-wide device context of the host classes (the reference is single-process,
// single-thread, one System per process — SURVEY.md §8b "Threading").
#pragma once
#include "../../include/vdo_slam_hip.h"

namespace VDO_SLAM {
// Lazily creates the context on device 0 (env VDO_DEVICE overrides).  Exits like the reference
// does on unrecoverable setup errors (src/System.cc:35-39) — there is no CPU fallback.
vdo_ctx* HostContext();
}  // namespace VDO_SLAM
