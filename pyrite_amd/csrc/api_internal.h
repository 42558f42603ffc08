// api_internal.h -- what the host-side translation units of libpyrite_gpu.so share besides the public ABI.
#pragma once
#include <string>

#include "../../include/pyrite_gpu.h"

namespace pyr {

// Records the thread-local message pyr_last_error() returns and hands `code` back.
int api_fail(int code, const std::string& message);
// Device a scene was created on.
int scene_device(const PyrScene* scene);

} // namespace pyr
