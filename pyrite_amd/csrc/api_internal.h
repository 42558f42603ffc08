// api_internal.h -- what the host-side translation units of libpyrite_gpu.so share besides the public ABI.
#pragma once
#include <string>

#include "../../include/pyrite_gpu.h"

namespace pyr {

// Records the thread-local message pyr_last_error() returns and hands `code` back.
int api_fail(int code, const std::string& message);
// Device a scene was created on.
int scene_device(const PyrScene* scene);
// The device word the kernels set when a render's film is invalid (a path outgrew the spectral tape: 1; a wave
// gave up waiting on its workgroup's LDS queues: 2), or nullptr while the scene has none. Stays set until somebody clears it.
uint32_t* scene_overflow_word(PyrScene* scene);
// Reads that word after the scene's renders have been waited for: PYR_OK, or PYR_ERR_DEVICE with the message set (and the
// word cleared). What pyr_render_simple does before it hands the film back.
int scene_check_overflow(PyrScene* scene);

} // namespace pyr
