// exact_math.h -- correctly rounded f32 square root (and nothing else yet) in fewer instructions than the compiler's expansion.
//
// The kernels must round like the CPU oracle (IEEE-754 sqrt), so they cannot use v_sqrt_f32 (1 ulp) alone. The compiler's
// sqrtf is v_sqrt_f32, the two neighbour residuals s -+ 1 ulp by fma, two selects -- and around that a scaling by 2^32 for
// arguments below 2^-96, the un-scaling, and a class test for 0 / inf: 16 vector instructions, of which the kernels' arguments
// never need the last seven. sqrt32 is the middle part alone (9 instructions). Verified on an MI355X for EVERY float bit
// pattern (tests/probes/exact_math_probe.hip, tests/test_gpu_exact_math.py): bit-identical to sqrtf -- including 0, -0, inf,
// NaN and negative arguments -- except for 0 < |x| <= 4.6e-32 (the range the scaling exists for), where it may be an ulp off.
// What the kernels take roots of are squared lengths, 1 - cos^2-like differences of numbers near 1 (zero or >= 2^-24) and
// radicands of the same kind: zero exactly, or far above 1e-31 for any scene whose features are larger than 1e-15 units (the
// reference itself ignores everything below DIST_EPSILON = 1e-4, math.rs:4).
#pragma once
#include <hip/hip_runtime.h>

namespace pyr {

__device__ __forceinline__ float sqrt32(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float residual_below = __builtin_fmaf(-below, s, x), residual_above = __builtin_fmaf(-above, s, x);
    float r = (0.0f >= residual_below) ? below : s;
    r = (0.0f < residual_above) ? above : r;
    return r;
}

} // namespace pyr
