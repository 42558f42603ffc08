// exact_math.h -- correctly rounded f32 square root and reciprocal in fewer instructions than the compiler's expansions.
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

// 1.0f / x where neither x nor the quotient is zero, denormal, infinite or NaN. The compiler's division is v_div_scale x 2,
// v_rcp_f32, four fma, v_div_fmas, v_div_fixup: eleven vector instructions, nine of them one behind the other. For a
// RECIPROCAL, one Newton step on v_rcp_f32 (1 ulp) is already the correctly rounded quotient: checked on an MI355X for every
// float bit pattern (tests/probes/exact_math_probe.hip, tools/rcp_probe.hip) -- bit-identical to 1.0f / x for all biased
// exponents 1 .. 252, i.e. 2^-126 <= |x| < 2^126, and NaN stays NaN; outside that range it is not (v_rcp_f32 flushes denormals,
// and 0 * inf makes the residual NaN where IEEE gives inf or 0). The kernels use it for 1 / |v| in `normalize` -- |v| comes from
// sqrt32 above, i.e. it is exactly zero (the quotient is then multiplied by the zero vector: NaN either way) or above 2e-16 --
// and nowhere else. The triangle tests keep the compiler's division: there the short form was
// measured SLOWER (C3 566 against 580 Msamples/s) although it takes twelve instructions out of every leaf step.
// A general a / b is NOT a * rcp32(b): that rounds twice.
__device__ __forceinline__ float rcp32(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

} // namespace pyr
