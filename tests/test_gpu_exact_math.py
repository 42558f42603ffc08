"""pyrite_amd/csrc/exact_math.h on the hardware: the kernels' short square root against the compiler's correctly rounded sqrtf
for EVERY f32 bit pattern (the CPU oracle takes IEEE square roots; GPU == oracle rests on this equality)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_short_square_root_is_correctly_rounded_wherever_the_kernels_use_it(gpu_lib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "probes")])
    out = subprocess.run([os.path.join(HERE, "probes", "exact_math_probe")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["sqrt_inputs"] == 2 ** 32
    # identical for 0, -0, inf, NaN, negative and every positive argument above 4.6e-32; below that (the range the compiler's
    # expansion rescales for) it may be an ulp off -- squared lengths under 4.6e-32 are lengths under 2.2e-16
    assert r["sqrt_mismatches"] == 0 or (0.0 < r["sqrt_mismatch_lowest_abs"] and r["sqrt_mismatch_highest_abs"] <= 4.6e-32), r
    # the short reciprocal (normalize: 1 / |v|): bit-identical to 1.0f / x for every x with 2^-126 <= |x| < 2^126, NaN stays NaN;
    # zero, denormal, huge and infinite arguments are where it differs (exact_math.h says why the kernels never pass one)
    assert r["rcp_inputs"] == 2 ** 32 and r["rcp_mismatches_in_range"] == 0, r
    assert 0 < r["rcp_mismatches_outside"] <= 3 * 2 ** 24, r
