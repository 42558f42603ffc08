"""A guard on the code the compiler makes of the BASELINE kernel, run on the CPU (hipcc cross-compiles): render_kernel_sm for the
big non-interpreter scenes sits at 128 VGPRs with ~100 uniform words live across its stage loop against 104 SGPRs, and what does
not fit is parked in VGPR lanes and fetched back with v_readlane -- vector instructions in the hot phases. Round 4 saw ONE more
uniform bool kept across the loop turn 140 v_readlane into 387 and C3 from 584 into 559 Msamples/s while every test stayed
green; this test would have said so without a GPU."""
import os
import re
import subprocess

import pytest

from pyrite_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_the_baseline_kernel_keeps_its_registers(tmp_path):
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    out = tmp_path / "sm.s"
    subprocess.check_call([build.HIPCC] + flags + ["--cuda-device-only", "-DPYR_DEV_ONLY_SM", "-S", "kernels.hip", "-o", str(out)], cwd=build.CSRC,
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    start = re.search(r"^_ZN3pyr16render_kernel_smILb0ELb0ELb0ELb1E\w*:", text, re.M)
    assert start, "render_kernel_sm<false, false, false, true> is not in the listing"
    body = text[start.start():]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.splitlines()]
    valu = sum(1 for l in lines if l.startswith("v_"))
    readlane = sum(1 for l in lines if l.startswith("v_readlane"))
    meta = text[text.index(".amdhsa_kernel", text.index("render_kernel_smILb0ELb0ELb0ELb1E")):]
    vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
    # round 4: 8,278 vector instructions, 140 v_readlane, 128 VGPRs, 272 B of scratch (the traversal stack's deep end: no spill)
    assert vgprs <= 128, vgprs
    assert scratch <= 272, scratch
    assert readlane <= 170, "SGPR spills grew: %d v_readlane (140 in round 4): a uniform value too many is live across the stage loop" % readlane
    assert valu <= 8500, valu
