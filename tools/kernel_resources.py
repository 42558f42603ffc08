#!/usr/bin/env python3
"""Developer tool: registers / scratch / occupancy of every kernel in kernels.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.   python tools/kernel_resources.py [extra -D flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyrite_amd import build  # noqa: E402

flags = [f for f in build.FLAGS if f != "-shared"] + sys.argv[1:]
cmd = [build.HIPCC] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "kernels.hip", "-o", "/dev/null"]
text = subprocess.run(cmd, cwd=build.CSRC, capture_output=True, text=True).stderr
for block in re.split(r"remark: Function Name: ", text)[1:]:
    name = block.split(" ")[0]

    def field(key):
        m = re.search(key + r": (\d+)", block)
        return int(m.group(1)) if m else -1

    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    demangled = re.sub(r"\(pyr::DevScene.*", "", demangled).replace("void pyr::", "")
    print("VGPR %3d  scratch %5d B  SGPR %3d  waves/SIMD %d  %s" % (field("VGPRs"), field(r"ScratchSize \[bytes/lane\]"), field("TotalSGPRs"),
                                                                  field(r"Occupancy \[waves/SIMD\]"), demangled))
