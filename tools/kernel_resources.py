#!/usr/bin/env python3
"""Register / LDS / occupancy summary of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/kernel_resources.py quaff_amd/csrc/qf_fb.hip [name filter regex] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"] + sys.argv[3:]
if src.endswith("qf_fb.hip"):
    flags.append("-munsafe-fp-atomics")
out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src],
                     capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", out)[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not re.search(pat, dem):
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print("%-72s VGPR %3s AGPR %3s SGPR %3s scratch %4s occupancy %s LDS %6s" % (
        dem[:72], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
