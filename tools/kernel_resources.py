#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py [substring ...]   (compiles shogidrl_amd/csrc/ksh_lib.hip for gfx950, no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "shogidrl_amd", "csrc", "ksh_lib.hip")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c",
                        "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(d, "x.o"), src],
                       capture_output=True, text=True, cwd=d)
rows, cur = {}, None
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    k, _, v = m.group(1).partition(":")
    if k.strip() == "Function Name":
        cur = v.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k.strip()] = v.strip()
want = sys.argv[1:]
print("%-58s %5s %5s %7s %7s %8s %4s" % ("kernel", "VGPR", "SGPR", "spillV", "scratch", "LDS", "occ"))
for k, v in rows.items():
    if want and not any(w in k for w in want):
        continue
    print("%-58s %5s %5s %7s %7s %8s %4s" % (k[:58], v.get("VGPRs"), v.get("TotalSGPRs"), v.get("VGPRs Spill"),
                                             v.get("ScratchSize [bytes/lane]"), v.get("LDS Size [bytes/block]"),
                                             v.get("Occupancy [waves/SIMD]")))
