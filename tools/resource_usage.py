#!/usr/bin/env python
"""Per-kernel VGPR / occupancy / LDS table from `hipcc -Rpass-analysis=kernel-resource-usage` output (CPU only):
    hipcc ... -Rpass-analysis=kernel-resource-usage -o /dev/null 2> ru.txt ; tools/resource_usage.py ru.txt [filter]"""
import re
import subprocess
import sys

cur, d = None, {}
keys = {"VGPRs": "vgpr", "AGPRs": "agpr", r"Occupancy \[waves/SIMD\]": "occ", r"LDS Size \[bytes/block\]": "lds",
        "VGPRs Spill": "spill", r"ScratchSize \[bytes/lane\]": "scratch"}
for l in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1)
        d[cur] = {}
    for k, short in keys.items():
        m = re.search(r"\s" + k + r": (\d+)", l)
        if m and cur:
            d[cur][short] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, v), n in zip(d.items(), names):
    n = n.split("(")[0].replace("void gt::", "")
    if flt in n:
        print("%-70s %s" % (n, " ".join("%s=%d" % kv for kv in v.items())))
