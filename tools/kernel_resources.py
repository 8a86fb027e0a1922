#!/usr/bin/env python3
"""Summarise `-Rpass-analysis=kernel-resource-usage` output (make -C csrc asm): registers, scratch, occupancy."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/asm/bconv.resources.txt"
only = sys.argv[2] if len(sys.argv) > 2 else ""
t = open(path).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    n = n.replace("void bnn::", "").split("(")[0]
    if only and only not in n:
        continue
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{n:100s} v={g('VGPRs'):3d} a={g('AGPRs'):3d} s={g('SGPRs'):3d} scratch={scratch:4d} occ={occ}")
