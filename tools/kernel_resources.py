#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (VGPR/AGPR/scratch/occupancy/LDS per kernel)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
K = {"v": r"VGPRs", "a": r"AGPRs", "scr": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]",
     "lds": r"LDS Size \[bytes/block\]", "s": r"SGPRs"}
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()
    vals = {}
    for k, rx in K.items():
        m = re.search(rx + r': (\d+)', b)
        vals[k] = int(m.group(1)) if m else -1
    nm = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    nm = nm.replace('(anonymous namespace)::', '').replace('(rbnn_activation)', '')
    nm = re.sub(r'\((anonymous namespace::)?\w+Args( const)?\)', '', nm)[:100]
    if pat in nm:
        print(f"{nm:100s} " + " ".join(f"{k}={v}" for k, v in vals.items()))
