#!/usr/bin/env python3
"""Compile-time resource usage of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: tools/kres.py chain_heavy [chain_front ...] [-- extra hipcc flags]"""
import re, subprocess, sys, os
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "web-audio-mastering_amd", "csrc")
for f in args:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c",
                          os.path.join(src, f + ".hip"), "-o", "/tmp/kres_" + f + ".o"] + extra, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: +([A-Za-z][\w \[\]/]*): (.+?) \[-Rpass", line)
        if not m:
            if "error" in line: print(line)
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ("Function Name", "Name"):
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        cur[k] = v
        if k.startswith("LDS Size"):
            print(f"{cur['name'][:60]:60s} vgpr {cur.get('VGPRs'):>4s} agpr {cur.get('AGPRs'):>3s} scratch {cur.get('ScratchSize [bytes/lane]'):>4s} occ {cur.get('Occupancy [waves/SIMD]'):>2s} "
                  f"sspill {cur.get('SGPRs Spill'):>3s} vspill {cur.get('VGPRs Spill'):>3s} lds {v:>6s}")
