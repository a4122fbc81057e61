#!/usr/bin/env python3
"""VGPRs / occupancy / spills / static LDS of every kernel in one csrc/*.hip file (cross-compiled: no GPU needed).
    tools/kernel_resources.py plx_block.hip [name-substring]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simplex_gp_amd", "csrc")
src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I../../include", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + (["-ffp-contract=off"] if src == "plx_build.hip" else [])
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(rows, names):
    if filt in name:
        r = rows[mangled]
        print(f"{name.split('(')[0][:70]:70s} VGPR {r.get('VGPRs'):>4} occ {r.get('Occupancy [waves/SIMD]'):>2} spill {r.get('VGPRs Spill'):>3} LDS {r.get('LDS Size [bytes/block]')}")
