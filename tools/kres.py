#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of the extension, as hipcc reports them
(-Rpass-analysis=kernel-resource-usage).  usage: tools/kres.py [filter] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
flt = args[0] if args else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-c",
       os.path.join(ROOT, "varkoder_amd", "csrc", "vkimg.hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-58s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r in rows:
    if flt in r["name"]:
        print("%-58s %5s %5s %8s %4s %7s" % (r["name"][:58], r.get("vgpr"), r.get("sgpr"), r.get("scratch"), r.get("occ"), r.get("lds")))
