#!/usr/bin/env python3
"""A/B timing of builds on the k=9 spill path (BASELINE config 4: 100 distinct samples of 1M x 150 bp):
python tools/k9_ab.py lib1.so lib2.so ...   (each library in its own child process; default library first)"""
import subprocess
import sys

CHILD = r"""
import sys, time, hashlib
sys.path.insert(0, ".")
import numpy as np, torch
from varkoder_amd import _capi
if sys.argv[1] != "default":
    _capi.LIB_PATH = sys.argv[1]
from varkoder_amd.engine import ImageEngine
k, n, dist = 9, 100, int(sys.argv[2])
eng = ImageEngine(k=k, mapping="cgr")
fq, offs, lens = eng.synth(0, n, 1_000_000, 150, dist=dist)
hist = torch.empty((n, 4 ** k), dtype=torch.int32, device="cuda")
status = torch.empty((n,), dtype=torch.int32, device="cuda")
eng.count(fq, offs, lens, hist=hist, status=status); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.count(fq, offs, lens, hist=hist, status=status); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
h = hashlib.sha256(hist[:16].cpu().numpy().tobytes()).hexdigest()[:16]
print(f"{sys.argv[1]:28s} k=9 dist={dist} count {min(ts)*1e3:7.2f} ms (min of 5; {[round(t*1e3,2) for t in ts]}) bad={int((status!=0).sum())} sha={h}", flush=True)
"""

if __name__ == "__main__":
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    dist = 0
    for a in sys.argv[1:]:
        if a.startswith("--dist="):
            dist = int(a[7:])
    rc = 0
    for lib in ["default"] + libs:
        rc |= subprocess.run([sys.executable, "-c", CHILD, lib, str(dist)]).returncode
    sys.exit(rc)
