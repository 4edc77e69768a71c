#!/usr/bin/env python3
"""A/B timing of experimental builds of the count kernel: python tools/k1_ab.py lib1.so lib2.so ...
Each library runs in its own child process (one ctypes load per process).  Prints the K1 launch
time for 1000 samples x 1M x 150 bp (--pool distinct samples, default 64; --k, --samples, --dist), first for the
library in the tree with its default kernel and with VKIMG_K1_CLASSIC=1, then for every library named; the hash
of the first 64 histograms shows whether the builds agree."""
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
k = int(sys.argv[2]); samples = int(sys.argv[3]); dist = int(sys.argv[4]); pool = int(sys.argv[5])
eng = ImageEngine(k=k, mapping="cgr")
fq, po, pl = eng.synth(0, pool, 1_000_000, 150, dist=dist)
idx = np.arange(samples) % pool
offs, lens = po[idx].copy(), pl[idx].copy()
hist = torch.empty((samples, 4 ** k), dtype=torch.int32, device="cuda")
status = torch.empty((samples,), dtype=torch.int32, device="cuda")
eng.count(fq, offs, lens, hist=hist, status=status); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.count(fq, offs, lens, hist=hist, status=status); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
h = hashlib.sha256(hist[:64].cpu().numpy().tobytes()).hexdigest()[:16]
import os
print(f"{sys.argv[1] + ('@classic' if os.environ.get('VKIMG_K1_CLASSIC') == '1' else ''):40s} k={k} dist={dist} pool={pool} K1 {min(ts)*1e3:8.2f} ms (min of 3; {[round(t*1e3,1) for t in ts]}) bad={int((status!=0).sum())} sha={h}", flush=True)
"""

if __name__ == "__main__":
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    k = 7
    samples = 1000
    dist = 0
    pool = 64
    for a in sys.argv[1:]:
        if a.startswith("--k="): k = int(a[4:])
        if a.startswith("--samples="): samples = int(a[10:])
        if a.startswith("--dist="): dist = int(a[7:])
        if a.startswith("--pool="): pool = int(a[7:])
    rc = 0
    import os
    for lib in ["default", "default@classic"] + libs:
        env = dict(os.environ)
        if lib.endswith("@classic"):  # the same library with VKIMG_K1_CLASSIC=1 (every byte through the heavy stage)
            env["VKIMG_K1_CLASSIC"] = "1"
            lib = lib[:-8]
        r = subprocess.run([sys.executable, "-c", CHILD, lib, str(k), str(samples), str(dist), str(pool)], env=env)
        rc |= r.returncode
    sys.exit(rc)
