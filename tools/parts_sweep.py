#!/usr/bin/env python3
"""K1 launch time against the number of workgroups per sample: python tools/parts_sweep.py [--pool=512] [--dist=0] [--k=7] [parts ...]
(1000 samples x 1M x 150 bp; parts 0 = the library's own choice).  One process; min of 3 after a warm-up each."""
import hashlib
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from varkoder_amd.engine import ImageEngine

k, pool, dist, samples = 7, 512, 0, 1000
plist = []
for a in sys.argv[1:]:
    if a.startswith("--k="): k = int(a[4:])
    elif a.startswith("--pool="): pool = int(a[7:])
    elif a.startswith("--dist="): dist = int(a[7:])
    elif a.startswith("--samples="): samples = int(a[10:])
    else: plist.append(int(a))
plist = plist or [0, 1, 2, 3, 4, 8, 16]
eng = ImageEngine(k=k, mapping="cgr")
fq, po, pl = eng.synth(0, pool, 1_000_000, 150, dist=dist)
idx = np.arange(samples) % pool
offs, lens = po[idx].copy(), pl[idx].copy()
hist = torch.empty((samples, 4 ** k), dtype=torch.int32, device="cuda")
status = torch.empty((samples,), dtype=torch.int32, device="cuda")
for parts in plist:
    eng.count(fq, offs, lens, parts=parts, hist=hist, status=status); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.count(fq, offs, lens, parts=parts, hist=hist, status=status); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    h = hashlib.sha256(hist[:64].cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"parts={parts:3d} k={k} dist={dist} pool={pool} K1 {min(ts)*1e3:8.2f} ms ({[round(t*1e3,1) for t in ts]}) bad={int((status!=0).sum())} sha={h}", flush=True)
