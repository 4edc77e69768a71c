#!/usr/bin/env python3
"""Timing of the subsampling launch next to the plain count (1000 samples x 1M x 150 bp, k=7)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
eng = ImageEngine(k=k, mapping="cgr")
fq, po, pl = eng.synth(0, 64, 1_000_000, 150)
idx = np.arange(samples) % 64
offs, lens = po[idx].copy(), pl[idx].copy()
hist = torch.empty((samples, 4 ** k), dtype=torch.int32, device="cuda")
status = torch.empty((samples,), dtype=torch.int32, device="cuda")
sites = torch.empty((samples, 2), dtype=torch.int64, device="cuda")


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


print(f"k={k} plain count         {timed(lambda: eng.count(fq, offs, lens, hist=hist, status=status)):8.2f} ms")
for frac in (1.0, 0.25, 0.01):
    thr = min(1 << 32, int(frac * (1 << 32)))
    ms = timed(lambda: eng.count_sampled(fq, offs, lens, 5, thr, hist=hist, status=status, sites=sites))
    s = sites.cpu().numpy()
    print(f"k={k} sampled p={frac:<5} {ms:8.2f} ms   taken fraction {s[:, 1].sum() / s[:, 0].sum():.4f}  sites/sample {s[0, 0]}")
