#!/usr/bin/env python3
"""Image kernel (K2) time vs batch size and k (random histograms)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

for k, mapping in ((7, "varKode"), (8, "cgr"), (9, "cgr"), (9, "varKode")):
    eng = ImageEngine(k=k, mapping=mapping)
    for n in (1, 100, 256, 1000):
        hist = torch.randint(0, 3000, (n, 4 ** k), dtype=torch.int32, device="cuda")
        img = torch.empty((n, eng.side, eng.side), dtype=torch.uint8, device="cuda")
        eng.images(hist, img=img); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.images(hist, img=img); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"k={k} {mapping:8s} n={n:5d}  {min(ts) * 1e3:8.3f} ms  ({min(ts) * 1e3 / n:.4f} ms/sample)", flush=True)
    eng.close()
