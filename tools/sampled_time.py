#!/usr/bin/env python3
"""Throughput of the subsampled count (vk_count_sampled_device) next to the plain one, k=7:
python tools/sampled_time.py [samples] [fraction,fraction,...]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fracs = [float(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1.0", "0.5", "0.1", "0.001"))]
eng = ImageEngine(k=7, mapping="cgr")
fq, offs, lens = eng.synth(0, n, 1_000_000, 150)
tot = float(lens.sum())


def timed(f):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


t = timed(lambda: eng.count(fq, offs, lens))
print(f"plain count       : {t * 1e3:8.2f} ms = {tot / t / 1e9:7.0f} GB/s ({tot / t / 8e12:.3f} of peak)", flush=True)
for fr in fracs:
    thr = min(1 << 32, int(fr * (1 << 32)))
    t = timed(lambda: eng.count_sampled(fq, offs, lens, 7, thr))
    print(f"sampled, p = {fr:5.3f}: {t * 1e3:8.2f} ms = {tot / t / 1e9:7.0f} GB/s ({tot / t / 8e12:.3f} of peak)", flush=True)
