#!/usr/bin/env python3
"""Time of the subsample ladder (step C on the GPU, subsample.ladder_counts) over a batch of cleaned samples resident
in HBM: python tools/ladder_time.py [samples] [reads] [k]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402
from varkoder_amd.subsample import ladder_counts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 7
eng = ImageEngine(k=k, mapping="cgr")
fq, offs, lens = eng.synth(0, n, reads, 150)
for rep in range(3):
    t0 = time.perf_counter()
    recs = ladder_counts(eng, fq, offs, lens, seed=5, min_bp=50000, max_bp=None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(len(r["steps"]) for r in recs)
    print(f"k={k}: {n} samples x {reads} reads: {steps} ladder steps in {dt * 1e3:8.1f} ms "
          f"({float(lens.sum()) * steps / n / dt / 1e9:6.0f} GB/s of text passes)", flush=True)
