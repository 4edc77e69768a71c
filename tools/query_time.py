#!/usr/bin/env python3
"""BASELINE config 5 on one GPU, side measurement: 1000 samples (k=7 cgr) FASTQ -> images -> input
transform -> ViT-L/32 forward (random weights, the reference's default architecture) -> sigmoid."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from varkoder_amd import query as Q  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = ImageEngine(k=7, mapping="cgr")
fq, po, pl = eng.synth(0, 64, 1_000_000, 150)
idx = np.arange(n) % 64
offs, lens = po[idx].copy(), pl[idx].copy()
model = Q.vit().cuda().eval()
vocab = [str(i) for i in range(1000)]


def run(half):
    img, hist, status = eng.fastq_to_images(fq, offs, lens)
    return Q.probabilities(eng, img, model, batch_size=256, half=half)


for half in (True, False):
    run(half); torch.cuda.synchronize()
    t0 = time.perf_counter(); p = run(half); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    img, hist, status = eng.fastq_to_images(fq, offs, lens); torch.cuda.synchronize()
    t0 = time.perf_counter(); Q.probabilities(eng, img, model, batch_size=256, half=half); torch.cuda.synchronize()
    dq = time.perf_counter() - t0
    print(f"{'fp16 autocast' if half else 'fp32':14s} {n} samples: reads->probabilities {dt * 1e3:7.1f} ms "
          f"({n / dt:7.0f} samples/s, {n * 150e6 / dt / 1e9:6.0f} Gbases/s); transform+model alone {dq * 1e3:7.1f} ms "
          f"({n / dq:7.0f} images/s)", flush=True)
