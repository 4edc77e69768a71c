#!/usr/bin/env python3
"""Time of the k=9 image stage (vk_image_device over 100 histograms of 1M-read samples): python tools/image9_time.py
(VKIMG_LIB picks the build)"""
import sys
import time
sys.path.insert(0, ".")
import hashlib
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402
for mapping in ("cgr", "varKode"):
    eng = ImageEngine(k=9, mapping=mapping)
    fq, offs, lens = eng.synth(0, 100, 1_000_000, 150)
    hist, status = eng.count(fq, offs, lens)
    img = eng.images(hist)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); img = eng.images(hist, img=img); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"k=9 {mapping:8s}: images of 100 samples {min(ts) * 1e3:7.3f} ms  sha {hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
    eng.close()
