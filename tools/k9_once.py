#!/usr/bin/env python3
"""A few k=9 count launches over 100 distinct samples (for rocprofv3 --kernel-trace --stats; VKIMG_LIB picks the build)."""
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402
eng = ImageEngine(k=9, mapping="cgr")
fq, offs, lens = eng.synth(0, 100, 1_000_000, 150)
hist = torch.empty((100, 4 ** 9), dtype=torch.int32, device="cuda")
status = torch.empty((100,), dtype=torch.int32, device="cuda")
for _ in range(4):
    eng.count(fq, offs, lens, hist=hist, status=status)
torch.cuda.synchronize()
