#!/usr/bin/env python3
"""Diagnostic: where one wave's piece loop spends its cycles (stamped build, tools/libvkimg_stamps.so;
hipcc ... -DVK_STAMPS).  Shares only -- the stamped build's run time means nothing."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
from varkoder_amd import _capi  # noqa: E402

_capi.LIB_PATH = "tools/libvkimg_stamps.so"
from varkoder_amd.engine import ImageEngine  # noqa: E402

eng = ImageEngine(k=9, mapping="cgr")
fq, po, pl = eng.synth(0, 64, 1_000_000, 150)
idx = np.arange(64) % 64
eng.count(fq, po[idx].copy(), pl[idx].copy(), parts=8)
import torch  # noqa: E402
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
assert eng.L.vk_debug_read_stamps(out) == 0
v = list(out)
n = v[4] or 1
names = ["stage+transposing reads (incl. load wait)", "classify", "scan + line masks + ok", "window loop"]
tot = sum(v[:4])
for name, x in zip(names, v[:4]):
    print(f"{name:45s} {x / n:9.1f} cycles/piece/wave  {100 * x / tot:5.1f} %")
print("pieces", n, "total cycles/piece", tot / n)
