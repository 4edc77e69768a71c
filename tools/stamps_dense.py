#!/usr/bin/env python3
"""Diagnostic: the share of a wavefront's time the dense k=7 kernel spends waiting for its piece's bytes (stamped
build: hipcc ... -DVK_STAMPS -o tools/libvkimg_stamps.so).  Shares only -- the stamped build's run time means nothing.
python tools/stamps_dense.py [pool] [dist] [library]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
from varkoder_amd import _capi  # noqa: E402

_capi.LIB_PATH = sys.argv[3] if len(sys.argv) > 3 else "tools/libvkimg_stamps.so"
from varkoder_amd.engine import ImageEngine  # noqa: E402
import torch  # noqa: E402

pool = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = ImageEngine(k=7, mapping="varKode")
fq, po, pl = eng.synth(0, pool, 1_000_000, 150, dist=dist)
idx = np.arange(1000) % pool
import time  # noqa: E402
offs, lens = po[idx].copy(), pl[idx].copy()
eng.count(fq, offs, lens)
torch.cuda.synchronize()
eng.L.vk_debug_read_stamps((C.c_ulonglong * 8)())
t0 = time.perf_counter()
eng.count(fq, offs, lens)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
out = (C.c_ulonglong * 8)()
assert eng.L.vk_debug_read_stamps(out) == 0
v = list(out)
n = v[7] or 1
print(f"pool {pool} dist {dist}: waiting for the piece {v[5] / n:9.1f} ticks/piece/wave of {v[6] / n:9.1f} per iteration = {100 * v[5] / max(v[6], 1):5.1f} %  (pieces {n}, two launches); wall time of the second launch {wall * 1e3:.1f} ms (stamped build)")
