#!/usr/bin/env python3
"""Where a batch's time goes on the mapped upload route (GPU box): python tools/mapped_probe.py [nfiles]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from varkoder_amd import engine as E  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
size = 179_200_000
tmp = tempfile.mkdtemp(prefix="vk_mp_")
blob = (b"@r\n" + b"ACGT" * 37 + b"AC\n+\n" + b"I" * 150 + b"\n") * (size // 309 + 1)
files = []
for i in range(n):
    p = os.path.join(tmp, f"f{i}@00084000K.fq")
    with open(p, "wb") as f:
        f.write(blob[:size])
    files.append(p)
eng = ImageEngine(k=7, mapping="cgr")
for mapped in (True, False, True, False):
    E.USE_MAPPED_UPLOAD = mapped
    t0 = time.perf_counter()
    st = eng.stage_files(files)
    t1 = time.perf_counter()
    dev, offs, lens = eng.upload_staged(st)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"mapped={mapped}: stage {1e3 * (t1 - t0):7.1f} ms, upload {1e3 * (t2 - t1):7.1f} ms = {n * size / (t2 - t1) / 1e9:5.1f} GB/s", flush=True)
    del dev
import shutil
shutil.rmtree(tmp, ignore_errors=True)
