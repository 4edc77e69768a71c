#!/usr/bin/env python3
"""Worst case for the LDS histogram: low-complexity reads (every window the same k-mer)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

reads = 200_000
rec = ("@s00000.0000000\n" + "A" * 150 + "\n+\n" + "I" * 150 + "\n").encode()
blob = rec * reads
mixed = (("@s00000.0000000\n" + "ACGT" * 37 + "AC" + "\n+\n" + "I" * 150 + "\n").encode()) * reads
for k in (7, 9):
    eng = ImageEngine(k=k, mapping="cgr")
    for name, data in (("poly-A", blob), ("ACGT repeat", mixed)):
        fq, offs, lens = eng.upload([data])
        n = 512
        o, l = np.repeat(offs, n), np.repeat(lens, n)
        hist = torch.empty((n, 4 ** k), dtype=torch.int32, device="cuda")
        status = torch.empty((n,), dtype=torch.int32, device="cuda")
        eng.count(fq, o, l, parts=1, hist=hist, status=status); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.count(fq, o, l, parts=1, hist=hist, status=status); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gb = n * len(data) / dt / 1e9
        print(f"k={k} {name:12s}: {dt * 1e3:8.2f} ms for {n} x {len(data) / 1e6:.0f} MB  = {gb:7.0f} GB/s "
              f"({gb / 8000:.1%} of peak), windows ok: {int(hist[0].sum()) == reads * (150 - k + 1)}", flush=True)
    eng.close()
