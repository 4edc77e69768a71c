#!/usr/bin/env python3
"""One timed count launch over 512 x 200k identical low-complexity reads, for profiling:
python tools/repeat_once.py [k] [polyA|acgt|ag|mixed]   (mixed: every fourth read is a random one)"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
kind = sys.argv[2] if len(sys.argv) > 2 else "acgt"
reads = 200_000
unit = {"polyA": "A" * 150, "acgt": "ACGT" * 37 + "AC", "ag": "AG" * 75}.get(kind)
if kind == "mixed":
    rng = np.random.default_rng(1)
    recs = []
    for i in range(4096):
        seq = ("ACGT" * 37 + "AC") if i % 4 else "".join(rng.choice(list("ACGT"), 150))
        recs.append("@s00000.0000000\n" + seq + "\n+\n" + "I" * 150 + "\n")
    data = ("".join(recs) * (reads // 4096 + 1)).encode()[: 320 * reads]
else:
    data = (("@s00000.0000000\n" + unit + "\n+\n" + "I" * 150 + "\n").encode()) * reads
eng = ImageEngine(k=k, mapping="cgr")
fq, offs, lens = eng.upload([data])
n = 512
o, l = np.repeat(offs, n), np.repeat(lens, n)
hist = torch.empty((n, 4 ** k), dtype=torch.int32, device="cuda")
status = torch.empty((n,), dtype=torch.int32, device="cuda")
eng.count(fq, o, l, parts=1, hist=hist, status=status)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    eng.count(fq, o, l, parts=1, hist=hist, status=status)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = min(ts)
gb = n * len(data) / dt / 1e9
print(f"k={k} {kind:6s}: {dt * 1e3:8.2f} ms for {n} x {len(data) / 1e6:.0f} MB = {gb:7.0f} GB/s ({gb / 8000:.1%} of peak), "
      f"windows {int(hist[0].sum())}, bad {int((status != 0).sum())}", flush=True)
