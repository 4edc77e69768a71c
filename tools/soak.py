#!/usr/bin/env python3
"""One-off soak (GPU box): many full-size samples (1M x 150 bp) counted in one launch and compared
bit for bit with the oracle, for both base distributions and several workgroup splits; then a
10,000-sample launch (BASELINE config 3's per-node batch on one GPU) checked through status words
and the window checksum."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

eng = ImageEngine(k=7, mapping="varKode")
bad = 0
for dist in (0, 1, 2):
    n = 12
    fq, offs, lens = eng.synth(3000 + 100 * dist, n, 1_000_000, 150, dist=dist)
    host = fq.cpu().numpy()
    want = [oracle.count_fastq(host[int(o):int(o) + int(l)], 7)[0] for o, l in zip(offs, lens)]
    for parts in (0, 1, 2, 5):
        hist, status = eng.count(fq, offs, lens, parts=parts)
        got = hist.cpu().numpy().view(np.uint32)
        ok = all(np.array_equal(got[i], want[i]) for i in range(n)) and not status.cpu().numpy().any()
        print(f"dist {dist} parts {parts}: {'exact' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    del fq, host
    torch.cuda.empty_cache()

pool = 128
fq, po, pl = eng.synth(0, pool, 1_000_000, 150)
idx = np.arange(10000) % pool
offs, lens = po[idx].copy(), pl[idx].copy()
t0 = time.perf_counter()
img, hist, status = eng.fastq_to_images(fq, offs, lens)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sums = hist.to(torch.int64).sum(dim=1)
same = bool((sums.view(-1, pool)[0] == sums.view(-1, pool)).all()) if 10000 % pool == 0 else True
first = sums[:pool]
rep_ok = bool((sums == first.repeat((10000 + pool - 1) // pool)[:10000]).all())
print(f"10000 samples: {dt:.3f} s, bad status {int((status != 0).sum())}, repeats consistent {rep_ok}")
sys.exit(1 if bad or int((status != 0).sum()) or not rep_ok else 0)
