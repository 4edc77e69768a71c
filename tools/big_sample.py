#!/usr/bin/env python3
"""One-off: a single sample larger than 4 GiB (64-bit offsets everywhere) against the oracle."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

reads = 9_000_000                       # x 520 B = 4.68 GB > 2^32
eng = ImageEngine(k=7, mapping="cgr")
fq, offs, lens = eng.synth(42, 1, reads, 250, dist=1)
assert int(lens[0]) > 2 ** 32
for parts in (0, 3):
    t0 = time.perf_counter()
    hist, status = eng.count(fq, offs, lens, parts=parts)
    torch.cuda.synchronize()
    print(f"parts={parts}: {time.perf_counter() - t0:.3f} s, launch {eng.last_count_launch()}", flush=True)
    got = hist.cpu().numpy().view(np.uint32)[0]
    if parts == 0:
        host = fq[:int(lens[0])].cpu().numpy()
        t0 = time.perf_counter()
        want, nwin, st = oracle.count_fastq(host, 7)
        print(f"oracle: {time.perf_counter() - t0:.1f} s, {nwin} windows", flush=True)
        del host
    assert int(status.cpu()[0]) == 0 and st == 0
    assert int(got.sum(dtype=np.uint64)) == nwin
    assert np.array_equal(got, want)
    print("exact", flush=True)
