#!/usr/bin/env python3
"""One-off extended fuzz (GPU box): thousands of adversarial well-formed FASTQs (tests/fastq_cases
random_fastq) through every k, plain and subsampled, against the oracle."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from fastq_cases import random_fastq  # noqa: E402
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for k in (5, 6, 7, 8, 9):
    eng = ImageEngine(k=k, mapping="cgr")
    rng = np.random.default_rng(1000 + k)
    for r in range(rounds):
        blobs = [random_fastq(rng) for _ in range(96)]
        fq, offs, lens = eng.upload(blobs)
        parts = int(rng.integers(0, 4))
        hist, status = eng.count(fq, offs, lens, parts=parts)
        h = hist.cpu().numpy().view(np.uint32)
        st = status.cpu().numpy()
        seed, thr = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 32 + 1))
        hs, sts, sites = eng.count_sampled(fq, offs, lens, seed, thr, parts=parts)
        hs = hs.cpu().numpy().view(np.uint32)
        si = sites.cpu().numpy()
        for i, b in enumerate(blobs):
            want, _, wst = oracle.count_fastq(b, k)
            ws, _, wsst, wsites = oracle.count_fastq_sampled(b, k, seed, thr)
            ok = wst == 0 and st[i] == 0 and np.array_equal(h[i], want) and np.array_equal(hs[i], ws) and \
                tuple(int(x) for x in si[i]) == wsites
            if not ok:
                bad += 1
                print("MISMATCH", k, r, i, len(b), parts, seed, thr, flush=True)
    print(f"k={k}: {rounds * 96} inputs done, mismatches so far {bad}", flush=True)
    eng.close()
sys.exit(1 if bad else 0)
