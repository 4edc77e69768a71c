#!/usr/bin/env python3
"""One-off soak (GPU box) of the low-complexity paths: samples made of tandem repeats of random motifs (period 1..10, every
phase, random lengths, broken by N or by a second motif) mixed with ordinary reads in random proportion, every k,
several `parts`, against the oracle.  python tools/soak_repeats.py [rounds]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from fastq_cases import rec  # noqa: E402
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0


def sample(rng):
    nmotif = int(rng.integers(1, 5))
    motifs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 11)))) for _ in range(nmotif)]
    p_random = float(rng.choice([0.0, 0.05, 0.25, 0.6]))
    fixed = rng.random() < 0.5          # every record the same length (reads sit alike in their blocks) or ragged
    n_fixed = int(rng.integers(30, 300))
    hdr_pad = int(rng.integers(0, 30))
    recs = []
    for i in range(int(rng.integers(400, 2500))):
        n = n_fixed if fixed else int(rng.integers(8, 400))
        r = rng.random()
        if r < p_random:
            seq = "".join(rng.choice(list("ACGT"), size=n))
        else:
            m = motifs[int(rng.integers(0, nmotif))]
            ph = int(rng.integers(0, len(m)))
            seq = (m * (n // len(m) + 3))[ph: ph + n]
            r2 = rng.random()
            if r2 < 0.08:
                seq = seq[: n // 2] + "N" + seq[n // 2 + 1:]
            elif r2 < 0.14:
                m2 = motifs[int(rng.integers(0, nmotif))]
                seq = seq[: n // 3] + (m2 * (n // len(m2) + 3))[: n - n // 3]
        recs.append(rec("r%06d" % i + "y" * (hdr_pad if fixed else int(rng.integers(0, 30))), seq))
    return b"".join(recs)


for k in (5, 6, 7, 8, 9):
    eng = ImageEngine(k=k, mapping="cgr")
    rng = np.random.default_rng(4200 + k)
    for r in range(rounds):
        blobs = [sample(rng) for _ in range(24)]
        fq, offs, lens = eng.upload(blobs)
        for parts in (0, 1, 3):
            hist, status = eng.count(fq, offs, lens, parts=parts)
            h = hist.cpu().numpy().view(np.uint32)
            st = status.cpu().numpy()
            for i, b in enumerate(blobs):
                want, _, wst = oracle.count_fastq(b, k)
                if not (wst == 0 and st[i] == 0 and np.array_equal(h[i], want)):
                    bad += 1
                    print("MISMATCH", k, r, i, parts, len(b), flush=True)
    print(f"k={k}: {rounds * 24} samples x 3 launches done, mismatches so far {bad}", flush=True)
    eng.close()
sys.exit(1 if bad else 0)
