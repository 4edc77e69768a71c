#!/usr/bin/env python3
"""Fuzz of the k = 8, 9 quad route (GPU box): inputs of 0.2 .. 3 MB -- every wavefront's range has pieces beyond its first and
last, several workgroups per sample, waves that run out of pieces before others -- with the compositions that move the
route's switches: ordinary reads, reads riddled with N (listed quads), AT-rich and two-letter reads (queues that fill:
`tight`, overflow into the direct count), stretches of homopolymers and short tandem repeats (the shortcuts, steps without
appends), tiny and empty samples beside large ones.  Against the oracle, and against the pair route for the first batch.
python tools/fuzz_quads.py [rounds] [blobs per round]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from fastq_cases import random_fastq  # noqa: E402
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nblobs = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def blob(rng):
    flavour = int(rng.integers(0, 8))
    target = int(rng.integers(200_000, 3_000_000)) if flavour != 7 else int(rng.integers(0, 3000))
    probs = {0: [.25, .25, .25, .25, 0], 1: [.24, .24, .24, .24, .04], 2: [.45, .05, .05, .45, 0], 3: [.5, 0, 0, .5, 0],
             4: [.25, .25, .25, .25, 0], 5: [.25, .25, .25, .25, 0], 6: [.3, .2, .2, .3, 0], 7: [.25, .25, .25, .25, 0]}[flavour]
    recs, size, i = [], 0, 0
    while size < target:
        u = rng.random()
        n = int(rng.integers(0, 45)) if u < 0.03 else int(rng.integers(45, 300)) if u < 0.97 else int(rng.integers(300, 4000))
        if flavour == 4 and rng.random() < 0.7:            # homopolymer / tandem stretches among ordinary reads
            unit = str(rng.choice(["A", "T", "AC", "ACGT", "AAG", "ACGTTGCATC"]))
            seq = (unit * (n // len(unit) + 1))[:n]
        elif flavour == 5 and (i // 400) % 2 == 0:         # long runs of identical poly-A reads, then ordinary ones
            seq = "A" * n
        else:
            seq = "".join(rng.choice(list("ACGTN"), p=probs, size=n)) if n else ""
        qual = "".join(rng.choice(list("!#5@+IJ~"), size=n)) if n else ""
        hdr = "@r" + "".join(rng.choice(list("abcXYZ012:/ _"), size=int(rng.integers(0, 70))))
        r = f"{hdr}\n{seq}\n+\n{qual}\n".encode()
        recs.append(r)
        size += len(r)
        i += 1
    if flavour == 6:
        recs.append(random_fastq(rng, nrec=int(rng.integers(5, 40))))
    return b"".join(recs)


bad = 0
for k in (9, 8):
    eng = ImageEngine(k=k, mapping="cgr")
    rng = np.random.default_rng(5000 + k)
    for r in range(rounds):
        blobs = [blob(rng) for _ in range(nblobs)]
        if blobs and not blobs[-1].endswith(b"\n") and blobs[-1]:
            blobs[-1] += b"\n"
        fq, offs, lens = eng.upload(blobs)
        parts = int(rng.integers(0, 7))
        hist, status = eng.count(fq, offs, lens, parts=parts)
        h = hist.cpu().numpy().view(np.uint32)
        st = status.cpu().numpy()
        wants = [oracle.count_fastq(b, k) for b in blobs]
        for i, b in enumerate(blobs):
            want, _, wst = wants[i]
            if not (wst == 0 and st[i] == 0 and np.array_equal(h[i], want)):
                bad += 1
                print("MISMATCH quads", k, r, i, len(b), parts, int(st[i]), wst, flush=True)
        if r == 0:      # the pair route on the same batch
            os.environ["VKIMG_SPILL_PAIRS"] = "1"
            e2 = ImageEngine(k=k, mapping="cgr")
            del os.environ["VKIMG_SPILL_PAIRS"]
            f2, o2, l2 = e2.upload(blobs)
            h2 = e2.count(f2, o2, l2, parts=parts)[0].cpu().numpy().view(np.uint32)
            for i in range(len(blobs)):
                if not np.array_equal(h2[i], wants[i][0]):
                    bad += 1
                    print("MISMATCH pairs", k, r, i, flush=True)
            e2.close()
        print(f"k={k} round {r}: {nblobs} inputs, {sum(len(b) for b in blobs) / 1e6:.0f} MB, parts {parts}, mismatches so far {bad}", flush=True)
    eng.close()
sys.exit(1 if bad else 0)
