#!/usr/bin/env python3
"""Extended fuzz of the dense k <= 7 kernel's steady state (GPU box): inputs LARGE enough that every wavefront's byte
range holds pieces that are neither its first nor its last -- the pieces the line pass, the lanes set aside,
vk_aside_kernel and the read index work on (tools/fuzz_gpu.py's inputs are a few KB: first / last pieces only, the
general path) -- built from thousands of adversarial records (tests/fastq_cases.random_fastq), through count(),
count_index() and the read walker, against the oracle.   python tools/fuzz_dense.py [rounds] [blobs per round]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from fastq_cases import random_fastq  # noqa: E402
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nblobs = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def big_blob(rng):
    """0.3 .. 1.5 MB of records; one flavour per blob so that some blobs are nearly all short reads (many lanes set
    aside, pieces over the limit) and others ordinary"""
    target = int(rng.integers(300_000, 1_500_000))
    flavour = rng.random()
    out, size = [], 0
    while size < target:
        if flavour < 0.3:      # everything random_fastq makes (no CRLF mix inside one file: it picks per call)
            b = random_fastq(rng, nrec=int(rng.integers(20, 60)))
            if not b.endswith(b"\n"):
                b += b"\n"
        else:                  # mostly ordinary reads with a share of short / empty / very long ones
            recs = []
            short = 0.02 if flavour < 0.6 else 0.3
            for _ in range(200):
                u = rng.random()
                n = int(rng.integers(0, 45)) if u < short else int(rng.integers(45, 300)) if u < 0.97 else int(rng.integers(300, 5000))
                seq = "".join(rng.choice(list("ACGTN" if rng.random() < 0.1 else "ACGT"), size=n)) if n else ""
                qual = "".join(rng.choice(list("!#5@+IJ~"), size=n)) if n else ""
                hdr = "@r" + "".join(rng.choice(list("abcXYZ012:/ _"), size=int(rng.integers(0, 70))))
                recs.append(f"{hdr}\n{seq}\n+\n{qual}\n".encode())
            b = b"".join(recs)
        out.append(b)
        size += len(b)
    return b"".join(out)


bad = 0
for k in (5, 6, 7):
    eng = ImageEngine(k=k, mapping="cgr")
    rng = np.random.default_rng(4000 + k)
    for r in range(rounds):
        blobs = [big_blob(rng) for _ in range(nblobs)]
        fq, offs, lens = eng.upload(blobs)
        parts = int(rng.integers(1, 4))
        hist, status = eng.count(fq, offs, lens, parts=parts)
        general = eng.last_count_general()
        h = hist.cpu().numpy().view(np.uint32)
        st = status.cpu().numpy()
        hi, nsites, sti = eng.count_index(fq, offs, lens, parts=parts)
        hi = hi.cpu().numpy().view(np.uint32)
        seed, thr = int(rng.integers(0, 2 ** 40)), int(rng.integers(1, 2 ** 32))
        hs, sts, sites = eng.count_sampled(fq, offs, lens, seed, thr, parts=parts)   # the walker: the samples are indexed
        hs = hs.cpu().numpy().view(np.uint32)
        si = sites.cpu().numpy()
        for i, b in enumerate(blobs):
            want, _, wst = oracle.count_fastq(b, k)
            ws, _, wsst, wsites = oracle.count_fastq_sampled(b, k, seed, thr)
            ok = wst == 0 and st[i] == 0 and sti[i] == 0 and np.array_equal(h[i], want) and np.array_equal(hi[i], want) and \
                int(nsites[i]) == wsites[0] and np.array_equal(hs[i], ws) and tuple(int(x) for x in si[i]) == wsites
            if not ok:
                bad += 1
                print("MISMATCH", k, r, i, len(b), parts, seed, thr, int(st[i]), int(sti[i]), int(nsites[i]), wsites,
                      bool(np.array_equal(h[i], want)), bool(np.array_equal(hi[i], want)), bool(np.array_equal(hs[i], ws)), flush=True)
        print(f"k={k} round {r}: {nblobs} inputs, {sum(len(b) for b in blobs) / 1e6:.0f} MB, parts {parts}, "
              f"general-path pieces {general}, mismatches so far {bad}", flush=True)
    eng.close()
# k = 8, 9: the spill path on the same kind of input, and the walker on small fractions (the only calls it takes there)
for k in (8, 9):
    eng = ImageEngine(k=k, mapping="cgr")
    rng = np.random.default_rng(4000 + k)
    for r in range(max(1, rounds // 4)):
        blobs = [big_blob(rng) for _ in range(nblobs)]
        fq, offs, lens = eng.upload(blobs)
        parts = int(rng.integers(1, 4))
        hist, status = eng.count(fq, offs, lens, parts=parts)
        h = hist.cpu().numpy().view(np.uint32)
        st = status.cpu().numpy()
        nsites, sti = eng.read_index(fq, offs, lens, parts=parts)
        seed, thr = int(rng.integers(0, 2 ** 40)), int(rng.integers(1, 2 ** 32 // 32))
        hs, sts, sites = eng.count_sampled(fq, offs, lens, seed, thr, parts=parts)
        walker = eng.last_count_launch()["lds_bytes"] < 16384
        hs = hs.cpu().numpy().view(np.uint32)
        si = sites.cpu().numpy()
        for i, b in enumerate(blobs):
            want, _, wst = oracle.count_fastq(b, k)
            ws, _, wsst, wsites = oracle.count_fastq_sampled(b, k, seed, thr)
            ok = wst == 0 and st[i] == 0 and sti[i] == 0 and np.array_equal(h[i], want) and int(nsites[i]) == wsites[0] and \
                np.array_equal(hs[i], ws) and tuple(int(x) for x in si[i]) == wsites
            if not ok:
                bad += 1
                print("MISMATCH", k, r, i, len(b), parts, seed, thr, bool(np.array_equal(h[i], want)), bool(np.array_equal(hs[i], ws)), flush=True)
        print(f"k={k} round {r}: {nblobs} inputs, parts {parts}, walker {walker}, mismatches so far {bad}", flush=True)
    eng.close()
sys.exit(1 if bad else 0)
