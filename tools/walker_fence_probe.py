#!/usr/bin/env python3
"""Which reads does a walker build miscount?  The inputs of tests/test_subsample.py::test_walker_at_k8_and_k9 through the
library named by VKIMG_LIB (a build of vk_ladder.h without the optimiser fence on the loop-carried mask),
against the oracle; for every (sample, seed, threshold) that differs, the reads taken whose windows explain the difference,
with where they lie in their 64-byte sectors:   VKIMG_LIB=ab/nofence.so python tools/walker_fence_probe.py [k ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fastq_cases  # noqa: E402
from oracle import oracle  # noqa: E402
from varkoder_amd import synth  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402


def sample_hash(seed, anchor):
    m = 0xFFFFFFFF
    h = (anchor ^ seed) & m
    h = (h + ((anchor >> 32) * 0x9E3779B1 + (seed >> 32))) & m
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & m
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & m
    h ^= h >> 16
    return h


def reads_of(blob):
    """(anchor, sequence bytes) of every record."""
    out, pos = [], 0
    lines = blob.split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i + 1 < len(lines):
        anchor = pos + len(lines[i])
        seq = lines[i + 1]
        out.append((anchor, seq))
        for j in range(4):
            if i + j < len(lines):
                pos += len(lines[i + j]) + 1
        i += 4
    return out


def main():
    ks = [int(a) for a in sys.argv[1:]] or [8, 9]
    print("library:", os.environ.get("VKIMG_LIB", "(in tree)"))
    for k in ks:
        eng = ImageEngine(k=k, mapping="cgr")
        blobs = [synth.sample_fastq(11 + d, 20000, 150, dist=d).tobytes() for d in range(3)]
        blobs.append(b"".join(fastq_cases.rec(f"L{i}", fastq_cases.rand_seq(np.random.default_rng(i), 1700)) for i in range(400)))
        blobs.append(fastq_cases.random_fastq(np.random.default_rng(77), 3000))
        fq, offs, lens = eng.upload(blobs)
        print("k", k, "sample offsets mod 64:", [int(o) % 64 for o in np.asarray(offs)])
        for parts in (0, 1, 3):
            nsites, status = eng.read_index(fq, offs, lens, parts=parts)
            pairs = [(i, seed, den) for i in range(len(blobs)) for seed, den in ((7, 33), (8, 100), (9, 400))]
            idx = [i for i, _, _ in pairs]
            seeds = np.array([s_ for _, s_, _ in pairs], dtype=np.uint64)
            thr = np.array([(1 << 32) // den for _, _, den in pairs], dtype=np.uint64)
            hist, st, sites = eng.count_sampled(fq, offs[idx], lens[idx], seeds, thr)
            h = hist.cpu().numpy().view(np.uint32)
            for j, (i, seed, den) in enumerate(pairs):
                want, nwin, wst, wsites = oracle.count_fastq_sampled(blobs[i], k, seed, int(thr[j]))
                d = h[j].astype(np.int64) - want.astype(np.int64)
                if not d.any():
                    continue
                codes = np.nonzero(d)[0]
                print(f" k={k} parts={parts} sample={i} seed={seed} 1/{den}: {codes.size} codes differ, sum {int(d.sum())}, "
                      f"plus {int(d[d > 0].sum())} minus {int(-d[d < 0].sum())}")
                taken = [(a, s) for a, s in reads_of(blobs[i]) if sample_hash(seed, a) < int(thr[j])]
                bad = set(int(c) for c in codes)
                for a, s in taken:
                    rec = b"@x\n" + s + b"\n+\n" + b"I" * len(s) + b"\n"
                    hr, _, _ = oracle.count_fastq(rec, k)
                    hit = [int(c) for c in np.nonzero(hr)[0] if int(c) in bad]
                    if len(hit) >= 1:
                        # positions of the windows in the read
                        enc = {65: 0, 67: 1, 71: 2, 84: 3}
                        where = []
                        for t in range(len(s) - k + 1):
                            w = s[t:t + k]
                            if all(b in enc for b in w):
                                code = 0
                                for b in w:
                                    code = code * 4 + enc[b]
                                if code in bad:
                                    where.append((t, int(d[code])))
                        p = a + 1
                        print(f"   read at p={p} (p%64={p % 64}, p%16={p % 16}) len={len(s)} end%64={(p + len(s)) % 64} "
                              f"hits={len(hit)} windows(start,diff)={where[:12]}")
        eng.close()


if __name__ == "__main__":
    main()
