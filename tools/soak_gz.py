#!/usr/bin/env python3
"""Soak of vk_inflate_device: a few hundred gzip files of 0.1 - 40 MB of FASTQ-like text made with random zlib
levels, memory levels, strategies, member counts (up to BGZF-like thousands of small members) and flushes in
mid-stream, inflated in batches of mixed sizes (so that the direct and the chunked path, 128 and 256 KiB chunks
all occur) from device memory and from pinned host memory; every byte compared with zlib's answer.
python tools/soak_gz.py [seed] [nfiles]"""
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, ".")
from varkoder_amd import synth  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 240
rng = np.random.default_rng(seed)
eng = ImageEngine(k=7, mapping="cgr")
pool = [synth.sample_fastq(1000 + i, 130_000, 150, dist=i & 1).tobytes() for i in range(4)]   # 41 MB each


def text_of(n):
    src = pool[int(rng.integers(len(pool)))]
    at = int(rng.integers(0, len(src) - n + 1))
    t = src[at:at + n]
    kind = int(rng.integers(6))
    if kind == 0:      # realistic qualities: noise over a small alphabet in every fourth line
        a = np.frombuffer(t, dtype=np.uint8).copy()
        q = rng.integers(33, 74, size=a.size, dtype=np.uint8)
        sel = rng.random(a.size) < 0.3
        a[sel & (a > 64)] = q[sel & (a > 64)]
        t = a.tobytes()
    elif kind == 1:    # a stretch of incompressible bytes in the middle (stored blocks)
        m = n // 2
        t = t[:m] + rng.integers(0, 256, size=min(n // 5, 300_000), dtype=np.uint8).tobytes() + t[m:]
    return t


def gz_of(t):
    level = int(rng.integers(0, 10))
    mem = int(rng.integers(1, 10))
    strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.choice(5, p=[0.6, 0.1, 0.1, 0.1, 0.1]))]
    style = int(rng.integers(5))
    if style == 0:     # several members
        cuts = sorted(int(x) for x in rng.integers(0, len(t) + 1, size=int(rng.integers(1, 5))))
        parts = [t[a:b] for a, b in zip([0] + cuts, cuts + [len(t)])]
    elif style == 1:   # BGZF-like: members of at most 64 KB
        parts = [t[i:i + 65280] for i in range(0, len(t), 65280)] or [b""]
    else:
        parts = [t]
    out = []
    for p in parts:
        co = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strat)
        if style == 2 and len(p) > 1000:   # sync / full flushes in mid-stream (empty stored blocks, byte alignment)
            a = len(p) // 3
            out.append(co.compress(p[:a]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(p[a:2 * a]) + co.flush(zlib.Z_FULL_FLUSH) +
                       co.compress(p[2 * a:]) + co.flush())
        else:
            out.append(co.compress(p) + co.flush())
    return b"".join(out)


done = bad = 0
while done < nfiles:
    nb = int(rng.integers(1, 40))
    sizes = [int(10 ** rng.uniform(5, 7.6)) for _ in range(nb)]
    if sum(sizes) > 400_000_000:
        sizes = sizes[:max(1, len(sizes) // 4)]
    texts = [text_of(n) for n in sizes]
    files = [gz_of(t) for t in texts]
    offs, pos = [], 0
    for f in files:
        offs.append(pos)
        pos += (len(f) + 15) // 16 * 16
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, f in zip(offs, files):
        host[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
    src = torch.from_numpy(host)
    src = src.pin_memory() if rng.random() < 0.5 else src.cuda()
    ooffs, pos = [], 0
    for t in texts:
        ooffs.append(pos)
        pos += (len(t) + 15) // 16 * 16
    out = torch.full((pos + 16,), 0xEE, dtype=torch.uint8, device="cuda")
    lens, st = eng.inflate(src, np.array(offs, dtype=np.uint64), np.array([len(f) for f in files], dtype=np.uint64), out,
                           np.array(ooffs, dtype=np.uint64), np.array([len(t) for t in texts], dtype=np.uint64))
    res = out.cpu().numpy()
    for i, t in enumerate(texts):
        got = bytes(res[ooffs[i]:ooffs[i] + int(lens[i])])
        if st[i] != 0 or got != t:
            bad += 1
            print(f"MISMATCH file {done + i}: status {int(st[i])}, {len(got)} of {len(t)} bytes, gz {len(files[i])} bytes", flush=True)
    done += nb
    print(f"{done} files, {bad} bad; last batch {nb} files, {sum(len(f) for f in files) / 1e6:.1f} MB -> {sum(sizes) / 1e6:.1f} MB"
          f" from {'pinned host' if src.device.type == 'cpu' else 'device'} memory", flush=True)
print("OK" if bad == 0 else f"FAILED: {bad}")
sys.exit(1 if bad else 0)
