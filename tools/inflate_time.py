#!/usr/bin/env python3
"""Throughput of vk_inflate_device: N gzip files of synthetic FASTQ (zlib level L) resident in HBM ->
text in HBM.  python tools/inflate_time.py [nfiles] [reads] [level] [pinned]
(a fourth argument: the compressed bytes stay in pinned host memory and the kernels read them over PCIe)"""
import sys
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, ".")
from varkoder_amd import synth  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = ImageEngine(k=7, mapping="cgr")
texts = [synth.sample_fastq(i, reads, 150, dist=i & 1).tobytes() for i in range(min(nfiles, 8))]
t0 = time.perf_counter()
comp = []
for t in texts:
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    comp.append(co.compress(t) + co.flush())
host_s = time.perf_counter() - t0
t0 = time.perf_counter()
for c in comp:
    zlib.decompress(c, 31)
host_inflate = sum(len(t) for t in texts) / (time.perf_counter() - t0)
files = [comp[i % len(comp)] for i in range(nfiles)]
tl = [len(texts[i % len(texts)]) for i in range(nfiles)]
offs, pos = [], 0
for f in files:
    offs.append(pos)
    pos += (len(f) + 15) // 16 * 16
host = np.zeros(pos + 16, dtype=np.uint8)
for o, f in zip(offs, files):
    host[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
dev = torch.from_numpy(host).pin_memory() if len(sys.argv) > 4 else torch.from_numpy(host).cuda()
ooffs, pos = [], 0
for n in tl:
    ooffs.append(pos)
    pos += (n + 15) // 16 * 16
out = torch.empty(pos + 16, dtype=torch.uint8, device="cuda")
args = (dev, np.array(offs, dtype=np.uint64), np.array([len(f) for f in files], dtype=np.uint64), out,
        np.array(ooffs, dtype=np.uint64), np.array(tl, dtype=np.uint64))
eng.inflate(*args)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    lens, st = eng.inflate(*args)
    ts.append(time.perf_counter() - t0)
assert not st.any() and lens.tolist() == tl
total = sum(tl)
print(f"{nfiles} files x {reads} reads, level {level}: ratio {total / sum(len(f) for f in files):.2f}, "
      f"GPU inflate {min(ts) * 1e3:.1f} ms = {total / min(ts) / 1e9:.2f} GB/s of text "
      f"({total / min(ts) / nfiles / 1e6:.0f} MB/s per file); one host thread (zlib) {host_inflate / 1e6:.0f} MB/s")
