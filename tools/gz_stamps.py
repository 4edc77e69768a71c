#!/usr/bin/env python3
"""Diagnostic: what the chunk decoder's wavefronts spend their time on, and how evenly the chunks finish
(stamped build: hipcc ... -DVK_GZ_STAMPS -o tools/libvk_gzstamps.so).  Shares only -- the stamped build's run
time means nothing.  python tools/gz_stamps.py [nfiles] [reads] [level]"""
import ctypes as C
import os
import sys
import zlib

os.environ["VKIMG_LIB"] = "tools/libvk_gzstamps.so"
sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from varkoder_amd import synth  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = ImageEngine(k=7, mapping="cgr")
texts = [synth.sample_fastq(i, reads, 150, dist=i & 1).tobytes() for i in range(min(nfiles, 8))]
comp = []
for t in texts:
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    comp.append(co.compress(t) + co.flush())
files = [comp[i % len(comp)] for i in range(nfiles)]
tl = [len(texts[i % len(texts)]) for i in range(nfiles)]
offs, pos = [], 0
for f in files:
    offs.append(pos)
    pos += (len(f) + 15) // 16 * 16
host = np.zeros(pos + 16, dtype=np.uint8)
for o, f in zip(offs, files):
    host[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
dev = torch.from_numpy(host).cuda()
ooffs, pos = [], 0
for n in tl:
    ooffs.append(pos)
    pos += (n + 15) // 16 * 16
out = torch.empty(pos + 16, dtype=torch.uint8, device="cuda")
args = (dev, np.array(offs, dtype=np.uint64), np.array([len(f) for f in files], dtype=np.uint64), out,
        np.array(ooffs, dtype=np.uint64), np.array(tl, dtype=np.uint64))
lens, status = eng.inflate(*args)
assert eng.L.vk_debug_read_gz_res(None, 0, 1) == 0
lens, status = eng.inflate(*args)
assert not status.any()
nch = sum((len(f) + (1 << 18) - 1) >> 18 for f in files)
buf = (C.c_ulonglong * (8 * nch))()
assert eng.L.vk_debug_read_gz_stamps(buf, nch) == 0
v = np.frombuffer(buf, dtype=np.uint64).reshape(nch, 8).astype(np.float64)
v = v[v[:, 1] > 0]
t0 = v[:, 0].min()
dur = (v[:, 1] - v[:, 0]) / 100.0            # wall clock ticks of 10 ns -> microseconds
end = (v[:, 1] - t0) / 100.0
print(f"{len(v)} chunk wavefronts; kernel span {end.max() / 1e3:.1f} ms")
for name, x in (("lifetime of a wavefront (ms)", dur / 1e3), ("finishes at (ms)", end / 1e3), ("resolve rounds", v[:, 2]), ("tokens", v[:, 3]),
                ("tokens per round", v[:, 3] / np.maximum(v[:, 2], 1))):
    q = np.percentile(x, [0, 10, 50, 90, 99, 100])
    print(f"{name:32s} min {q[0]:10.1f}  p10 {q[1]:10.1f}  median {q[2]:10.1f}  p90 {q[3]:10.1f}  p99 {q[4]:10.1f}  max {q[5]:10.1f}  mean {x.mean():10.1f}")
# what a wavefront's lifetime goes with: its own work (tokens, cycles it counted itself), or the company it keeps
own = v[:, 4:8].sum(axis=1)
print("lifetime against the wavefront's own work: correlation with tokens %.3f, with its own counted cycles %.3f" %
      (np.corrcoef(dur, v[:, 3])[0, 1], np.corrcoef(dur, own)[0, 1]))
share = own / np.maximum(dur * 1e-6 * 2.1e9, 1)   # counted cycles / wall cycles at ~2.1 GHz
q = np.percentile(share, [0, 10, 50, 90, 100])
print("counted cycles / lifetime (the share of the wall clock the wavefront was in its stamped sections, incl. waits): "
      "min %.2f p10 %.2f median %.2f p90 %.2f max %.2f" % tuple(q))
order = np.argsort(v[:, 0])
k = len(order) // 8
for part in range(8):
    sel = order[part * k:(part + 1) * k]
    print("  started %5.2f-%5.2f ms: lifetime mean %6.1f ms, tokens mean %8.0f, own cycles mean %6.1f M" %
          ((v[sel, 0].min() - t0) / 1e5, (v[sel, 0].max() - t0) / 1e5, dur[sel].mean() / 1e3, v[sel, 3].mean(), own[sel].mean() / 1e6))
tot = v[:, 4:8].sum()
for name, col in (("gz_tokens", 4), ("gz_resolve", 6), ("headers + tables", 7)):
    print(f"{name:20s} {100 * v[:, col].sum() / tot:5.1f} %   {v[:, col].sum() / max(v[:, 3].sum(), 1):9.1f} cycles per token")
buf2 = (C.c_ulonglong * (4 * nch))()
assert eng.L.vk_debug_read_gz_res(buf2, nch, 0) == 0
r = np.frombuffer(buf2, dtype=np.uint64).reshape(nch, 4).astype(np.float64)
rt = r.sum()
rounds = max(v[:, 2].sum(), 1)
print("inside gz_resolve (the direct path's blocks, if any, are counted in too):")
for name, col in (("group set-up (ring, prefix sum, checks)", 0), ("round head + masks", 1), ("element loop (loads, stores issued)", 2), ("store drain", 3)):
    print(f"  {name:42s} {100 * r[:, col].sum() / rt:5.1f} %   {r[:, col].sum() / rounds:9.1f} cycles per round")
buf3 = (C.c_ulonglong * (8 * nch))()
assert eng.L.vk_debug_read_gz_find(buf3, nch) == 0
f = np.frombuffer(buf3, dtype=np.uint64).reshape(nch, 8).astype(np.float64)
f = f[f[:, 0] > 0]
print("vk_gzfind_kernel, per chunk:")
for name, x in (("wall time (ms)", f[:, 0] / 1e5), ("full header tests (Mcycles)", f[:, 1] / 1e6), ("verification (Mcycles)", f[:, 2] / 1e6), ("candidates tested in full", f[:, 3]),
                ("  of them above 30 kcycles", f[:, 4]), ("  their cycles (M)", f[:, 5] / 1e6), ("  the longest test (kcycles)", f[:, 6] / 1e3), ("scan iterations (64 positions)", f[:, 7])):
    q = np.percentile(x, [0, 50, 90, 99, 100])
    print(f"  {name:32s} min {q[0]:9.2f}  median {q[1]:9.2f}  p90 {q[2]:9.2f}  p99 {q[3]:9.2f}  max {q[4]:9.2f}  mean {x.mean():9.2f}")
