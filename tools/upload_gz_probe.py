#!/usr/bin/env python3
"""Where upload_staged spends its time on .fq.gz files: staging, the copy of the compressed bytes, the
text allocation and vk_inflate_device, each bracketed by a device synchronise.
python tools/upload_gz_probe.py [nfiles] [reads]"""
import shutil
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, ".")
from varkoder_amd.engine import ImageEngine  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
tmp = Path(tempfile.mkdtemp(prefix="vk_upprobe_"))
eng = ImageEngine(k=7, mapping="varKode")
fq, offs, lens = eng.synth(0, nfiles, reads, 150)
host = fq.cpu().numpy()
del fq
files = [tmp / f"s{i:04d}.fq.gz" for i in range(nfiles)]


def write(i):
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(files[i], "wb") as f:
        f.write(co.compress(host[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()) + co.flush())


with ThreadPoolExecutor(16) as ex:
    list(ex.map(write, range(nfiles)))
    del host
    for rep in range(3):
        sync = torch.cuda.synchronize
        t = [time.perf_counter()]
        st = eng.stage_files(files, ex)
        t.append(time.perf_counter())
        dev = torch.empty(st["text_total"], dtype=torch.uint8, device="cuda")
        sync(); t.append(time.perf_counter())
        gzdev = torch.empty(st["stage_total"] - st["plain_total"], dtype=torch.uint8, device="cuda")
        gzdev.copy_(st["pinned"][st["plain_total"]:st["stage_total"]], non_blocking=True)
        sync(); t.append(time.perf_counter())
        got, status = eng.inflate(gzdev, st["src"] - np.uint64(st["plain_total"]), st["disk"], dev, st["offs"], st["caps"])
        sync(); t.append(time.perf_counter())
        del dev, gzdev
        t0 = time.perf_counter()
        d2, o2, l2 = eng.upload_staged(st)
        sync(); whole = time.perf_counter() - t0
        del d2
        names = ["stage_files", "alloc text", "copy gz", "inflate"]
        print(f"rep {rep}: " + ", ".join(f"{n} {1e3 * (b - a):.1f} ms" for n, a, b in zip(names, t, t[1:])) +
              f"; upload_staged as a whole {1e3 * whole:.1f} ms; gz {int(st['disk'].sum()) / 1e9:.2f} GB -> text {int(got.sum()) / 1e9:.2f} GB,"
              f" bad {int((status != 0).sum())}", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
