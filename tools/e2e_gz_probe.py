#!/usr/bin/env python3
"""Where the time of the .fq.gz file pipeline goes: per-batch upload / kernel times (verbose) for
two batch sizes.  python tools/e2e_gz_probe.py [nfiles] [reads]"""
import shutil
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, ".")
from varkoder_amd import pipeline  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
tmp = Path(tempfile.mkdtemp(prefix="vk_gzprobe_"))
eng = ImageEngine(k=7, mapping="varKode")
fq, offs, lens = eng.synth(0, nfiles, reads, 150)
host = fq.cpu().numpy()
del fq
files = [tmp / f"s{i:04d}@{reads * 150 // 1000:08d}K.fq.gz" for i in range(nfiles)]


def write(i):
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(files[i], "wb") as f:
        f.write(co.compress(host[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()) + co.flush())


with ThreadPoolExecutor(16) as ex:
    list(ex.map(write, range(nfiles)))
del host
for bb in [int(x) << 30 for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("3", "5", "9", "16"))]:
    for rep in range(2):
        t0 = time.perf_counter()
        st = pipeline.fastqs_to_images(files, tmp / f"img{bb}_{rep}", k=7, mapping_code="varKode", io_threads=16, engine=eng,
                                       batch_bytes=bb, verbose=(rep == 1))
        dt = time.perf_counter() - t0
        print(f"batch_bytes {bb >> 30} GiB rep {rep}: {dt:.3f} s = {nfiles * reads * 150 / dt / 1e9:.2f} Gbases/s", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
