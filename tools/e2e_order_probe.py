#!/usr/bin/env python3
"""bench.py's end_to_end in the same order (plain text first, then .fq.gz), with the pipeline's per-batch
times printed: python tools/e2e_order_probe.py [gz_first]"""
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

nfiles, reads = 64, 400_000
tmp = Path(tempfile.mkdtemp(prefix="vk_order_"))
eng = ImageEngine(k=7, mapping="varKode")
fq, offs, lens = eng.synth(1 << 20, nfiles, reads, 150)
host = fq.cpu().numpy()
del fq
plain = [tmp / f"s{i:04d}@{reads * 150 // 1000:08d}K.fq" for i in range(nfiles)]
(tmp / "gz").mkdir()
gz = [tmp / "gz" / f"s{i:04d}@{reads * 150 // 1000:08d}K.fq.gz" for i in range(nfiles)]


def write(i):
    blob = host[int(offs[i]):int(offs[i]) + int(lens[i])]
    blob.tofile(plain[i])
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(gz[i], "wb") as f:
        f.write(co.compress(blob.tobytes()) + co.flush())


with ThreadPoolExecutor(16) as ex:
    list(ex.map(write, range(nfiles)))
del host
order = (("fq_gz", gz, 16 << 30), ("plain_text", plain, 2 << 30))
if 'gz_first' not in sys.argv:
    order = order[::-1]
for name, files, bb in order + order:
    if 'fresh' in sys.argv:
        eng.__dict__.pop('_pinned_slots', None)  # a new staging buffer for every kind of input
    for rep in range(3):
        t0 = time.perf_counter()
        pipeline.fastqs_to_images(files, tmp / f"img_{name}_{rep}_{time.time_ns()}", k=7, mapping_code="varKode", io_threads=16, engine=eng,
                                  batch_bytes=bb, verbose=(rep == 2))
        dt = time.perf_counter() - t0
        print(f"{name} rep {rep}: {dt:.3f} s = {nfiles * reads * 150 / dt / 1e9:.2f} Gbases/s", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
