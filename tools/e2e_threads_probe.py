#!/usr/bin/env python3
"""The file pipeline on plain-text files with few I/O threads (a rank that shares its host's cores with others):
the staged route against the mapped one.  python tools/e2e_threads_probe.py [nfiles] [threads,threads,...] [gz]"""
import shutil
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, ".")
from varkoder_amd import engine as E, pipeline  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("2", "4", "16"))]
reads = 560_000
tmp = Path(tempfile.mkdtemp(prefix="vk_thr_"))
eng = ImageEngine(k=7, mapping="varKode")
fq, offs, lens = eng.synth(0, 8, reads, 150)
host = fq.cpu().numpy()
files = []
for i in range(nfiles):
    p = tmp / f"s{i:04d}@{reads * 150 // 1000:08d}K.fq"
    p.write_bytes(host[int(offs[i % 8]):int(offs[i % 8]) + int(lens[i % 8])].tobytes())
    files.append(p)
if len(sys.argv) > 3 and sys.argv[3] == "gz":      # the same files as .fq.gz (zlib level 1)
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def pack(p):
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        q = Path(str(p) + ".gz")
        q.write_bytes(co.compress(p.read_bytes()) + co.flush())
        p.unlink()
        return q
    with ThreadPoolExecutor(16) as ex:
        files = list(ex.map(pack, files))
del host, fq
for t in threads:
    for route in (False, True, False, True):
        E.USE_MAPPED_UPLOAD = route
        out = tmp / f"img_{t}_{route}_{time.time_ns()}"
        tm = {}
        t0 = time.perf_counter()
        pipeline.fastqs_to_images(files, out, k=7, mapping_code="varKode", io_threads=t, engine=eng, timings=tm)
        dt = time.perf_counter() - t0
        print(f"io_threads {t:2d} {'mapped' if route else 'staged'}: {dt:.3f} s = {nfiles * reads * 150 / dt / 1e9:5.2f} Gbases/s "
              f"(stage wait {tm['stage_wait_s']:.3f}, upload {tm['upload_s']:.3f})", flush=True)
        shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(tmp, ignore_errors=True)
