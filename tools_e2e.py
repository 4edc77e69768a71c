#!/usr/bin/env python3
"""End-to-end file pipeline (split FASTQ files on disk -> PNGs on disk) on one GPU; a side
measurement quoted in DESIGN.md, not part of the bench contract."""
import json
import shutil
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, ".")
from varkoder_amd import pipeline  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402

nfiles, reads = int(sys.argv[1]) if len(sys.argv) > 1 else 48, int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
tmp = Path(tempfile.mkdtemp(prefix="vk_e2e_"))
eng = ImageEngine(k=7, mapping="varKode")
fq, offs, lens = eng.synth(0, nfiles, reads, 150)
host = fq.cpu().numpy()
files = []
for i in range(nfiles):
    f = tmp / f"s{i:04d}@{reads * 150 // 1000:08d}K.fq"
    host[int(offs[i]):int(offs[i]) + int(lens[i])].tofile(f)
    files.append(f)
del host
res = {}
for threads in (4, 16):
    out = tmp / f"img{threads}"
    t0 = time.perf_counter()
    stats = pipeline.fastqs_to_images(files, out, k=7, mapping_code="varKode", io_threads=threads, engine=eng,
                                      batch_bytes=1 << 30)
    dt = time.perf_counter() - t0
    assert len(stats) == nfiles and all("failed_step" not in v for v in stats.values())
    res[f"io_threads_{threads}"] = {"files": nfiles, "reads_per_file": reads, "seconds": dt,
                                    "files_per_s": nfiles / dt, "gbases_per_s": nfiles * reads * 150 / dt / 1e9}
shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps(res))
