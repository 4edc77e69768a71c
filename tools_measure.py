#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not part of the bench contract): other configs,
the skewed base distribution, and the PCIe-inclusive rate when the boundary hands over host
buffers.  Usage on the GPU box: python tools_measure.py > gpurun_out/measure.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from varkoder_amd.engine import ImageEngine  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def config(k, mapping, samples, pool, dist, reads=1_000_000, parts=0, reps=2):
    eng = ImageEngine(k=k, mapping=mapping)
    fq, po, pl = eng.synth(0, pool, reads, 150, dist=dist)
    idx = np.arange(samples) % pool
    offs, lens = po[idx].copy(), pl[idx].copy()
    hist = torch.empty((samples, 4 ** k), dtype=torch.int32, device="cuda")
    status = torch.empty((samples,), dtype=torch.int32, device="cuda")
    img = torch.empty((samples, eng.side, eng.side), dtype=torch.uint8, device="cuda")
    tc = timed(lambda: eng.count(fq, offs, lens, parts=parts, hist=hist, status=status), reps)
    ti = timed(lambda: eng.images(hist, img=img), reps)
    bases = samples * reads * 150
    out = {"k": k, "mapping": mapping, "samples": samples, "pool": pool, "dist": dist, "parts": parts,
           "count_ms": tc * 1e3, "image_ms": ti * 1e3, "gbases_per_s": bases / (tc + ti) / 1e9,
           "count_GBps": samples * (320e6 + 4 * 4 ** k) / tc / 1e9, "launch": eng.last_count_launch(),
           "bad": int((status != 0).sum().item())}
    eng.close()
    del fq
    torch.cuda.empty_cache()
    return out


def pcie_inclusive(nsamples=8, reads=1_000_000):
    """Host (pinned) FASTQ -> H2D -> kernels -> D2H images, double buffered on two streams."""
    eng = ImageEngine(k=7, mapping="varKode")
    fq, po, pl = eng.synth(0, 1, reads, 150)
    n = int(pl[0])
    host = torch.empty(n + 16, dtype=torch.uint8).pin_memory()
    host[:n + 16].copy_(fq[:n + 16])
    bufs = [torch.empty(n + 16, dtype=torch.uint8, device="cuda") for _ in range(2)]
    out_host = torch.empty((eng.side, eng.side), dtype=torch.uint8).pin_memory()
    copy_stream = torch.cuda.Stream()
    offs, lens = np.zeros(1, np.uint64), np.array([n], np.uint64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [None, None]
    for i in range(nsamples):
        b = i & 1
        with torch.cuda.stream(copy_stream):
            bufs[b].copy_(host, non_blocking=True)
            e = torch.cuda.Event()
            e.record()
        torch.cuda.current_stream().wait_event(e)
        img, hist, st = eng.fastq_to_images(bufs[b], offs, lens)
        out_host.copy_(img[0], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        copy_stream.wait_event(done)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.close()
    return {"samples": nsamples, "seconds": dt, "gbases_per_s": nsamples * reads * 150 / dt / 1e9,
            "h2d_GBps": nsamples * n / dt / 1e9}


if __name__ == "__main__":
    res = {}
    res["k7_varKode_uniform"] = config(7, "varKode", 1000, 256, 0)
    res["k7_varKode_skew"] = config(7, "varKode", 1000, 256, 1)
    res["k7_cgr_uniform"] = config(7, "cgr", 1000, 256, 0)
    res["k9_cgr_100"] = config(9, "cgr", 100, 100, 0, reps=1)
    res["k5_cgr"] = config(5, "cgr", 1000, 256, 0)
    res["single_sample_latency_parts_auto"] = config(7, "varKode", 1, 1, 0, reps=5)
    res["pcie_inclusive"] = pcie_inclusive()
    print(json.dumps(res, indent=1))
