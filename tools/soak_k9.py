#!/usr/bin/env python3
"""One-off soak (GPU box): full-size samples (1M x 150 bp) through the k=9 and k=8 spill path and
through the subsampling launch (k=7 and k=9), bit for bit against the oracle."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from varkoder_amd.engine import ImageEngine  # noqa: E402
from varkoder_amd.mapping import pixel_lut, side  # noqa: E402

bad = 0
for k in (9, 8):
    eng = ImageEngine(k=k, mapping="cgr")
    for dist in (0, 1):
        n = 6
        fq, offs, lens = eng.synth(7000 + 10 * dist, n, 1_000_000, 150, dist=dist)
        host = fq.cpu().numpy()
        blobs = [host[int(o):int(o) + int(l)] for o, l in zip(offs, lens)]
        want = [oracle.count_fastq(b, k)[0] for b in blobs]
        for parts in (0, 3):
            img, hist, status = eng.fastq_to_images(fq, offs, lens, parts=parts)
            got = hist.cpu().numpy().view(np.uint32)
            ok = all(np.array_equal(got[i], want[i]) for i in range(n)) and not status.cpu().numpy().any()
            lut, s = pixel_lut(k, "cgr"), side(k, "cgr")
            im = img.cpu().numpy()
            ok_img = all(np.array_equal(im[i].ravel(), oracle.image(oracle.strand_merge(want[i], k), k, lut, s * s))
                         for i in range(2))
            print(f"k={k} dist {dist} parts {parts}: counts {'exact' if ok else 'MISMATCH'}, images "
                  f"{'exact' if ok_img else 'MISMATCH'}", flush=True)
            bad += (not ok) + (not ok_img)
        if k == 9:
            for seed, thr in ((3, 1 << 30), (4, (1 << 32) // 50)):
                h, st, sites = eng.count_sampled(fq, offs, lens, seed, thr)
                g = h.cpu().numpy().view(np.uint32)
                si = sites.cpu().numpy()
                ok = True
                for i in range(n):
                    w, _, wst, ws = oracle.count_fastq_sampled(blobs[i], k, seed, thr)
                    ok &= np.array_equal(g[i], w) and tuple(int(x) for x in si[i]) == ws and wst == 0
                print(f"k={k} dist {dist} sampled seed {seed}: {'exact' if ok else 'MISMATCH'}", flush=True)
                bad += not ok
        del fq, host, blobs
        torch.cuda.empty_cache()
    eng.close()

eng = ImageEngine(k=7, mapping="cgr")
fq, offs, lens = eng.synth(9000, 8, 1_000_000, 150, dist=1)
host = fq.cpu().numpy()
for seed, thr, parts in ((11, 1 << 31, 0), (12, (1 << 32) // 300, 2)):
    h, st, sites = eng.count_sampled(fq, offs, lens, seed, thr, parts=parts)
    g = h.cpu().numpy().view(np.uint32)
    si = sites.cpu().numpy()
    ok = True
    for i in range(8):
        w, _, wst, ws = oracle.count_fastq_sampled(host[int(offs[i]):int(offs[i]) + int(lens[i])], 7, seed, thr)
        ok &= np.array_equal(g[i], w) and tuple(int(x) for x in si[i]) == ws and wst == 0
    print(f"k=7 sampled seed {seed} parts {parts}: {'exact' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
