"""Batched steps D+E of run_clean2img (commands/image.py:1054-1127) for already cleaned and
split FASTQ files: many files per launch, images written by a host thread pool, optional
sharding over the ranks of a torch.distributed job.

Steps B/C of the reference (fastp, reformat.sh) are external tools that are out of this
path's scope; this module enters where the reference enters step D: with
files named `<sample>@<bp>K.fq[.gz]` as split_fastq leaves them (image.py:699-709).
"""
import time
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path


from .config import QUAL_THRESH
from .image import counts_name, eprint, png_name, shard_folder, write_png
from .shard import TailQueue, agreed_weights, file_weights, gz_text_bytes, shard_by_size, split_head_tail


# Text bytes in HBM per batch.  Plain files: small enough that reading the next batch from disk overlaps
# the copy of this one.  gzip files are inflated on the GPU, which wants many files in flight, and their
# staging is cheap: larger batches.  The first batch of a run is cut short so that the device has work
# while the bulk of the files is still being read.  (bench.py's end-to-end leg runs with these defaults.)
DEFAULT_BATCH_BYTES = 2 << 30
DEFAULT_GZ_BATCH_BYTES = 16 << 30
FIRST_BATCH_BYTES = 512 << 20


def image_name(fastq_path, k, mapping_code):
    """`<sample>@<bp>K+<mapping>+k<k>.png` straight from a split FASTQ's name."""
    return png_name(counts_name(fastq_path, k), mapping_code)


class RouteChooser:
    """Plain-text files reach HBM by one of two routes (engine.plain_route): STAGED -- the I/O threads read() them into a
    pinned buffer, one DMA per batch -- or MAPPED -- the files' page-cache pages are mapped and the GPU copies out of
    them, no read() at all.  Which is faster depends on the host: with 16 fast cores the staged route runs at the link's
    rate (BENCH_r04: 56 of 57.6 GB/s); on a slower host the read() copies -- and the memory bandwidth they take from the DMA
    -- hold it at 44 (BENCH_r05).  So the run measures itself: three batches on the route it starts with (behind two that
    do not count); if they moved less than 85 % of what the link carries (measured once, 128 MiB pinned -> device), the
    same on the other route, and the rest on whichever was faster.  VARKODER_AMD_MMAP=0/1 pins the route and switches this off."""

    def __init__(self, eng, tm):
        from . import engine as E
        self.eng, self.tm, self.E = eng, tm, E
        self.on = E.USE_MAPPED_UPLOAD is None and getattr(eng, "route_override", None) is None and hasattr(eng, "h2d_link_rate")
        if getattr(eng, "route_rates", None):    # (an earlier pass of this engine measured and chose: say so in this pass's record)
            tm["plain_route_rates_gb_s"] = dict(eng.route_rates)
        self.rate = {}          # route -> [plain bytes, seconds, batches] over the batches that count
        self.seen = {}          # route -> batches seen
        self.first = None       # the route the run began on
        self.trial = False      # the other route is being tried

    def batch_done(self, bi, staged, seconds):
        if not isinstance(staged, dict) or "disk" not in staged:
            return
        route = staged.get("plain_route", "staged")   # (as stage_files brought this batch in)
        self.tm["plain_route"] = route
        if not self.on:
            return
        plain = int(staged["disk"][~staged["is_gz"]].sum()) if len(staged["disk"]) else 0
        if plain == 0:
            return
        # the first batches of a route do not count: the run's short first batch, the pinned staging buffers growing to
        # their size, the first registrations of mapped pages (BENCH r06, first cut: 21.8 GB/s "measured" for the staged route
        # in a pass that then ran at 51)
        self.seen[route] = self.seen.get(route, 0) + 1
        if self.seen[route] <= (3 if not self.trial else 2):   # (batch 0 is short; each of the two staging buffers grows once more after it)
            return
        acc = self.rate.setdefault(route, [0, 0.0, 0])
        acc[0] += plain
        acc[1] += seconds
        acc[2] += 1
        if self.first is None:
            self.first = route
        other = "staged" if self.first == "mapped" else "mapped"
        if not self.trial:
            if route == self.first and acc[2] >= 3:
                link = self.eng.h2d_link_rate()
                got = acc[0] / acc[1]
                self.tm["plain_route_rates_gb_s"] = {route: got / 1e9, "link": link / 1e9}
                if got < 0.85 * link:
                    self.eng.route_override = other
                    self.trial = True
                else:
                    self.eng.route_override = route   # (settled: later passes of this engine do not measure again)
                    self.tm["plain_route_rates_gb_s"]["chosen"] = route
                    self.on = False
                self.eng.route_rates = dict(self.tm["plain_route_rates_gb_s"])
        elif route == other and acc[2] >= 3:
            a, b = self.rate[self.first], self.rate[other]
            best = other if b[0] / b[1] > a[0] / a[1] else self.first
            self.eng.route_override = best
            self.tm["plain_route_rates_gb_s"].update({other: b[0] / b[1] / 1e9, "chosen": best})
            self.eng.route_rates = dict(self.tm["plain_route_rates_gb_s"])
            self.on = False


def fastqs_to_images(files, outdir, k=7, mapping_code="cgr", labels=None, base_sd=None, overwrite=False,
                     subfolder_levels=0, device=0, rank=0, world=1, batch_bytes=None, io_threads=8,
                     engine=None, verbose=False, timings=None, weights=None, tail_frac=0.1):
    """Process this rank's share of `files`.  Returns {sample_file_stem: OrderedDict(stats)}
    with the reference's stats keys `<k>mer_counting_time` and `k<k>_img_time` (per-file
    share of the batch wall time) or `failed_step` for files whose FASTQ framing is bad.
    batch_bytes: None = DEFAULT_BATCH_BYTES (DEFAULT_GZ_BATCH_BYTES when every file is gzip).
    weights: the files' work estimates as every rank of the job uses them (shard.agreed_weights -- a collective: pass them in
    when this call sits in a try block that a failing rank would leave early); None: agreed on here.
    timings: optional dict that receives where this thread's wall time went (seconds): waiting for the
    staging thread (`stage_wait_s`), the copy to the device and the inflate (`upload_s`, of which
    `inflate_s`), kernels + copies back (`kernels_s`) and waiting for the last hand-overs and PNGs (`png_tail_s`);
    `png_submit_s` = handing images to the PNG pool, on a thread of its own since round 5 (not this thread's time); `batches`."""
    from .engine import ImageEngine
    files = [Path(f) for f in files]
    labels = labels or {}
    base_sd = base_sd or {}
    if weights is None:   # (a collective only when this call itself is one rank's share of a job: bench.py's per-rank legs call with world = 1 inside a group)
        weights = agreed_weights(files) if world > 1 else file_weights(files)
    # Size-aware (rank 0's view of the sizes: shard.agreed_weights).  With a process group, the longest files -- nine tenths
    # of the weight -- are dealt statically, and the many small ones at the end of the order are pulled from a shared
    # cursor by whichever rank gets there first (shard.split_head_tail, TailQueue): the deal balances estimates, the
    # tail what they got wrong.  (The queue is made here, before anything a single rank could fail in: every rank
    # constructs it, none has to reach it.)
    import torch.distributed as dist
    grouped = world > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size() == world
    head, tail = split_head_tail(weights, tail_frac if grouped else 0.0)
    mine = [files[head[j]] for j in shard_by_size([weights[i] for i in head], rank, world)]
    tail_files = [files[i] for i in tail]
    tail_queue = TailQueue(len(tail_files)) if tail_files else None
    tail_chunk = max(1, len(tail_files) // (6 * world))   # files per claim: ~6 claims per rank
    eng = engine or ImageEngine(k=k, mapping=mapping_code, device=device)
    outdir = Path(outdir)
    outdir.mkdir(parents=True, exist_ok=True)
    stats = OrderedDict()
    pool = ThreadPoolExecutor(io_threads)
    pending = []

    def target(f):
        name = image_name(f, k, mapping_code)
        return shard_folder(outdir, name, subfolder_levels), name

    def wanted(fs):
        out = []
        for f in fs:
            d, name = target(f)
            if not overwrite and (d / name).is_file():
                eprint("File exists. Skipping image for file:", str(f))
                continue
            out.append(f)
        return out

    todo = wanted(mine)
    import os
    if batch_bytes is None:
        every = todo + tail_files
        batch_bytes = DEFAULT_GZ_BATCH_BYTES if every and all(f.suffix == ".gz" for f in every) else DEFAULT_BATCH_BYTES
    tm = timings if timings is not None else {}
    for key in ("stage_wait_s", "upload_s", "inflate_s", "kernels_s", "png_submit_s", "png_tail_s"):
        tm.setdefault(key, 0.0)

    def text_bytes(f):
        return gz_text_bytes(f) if f.suffix == ".gz" else os.path.getsize(f)

    def batch_source():
        """This rank's batches, by text size in HBM (a gzip file counts the text its framing names, shard.gz_text_bytes: it is
        inflated on the GPU): its static share first, then what it claims of the tail, a chunk per batch."""
        batch, nbytes, first = [], 0, True
        for f in todo:
            sz = text_bytes(f)
            if batch and nbytes + sz > (min(batch_bytes, FIRST_BATCH_BYTES) if first else batch_bytes):
                yield batch, nbytes
                batch, nbytes, first = [], 0, False
            batch.append(f)
            nbytes += sz
        if batch:
            yield batch, nbytes
        while tail_queue is not None:
            got = tail_queue.next(tail_chunk)
            if len(got) == 0:
                break
            batch = wanted([tail_files[i] for i in got])
            tm["tail_files"] = tm.get("tail_files", 0) + len(batch)
            if batch:
                yield batch, sum(text_bytes(f) for f in batch)
    # The host half of a batch (file reads into a pinned buffer) runs one batch ahead on its own
    # thread, into the other of two staging buffers, while this thread copies and processes.
    stager = ThreadPoolExecutor(1)
    finisher, handed = ThreadPoolExecutor(1), []
    done = False
    chooser = RouteChooser(eng, tm)
    try:
        source = batch_source()
        cur = next(source, None)
        staged = stager.submit(eng.stage_files, cur[0], pool, 0) if cur else None
        bi = -1
        while cur is not None:
            bi += 1
            batch, nbytes = cur
            t0 = time.perf_counter()
            ready = staged.result()
            tm["stage_wait_s"] += time.perf_counter() - t0
            cur = next(source, None)          # (a chunk of the tail is claimed here: one batch ahead, like the staging)
            if cur is not None:
                staged = stager.submit(eng.stage_files, cur[0], pool, (bi + 1) & 1)
            for h in handed:   # a hand-over that failed (a folder that cannot be made) stops the pass here, not after the last batch
                if h.done() and h.exception() is not None:
                    raise h.exception()
            tu = time.perf_counter()
            dev, offs, lens = eng.upload_staged(ready, timings=tm)
            t1 = time.perf_counter()
            tm["upload_s"] += t1 - tu
            img, hist, status = eng.fastq_to_images(dev, offs, lens)
            st = status.cpu().numpy()
            imgs = img.cpu().numpy()
            nz = (hist != 0).any(dim=1).cpu().numpy()
            t2 = time.perf_counter()
            tm["kernels_s"] += t2 - t1
            chooser.batch_done(bi, ready, t2 - t0)

            def hand_over(batch=batch, st=st, imgs=imgs, nz=nz, per_file=(t2 - t0) / len(batch)):
                # stats rows and PNG jobs of one batch: on a thread of its own (batches in order), beside the next batch's
                # upload and inflate, in whose C calls this thread's interpreter lock is free (9 % of a .fq.gz pass before)
                th = time.perf_counter()
                for j, f in enumerate(batch):
                    key = str(f.name.removesuffix("".join(f.suffixes)))
                    s = stats.setdefault(key, OrderedDict())
                    if st[j] or not nz[j]:
                        eprint("K-MER COUNTING FAIL, SKIPPING FILE:", f)
                        s["failed_step"] = "image"
                        continue
                    s[str(k) + "mer_counting_time"] = per_file
                    d, name = target(f)
                    d.mkdir(parents=True, exist_ok=True)
                    sample = key.split("@")[0]
                    sd = base_sd.get(sample, 0)
                    pending.append((key, time.perf_counter(),
                                    pool.submit(write_png, imgs[j].copy(), d / name, labels.get(sample, []), sd,
                                                QUAL_THRESH, mapping_code)))
                tm["png_submit_s"] += time.perf_counter() - th   # (this thread's time: off the main thread's path)

            handed.append(finisher.submit(hand_over))
            if verbose:
                eprint(f"batch of {len(batch)} files, {nbytes} bytes: upload {t1 - t0:.3f}s kernels {t2 - t1:.3f}s")
        tt = time.perf_counter()
        for h in handed:
            h.result()   # (an exception of a batch's hand-over surfaces here)
        for key, t, fut in pending:
            fut.result()
            stats[key]["k" + str(k) + "_img_time"] = time.perf_counter() - t
        done = True
    finally:
        # An error anywhere above (a copy that runs out of memory, a PNG that cannot be written) must not leave the staging
        # thread reading the next batch or queued hand-overs writing PNGs into outdir behind the caller's back.
        for ex in (stager, finisher, pool):
            ex.shutdown(wait=True, cancel_futures=not done)
        if not done and engine is None:
            eng.close()
    tm["png_tail_s"] += time.perf_counter() - tt
    tm["batches"] = tm.get("batches", 0) + bi + 1
    if engine is None:
        eng.close()
    return stats


def clean_to_images(files, outdir, k=7, mapping_code="cgr", min_bp=50000, max_bp=None, is_query=False, seeds=None,
                    labels=None, base_sd=None, subfolder_levels=0, device=0, rank=0, world=1, batch_bytes=None,
                    io_threads=8, engine=None, verbose=False, weights=None):
    """Steps C+D+E of run_clean2img (commands/image.py:1006-1127) for cleaned, UNSPLIT read files
    `<sample>.fq[.gz]` (the reference's `<int_folder>/clean_reads/`): the 1-2-5 ladder of subsamples
    is drawn on the GPU (subsample.ladder_counts) instead of writing one file per size with
    reformat.sh, and every step becomes `<sample>@<bp>K+<mapping>+k<k>.png`.

    seeds: {sample: int} (default 0).  Returns {sample: OrderedDict(stats)} with the reference's
    keys `splitting_bp_per_file`, `<k>mer_counting_time`, `k<k>_img_time`, or `failed_step`."""
    import os

    import torch

    from .engine import ImageEngine
    from .subsample import ladder_counts, split_name
    files = [Path(f) for f in files]
    labels, base_sd, seeds = labels or {}, base_sd or {}, seeds or {}
    if weights is None:
        weights = agreed_weights(files) if world > 1 else file_weights(files)   # (a collective when sharded: see fastqs_to_images)
    mine = [files[i] for i in shard_by_size(weights, rank, world)]   # size-aware; rank 0's view of the sizes: see shard.agreed_weights
    eng = engine or ImageEngine(k=k, mapping=mapping_code, device=device)
    outdir = Path(outdir)
    outdir.mkdir(parents=True, exist_ok=True)
    stats = OrderedDict()
    pool = ThreadPoolExecutor(io_threads)
    pending = []
    if batch_bytes is None:
        batch_bytes = DEFAULT_GZ_BATCH_BYTES if mine and all(f.suffix == ".gz" for f in mine) else DEFAULT_BATCH_BYTES
    done = False
    try:   # (an error below must not leave the pool writing PNGs into outdir behind the caller's back: see fastqs_to_images)
        i = 0
        while i < len(mine):
            batch, nbytes = [], 0
            t0 = time.perf_counter()
            for f in mine[i:]:
                sz = gz_text_bytes(f) if f.suffix == ".gz" else os.path.getsize(f)
                if batch and nbytes + sz > batch_bytes:
                    break
                batch.append(f)
                nbytes += sz
            i += len(batch)
            names = [str(f.name.removesuffix("".join(f.suffixes))) for f in batch]
            dev, offs, lens = eng.upload_files(batch, pool)
            recs = []
            # one seed per launch: samples with different seeds go in separate calls
            by_seed = OrderedDict()
            for j, s in enumerate(names):
                by_seed.setdefault(int(seeds.get(s, 0)), []).append(j)
            recs = [None] * len(batch)
            for seed, idx in by_seed.items():
                for j, r in zip(idx, ladder_counts(eng, dev, offs[idx], lens[idx], seed=seed, min_bp=min_bp,
                                                   max_bp=max_bp, is_query=is_query)):
                    recs[j] = r
            t1 = time.perf_counter()
            flat = [(j, bp, h) for j, r in enumerate(recs) for bp, h, _ in r["steps"]]
            imgs = eng.images(torch.stack([h for _, _, h in flat])).cpu().numpy() if flat else []
            nz = [bool((h != 0).any().item()) for _, _, h in flat]
            t2 = time.perf_counter()
            for j, s in enumerate(names):
                st = stats.setdefault(s, OrderedDict())
                if recs[j]["error"]:
                    eprint("SPLIT FAIL:", batch[j], "-", recs[j]["error"])
                    st["failed_step"] = "split"
                    continue
                st["splitting_time"] = (t1 - t0) / len(batch)
                st["splitting_bp_per_file"] = ",".join(str(bp) for bp, _, _ in recs[j]["steps"])
                st[str(k) + "mer_counting_time"] = (t1 - t0) / len(batch)
            for n, (j, bp, _) in enumerate(flat):
                s = names[j]
                if not nz[n]:
                    eprint("IMAGE FAIL:", split_name(s, bp))
                    stats[s]["failed_step"] = "image"
                    continue
                name = png_name(split_name(s, bp) + "+k" + str(k) + ".fq.h5", mapping_code)
                d = shard_folder(outdir, name, subfolder_levels)
                d.mkdir(parents=True, exist_ok=True)
                pending.append((s, time.perf_counter(),
                                pool.submit(write_png, imgs[n].copy(), d / name, labels.get(s, []), base_sd.get(s, 0),
                                            QUAL_THRESH, mapping_code)))
            if verbose:
                eprint(f"batch of {len(batch)} samples, {nbytes} bytes: upload+ladder {t1 - t0:.3f}s images {t2 - t1:.3f}s")
        for s, t, fut in pending:
            fut.result()
            key = "k" + str(k) + "_img_time"
            stats[s][key] = stats[s].get(key, 0) + (time.perf_counter() - t)
        done = True
    finally:
        pool.shutdown(wait=True, cancel_futures=not done)
        if not done and engine is None:
            eng.close()
    if engine is None:
        eng.close()
    return stats
