#!/usr/bin/env python3
"""bench.py -- throughput of the `image` hot path on MI355X.

Metric (BASELINE.json): Gbases/s (+ samples/s) for `varKoder image`, k=7, 150 bp reads.
Workload at N=1 = BASELINE.json configs[1]: 1000 synthetic samples x 1M 150 bp reads,
k=7 varKode (91x91), 1x MI355X.  One "step" = one pass of the whole hot path
(FASTQ text resident in HBM -> k-mer histograms -> uint8 images) over that batch.

320 GB of distinct text does not fit one GPU, so -- as SURVEY.md 8d prescribes -- a pool
of `--pool` distinct samples (default 512 = 164 GB: one per workgroup the 256 CUs keep resident, two
per CU, so that no two workgroups running at the same time read the same bytes) is generated on the
device and the batch of 1000 cycles through it; every batch entry still gets its own histogram and
image.  Generation is outside the timed region.

Side legs at N=1, each in the same JSON line and none of them `value`: `config4` (BASELINE.json
configs[3]: k=9 cgr, 100 samples, both base distributions -- the LDS-spill path), `cpu_baseline`,
`end_to_end` (files on disk -> PNG files).

N>1 = BASELINE.json configs[2]: 10000 samples sharded over the N GPUs (ceil(10000/N) per rank per
step, no data-path collective: samples are independent, as in the reference's sample-level pool,
varKoder/commands/image.py:1281-1284); every rank keeps its own pool of distinct samples.  The only
collectives are the barrier and the MAX-reduce of the elapsed time.  `--gpus N` without a launcher
(no WORLD_SIZE in the environment) starts the N ranks itself, as fresh child processes, before
anything in this process touches the GPU; under torchrun (the driver's way) it must equal WORLD_SIZE.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=0, help="samples per GPU per step (default: 1000 at N=1 = "
                    "configs[1]; ceil(total/N) at N>1 = configs[2])")
    ap.add_argument("--total-samples", type=int, default=10000, help="samples per step over all ranks at N>1")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--readlen", type=int, default=150)
    ap.add_argument("--k", type=int, default=7)
    ap.add_argument("--mapping", default="varKode")
    ap.add_argument("--pool", type=int, default=512, help="distinct samples resident in HBM (512 = one per resident workgroup)")
    ap.add_argument("--dist", type=int, default=0, help="0 uniform, 1 GC-skew + homopolymers")
    ap.add_argument("--parts", type=int, default=0, help="workgroups per sample (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> PNGs measurement (N=1 only)")
    ap.add_argument("--no-config4", action="store_true", help="skip the k=9 side leg (N=1 only)")
    ap.add_argument("--config4-samples", type=int, default=100)
    ap.add_argument("--config4-steps", type=int, default=8)
    ap.add_argument("--no-ladder", action="store_true", help="skip the subsample-ladder leg (N=1 only)")
    ap.add_argument("--ladder-shard-samples", type=int, default=256, help="samples of the N > 1 leg with ladder-shaped (unequal) units; 0 = skip")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run this script again under rocprofv3 --pmc for roofline.traffic "
                    "(the figure then comes from profiles/traffic_latest.json when that matches the configuration)")
    ap.add_argument("--no-query", action="store_true", help="skip the images -> preprocess -> forward leg (BASELINE configs[4], N=1 only)")
    ap.add_argument("--query-samples", type=int, default=512)
    ap.add_argument("--query-batch", type=int, default=256)
    ap.add_argument("--query-steps", type=int, default=3)
    ap.add_argument("--ladder-samples", type=int, default=32)
    ap.add_argument("--no-realistic", action="store_true", help="skip the fastp-shaped read-length leg (N=1 only)")
    ap.add_argument("--realistic-pool", type=int, default=256)
    ap.add_argument("--realistic-steps", type=int, default=3)
    ap.add_argument("--e2e-files", type=int, default=256)
    ap.add_argument("--e2e-reads", type=int, default=560_000, help="reads per file of the end-to-end measurement "
                    "(256 x 560k x 150 bp = 21.5 Gbases: about a second per pass)")
    ap.add_argument("--e2e-passes", type=int, default=3)
    ap.add_argument("--e2e-files-per-rank", type=int, default=24, help="files of the end-to-end leg at N > 1 (per rank)")
    ap.add_argument("--e2e-io-threads", type=int, default=0, help="I/O threads per rank of the N > 1 end-to-end leg (0 = this "
                    "rank's share of the usable cores; a rehearsal with fewer ranks than the node has GPUs sets the node's share)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every core this "
                    "process may use: physical cores, capped by affinity and the cgroup's CPU quota)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N>1 (nccl = RCCL); "
                    "gloo allows a 2-rank rehearsal on a single GPU together with --all-on-device0")
    ap.add_argument("--all-on-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    return ap.parse_args()


def cpu_budget():
    """Cores this process can really use: physical cores of the host (unique physical id / core id
    pairs in /proc/cpuinfo), capped by the scheduler affinity and by the cgroup's CPU quota."""
    host = os.cpu_count() or 1
    phys = set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pid = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    physical = len(phys) or host
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = host
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    quota = float(tok[0]) / float(tok[1])
            else:
                q = float(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        quota = q / float(g.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = min(physical, affinity)
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return {"host_cpus": host, "physical_cores": physical, "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
            "usable_cores": usable}


class VerifyError(RuntimeError):
    pass


def cpu_baseline(eng, fastq_dev, offs, lens, args, seconds, gpu_hist=None, gpu_img=None):
    """The oracle's C restatement ("port") timed on this host's cores on a bounded sample
    of the same workload: whole samples of the device-generated pool, copied back.  Runs on every
    core this process may use (cpu_budget) and, for the record, on one thread.
    The samples it counts are the first entries of the timed batch: their histograms and images as the GPU
    left them in the last timed step (gpu_hist, gpu_img) must equal the oracle's -- `verified_samples`; a
    difference raises VerifyError and the bench exits non-zero."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from varkoder_amd.mapping import pixel_lut, side
    oracle.lib()
    lut = pixel_lut(args.k, args.mapping)
    n = side(args.k, args.mapping)
    nbuf = min(4, len(offs))
    bufs = [fastq_dev[int(offs[i]):int(offs[i]) + int(lens[i])].cpu().numpy() for i in range(nbuf)]
    bases_per_sample = args.reads * args.readlen
    budget = cpu_budget()
    cores = args.cpu_threads if args.cpu_threads > 0 else budget["usable_cores"]
    verified = 0
    if gpu_hist is not None and gpu_img is not None:
        with ThreadPoolExecutor(nbuf) as ex:
            want = list(ex.map(lambda i: (oracle.count_fastq(bufs[i], args.k), oracle.fastq_to_image(bufs[i], args.k, lut, n * n)), range(nbuf)))
        for i, ((wh, _, st), (wi, _, _)) in enumerate(want):
            gh = gpu_hist[i].cpu().numpy().view(np.uint32)
            gi = gpu_img[i].cpu().numpy().ravel()
            if st != 0 or not np.array_equal(gh, wh) or not np.array_equal(gi, wi):
                raise VerifyError("batch entry %d: the GPU's histogram / image differs from the oracle's" % i)
            verified += 1

    def one(i):
        img, nwin, st = oracle.fastq_to_image(bufs[i % nbuf], args.k, lut, n * n)
        assert st == 0
        return nwin

    # single thread first (one sample), to size the multi-thread run
    t0 = time.perf_counter()
    one(0)
    t1 = time.perf_counter() - t0
    per_thread = max(1, int(seconds / max(t1, 1e-3)))
    nsamp = min(cores * per_thread, 64 * cores)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(one, range(nsamp)))
    dt = time.perf_counter() - t0
    out = {"value": nsamp * bases_per_sample / dt / 1e9, "unit": "Gbases/s", "cores": cores,
           "kind": "port", "single_thread_value": bases_per_sample / t1 / 1e9, "cpu_model": cpu_model(),
           "speedup_over_one_thread": (nsamp * bases_per_sample / dt) / (bases_per_sample / t1),
           "sample": f"{nsamp} samples of {args.reads} x {args.readlen} bp (FASTQ->counts->image, "
                     f"oracle/vk_oracle.c, {cores} threads, {dt:.1f} s)",
           "verified_samples": verified}
    out.update(budget)
    ref = dsk_reference(bufs[0], args, cores)
    if ref:
        out["dsk"] = ref
    return out


def end_to_end(eng, args):
    """SURVEY 8d's second timing: real files on disk -> PNG files on disk through the file pipeline
    (parallel reads into a pinned buffer, one H2D DMA per batch, kernels, PNG encode), once for plain
    FASTQ text and once for the .fq.gz files step C of the reference really hands over
    (commands/image.py:696-708).  Files come from the synthetic generator (written here, outside the
    timed region, and read back from the page cache)."""
    import shutil
    import tempfile
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from pathlib import Path

    import torch
    from varkoder_amd import pipeline
    nfiles, reads = args.e2e_files, args.e2e_reads
    threads = cpu_budget()["usable_cores"]
    tmp = Path(tempfile.mkdtemp(prefix="vk_e2e_"))
    # the files (plain + gzip, ~1.3x the text) must fit the temporary directory, and the text the page cache
    rec = 2 * args.readlen + 20
    free = shutil.disk_usage(tmp).free
    try:
        with open("/proc/meminfo") as f:
            avail = next(int(l.split()[1]) * 1024 for l in f if l.startswith("MemAvailable"))
    except Exception:
        avail = free
    room = min(free, avail // 2)
    while nfiles > 16 and nfiles * reads * rec * 1.35 > room:
        nfiles //= 2
    from varkoder_amd.engine import plain_route
    out = {"files": nfiles, "files_asked": args.e2e_files, "reads_per_file": reads, "read_len": args.readlen,
           "io_threads": threads, "plain_text_route": plain_route(threads, eng) + " at the start (a run that waits for its staging "
                                                      "threads goes over to the mapped route: where_the_median_pass_went_s."
                                                      "plain_route_switched_at_batch)"}
    try:
        # generated on the device in slabs of 32 files and copied back (a 46 GB tensor at once would also do,
        # but the pool of the main measurement is still resident)
        host_parts, offs, lens = [], [], []
        for f0 in range(0, nfiles, 32):
            nf = min(32, nfiles - f0)
            fq, o, l = eng.synth((1 << 20) + f0, nf, reads, args.readlen, dist=args.dist)
            host_parts.append((fq.cpu().numpy(), o, l))
            del fq
        torch.cuda.empty_cache()
        kb = reads * args.readlen // 1000
        plain = [tmp / f"s{i:04d}@{kb:08d}K.fq" for i in range(nfiles)]
        gz = [tmp / "gz" / f"s{i:04d}@{kb:08d}K.fq.gz" for i in range(nfiles)]
        (tmp / "gz").mkdir()

        def write(i):
            host, o, l = host_parts[i // 32]
            blob = host[int(o[i % 32]):int(o[i % 32]) + int(l[i % 32])]
            blob.tofile(plain[i])
            co = zlib.compressobj(1, zlib.DEFLATED, 31)       # gzip container, fast level
            with open(gz[i], "wb") as f:
                f.write(co.compress(blob.tobytes()) + co.flush())
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(write, range(nfiles)))
        text_bytes = nfiles * reads * rec
        gz_bytes = sum(p.stat().st_size for p in gz)
        del host_parts
        bases = nfiles * reads * args.readlen
        # batch sizes are the pipeline's own defaults (what `python -m varkoder_amd image` runs with)
        for name, files, moved in (("plain_text", plain, text_bytes), ("fq_gz", gz, gz_bytes)):
            # one untimed pass first: staging buffers and device workspaces are allocated once per process
            pipeline.fastqs_to_images(files, tmp / ("warm_" + name), k=args.k, mapping_code=args.mapping,
                                      io_threads=threads, engine=eng)
            shutil.rmtree(tmp / ("warm_" + name), ignore_errors=True)
            # timed passes, each into a fresh folder; the MEDIAN is quoted, all are listed
            passes, parts = [], []
            for rep in range(max(1, args.e2e_passes)):
                dst = tmp / ("img%d_%s" % (rep, name))
                tm = {}
                t0 = time.perf_counter()
                stats = pipeline.fastqs_to_images(files, dst, k=args.k, mapping_code=args.mapping, io_threads=threads,
                                                  engine=eng, timings=tm)
                passes.append(time.perf_counter() - t0)
                parts.append(tm)
            order = sorted(range(len(passes)), key=lambda i: passes[i])
            mid = order[len(order) // 2]
            dt = passes[mid]
            dst = tmp / ("img0_" + name)
            ok = len(stats) == nfiles and all("failed_step" not in v for v in stats.values())
            out[name] = {"seconds": dt, "passes_s": passes, "gbases_per_s": bases / dt / 1e9, "files_per_s": nfiles / dt,
                         "file_bytes": moved, "file_gb_per_s": moved / dt / 1e9, "text_gb_per_s": text_bytes / dt / 1e9,
                         "batch_bytes": "pipeline default", "all_files_ok": ok, "pngs": len(list(dst.rglob("*.png"))),
                         "where_the_median_pass_went_s": {k: round(v, 4) if isinstance(v, float) else v
                                                          for k, v in parts[mid].items()}}
        # the two routes must give the same images
        same = 0
        for p in sorted((tmp / "img0_plain_text").rglob("*.png")):
            q = tmp / "img0_fq_gz" / p.name
            same += int(q.is_file() and q.read_bytes() == p.read_bytes())
        out["gz_pngs_identical_to_plain"] = same == nfiles
        # the link itself, for scale: one pinned 1 GiB buffer, host to device
        pin = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
        dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        out["pcie_h2d_gb_per_s_pinned_1GiB"] = 4 * (1 << 30) / (time.perf_counter() - t0) / 1e9
        out["note"] = ("files -> PNGs, page cache warm, one GPU, the median of the timed passes; the breakdown is the "
                       "main thread's wall time (staging of the next batch runs beside it); .fq.gz files stay compressed in the "
                       "pinned staging buffer, the inflate kernels read them over PCIe and write the text to HBM "
                       "(vk_inflate_device)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def end_to_end_ranks(eng, args, rank, world, dist, red_dev):
    """The end-to-end leg at N > 1: every rank runs the file pipeline on its OWN set of files (written by the
    rank, page cache warm) at the same time -- what a node-level run of `torchrun -m varkoder_amd image` does
    to the host: `world` staging pools reading the page cache and `world` H2D streams at once.  Passes start
    at a barrier; a pass's aggregate is all ranks' bases over the slowest rank's time; the median pass is
    quoted, with every rank's time.  Fewer and smaller than the N=1 leg (the node's disk and page cache hold
    every rank's files).  Plain text only."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from pathlib import Path

    import torch
    from varkoder_amd import pipeline
    from varkoder_amd.shard import usable_cores
    nfiles, reads = args.e2e_files_per_rank, args.e2e_reads
    threads = args.e2e_io_threads if args.e2e_io_threads > 0 else max(1, usable_cores() // world)
    tmp = Path(tempfile.mkdtemp(prefix="vk_e2e_r%d_" % rank))

    def everyone(ok):
        """Did every rank get here in one piece?  (A collective every rank reaches whatever happened to it: a rank
        that failed must not leave the others waiting in the next barrier.)"""
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    res, err, files = None, None, []
    try:
        try:
            fq, offs, lens = eng.synth((2 << 20) + rank * nfiles, nfiles, reads, args.readlen, dist=args.dist)
            host = fq.cpu().numpy()
            del fq
            torch.cuda.empty_cache()
            kb = reads * args.readlen // 1000
            files = [tmp / f"r{rank}s{i:04d}@{kb:08d}K.fq" for i in range(nfiles)]
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(lambda i: host[int(offs[i]):int(offs[i]) + int(lens[i])].tofile(files[i]), range(nfiles)))
            del host
            pipeline.fastqs_to_images(files, tmp / "warm", k=args.k, mapping_code=args.mapping, io_threads=threads, engine=eng)
            shutil.rmtree(tmp / "warm", ignore_errors=True)
        except Exception as e:  # noqa: BLE001 -- a side measurement: never lose the bench line over it
            err = repr(e)
        if not everyone(err is None):
            return {"error": err or "another rank failed while writing its files"}
        per_pass = []
        for rep in range(max(1, args.e2e_passes)):
            dist.barrier()
            dt = -1.0
            try:
                t0 = time.perf_counter()
                stats = pipeline.fastqs_to_images(files, tmp / ("img%d" % rep), k=args.k, mapping_code=args.mapping,
                                                  io_threads=threads, engine=eng)
                dt = time.perf_counter() - t0
                if not (len(stats) == nfiles and all("failed_step" not in v for v in stats.values())):
                    dt = -1.0
            except Exception as e:  # noqa: BLE001
                err = repr(e)
            mine = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_pass.append([float(x.item()) for x in every])
            if min(per_pass[-1]) < 0:      # (every rank sees the same list: all leave together)
                break
        bases = world * nfiles * reads * args.readlen
        good = [p for p in per_pass if min(p) > 0]
        from varkoder_amd.engine import plain_route
        res = {"files_per_rank": nfiles, "reads_per_file": reads, "io_threads_per_rank": threads,
               "plain_text_route": plain_route(threads), "passes_s_by_rank": per_pass,
               "all_files_ok": len(good) == len(per_pass),
               "note": "plain-text files -> PNGs on every rank at once, page cache warm; aggregate = all ranks' bases / "
                       "slowest rank, median pass"}
        if good:
            worst = [max(p) for p in good]
            mid = sorted(range(len(worst)), key=lambda i: worst[i])[len(worst) // 2]
            res.update(gbases_per_s=bases / worst[mid] / 1e9, seconds=worst[mid])
        if err:
            res["error"] = err
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def live_traffic(args, k, mapping, samples, pool, dist_code, timeout=180.0):
    """HBM bytes per count launch, measured NOW: this script run again as a child of `rocprofv3 --pmc FETCH_SIZE` and of
    `rocprofv3 --pmc WRITE_SIZE` (separate passes, counters only: MI355X_MICROARCH.md "HBM"), one warm-up and one timed launch
    of the same configuration, every side leg off.  FETCH_SIZE (KB) x 2 -- gfx950 tallies a wide streaming read at half --
    + WRITE_SIZE (KB), per dispatch, summed over the count path's kernels (k <= 7: the count kernel alone, as
    profiles/summarize.py does; k = 8, 9: every vk_bucket_* / vk_quad_* kernel, as profiles/summarize_k9.py does).
    Returns (bytes or None, note).  The parent must have freed its sample pool: the child allocates its own."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--samples", str(samples), "--pool", str(pool),
             "--reads", str(args.reads), "--readlen", str(args.readlen), "--k", str(k), "--mapping", mapping, "--dist", str(dist_code),
             "--parts", str(args.parts), "--no-cpu-baseline", "--no-e2e", "--no-config4", "--no-realistic", "--no-ladder", "--no-query",
             "--no-live-traffic"]
    pat = re.compile(r"vk_count_dense_kernel|vk_count_kernel" if k <= 7 else r"vk_bucket\w*|vk_quad\w*")
    env = dict(os.environ, TMPDIR="/tmp")
    for drop in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        env.pop(drop, None)
    total = {}
    child_ms = {}
    with tempfile.TemporaryDirectory(prefix="vk_traffic_", dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            try:
                r = subprocess.run([prof, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + child,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env, cwd="/tmp")
            except Exception as e:  # noqa: BLE001
                return None, "rocprofv3 child failed: %r" % (e,)
            path = None
            for root, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if r.returncode != 0 or path is None:
                return None, "rocprofv3 child: rc %d, %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])
            try:   # the child's own bench line: its launch time under the counters, beside the timed steps' of this process
                line = json.loads(r.stdout.decode(errors="replace").strip().splitlines()[-1])
                child_ms[ctr] = float(line["roofline"]["avg_launch_ms"])
            except Exception:  # noqa: BLE001
                child_ms[ctr] = None
            per_kernel = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    m = pat.search(row["Kernel_Name"])
                    if m and row["Counter_Name"] == ctr:
                        per_kernel.setdefault(m.group(0), []).append(float(row["Counter_Value"]))
            if not per_kernel:
                return None, "no %s rows for the count kernels" % ctr
            total[ctr] = sum(sum(v) / len(v) for v in per_kernel.values())
    return total["FETCH_SIZE"] * 1024 * 2 + total["WRITE_SIZE"] * 1024, \
        "measured_in_this_run (this script under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate child processes, the same " \
        "configuration; FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md); the children's own count launch under the counters took " \
        "%s / %s ms (avg_launch_ms of this line: the timed steps of this process)" % tuple(
            "%.2f" % child_ms[c] if child_ms.get(c) else "?" for c in ("FETCH_SIZE", "WRITE_SIZE"))


def ladder_shard(eng, args, rank, world, dist, red_dev):
    """N > 1: units of UNEQUAL size -- the files `split_fastqs/` holds for `--ladder-shard-samples` samples, one per rung of
    the reference's test ladder (-m 500K -M 20M: 500K, 1M, 2M, 5M, 10M, 20M bases, commands/image.py:682-708,
    tests/02_constants.sh:32), sorted by name as the CLI lists them -- sharded over the ranks by size (shard.shard_by_size,
    what `varkoder_amd image` does) and counted + imaged from HBM, every rank its own share at the same time.  Reported:
    every rank's bytes and milliseconds, and the bytes the static i % world rule would have dealt (not run)."""
    import torch
    from varkoder_amd import shard
    rungs_kbp = [500, 1000, 2000, 5000, 10000, 20000]
    ns = args.ladder_shard_samples
    names = sorted("s%03d@%08dK" % (s, kb) for s in range(ns) for kb in rungs_kbp)
    reads = [max(1, int(n.split("@")[1][:8]) * 1000 // args.readlen) for n in names]
    rec = 2 * args.readlen + 20
    weights = [r * rec for r in reads]
    # as pipeline.fastqs_to_images deals files: the head of the longest-first order statically, the last tenth of the weight
    # pulled from a shared cursor by whichever rank is free (every rank keeps the tail's units in HBM: a tenth of the bytes)
    head, tail = shard.split_head_tail(weights, 0.1)
    mine = [head[j] for j in shard.shard_by_size([weights[i] for i in head], rank, world)]
    chunk = max(1, len(tail) // (6 * world))
    static_loads = shard.rank_loads(weights, world)
    rr = [sum(weights[i] for i in shard.shard_indices(len(names), r, world)) for r in range(world)]
    err, per_pass, took_bytes = None, [], []
    try:
        units = mine + tail
        offs = np.zeros(len(units), dtype=np.uint64)
        lens = np.array([weights[i] for i in units], dtype=np.uint64)
        if len(units) > 1:
            offs[1:] = np.cumsum((lens[:-1] + np.uint64(15)) // np.uint64(16) * np.uint64(16))
        total = int(offs[-1] + lens[-1]) if len(units) else 0
        buf = torch.empty(((total + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=eng.device)
        for j, i in enumerate(units):   # unit i is "sample" (3 << 20) + i of the generator: the same text whoever gets it
            eng.synth((3 << 20) + i, 1, reads[i], args.readlen, dist=args.dist, out=buf[int(offs[j]):])
        img, hist, status = eng.fastq_to_images(buf, offs, lens)     # warm-up: workspaces
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001 -- a side measurement: never lose the bench line over it
        err = repr(e)
    flag = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=red_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if flag.item() < 0.5:
        return {"error": err or "another rank failed while making its units"}
    bad = 0
    nh = len(mine)
    for rep in range(3):
        queue = None
        try:
            queue = shard.TailQueue(len(tail))    # (every rank, in the same order: the cursor's key is a sequence number)
        except Exception as e:  # noqa: BLE001 -- (no store behind this group: the pass reads NaN, the bench line lives)
            err = repr(e)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nbytes = 0
        try:   # (a rank that fails here still takes part in the pass's collectives: its time reads NaN)
            if queue is None:
                raise RuntimeError(err)
            if nh:
                img, hist, status = eng.fastq_to_images(buf, offs[:nh], lens[:nh])
                bad = int((status != 0).sum().item())       # (the copy back is the launch's synchronisation point)
                nbytes += int(lens[:nh].sum())
            while True:
                got = queue.next(chunk)
                if len(got) == 0:
                    break
                sel = np.array([nh + j for j in got])
                img, hist, status = eng.fastq_to_images(buf, offs[sel], lens[sel])
                bad += int((status != 0).sum().item())
                nbytes += int(lens[sel].sum())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            err, dt = repr(e), float("nan")
        t = torch.tensor([dt, float(nbytes)], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_pass.append([float(x[0].item()) * 1e3 for x in every])
        took_bytes.append([int(x[1].item()) for x in every])
    if any(x != x for p in per_pass for x in p):
        return {"error": err or "another rank failed in a timed pass", "ms_by_rank_all_passes": per_pass}
    worst = [max(p) for p in per_pass]
    mid = sorted(range(len(worst)), key=lambda i: worst[i])[len(worst) // 2]
    mean = sum(weights) / world
    ms = per_pass[mid]
    return {"units": len(names), "samples": ns, "rungs_kbp": rungs_kbp,
            "rule": "shard.split_head_tail + shard_by_size + TailQueue (longest first to the least-loaded rank; the last tenth of the "
                    "bytes pulled from a shared cursor, %d units a claim)" % chunk,
            "head_units": len(head), "tail_units": len(tail),
            "bytes_by_rank": took_bytes[mid], "max_over_mean_bytes": max(took_bytes[mid]) / mean,
            "ms_by_rank_median_pass": ms, "max_over_mean_ms": max(ms) / (sum(ms) / world), "ms_by_rank_all_passes": per_pass,
            "gbases_per_s": sum(r * args.readlen for r in reads) / (worst[mid] * 1e-3) / 1e9,
            "static_deal_bytes_by_rank": static_loads, "static_deal_max_over_mean_bytes": max(static_loads) / mean,
            "round_robin_bytes_by_rank": rr, "round_robin_max_over_mean_bytes": max(rr) / mean,
            "bad_status_units_this_rank": bad}


def config4(args, device_index):
    """BASELINE.json configs[3] as a side leg of the N=1 line: k=9 cgr images (512x512), 1M 150 bp reads
    per sample, 100 samples, every sample distinct (pool = samples), for the uniform and the GC-skew +
    homopolymer base distribution.  4^9 u32 counters do not fit LDS: this is the spill path
    (the quad route: vk_bucket_kernel<9, 3> -> vk_quad_list_kernel -> vk_quad_count_kernel -> vk_quad_merge_kernel).
    Times are HIP events on the launch stream around the count and the image call; `frac` is against the
    same HBM peak as the main line (algorithmic bytes = text read once + the 1 MiB histogram written once
    per sample); `traffic` and the per-kernel times come from profiles/k9_latest.json when that file was
    taken on this configuration, else null."""
    import torch
    from varkoder_amd.engine import ImageEngine
    n, k = args.config4_samples, 9
    eng = ImageEngine(k=k, mapping="cgr", device=device_index)
    dev = torch.device("cuda", device_index)
    ncode = 4 ** k
    out = {"workload": "BASELINE configs[3]: k=9 cgr (%dx%d), %d distinct samples x %d x %d bp reads, 1 GPU, FASTQ "
                       "text resident in HBM" % (eng.side, eng.side, n, args.reads, args.readlen),
           "samples": n, "steps": args.config4_steps, "k": k, "mapping": "cgr"}
    prof = None
    try:
        with open(os.path.join(ROOT, "profiles", "k9_latest.json")) as f:
            prof = json.load(f)
    except Exception:
        prof = None
    hist = torch.empty((n, ncode), dtype=torch.int32, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    img = torch.empty((n, eng.side, eng.side), dtype=torch.uint8, device=dev)
    buf = None
    for dist_code in (0, 1, 2):    # uniform; GC-skew + homopolymers; reads of the lengths fastp writes (synth.py)
        if dist_code == 2:
            buf = None
            torch.cuda.empty_cache()
        buf, offs, lens = eng.synth(7000, n, args.reads, args.readlen, dist=dist_code, out=buf)
        for _ in range(2):    # warm-up: workspaces are allocated in the first launch; the one behind it still runs ~2 % slow
            eng.count(buf, offs, lens, hist=hist, status=status)
            eng.images(hist, img=img)
        torch.cuda.synchronize()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.config4_steps)]
        t0 = time.perf_counter()
        for e in ev:
            e[0].record()
            eng.count(buf, offs, lens, hist=hist, status=status)
            e[1].record()
            eng.images(hist, img=img)
            e[2].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / len(ev)
        count_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        image_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        alg = int(np.sum(lens)) + n * 4 * ncode
        bases = n * args.reads * args.readlen
        if dist_code == 2:
            bases = int(round(n * args.reads * 152.3))     # mean read length of the mix (synth.py, dist 2); the text decides `frac`
        leg = {"ms_per_step": wall * 1e3, "count_ms": count_ms, "image_ms": image_ms,
               "count_ms_by_step": [float(e[0].elapsed_time(e[1])) for e in ev],
               "gbases_per_s": bases / wall / 1e9, "fastq_bytes": int(np.sum(lens)),
               "bad_status_samples": int((status != 0).sum().item()), "count_launch": eng.last_count_launch(),
               "roofline": {"bound": "hbm", "kernel": "vk_bucket_kernel<9,3> + vk_quad_list_kernel + vk_quad_count_kernel + vk_quad_merge_kernel",
                            "achieved": alg / (count_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (count_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg,
                            "traffic": None, "kernel_ms_profiled": None,
                            "traffic_source": "from_profile_file (profiles/k9_latest.json: separate rocprofv3 runs of this configuration, "
                                              "not measured in this run); count_ms / image_ms / ms_per_step are this run's HIP events"}}
        want = {"k": k, "samples": n, "reads": args.reads, "readlen": args.readlen, "pool": n, "dist": dist_code}
        for entry in (prof or {}).get("legs", []):
            if entry.get("config") == want:
                leg["roofline"]["traffic"] = entry.get("hbm_bytes_per_launch")
                leg["roofline"]["kernel_ms_profiled"] = entry.get("kernel_ms")
                leg["roofline"]["profile"] = entry.get("tag")
        out["dist%d" % dist_code] = leg
    eng.close()
    del buf, hist, img, status
    torch.cuda.empty_cache()
    return out


def realistic(args, device_index):
    """The k <= 7 count on text shaped like what step B of the reference hands to step D (fastp --merge
    --include_unmerged --disable_length_filtering, commands/image.py:405,426-427,494-495): reads of 0 .. 290 bases
    under 40 .. 70 byte headers (synth.py dist 2).  A batch of `samples` cycling through a pool of distinct samples;
    `frac` is against this text's own algorithmic bytes; `general_piece_fraction` = 4 KiB pieces that left the
    sequence-only fast path (short reads, see vk_count_dense_kernel) / all pieces."""
    import torch
    from varkoder_amd.engine import ImageEngine
    eng = ImageEngine(k=args.k, mapping=args.mapping, device=device_index)
    dev = torch.device("cuda", device_index)
    n, pool = args.samples, min(args.realistic_pool, args.samples)
    ncode = 4 ** args.k
    buf, poffs, plens = eng.synth(3000, pool, args.reads, args.readlen, dist=2)
    idx = np.arange(n) % pool
    offs, lens = poffs[idx].copy(), plens[idx].copy()
    hist = torch.empty((n, ncode), dtype=torch.int32, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    out = {"workload": "%d samples x %d reads of 0 .. %d bases (65 %% %d, 20 %% merged pairs, 10 %% trimmed, 5 %% under 45; "
                       "headers of 40 .. 70 bytes), k=%d, %d distinct samples in HBM" % (n, args.reads, 2 * args.readlen - 10, args.readlen, args.k, pool),
           "fastq_bytes_per_launch": int(np.sum(lens)), "steps": args.realistic_steps}
    for name, env in (("dense", None), ("classic", "1")):
        if env:
            os.environ["VKIMG_K1_CLASSIC"] = env
            e2 = ImageEngine(k=args.k, mapping=args.mapping, device=device_index)
        else:
            e2 = eng
        try:
            e2.count(buf, offs, lens, hist=hist, status=status)
            torch.cuda.synchronize()
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.realistic_steps)]
            for e in ev:
                e[0].record()
                e2.count(buf, offs, lens, hist=hist, status=status)
                e[1].record()
            torch.cuda.synchronize()
            ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
            alg = int(np.sum(lens)) + n * 4 * ncode
            leg = {"count_ms": ms, "achieved": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "bad_status_samples": int((status != 0).sum().item())}
            if not env:
                g, pcs = e2.last_count_general()
                leg.update(general_pieces=g, pieces=pcs, general_piece_fraction=g / max(1, pcs))
            out[name] = leg
        finally:
            if env:
                os.environ.pop("VKIMG_K1_CLASSIC", None)
                e2.close()
    eng.close()
    del buf, hist, status
    torch.cuda.empty_cache()
    return out


def ladder(args, device_index):
    """The subsample ladder of step C on the GPU (split_fastq's 1-2-5 ladder, commands/image.py:682-695, every step a
    Bernoulli subsample of the reads with its own seed): `--ladder-samples` samples of the main workload's shape, all steps
    of all samples -- read index + full count in one pass, then ONE walker launch over every (sample, step) pair
    (subsample.ladder_counts).  Wall time of the whole call, host logic included; the median of three."""
    import torch
    from varkoder_amd.engine import ImageEngine
    from varkoder_amd.subsample import ladder_counts
    eng = ImageEngine(k=args.k, mapping=args.mapping, device=device_index)
    n = args.ladder_samples
    buf, offs, lens = eng.synth(5000, n, args.reads, args.readlen, dist=args.dist)
    times, steps = [], 0
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = ladder_counts(eng, buf, offs, lens, seed=5, min_bp=50000, max_bp=None)
        torch.cuda.synchronize()
        if rep:
            times.append(time.perf_counter() - t0)
        steps = sum(len(r["steps"]) for r in recs)
        bad = sum(1 for r in recs if r["error"])
    eng.close()
    del buf
    torch.cuda.empty_cache()
    ms = sorted(times)[len(times) // 2] * 1e3
    return {"samples": n, "steps": steps, "ms": ms, "ms_per_sample_ladder": ms / n, "passes_ms": [t * 1e3 for t in times],
            "failed_samples": bad, "bases_of_one_full_pass": n * args.reads * args.readlen,
            "note": "k=%d; every step of every sample: one pass for the read index + the full count, one walker launch for the "
                    "rest (vk_count_index_device, vk_walk_kernel); round 3 streamed the text once per step (38.2 ms for 32 x 12)" % args.k}


def query_leg(args, device_index):
    """BASELINE.json configs[4] on ONE GPU: FASTQ resident in HBM -> count -> image (stays on the device) ->
    vk_preprocess_device (PIL's BOX squish to 224 x 224, /255, normalise: query.py) -> batched forward of a model with the
    shape of the reference's default architecture (timm vit_large_patch32_224, core/config.py:51-52; random weights --
    fastai / timm / the hub weights need a network) at the reference's `get_preds` precision (fp32, commands/query.py:283-324)
    -> sigmoid >= threshold.  Times: HIP events on the current stream (the engine launches on it); the preprocess kernel's
    rate is against its own bytes (u8 image read once, three float32 planes written).  `verified`: the float tensors of
    four images must equal a PIL + NumPy restatement of the reference's item transform bit for bit, else the leg fails."""
    import torch
    from varkoder_amd import query as Q
    from varkoder_amd.engine import ImageEngine
    n, pool, k, mapping, bs = args.query_samples, min(64, args.query_samples), 7, "cgr", args.query_batch
    eng = ImageEngine(k=k, mapping=mapping, device=device_index)
    dev = torch.device("cuda", device_index)
    buf, poffs, plens = eng.synth(9000, pool, args.reads, args.readlen, dist=args.dist)
    idx = np.arange(n) % pool
    offs, lens = poffs[idx].copy(), plens[idx].copy()
    torch.manual_seed(0)
    model = Q.vit().to(dev).eval()
    nparam = sum(p.numel() for p in model.parameters())
    out_size, thr = 224, 0.7

    def one_pass(ev=None):
        if ev:
            ev[0].record()
        img, hist, status = eng.fastq_to_images(buf, offs, lens)
        if ev:
            ev[1].record()
        pre_ms = fwd_ms = 0.0
        hits = 0
        pev = []
        with torch.no_grad():
            for i in range(0, n, bs):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                x = Q.preprocess(eng, img[i:i + bs].contiguous(), out_size=out_size)
                e[1].record()
                probs = torch.sigmoid(model(x))
                hits_t = (probs >= thr).sum()
                e[2].record()
                pev.append((e, hits_t))
        if ev:
            ev[2].record()
        torch.cuda.synchronize()
        for e, h in pev:
            pre_ms += e[0].elapsed_time(e[1])
            fwd_ms += e[1].elapsed_time(e[2])
            hits += int(h.item())
        return img, status, pre_ms, fwd_ms, hits

    one_pass()                                             # warm-up: workspaces, GEMM algorithm selection
    res = []
    for _ in range(args.query_steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t0 = time.perf_counter()
        img, status, pre_ms, fwd_ms, hits = one_pass(ev)
        wall = time.perf_counter() - t0
        res.append((wall, ev[0].elapsed_time(ev[1]), pre_ms, fwd_ms, hits))
    wall, img_ms, pre_ms, fwd_ms, hits = sorted(res)[len(res) // 2]
    # the four images the check looks at: the device transform against PIL's resize + float32 arithmetic
    from PIL import Image
    host = img[:4].cpu().numpy()
    got = Q.preprocess(eng, img[:4].contiguous(), out_size=out_size).cpu().numpy()
    verified = 0
    for i in range(4):
        r = np.array(Image.fromarray(host[i]).convert("RGB").resize((out_size, out_size), resample=Image.Resampling.BOX))
        want = np.ascontiguousarray(((r.astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1))
        if not np.array_equal(got[i], want):
            raise VerifyError("query leg: preprocess tensor of image %d differs from the PIL pipeline" % i)
        verified += 1
    pre_bytes = n * (eng.side * eng.side + 3 * out_size * out_size * 4)
    bad = int((status != 0).sum().item())
    eng.close()
    del buf, model
    torch.cuda.empty_cache()
    return {"workload": "BASELINE configs[4] on one GPU: %d samples x %d x %d bp reads (pool %d), k=%d %s -> %dx%d images on the device -> "
                        "BOX squish to %dx%d + normalise -> ViT-L/32-224-shaped forward (%d M parameters, random weights, fp32, batch %d) -> "
                        "sigmoid >= %.1f" % (n, args.reads, args.readlen, pool, k, mapping, eng.side, eng.side, out_size, out_size, nparam // 1000000, bs, thr),
            "samples": n, "steps": args.query_steps, "ms_per_pass": wall * 1e3, "images_per_s": n / wall,
            "gbases_per_s": n * args.reads * args.readlen / wall / 1e9,
            "count_and_image_ms": img_ms, "preprocess_ms": pre_ms, "forward_ms": fwd_ms,
            "forward_share": fwd_ms / (wall * 1e3), "preprocess_gb_per_s": pre_bytes / (pre_ms * 1e-3) / 1e9,
            "preprocess_bytes": pre_bytes, "labels_over_threshold": hits, "bad_status_samples": bad,
            "verified_preprocess_images": verified, "dtype": "u32 counts, u8 images, float32 transform and forward"}


class ClockSampler:
    """The GPU's shader clock while the timed steps run: the starred line of the card's `pp_dpm_sclk` (found by the device's
    PCI address), read every 10 ms by a host thread.  Boxes of the pool run the same binary a few per cent apart; with the
    clock in the record a slow box can be told from a slow kernel.  Never raises: {"error": ...} when the file is not there."""

    def __init__(self, device_index):
        import threading
        self.path, self.err, self.vals, self.stop_flag, self.thread = None, None, [], threading.Event(), None
        self.marked = False    # samples count from mark() on: the timed steps
        try:
            import glob
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            cand = glob.glob("/sys/bus/pci/devices/%s/pp_dpm_sclk" % bdf)
            if cand:
                self.path = cand[0]
            else:
                self.err = "no pp_dpm_sclk for PCI device %s" % bdf
        except Exception as e:  # noqa: BLE001
            self.err = repr(e)

    def read(self):
        with open(self.path) as f:
            for line in f:
                if "*" in line:
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        return None

    def mark(self):
        self.marked = True

    def start(self):
        if self.path is None:
            return
        import threading

        def run():
            while not self.stop_flag.is_set():
                try:
                    v = self.read()
                    if v and self.marked:
                        self.vals.append(v)
                except Exception as e:  # noqa: BLE001
                    self.err = repr(e)
                    return
                self.stop_flag.wait(0.01)
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def stop(self):
        if self.thread is not None:
            self.stop_flag.set()
            self.thread.join(timeout=2.0)
        if not self.vals:
            return {"error": self.err or "no samples"}
        v = sorted(self.vals)
        return {"samples": len(v), "min": v[0], "median": v[len(v) // 2], "max": v[-1], "source": self.path}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def dsk_argv(threads, k, infile, tmpdir, outpath):
    """The reference's dsk command line (commands/image.py:771-790; pinned by tests/golden tool_argv)."""
    return ["dsk", "-nb-cores", str(threads), "-kmer-size", str(k), "-abundance-min", "1",
            "-abundance-min-threshold", "1", "-max-memory", "1000", "-file", str(infile), "-out-tmp", str(tmpdir),
            "-out", str(outpath)]


def dsk2ascii_argv(threads, counts, tmpdir):
    """The reference's dsk2ascii command line (commands/image.py:875-891)."""
    return ["dsk2ascii", "-c", "-file", str(counts), "-nb-cores", str(threads), "-out",
            os.path.join(str(tmpdir), "dsk.txt"), "-verbose", "0"]


def dsk_reference(buf, args, cores):
    """If GATB dsk is on PATH (it is not in the build image), time the reference's exact
    invocation (varKoder/commands/image.py:771-790, :875-886) on one sample of the workload."""
    import shutil
    import subprocess
    import tempfile
    if not (shutil.which("dsk") and shutil.which("dsk2ascii")):
        return None
    with tempfile.TemporaryDirectory(prefix="vkbench_") as tmp:
        fq = os.path.join(tmp, "s@00150000K.fq")
        buf.tofile(fq)
        res = {}
        for nc in (1, cores):
            h5 = os.path.join(tmp, f"s_{nc}.h5")
            t0 = time.perf_counter()
            subprocess.run(dsk_argv(nc, args.k, fq, tmp, h5), check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            subprocess.run(dsk2ascii_argv(nc, h5, tmp), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            res[f"gbases_per_s_{nc}_cores"] = args.reads * args.readlen / (time.perf_counter() - t0) / 1e9
        res["kind"] = "reference (dsk + dsk2ascii, one sample)"
        return res


def spawn_ranks(args):
    """`--gpus N` with no launcher: start N fresh child processes (one rank per GPU) BEFORE this
    process has imported torch or touched the GPU, relay rank 0's JSON line, return the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                         f"--nproc-per-node {args.gpus} or drop the launcher\n")
        sys.exit(2)
    if args.samples <= 0:
        args.samples = 1000 if world == 1 else -(-args.total_samples // world)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.all_on_device0:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    from varkoder_amd.engine import ImageEngine
    eng = ImageEngine(k=args.k, mapping=args.mapping, device=local_rank)

    pool = min(args.pool, args.samples)
    fastq, poffs, plens = eng.synth(rank * pool, pool, args.reads, args.readlen, dist=args.dist)
    idx = np.arange(args.samples) % pool
    offs, lens = poffs[idx].copy(), plens[idx].copy()
    ncode = 4 ** args.k
    hist = torch.empty((args.samples, ncode), dtype=torch.int32, device=dev)
    status = torch.empty((args.samples,), dtype=torch.int32, device=dev)
    img = torch.empty((args.samples, eng.side, eng.side), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(e=None):
        if e:
            e[0].record()
        eng.count(fastq, offs, lens, parts=args.parts, hist=hist, status=status)
        if e:
            e[1].record()
        eng.images(hist, img=img)
        if e:
            e[2].record()

    clocks = ClockSampler(local_rank)      # (a host thread reading one sysfs file every 10 ms: nothing on the GPU's side)
    clocks.start()
    for _ in range(args.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    clocks.mark()     # (the thread has been running since before the warm-up: nothing between the fence and the first step)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    fence()
    elapsed = time.perf_counter() - t0
    sclk = clocks.stop()
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # per-rank times: an imbalance must be visible in the record
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bad = int((status != 0).sum().item())
    e2e_ranks = None
    if world > 1 and not args.no_e2e:   # (every rank takes part: it is a node-level measurement)
        hist = img = None
        torch.cuda.empty_cache()
        e2e_ranks = end_to_end_ranks(eng, args, rank, world, dist, red_dev)
    lshard = None
    if world > 1 and args.ladder_shard_samples > 0 and args.k <= 7:   # (every rank takes part)
        hist = img = None
        torch.cuda.empty_cache()
        lshard = ladder_shard(eng, args, rank, world, dist, red_dev)
    count_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    image_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    if rank == 0:
        bases_per_step = args.samples * args.reads * args.readlen * world
        value = bases_per_step * args.steps / elapsed / 1e9
        fastq_bytes = int(lens[0])
        # algorithmic bytes of the dominant kernel (vk_count_kernel), SURVEY 8d:
        # FASTQ text read once + the 4^k u32 histogram written once, per sample
        alg_bytes = args.samples * (fastq_bytes + 4 * ncode)
        achieved = alg_bytes / (count_ms * 1e-3) / 1e9
        # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/): only quoted when
        # that profile was taken on this kernel and this configuration, otherwise null
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                want = {"k": args.k, "samples": args.samples, "reads": args.reads, "readlen": args.readlen,
                        "pool": pool, "dist": args.dist}
                if tj.get("config") == want:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the kernel the count call launches (rocprofv3's name for it): k <= 7 the sequence-only dense kernel
        # unless VKIMG_K1_CLASSIC=1 asks for the kernel that classifies every byte; k = 8, 9 the spill path
        count_kernel = ("vk_bucket_kernel+vk_quad_list_kernel+vk_quad_count_kernel+vk_quad_merge_kernel" if args.k > 7 else
                        "vk_count_kernel" if os.environ.get("VKIMG_K1_CLASSIC") == "1" else "vk_count_dense_kernel")
        out = {
            "metric": "Gbases/s for `varKoder image` k=%d, %d bp reads" % (args.k, args.readlen),
            "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "samples_per_s": args.samples * world * args.steps / elapsed,
            "config": {"workload": "%s: %d synthetic samples x %d x %d bp reads per GPU per step, k=%d %s (%dx%d), "
                                   "FASTQ text resident in HBM" % (
                                       "BASELINE configs[1]" if world == 1 else
                                       "BASELINE configs[2] (%d samples sharded over %d GPUs)" % (args.samples * world, world),
                                       args.samples, args.reads, args.readlen, args.k, args.mapping, eng.side, eng.side),
                       "process_group": {"backend": dist.get_backend() if world > 1 else None,
                                         "world_size": dist.get_world_size() if world > 1 else 1},
                       "samples_per_gpu": args.samples, "reads_per_sample": args.reads,
                       "read_len": args.readlen, "k": args.k, "mapping": args.mapping,
                       "distinct_samples_in_hbm": pool, "base_distribution": args.dist,
                       "fastq_bytes_per_sample": fastq_bytes, "parallelism": "samples sharded x%d" % world,
                       "count_launch": eng.last_count_launch()},
            "ms_per_step_by_rank": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            **({"rehearsal": "all %d ranks share cuda:0 (--all-on-device0): `value`, ms_per_step and the kernel times say nothing about "
                             "N GPUs; what this run measures is the HOST side of a node-level run -- %d ranks' staging pools, page cache "
                             "and H2D streams at once, with the I/O threads each rank gets of this box's cores (end_to_end)" % (world, world)}
               if args.all_on_device0 and world > 1 else {}),
            "kernel_ms": {count_kernel + "(+check)": count_ms, "vk_image_kernel": image_ms},
            "bad_status_samples": bad,
            "roofline": {"bound": "hbm", "kernel": count_kernel, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": count_ms,
                         "count_ms_by_step": [float(e[0].elapsed_time(e[1])) for e in ev],
                         "shader_clock_mhz_during_timed_steps": sclk,
                         "traffic_source": "from_profile_file (profiles/traffic_latest.json: separate rocprofv3 --pmc passes "
                                           "of this configuration); achieved / avg_launch_ms are this run's HIP events"},
        }
        if e2e_ranks is not None:
            out["end_to_end"] = e2e_ranks
        if lshard is not None:
            out["ladder_shard"] = lshard
        if world == 1 and not args.no_config4 and args.k <= 7:
            try:
                out["config4"] = config4(args, local_rank)
            except Exception as e:  # a side measurement: never lose the bench line over it
                out["config4"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                # (hist / img: as the last timed step left them; batch entry i = pool sample i for i < pool)
                out["cpu_baseline"] = cpu_baseline(eng, fastq, poffs, plens, args, args.cpu_seconds, hist, img)
                out["verified_samples"] = out["cpu_baseline"].get("verified_samples", 0)
            except VerifyError as e:
                out["cpu_baseline"] = {"error": repr(e)}
                out["verified_samples"] = 0
                out["verify_failed"] = str(e)
                bad = max(bad, 1)
            except Exception as e:  # the baseline is a reported side figure, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        hist = img = fastq = None    # make room: the legs below allocate their own batches
        torch.cuda.empty_cache()
        if world == 1 and not args.no_realistic and args.k <= 7:
            try:
                out["realistic"] = realistic(args, local_rank)
            except Exception as e:  # a side measurement: never lose the bench line over it
                out["realistic"] = {"error": repr(e)}
        if world == 1 and not args.no_ladder and args.k <= 7:
            try:
                out["ladder"] = ladder(args, local_rank)
            except Exception as e:  # a side measurement: never lose the bench line over it
                out["ladder"] = {"error": repr(e)}
        if world == 1 and not args.no_query and args.k <= 7:
            try:
                out["query"] = query_leg(args, local_rank)
            except VerifyError as e:
                out["query"] = {"error": repr(e)}
                out["verify_failed"] = str(e)
                bad = max(bad, 1)
            except Exception as e:  # a side measurement: never lose the bench line over it
                out["query"] = {"error": repr(e)}
        if world == 1 and not args.no_e2e:
            try:
                out["end_to_end"] = end_to_end(eng, args)
            except Exception as e:  # a side measurement: never lose the bench line over it
                out["end_to_end"] = {"error": repr(e)}
        under_profiler = any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES")) or \
            any(v.startswith(("ROCPROF", "ROCPROFILER")) for v in os.environ)
        if world == 1 and not args.no_live_traffic and not under_profiler:
            # roofline.traffic measured in THIS run: the sample pools of every leg above are freed by now, and so is this
            # process's engine (its workspaces would sit beside the children's on the card)
            eng.close()
            eng = None
            torch.cuda.empty_cache()
            t0 = time.perf_counter()
            try:
                got, note = live_traffic(args, args.k, args.mapping, args.samples, pool, args.dist)
            except Exception as e:  # noqa: BLE001 -- never lose the bench line over it
                got, note = None, "live measurement failed: %r" % (e,)
            out["roofline"]["traffic_from_profile_file"] = out["roofline"]["traffic"]
            if got is not None:
                out["roofline"]["traffic"] = got
                out["roofline"]["traffic_source"] = note
            else:
                out["roofline"]["traffic_live_error"] = note
            c4 = out.get("config4")
            if got is not None and isinstance(c4, dict) and "error" not in c4:   # (a profiler that cannot run here is not asked six more times)
                for dcode in (0, 1, 2):
                    leg = c4.get("dist%d" % dcode)
                    if not isinstance(leg, dict) or time.perf_counter() - t0 > 150.0:
                        continue
                    try:
                        got, note = live_traffic(args, 9, "cgr", c4["samples"], c4["samples"], dcode)
                    except Exception as e:  # noqa: BLE001
                        got, note = None, "live measurement failed: %r" % (e,)
                    leg["roofline"]["traffic_from_profile_file"] = leg["roofline"]["traffic"]
                    if got is not None:
                        leg["roofline"]["traffic"] = got
                        leg["roofline"]["traffic_source"] = note
                    else:
                        leg["roofline"]["traffic_live_error"] = note
            out["live_traffic_seconds"] = time.perf_counter() - t0
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if eng is not None:
        eng.close()
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
