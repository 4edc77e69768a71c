"""N>1 path on CPU: two ranks over gloo shard the samples, process their share, and rank 0
merges the stats; the timing protocol's MAX-reduce is exercised too.  The per-sample work
is done by the oracle here (this is a test of the sharding, not of the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_samples, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle
    from varkoder_amd import shard, synth
    from varkoder_amd.mapping import pixel_lut
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_indices(n_samples, rank, world)
    lut = pixel_lut(5, "cgr")
    stats = {}
    for s in mine:
        img, nwin, st = oracle.fastq_to_image(synth.sample_fastq(s, 200, 150), 5, lut, 1024)
        stats[f"s{s:03d}"] = {"windows": int(nwin), "img_sum": int(img.astype(np.int64).sum()), "rank": rank}
    dist.barrier()
    slow = shard.max_over_ranks(1.0 + rank)          # the bench's "MAX over ranks" timing rule
    merged = shard.gather_stats(stats, dst=0)
    if rank == 0:
        out.put((slow, merged))
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_covers_every_sample_once():
    from oracle import oracle
    from varkoder_amd import shard, synth
    from varkoder_amd.mapping import pixel_lut
    n, world = 7, 2
    assert shard.shard_indices(n, 0, 2) == [0, 2, 4, 6] and shard.shard_indices(n, 1, 2) == [1, 3, 5]
    with pytest.raises(ValueError):
        shard.shard_indices(n, 2, 2)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    slow, merged = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slow == 2.0
    assert list(merged.keys()) == [f"s{s:03d}" for s in range(n)]
    lut = pixel_lut(5, "cgr")
    for s in range(n):
        img, nwin, st = oracle.fastq_to_image(synth.sample_fastq(s, 200, 150), 5, lut, 1024)
        got = merged[f"s{s:03d}"]
        assert got["windows"] == nwin and got["img_sum"] == int(img.astype(np.int64).sum())
        assert got["rank"] == s % world


LADDERS = {6: [500, 1000, 2000, 5000, 10000, 20000], 8: [500, 1000, 2000, 5000, 10000, 20000, 50000, 100000],
           9: [500, 1000, 2000, 5000, 10000, 20000, 50000, 100000, 200000]}   # Kbp per rung (commands/image.py:682-695)


@pytest.mark.parametrize("rungs", [6, 8, 9])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_size_aware_shard_balances_the_reference_ladder(rungs, world):
    """`split_fastqs/` of 100 samples: one file per rung and sample, sorted by name = by size within a sample
    (commands/image.py:682-708).  i % world pins rungs to ranks (8 rungs on 8 ranks: one rank gets every 100M
    file); shard_by_size keeps every rank within 5 % of the mean and still covers every file exactly once."""
    from varkoder_amd import shard
    names = sorted(f"s{s:03d}@{kbp:08d}K.fq" for s in range(100) for kbp in LADDERS[rungs])
    weights = [int(n.split("@")[1][:8]) * 2133 for n in names]   # ~2.13 bytes of FASTQ per base
    loads = shard.rank_loads(weights, world)
    assert max(loads) / (sum(loads) / world) <= 1.05
    parts = [shard.shard_by_size(weights, r, world) for r in range(world)]
    assert sorted(i for p in parts for i in p) == list(range(len(names)))
    assert all(p == sorted(p) for p in parts)
    # what the static round-robin did with the same list (the verdict's numbers: 1.48x of 2 at world 2 on six rungs)
    rr = [sum(weights[i] for i in shard.shard_indices(len(names), r, world)) for r in range(world)]
    if rungs % world == 0 or world % rungs == 0:
        assert max(rr) / (sum(rr) / world) > 1.3


@pytest.mark.parametrize("world", [2, 4, 8])
def test_size_aware_shard_balances_unsplit_clean_files(world):
    """--from-clean inputs: one file per sample, sizes spread over two orders of magnitude."""
    from varkoder_amd import shard
    rng = np.random.default_rng(11)
    weights = [int(w) for w in np.exp(rng.normal(18.0, 1.0, size=100))]
    loads = shard.rank_loads(weights, world)
    assert max(loads) / (sum(loads) / world) <= 1.05
    assert shard.shard_by_size(weights, 0, 1) == list(range(100))
    assert shard.shard_by_size([], 0, world) == []
    with pytest.raises(ValueError):
        shard.shard_by_size(weights, world, world)


def test_file_weights_count_text_bytes(tmp_path):
    """A plain file weighs its bytes; a gzip file the text its own framing names: the size word of a one-member file, the
    first block's ratio of a BGZF file; members glued together (the last size word says nothing) six times the file."""
    import gzip
    import struct
    import zlib
    from varkoder_amd import shard, synth
    text = synth.sample_fastq(5, 4000, 150).tobytes()
    (tmp_path / "a.fq").write_bytes(b"x" * 100)
    (tmp_path / "b.fq.gz").write_bytes(b"y" * 10)                       # (under 18 bytes: not a gzip file)
    (tmp_path / "one.fq.gz").write_bytes(gzip.compress(text, 1))
    glued = gzip.compress(text, 6) + gzip.compress(text[:1000], 6)
    (tmp_path / "glued.fq.gz").write_bytes(glued)

    def bgzf(data, block=30000):
        out = []
        for i in list(range(0, len(data), block)) + [None]:
            piece = b"" if i is None else data[i:i + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = co.compress(piece) + co.flush()
            bsize = 12 + 6 + len(body) + 8
            out.append(b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
                       struct.pack("<II", zlib.crc32(piece), len(piece)))
        return b"".join(out)
    bz = bgzf(text)
    (tmp_path / "bg.fq.gz").write_bytes(bz)
    w = shard.file_weights([tmp_path / n for n in ("a.fq", "b.fq.gz", "missing.fq", "one.fq.gz", "glued.fq.gz", "bg.fq.gz")])
    assert w[:3] == [100, 0, 0]
    assert w[3] == len(text)
    assert w[4] == 6 * len(glued)
    assert abs(w[5] - len(text)) <= 0.1 * len(text)


def test_single_process_helpers_are_identity():
    from varkoder_amd import shard
    assert shard.max_over_ranks(3.5) == 3.5
    assert shard.gather_stats({"a": {"x": 1}}) == {"a": {"x": 1}}
    assert shard.world_info()[1] >= 1


def test_merge_stats_adds_the_ranks_shares_of_a_sample():
    """Files are dealt by size, so one sample's ladder rungs land on several ranks: a timing is the SUM of the
    ranks' shares (commands/image.py:1078 accumulates over a sample's files), a failed step stays, the rest is
    taken as it comes."""
    from collections import OrderedDict
    from varkoder_amd import shard
    a = {"s1": OrderedDict([("7mer_counting_time", 1.5), ("k7_img_time", 0.25), ("base_frequencies_sd", 0.1)]),
         "s2": OrderedDict([("7mer_counting_time", 1.0), ("failed_step", "image")])}
    b = {"s1": OrderedDict([("7mer_counting_time", 2.0), ("k7_img_time", 0.5), ("base_frequencies_sd", 0.1)]),
         "s2": OrderedDict([("7mer_counting_time", 4.0), ("k7_img_time", 1.0)]),
         "s3": OrderedDict([("failed_step", "split")])}
    m = shard.merge_stats([a, b, None])
    assert list(m) == ["s1", "s2", "s3"]
    assert m["s1"] == {"7mer_counting_time": 3.5, "k7_img_time": 0.75, "base_frequencies_sd": 0.1}
    assert m["s2"] == {"7mer_counting_time": 5.0, "failed_step": "image", "k7_img_time": 1.0}
    assert m["s3"] == {"failed_step": "split"}
    assert shard.gather_stats(a, error="boom", with_errors=True) == (shard.merge_stats([a]), ["boom"])


def _failing_job_worker(rank, world, port, tmp, out):
    """One rank of `varkoder_amd image`'s closing half (cli.finish_image_job); rank 1's share "raised"."""
    sys.path.insert(0, ROOT)
    import argparse
    import time
    from collections import OrderedDict
    from pathlib import Path
    import torch.distributed as dist
    from varkoder_amd import cli
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = argparse.Namespace(stats_file=os.path.join(tmp, "stats.csv"), label_table=True)
    stats = {"s1": OrderedDict([("7mer_counting_time", 1.0 + rank), ("k7_img_time", 0.5)])}
    error = OSError("cannot write into the output folder") if rank == 1 else None
    if rank == 1:
        stats = {}
    t0 = time.time()
    try:
        cli.finish_image_job(args, Path(tmp), rank, world, stats, error, ["s1"], {}, {})
        out.put((rank, "ok", time.time() - t0))
    except Exception as e:   # noqa: BLE001
        out.put((rank, repr(e), time.time() - t0))
        sys.exit(3)


def test_a_failing_rank_ends_the_image_job_at_once_on_every_rank(tmp_path):
    """A rank whose share raised still goes to the gather and the barrier (its error rides with its stats): nobody waits
    for the backend's timeout, rank 0 still writes the stats of what finished, every rank exits non-zero."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_job_worker, args=(r, world, port, str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 3
    assert "cannot write" in got[0][1] and "cannot write" in got[1][1]
    assert max(t for _, _, t in got) < 60
    import pandas as pd
    df = pd.read_csv(tmp_path / "stats.csv")
    assert list(df["sample"]) == ["s1"] and float(df["7mer_counting_time"][0]) == 1.0
    assert not (tmp_path / "labels.csv").exists()


def _weights_worker(rank, world, port, tmp, out):
    sys.path.insert(0, ROOT)
    from pathlib import Path
    import torch.distributed as dist
    from varkoder_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    files = sorted(Path(tmp).glob("*.fq"))
    if rank == 1:   # this rank's view of the sizes differs (a file still being written, an attribute cache)
        real = shard.file_weights
        shard.file_weights = lambda fs: list(reversed(real(fs)))
    own = shard.file_weights(files)
    w = shard.agreed_weights(files)
    out.put((rank, own, w, shard.shard_by_size(w, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_see_different_sizes_still_partition_the_files(tmp_path):
    """shard_by_size is only a partition when every rank feeds it the same weights: rank 0's view is broadcast."""
    for i, n in enumerate([1000, 900, 800, 700, 50]):
        (tmp_path / f"f{i}.fq").write_bytes(b"a" * n)
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_weights_worker, args=(r, world, port, str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, own0, w0, mine0), (_, own1, w1, mine1) = got
    assert own1 != own0 and w0 == w1 == own0      # rank 0's view, on both
    assert sorted(mine0 + mine1) == list(range(5))   # every file exactly once


def _tail_worker(rank, world, port, weights, cost, scale, out):
    """One rank of a job whose items cost `cost` seconds (x scale) -- which the scheduler does NOT know: it deals by `weights`."""
    sys.path.insert(0, ROOT)
    import time
    import torch.distributed as dist
    from varkoder_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    head, tail = shard.split_head_tail(weights, 0.1)
    mine = [head[j] for j in shard.shard_by_size([weights[i] for i in head], rank, world)]
    queue = shard.TailQueue(len(tail))
    chunk = max(1, len(tail) // (6 * world))
    dist.barrier()
    took = []
    for i in mine:
        time.sleep(cost[i] * scale)
        took.append(i)
    while True:
        got = queue.next(chunk)
        if len(got) == 0:
            break
        for j in got:
            time.sleep(cost[tail[j]] * scale)
            took.append(tail[j])
    out.put((rank, took))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_the_tail_queue_balances_what_the_weights_got_wrong(world):
    """Mixed inputs: plain files, gzip files whose inflate makes a text byte dearer, k = 9 samples whose count time depends on
    their bases -- the weights (text bytes) are off by up to 30 % per item, in a way that lines up with the deal (the dearer
    kind sorts together).  Dealt statically, the ranks' real time differs by more than the bound; with the last tenth of
    the weight pulled from the shared cursor by whoever is free, max / mean of the MODELLED time (the sum of the real
    costs of what a rank processed) is within 5 %."""
    from varkoder_amd import shard
    rng = np.random.default_rng(world)
    n = 40 * world
    weights = [int(w) for w in rng.integers(50, 400, size=n // 2)] + [int(w) for w in rng.integers(2, 30, size=n - n // 2)]
    kind = rng.integers(0, 3, size=n)                       # 0 plain, 1 gzip, 2 k = 9 on skewed bases
    factor = np.array([1.0, 1.3, 1.15])[kind]
    cost = [float(w * f) for w, f in zip(weights, factor)]
    # static deal alone, for the record: its imbalance in real cost
    static = [sum(cost[i] for i in shard.shard_by_size(weights, r, world)) for r in range(world)]
    scale = 2.0 / (sum(cost) / world)                       # ~2 s of sleeping per rank
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tail_worker, args=(r, world, port, weights, cost, scale, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(i for took in got.values() for i in took) == list(range(n))      # every item exactly once
    modelled = [sum(cost[i] for i in got[r]) for r in range(world)]
    assert max(modelled) / (sum(modelled) / world) <= 1.05, (modelled, static)


def _giant_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from varkoder_amd import shard, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fq = synth.sample_fastq(77, 3000, 150, dist=1).tobytes()
    start, end = shard.split_at_records(fq, world)[rank]
    part = fq[start:end]
    h, nwin, st = oracle.count_fastq(part, 7)          # stands in for the GPU count of this range
    t = torch.from_numpy(h.astype(np.int64))
    shard.allreduce_sum_(t)
    if rank == 0:
        out.put((st, t.numpy().copy(), (start, end)))
    dist.barrier()
    dist.destroy_process_group()


def test_giant_sample_split_and_allreduce_is_exact():
    """One sample over two ranks: record-aligned split + SUM all-reduce of the histograms equals
    the histogram of the whole sample (integer, order-independent)."""
    from oracle import oracle
    from varkoder_amd import shard, synth
    from fastq_cases import rec
    fq = synth.sample_fastq(77, 3000, 150, dist=1).tobytes()
    want = oracle.count_fastq(fq, 7)[0].astype(np.int64)
    for world in (2, 3, 7):
        ranges = shard.split_at_records(fq, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(fq)
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        total = np.zeros_like(want)
        for s, e in ranges:
            assert e == s or fq[s:s + 1] == b"@"
            h, _, st = oracle.count_fastq(fq[s:e], 7)
            assert st == 0
            total += h
        assert np.array_equal(total, want)
    # quality lines that start with '@' must not be taken for headers
    tricky = b"".join(rec(f"r{i}", "ACGTACGTAC" * 3, qual="@" + "+" * 29) for i in range(400))
    for s, e in shard.split_at_records(tricky, 5):
        assert oracle.count_fastq(tricky[s:e], 5)[2] == 0
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_giant_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    st, total, _ = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert st == 0 and np.array_equal(total, want)


def test_u32_counts_widen_unsigned():
    """A per-rank count of 2^31 or more must not turn negative on the way into the 64-bit all-reduce."""
    import torch
    from varkoder_amd import shard
    raw = np.array([0x80000001, 0xFFFFFFFF, 5, 0], dtype=np.uint32)
    t = torch.from_numpy(raw.view(np.int32).copy())
    assert t[0].item() < 0                                   # what a plain .to(int64) would sign-extend
    w = shard.widen_u32(t)
    assert w.dtype == torch.int64 and w.tolist() == [0x80000001, 0xFFFFFFFF, 5, 0]


def _giant_gpu_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from varkoder_amd import synth
    from varkoder_amd.engine import ImageEngine, count_giant_sample
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = ImageEngine(k=7, mapping="cgr", device=0)           # both ranks on cuda:0: a rehearsal of the data path
    fq = synth.sample_fastq(77, 30000, 150, dist=1)
    h, st = count_giant_sample(eng, fq, rank=rank, world=world)
    if rank == 0:
        out.put((st, h.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.gpu
def test_giant_sample_over_two_ranks_through_the_kernel():
    """engine.count_giant_sample at world = 2 with the HIP count kernel doing each rank's range
    (both ranks on the one GPU of the test box, gloo as the process group)."""
    from oracle import oracle
    from varkoder_amd import synth
    fq = synth.sample_fastq(77, 30000, 150, dist=1)
    want = oracle.count_fastq(fq, 7)[0].astype(np.int64)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_giant_gpu_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    st, total = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert st == 0 and np.array_equal(total, want)


def test_without_a_tail_weightless_files_stay_in_the_static_deal():
    """tail_frac = 0 (ranks without a process group: each has only a local cursor): nothing may land in the tail, not
    even files of weight 0 (missing or empty ones) -- every rank would claim all of them from its own cursor."""
    from varkoder_amd import shard
    weights = [500, 0, 300, 0, 200]
    head, tail = shard.split_head_tail(weights, 0.0)
    assert tail == [] and head == [0, 2, 4, 1, 3]
    dealt = [sorted(head[j] for j in shard.shard_by_size([weights[i] for i in head], r, 2)) for r in range(2)]
    assert sorted(dealt[0] + dealt[1]) == list(range(5))
    head, tail = shard.split_head_tail(weights, 0.1)      # with a tail, the weightless ones are its first members
    assert sorted(head + tail) == list(range(5)) and set(tail) >= {1, 3}
