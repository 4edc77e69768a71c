"""Multi-GPU sharding of the `image` hot path: samples are independent units, so ranks
(one process per GPU) take disjoint sample sets and never exchange data-path bytes.

The reference maps samples over a multiprocessing.Pool (commands/image.py:1281-1284) and
lets the parent merge the per-sample stats (:1144-1170); here the pool is the set of
ranks of a torch.distributed job and rank 0 merges.  The only collectives are control
plane: a barrier, a MAX-reduce of elapsed time (bench.py) and an object gather of the
small stats dicts -- over gloo on CPU tensors, or RCCL when the group is `nccl`.
"""
import os
from collections import OrderedDict


def world_info():
    """(rank, world size, device index of this rank).  The device is LOCAL_RANK -- one process per GPU --
    unless VARKODER_AMD_DEVICE names one for every rank (a two-rank rehearsal on a one-GPU box)."""
    dev = os.environ.get("VARKODER_AMD_DEVICE")
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(dev) if dev not in (None, "") else int(os.environ.get("LOCAL_RANK", "0")))


def usable_cores():
    """Cores this process may really use: scheduler affinity, capped by the cgroup's CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(float(tok[0]) / float(tok[1]) + 0.5)))
            else:
                q = float(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / float(g.read().split()[0]) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def io_threads_per_rank(n_threads):
    """Host threads of one rank's file pipeline: the reference's -n is a per-node figure
    (commands/image.py:1281-1284 sizes ONE pool with it), so the ranks of a node share it -- 4 I/O
    threads per requested core as before, but never more than this rank's share of the cores the
    job may use (usable cores // LOCAL_WORLD_SIZE): eight ranks on a 16-core quota get 2 each, not 8 x 4n."""
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    share = max(1, usable_cores() // local_world)
    return max(1, min(max(1, int(n_threads)) * 4, share if local_world > 1 else max(share, 1) * 4))


def shard_indices(n_items, rank, world):
    """Round-robin shard for units of EQUAL size (bench.py's synthetic samples): item i belongs to rank
    i % world, balanced to within one item.  Files are sharded by size instead: shard_by_size."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_items, world))


def shard_by_size(weights, rank, world):
    """This rank's items (ascending indices) under the longest-processing-time rule: items in order of
    decreasing weight (ties: lower index first), each to the rank with the least weight so far (ties:
    lower rank).  Every rank computes the same assignment from the same weights -- no collective, no
    shared queue -- and no rank carries more than the mean load plus one item.

    Why not i % world: the reference hands samples to whichever pool worker is free
    (`pool.imap_unordered`, commands/image.py:1281-1284), so sizes do not matter to it; a static
    round-robin over SORTED file names does not have that property -- `split_fastqs/` holds one file
    per rung of the 1-2-5 ladder and sample (`<sample>@<bp 8 digits>K`, :682-708), the names sort by
    size within a sample, and i % world pins rungs to ranks (6 rungs on 2 ranks: 12.5 M against 26 M
    bases per sample)."""
    import heapq
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    heap = [(0, r) for r in range(world)]   # (load, rank): heapq pops the least load, then the lowest rank
    mine = []
    for i in order:
        load, r = heapq.heappop(heap)
        if r == rank:
            mine.append(i)
        heapq.heappush(heap, (load + int(weights[i]), r))
    return sorted(mine)


def split_head_tail(weights, tail_frac=0.1):
    """(head, tail): the items in longest-first order (ties: lower index), cut where the weight still to come drops to
    `tail_frac` of the total.  The head is dealt statically (shard_by_size over the head's weights); the tail -- the many
    small items at the end of the order -- is what ranks that finish early pull from a shared counter (TailQueue), the
    way the reference's pool hands the next sample to whichever worker is free (`imap_unordered`,
    commands/image.py:1281-1284): a static deal balances ESTIMATES (a .gz's text bytes, a k = 9 count whose time
    depends on the bases), the tail absorbs what the estimates got wrong."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    if tail_frac <= 0:   # (no tail asked for: weightless items -- missing or empty files -- stay in the static deal too)
        return order, []
    total = sum(int(w) for w in weights)
    left, cut = total, len(order)
    for n, i in enumerate(order):
        if left <= tail_frac * total:
            cut = n
            break
        left -= int(weights[i])
    return order[:cut], order[cut:]


class TailQueue:
    """A shared cursor over `n` items for the ranks of the default process group: next(count) claims the next `count`
    of them atomically (range, possibly empty at the end).  Control plane only: one counter in the group's own
    key-value store (the TCPStore torchrun's rendezvous set up) -- no tensor, no data-path collective.  Every rank must
    construct its queues in the same order (the key is a per-process sequence number).  Without a process group: a
    local cursor."""
    _seq = 0

    def __init__(self, n):
        import torch.distributed as dist
        TailQueue._seq += 1
        self.n = int(n)
        self.key = "varkoder_amd_tail_%d" % TailQueue._seq
        self.store = None
        self.local = 0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d
            self.store = distributed_c10d._get_default_store()

    def next(self, count=1):
        count = max(1, int(count))
        if self.store is None:
            start = self.local
            self.local += count
        else:
            start = int(self.store.add(self.key, count)) - count
        return range(min(start, self.n), min(start + count, self.n))


def rank_loads(weights, world):
    """Total weight of every rank under shard_by_size (tests, bench.py's report)."""
    return [sum(int(weights[i]) for i in shard_by_size(weights, r, world)) for r in range(world)]


GZ_TEXT_RATIO = 6   # text bytes per compressed byte assumed for a gzip file that does not say (see gz_text_bytes)


def gz_text_bytes(path, size=None):
    """Text bytes of a gzip file without inflating it -- what its own framing says, read from a few dozen bytes:
      * one member (what split_fastq / pigz / gzip write): the ISIZE word that ends the file, + 2^32 as often as it takes
        to reach the file's own size (ISIZE is the text size modulo 2^32, and FASTQ text is never smaller than its gzip);
      * BGZF (bgzip, BBTools through bgzip; the `BC` extra field): members of at most 64 KiB of text each, the last an
        empty end marker whose ISIZE is 0 -- the text-to-file ratio of the FIRST block (its ISIZE over its BSIZE)
        times the file size (the engine walks every block header when it stages the file; a weight needs no more);
      * anything else (several members glued together, a size word below the file size that is no wrap): the file
        size times GZ_TEXT_RATIO, the rule of round 5.
    Never raises: an unreadable file weighs its size times GZ_TEXT_RATIO (it fails later, on the rank that gets it)."""
    try:
        if size is None:
            size = os.path.getsize(path)
        if size < 18:
            return 0
        with open(path, "rb") as f:
            head = f.read(18)
            if head[:2] != b"\x1f\x8b":
                return size                                   # not gzip after all: plain text under a .gz name
            if head[3] & 4 and head[12:14] == b"BC" and int.from_bytes(head[10:12], "little") >= 6:
                bsize = int.from_bytes(head[16:18], "little") + 1
                if 26 <= bsize <= size:
                    f.seek(bsize - 4)
                    isize = int.from_bytes(f.read(4), "little")
                    if 0 < isize <= 65536:
                        return int(size * (isize / bsize))
                return size * GZ_TEXT_RATIO
            f.seek(size - 4)
            isize = int.from_bytes(f.read(4), "little")
        if isize >= size // 2:          # one member whose size word is plausible for text (ratio >= 0.5)
            return isize
        if size >= (1 << 31):           # a wrap: text of 4 GiB and more
            est = isize
            while est < size:
                est += 1 << 32
            return est
        return size * GZ_TEXT_RATIO      # several members glued together: the last one's size says nothing
    except OSError:
        try:
            return (size if size is not None else os.path.getsize(path)) * GZ_TEXT_RATIO
        except OSError:
            return 0


def file_weights(files):
    """Work estimate of each input file: its text bytes (a .gz: gz_text_bytes; a missing file 0 -- it fails later, on
    the rank that gets it)."""
    out = []
    for f in files:
        try:
            sz = os.path.getsize(f)
        except OSError:
            sz = 0
        out.append(gz_text_bytes(f, sz) if str(f).endswith(".gz") and sz else sz)
    return out


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over the default process group (1 rank: identity)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def merge_stats(parts):
    """Fold the ranks' {sample: OrderedDict(stats)} dicts into one, the way run_clean2img accumulates a sample's
    stats while it walks the sample's files (commands/image.py:1078: `stats[k] = stats.get(k, 0) + v` for the
    timings; :1070-1075: a failed step is recorded and stays).  Files are dealt by size (shard_by_size), so the
    rungs of ONE sample's ladder land on different ranks by design: a `*_time` value is the SUM over the ranks
    that held a file of the sample, `failed_step` survives whichever rank reported it (the first in rank order
    when several did), every other key (`base_frequencies_sd`, `splitting_bp_per_file`: the same on every rank
    that has it) is taken from the last rank that holds it."""
    merged = OrderedDict()
    for part in parts:
        for name in sorted(part or {}):
            row = merged.setdefault(name, OrderedDict())
            for key, val in part[name].items():
                if key.endswith("_time") and key in row:
                    row[key] = row[key] + val
                elif key == "failed_step" and key in row:
                    continue
                else:
                    row[key] = val
    return OrderedDict(sorted(merged.items()))


def gather_stats(local_stats, dst=0, error=None, with_errors=False):
    """Merge {sample: OrderedDict(...)} dicts of all ranks on rank `dst` (others get None) with merge_stats,
    the counterpart of process_stats' all_stats.update (commands/image.py:1167-1168).

    error: what went wrong on THIS rank (a string), or None.  It travels with the rank's dict, so a rank whose
    work raised still takes part in the collective -- a rank that stayed away would leave the others blocked
    here until the backend's timeout (the reference's pool loses one sample when a worker dies,
    commands/image.py:1281-1284, never the whole run).  with_errors: return (merged, [errors of all ranks])
    -- on `dst`; (None, None) elsewhere with gloo; every rank gets both under nccl (all_gather)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        merged = merge_stats([local_stats])
        return (merged, [error] if error else []) if with_errors else merged
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = (dict(local_stats), error)
    if dist.get_backend() == "nccl":
        # object collectives need CPU tensors: all_gather_object handles the device hop itself
        bucket = [None] * world
        dist.all_gather_object(bucket, mine)
    else:
        bucket = [None] * world if rank == dst else None
        dist.gather_object(mine, bucket, dst=dst)
    if rank != dst:
        return (None, None) if with_errors else None
    merged = merge_stats([part for part, _ in bucket])
    errors = [e for _, e in bucket if e]
    return (merged, errors) if with_errors else merged


def agreed_weights(files, src=0):
    """file_weights(files) as EVERY rank of the job will use them: rank `src` looks at the files and broadcasts
    what it saw (control plane: one small object).  shard_by_size is only a partition when all ranks feed it
    the same numbers, and two ranks that stat a file a moment apart -- one still being written, an NFS
    attribute cache, another node -- need not see the same size: files would be imaged twice, or by nobody,
    without an error.  Without a process group: this process's own view."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return file_weights(files)
    box = [file_weights(files) if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    if len(box[0]) != len(files):
        raise RuntimeError("ranks disagree about the list of input files (%d here, %d on rank %d)"
                           % (len(files), len(box[0]), src))
    return box[0]


def split_at_records(data, nparts, window=1 << 16):
    """Byte ranges [(start, end)] that cut one FASTQ text into `nparts` pieces at record starts, so
    that every piece is a well-formed FASTQ on its own (used to spread ONE giant sample over the
    ranks of a job; the kernels do the same recovery per wavefront, csrc/vkimg.hip sync_phase).

    A line start l is a header iff byte[l] == '@' and the line two lines later starts with '+'
    (a quality line may start with '@', but then that later line is a sequence line).  `data` is
    anything sliceable into bytes (bytes, memoryview, numpy uint8 array).
    """
    n = len(data)
    cuts = [0]
    for p in range(1, nparts):
        target = n * p // nparts
        if target <= cuts[-1]:
            cuts.append(cuts[-1])
            continue
        chunk = bytes(data[target:min(n, target + window)])
        nl = []
        pos = -1
        while len(nl) < 6:
            pos = chunk.find(b"\n", pos + 1)
            if pos < 0:
                break
            nl.append(pos)
        cut = None
        for i in range(max(0, min(4, len(nl) - 2))):
            li, lj = nl[i] + 1, nl[i + 2] + 1
            if lj < len(chunk) and chunk[li:li + 1] == b"@" and chunk[lj:lj + 1] == b"+":
                cut = target + li
                break
        if cut is None:          # no record start in the window (end of file, or giant lines)
            cut = n
        cuts.append(max(cut, cuts[-1]))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(nparts)]


def widen_u32(t):
    """int64 copy of a tensor that holds UNSIGNED 32-bit counts in int32 storage (torch has no u32
    arithmetic): a plain .to(int64) would sign-extend a count of 2^31 or more into a negative number."""
    import torch
    return t.to(torch.int64) & 0xFFFFFFFF


def allreduce_sum_(hist):
    """In-place SUM all-reduce of a histogram tensor over the default group: the one data-path
    collective of this package (RCCL over xGMI when the group is `nccl`; 64 KB at k=7, 1 MB at
    k=9 -- latency-bound, any algorithm fits one link).  Integer addition is associative, so the
    result is exact and independent of the reduction order."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if hist.is_cuda and dist.get_backend() != "nccl":   # gloo rehearsal of a GPU job: reduce on the host
            h = hist.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            hist.copy_(h)
        else:
            dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist
