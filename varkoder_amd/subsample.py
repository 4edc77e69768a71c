"""Step C of `varKoder image` on the GPU (opt-in, statistical equivalent).

The reference counts the bases of a cleaned read file, derives a 1-2-5 ladder of output sizes and
runs `reformat.sh samplebasestarget=<bp> sampleseed=<seed+i>` once per size to write
`<sample>@<bp>K.fq.gz`, which dsk then counts (split_fastq / run_parallel_reformats,
commands/image.py:577-725).  Here the cleaned reads stay in HBM: one pass per sample lists its reads
and counts its sites (vk_read_index_device), and ONE launch counts every further ladder step of every
sample, each over a pseudo-random subset of the reads (vk_count_sampled_device: Bernoulli per read
with probability bp / nsites and seed `seed + i`, so the expected -- not the exact -- number of bases
is bp; BBTools' own RNG stream is not reproduced), walking only the reads it takes.  Names, ladder
arithmetic and the stats keys are the reference's.
"""
import math
from collections import OrderedDict

import numpy as np

from .config import SAMPLE_BP_SEP

ALL_READS = 1 << 32
WALK_MAX_THRESHOLD_SPILL = (1 << 32) // 32     # vkimg.hip kWalkMaxThresholdSpill


def sites_ladder(nsites, min_bp=50000, max_bp=None, is_query=False):
    """Sizes (bp) of the files split_fastq would write for a cleaned file of `nsites` bases
    (commands/image.py:677-701), largest first.  Raises like the reference when the file holds
    less than min_bp."""
    nsites = int(nsites)
    if max_bp is None:
        sizes = [nsites]
    elif is_query or nsites > min_bp:
        sizes = [min(nsites, int(max_bp))]
    else:
        raise Exception("Input file has less than minimum data.")
    if not is_query:
        while sizes[-1] > min_bp:
            oneless = sizes[-1] - 1
            nzeros = int(math.log10(oneless))
            first_digit = int(oneless / (10 ** nzeros))
            if first_digit in (1, 2, 5):
                sizes.append(first_digit * (10 ** nzeros))
            else:
                sizes.append(max(x for x in (1, 2, 5) if x < first_digit) * (10 ** nzeros))
        if sizes[-1] < min_bp:
            del sizes[-1]
    return sizes


def split_name(prefix, bp):
    """`<prefix>@<bp/1000, 8 digits>K` (commands/image.py:704-713, without the .fq.gz)."""
    return prefix + SAMPLE_BP_SEP + str(int(bp / 1000)).rjust(8, "0") + "K"


def threshold(bp, nsites):
    """Per-read probability bp / nsites as the kernel's 32.32 threshold (2^32 = every read)."""
    if nsites <= 0 or bp >= nsites:
        return ALL_READS
    return min(ALL_READS, int(bp) * ALL_READS // int(nsites))


def ladder_counts(engine, fastq, offsets, lengths, seed=0, min_bp=50000, max_bp=None, is_query=False, parts=0):
    """All ladder steps of a batch of cleaned samples resident in HBM.

    Returns one record per sample: OrderedDict(nsites=..., status=..., steps=[(bp, hist uint32
    tensor [4^k] on the device, sites_taken)], error=None | str).  Step i uses seed + i like the
    reference's `sampleseed` (image.py:585)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    lengths = np.asarray(lengths, dtype=np.uint64)
    n = len(offsets)
    # one pass per sample: the read index (every read's anchor) and the number of sites; the subsamples then walk the
    # reads they take (vk_read_index_device / vk_ladder.h) instead of streaming the text once each
    # (when the ladder is not capped every sample's first step takes everything: the plain count comes out of the same pass)
    full_hist = None
    if max_bp is None:
        full_hist, nsites, status_h = engine.count_index(fastq, offsets, lengths, parts=parts)
    else:
        nsites, status_h = engine.read_index(fastq, offsets, lengths, parts=parts)
    out, plans = [], []
    for i in range(n):
        rec = OrderedDict(nsites=int(nsites[i]), status=int(status_h[i]), steps=[], error=None)
        try:
            sizes = sites_ladder(nsites[i], min_bp, max_bp, is_query) if not status_h[i] else []
            if status_h[i]:
                rec["error"] = "inconsistent FASTQ framing"
        except Exception as e:  # noqa: BLE001 - the reference's "less than minimum data"
            sizes, rec["error"] = [], str(e)
        plans.append(sizes)
        out.append(rec)
    # steps that take everything: the plain count of those samples, one launch
    whole = [i for i in range(n) if plans[i] and plans[i][0] >= nsites[i]]
    if whole and full_hist is not None:
        for i in whole:
            out[i]["steps"].append((plans[i][0], full_hist[i], int(nsites[i])))
    elif whole:
        h, _ = engine.count(fastq, offsets[whole], lengths[whole], parts=parts)
        for j, i in enumerate(whole):
            out[i]["steps"].append((plans[i][0], h[j], int(nsites[i])))
    # every further step of every sample in ONE launch: a (sample, step) pair is a sample of its own to the kernel
    # (same bytes, its own seed and threshold)
    pairs = [(i, level) for i in range(n) for level, bp in enumerate(plans[i]) if not (level == 0 and bp >= nsites[i])]
    if pairs:
        thr_all = np.array([threshold(plans[i][level], nsites[i]) for i, level in pairs], dtype=np.uint64)
        # k = 8, 9: the walker only pays for subsamples of a few per cent of the reads (WALK_MAX_THRESHOLD_SPILL, the
        # library's rule for a whole call): the larger steps go in a call of their own, which streams
        small = thr_all <= WALK_MAX_THRESHOLD_SPILL if engine.k > 7 else np.ones(len(pairs), dtype=bool)
        got = {}
        for sel in (np.flatnonzero(small), np.flatnonzero(~small)):
            if sel.size == 0:
                continue
            idx = [pairs[j][0] for j in sel]
            seeds = np.array([seed + pairs[j][1] for j in sel], dtype=np.uint64)
            h, _, st = engine.count_sampled(fastq, offsets[idx], lengths[idx], seeds, thr_all[sel], parts=parts)
            taken = st[:, 1].cpu().numpy()
            for jj, j in enumerate(sel):
                got[int(j)] = (h[jj], int(taken[jj]))
        for j, (i, level) in enumerate(pairs):
            out[i]["steps"].append((plans[i][level], got[j][0], got[j][1]))
        for rec in out:      # (steps in the ladder's order, largest first)
            rec["steps"].sort(key=lambda t: -t[0])
    return out
