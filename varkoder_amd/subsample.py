"""Step C of `varKoder image` on the GPU (opt-in, statistical equivalent).

The reference counts the bases of a cleaned read file, derives a 1-2-5 ladder of output sizes and
runs `reformat.sh samplebasestarget=<bp> sampleseed=<seed+i>` once per size to write
`<sample>@<bp>K.fq.gz`, which dsk then counts (split_fastq / run_parallel_reformats,
commands/image.py:577-725).  Here the cleaned reads stay in HBM: one launch counts everything and
returns the number of sites, and one launch per further ladder step counts a pseudo-random subset
of the reads (vk_count_sampled_device: Bernoulli per read with probability bp / nsites and seed
`seed + i`, so the expected -- not the exact -- number of bases is bp; BBTools' own RNG stream is
not reproduced).  Names, ladder arithmetic and the stats keys are the reference's.
"""
import math
from collections import OrderedDict

import numpy as np

from .config import SAMPLE_BP_SEP

ALL_READS = 1 << 32


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
    full_hist, status, sites = engine.count_sampled(fastq, offsets, lengths, seed, ALL_READS, parts=parts)
    nsites = sites[:, 0].cpu().numpy()
    status_h = status.cpu().numpy()
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
    # every further step of every sample in ONE launch: a (sample, step) pair is a sample of its own to the kernel
    # (same bytes, its own seed and threshold); the pairs of a sample run side by side and share its lines in L2
    pairs = []
    for i in range(n):
        for level, bp in enumerate(plans[i]):
            if level == 0 and bp >= nsites[i]:
                out[i]["steps"].append((bp, full_hist[i], int(nsites[i])))      # everything: already counted
            else:
                pairs.append((i, level))
    if pairs:
        idx = [i for i, _ in pairs]
        thr = np.array([threshold(plans[i][level], nsites[i]) for i, level in pairs], dtype=np.uint64)
        seeds = np.array([seed + level for _, level in pairs], dtype=np.uint64)
        h, _, st = engine.count_sampled(fastq, offsets[idx], lengths[idx], seeds, thr, parts=parts)
        taken = st[:, 1].cpu().numpy()
        for j, (i, level) in enumerate(pairs):
            out[i]["steps"].append((plans[i][level], h[j], int(taken[j])))
    return out
