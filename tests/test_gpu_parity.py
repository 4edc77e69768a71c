"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.

Integer / byte work: the bar is bit-exact equality.
"""
import hashlib

import numpy as np
import pytest

import vectors
from fastq_cases import edge_cases
from oracle import oracle
from varkoder_amd import synth
from varkoder_amd.mapping import pixel_lut, side

pytestmark = pytest.mark.gpu

KS = (5, 6, 7, 8, 9)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_library_loaded_and_gpu_visible(engines):
    import torch
    assert torch.cuda.is_available()
    eng = engines(7)
    assert eng.L.vk_abi_version() == 1


@pytest.mark.parametrize("k", KS)
def test_count_edge_cases_match_oracle(engines, k):
    eng = engines(k)
    for name, fq in edge_cases().items():
        want, nwin, st = oracle.count_fastq(fq, k)
        got, status = eng.count_host(fq)
        assert status == 0, (name, status)
        assert st == 0, name
        assert int(got.sum(dtype=np.uint64)) == nwin, name
        assert np.array_equal(got, want), name


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("dist", (0, 1, 2))
def test_count_synthetic_batch_matches_oracle(engines, k, dist):
    """Small parity set of SURVEY 8d: 8 samples x 10,000 reads x 150 bp, generated on the
    host, counted in one batched launch (several workgroup splits).  dist 2: reads of every length
    from 0 to 290 under long headers, as fastp hands them to step D (synth.py)."""
    eng = engines(k)
    samples = [synth.sample_fastq(s, 10000, 150, dist=dist) for s in range(8)]
    dev, offs, lens = eng.upload(samples)
    want = np.stack([oracle.count_fastq(s, k)[0] for s in samples])
    for parts in (0, 1, 3, 8):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any(), parts
        got = hist.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), parts


def test_device_generator_equals_host_generator(engines):
    eng = engines(7)
    for dist in (0, 1, 2):
        dev, offs, lens = eng.synth(5, 3, 2000, 150, dist=dist)
        got = dev.cpu().numpy()
        for j in range(3):
            want = synth.sample_fastq(5 + j, 2000, 150, dist=dist)
            o = int(offs[j])
            assert int(lens[j]) == want.size, (dist, j)
            assert np.array_equal(got[o:o + want.size], want), (dist, j)
    # dist 2 across the generator's slabs of 64 samples and chunks of 256 reads, padding zeroed
    dev, offs, lens = eng.synth(60, 70, 700, 150, dist=2)
    got = dev.cpu().numpy()
    for j in (0, 63, 64, 69):
        want = synth.sample_fastq(60 + j, 700, 150, dist=2)
        o = int(offs[j])
        assert int(lens[j]) == want.size and np.array_equal(got[o:o + want.size], want), j
        assert not got[o + want.size:(o + want.size + 15) // 16 * 16].any(), j


@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("mapping", ("cgr", "varKode"))
def test_image_matches_reference_golden(engines, manifest, golden_small, k, mapping):
    """Expected pixels were produced by the reference's own make_image (oracle/gen_golden.py)."""
    eng = engines(k, mapping)
    for dist in vectors.DISTS:
        fwd = vectors.fwd_hist(k, dist)
        img = eng.image_host(fwd)
        key = f"k{k}_{mapping}_{dist}"
        assert list(img.shape) == manifest["image_cases"][key]["shape"]
        if key in golden_small:
            assert np.array_equal(img, golden_small[key]), key
        assert sha(img) == manifest["image_cases"][key]["sha256"], key


@pytest.mark.parametrize("k,mapping", [(7, "cgr"), (7, "varKode"), (9, "cgr"), (5, "varKode")])
def test_fastq_to_image_batch_matches_oracle(engines, k, mapping):
    eng = engines(k, mapping)
    samples = [synth.sample_fastq(100 + s, 3000 + 500 * s, 150, dist=s & 1) for s in range(5)]
    dev, offs, lens = eng.upload(samples)
    img, hist, status = eng.fastq_to_images(dev, offs, lens)
    assert not status.cpu().numpy().any()
    lut, n = pixel_lut(k, mapping), side(k, mapping)
    got = img.cpu().numpy()
    for i, s in enumerate(samples):
        want, _, st = oracle.fastq_to_image(s, k, lut, n * n)
        assert st == 0
        assert np.array_equal(got[i].ravel(), want), i


def test_bad_framing_sets_status(engines):
    eng = engines(7)
    good = synth.sample_fastq(1, 50, 150).tobytes()
    _, st = eng.count_host(good[1:])            # does not start with '@'
    assert st & 1
    _, st = eng.count_host(good[:-200])         # truncated inside a record
    assert st & 2
    _, st = eng.count_host(good)
    assert st == 0
    # wrapped (multi-line) FASTQ: 6 lines per record can look consistent mod 4; the first record gives it away
    wrapped = b"".join(b"@r%d\nACGTACGTAC\nGGGTTTAAAC\n+\nIIIIIIIIII\nIIIIIIIIII\n" % i for i in range(40))
    _, st = eng.count_host(wrapped)
    assert st & 1
    _, st = eng.count_host(b">seq1\nACGTACGTACGTACGT\n>seq2\nACGTACGTAAAA\n")        # FASTA
    assert st & 1


@pytest.mark.parametrize("k,mapping", [(7, "varKode"), (9, "cgr")])
def test_full_size_samples_match_oracle(engines, k, mapping):
    """BASELINE.json sizes: 1,000,000 reads x 150 bp per sample (320 MB of FASTQ text each),
    generated on the device.  The oracle counts such a sample in well under a second, so the
    check is exact: histograms and images bit-identical, for several workgroup splits, plus
    the size-independent properties (window checksum, linearity under concatenation)."""
    import torch
    eng = engines(k, mapping)
    reads, L = 1_000_000, 150
    dev, offs, lens = eng.synth(900, 2, reads, L, dist=1)
    host = dev.cpu().numpy()
    lut, n = pixel_lut(k, mapping), side(k, mapping)
    want_h, want_i, nwins = [], [], []
    for j in range(2):
        buf = host[int(offs[j]):int(offs[j]) + int(lens[j])]
        h, nwin, st = oracle.count_fastq(buf, k)
        assert st == 0
        want_h.append(h)
        nwins.append(nwin)
        want_i.append(oracle.image(oracle.strand_merge(h, k), k, lut, n * n))
    for parts in (1, 4, 0):
        img, hist, status = eng.fastq_to_images(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        got_h = hist.cpu().numpy().view(np.uint32)
        got_i = img.cpu().numpy()
        for j in range(2):
            assert int(got_h[j].sum(dtype=np.uint64)) == nwins[j]          # window checksum
            assert np.array_equal(got_h[j], want_h[j]), (parts, j)
            assert np.array_equal(got_i[j].ravel(), want_i[j]), (parts, j)
    # linearity: the two samples counted as ONE 640 MB sample = sum of the two histograms
    both_off = np.array([0], dtype=np.uint64)
    both_len = np.array([int(lens[0] + lens[1])], dtype=np.uint64)
    hist, status = eng.count(dev, both_off, both_len)
    assert not status.cpu().numpy().any()
    assert np.array_equal(hist.cpu().numpy().view(np.uint32)[0], want_h[0] + want_h[1])
    # expected window count from the text itself: reads*(L-k+1) minus windows touching an N
    seq = dev[:int(lens[0])].view(reads, 2 * L + 20)[:, 16:16 + L]
    isn = (seq == ord("N")).to(torch.int32)
    cs = torch.cumsum(isn, dim=1)
    cs = torch.cat([torch.zeros((reads, 1), dtype=cs.dtype, device=cs.device), cs], dim=1)
    clean = ((cs[:, k:] - cs[:, :-k]) == 0).sum().item()
    assert clean == nwins[0]


@pytest.mark.parametrize("k,mapping", [(7, "varKode"), (9, "cgr")])
def test_full_size_fastp_shaped_sample_matches_oracle(engines, k, mapping):
    """One sample of 1,000,000 reads of the lengths fastp really writes (0 .. 290 bases, 40 .. 70 byte headers,
    '@' and '+' in the quality lines; synth.py dist 2), 364 MB of text: histogram and image bit-identical."""
    eng = engines(k, mapping)
    dev, offs, lens = eng.synth(950, 1, 1_000_000, 150, dist=2)
    host = dev.cpu().numpy()[:int(lens[0])]
    lut, n = pixel_lut(k, mapping), side(k, mapping)
    want_h, nwin, st = oracle.count_fastq(host, k)
    assert st == 0
    want_i = oracle.image(oracle.strand_merge(want_h, k), k, lut, n * n)
    for parts in (0, 3):
        img, hist, status = eng.fastq_to_images(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        got_h = hist.cpu().numpy().view(np.uint32)[0]
        assert int(got_h.sum(dtype=np.uint64)) == nwin
        assert np.array_equal(got_h, want_h), parts
        assert np.array_equal(img.cpu().numpy()[0].ravel(), want_i), parts
    if k <= 7:
        general, pieces = eng.last_count_general()
        assert 0 < general <= pieces


@pytest.fixture
def route_engine(monkeypatch):
    """An engine of its own for a k = 8, 9 test under extra environment: route "pairs" = VKIMG_SPILL_PAIRS=1 (the pair
    route of rounds 1-4, which subsampled and packed launches still use), "quads" = the shipped quad route."""
    from varkoder_amd.engine import ImageEngine
    made = []

    def make(k, route, **env):
        if route == "pairs":
            monkeypatch.setenv("VKIMG_SPILL_PAIRS", "1")
        else:
            monkeypatch.delenv("VKIMG_SPILL_PAIRS", raising=False)
        for name, value in env.items():
            monkeypatch.setenv(name, str(value))
        made.append(ImageEngine(k=k, mapping="cgr", device=0))
        return made[-1]
    yield make
    for e in made:
        e.close()


@pytest.mark.parametrize("route", ("quads", "pairs"))
@pytest.mark.parametrize("k", (8, 9))
def test_spill_path_skewed_input_takes_the_exact_fallbacks(route_engine, k, route):
    """k = 8, 9 bucket windows through LDS queues into per-part streams.  A low-complexity sample
    sends (almost) every window to ONE part: its queue and its bucket overflow, and the overflow
    must still be counted exactly (global-atomic fallbacks)."""
    from fastq_cases import rec
    rng = np.random.default_rng(k)
    reads = []
    for i in range(30000):
        r = rng.random()
        if r < 0.7:
            seq = "A" * 150
        elif r < 0.85:
            seq = "ACACACACAC" * 15
        else:
            seq = "".join(rng.choice(list("ACGT"), size=150))
        reads.append(rec(f"r{i}", seq))
    fq = b"".join(reads)
    eng = route_engine(k, route)
    want, nwin, st = oracle.count_fastq(fq, k)
    assert st == 0
    dev, offs, lens = eng.upload([fq, fq[: len(fq) // 2 - (len(fq) // 2) % 1 ]])
    # second sample: cut at a record boundary
    cut = fq.rfind(b"\n@", 0, len(fq) // 2) + 1
    dev, offs, lens = eng.upload([fq, fq[:cut]])
    want2 = oracle.count_fastq(fq[:cut], k)[0]
    for parts in (1, 4):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        got = hist.cpu().numpy().view(np.uint32)
        assert int(got[0].sum(dtype=np.uint64)) == nwin
        assert np.array_equal(got[0], want), parts
        assert np.array_equal(got[1], want2), parts


@pytest.mark.parametrize("k", (5, 6, 7))
def test_low_complexity_runs_switch_the_window_loop_and_stay_exact(engines, k):
    """Stretches of identical reads (poly-A, short tandem repeats) make the LDS-histogram kernel count
    whole groups of lanes with one add (windows_lds_hot), entered and left piece by piece as the input
    changes: histograms equal the oracle's whatever the mix."""
    from fastq_cases import rec
    rng = np.random.default_rng(100 + k)
    recs = []
    i = 0
    for block in range(24):
        kind = block % 4
        for _ in range(int(rng.integers(40, 400))):
            if kind == 0:
                seq = "A" * 150
            elif kind == 1:
                seq = "".join(rng.choice(list("ACGT"), size=int(rng.integers(30, 200))))
            elif kind == 2:
                seq = ("ACGT" * 40)[: int(rng.integers(60, 160))]
            else:
                seq = "T" * 70 + "N" + "G" * 79
            recs.append(rec(f"r{i}", seq))
            i += 1
    fq = b"".join(recs)
    eng = engines(k)
    want, nwin, st = oracle.count_fastq(fq, k)
    assert st == 0
    dev, offs, lens = eng.upload([fq, b"".join(recs[::-1])])
    want2 = oracle.count_fastq(b"".join(recs[::-1]), k)[0]
    for parts in (1, 3):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        got = hist.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[0], want), parts
        assert np.array_equal(got[1], want2), parts


@pytest.mark.parametrize("route", ("quads", "pairs"))
@pytest.mark.parametrize("k", (8, 9))
def test_spill_path_with_a_full_arena_counts_directly(route_engine, k, route):
    """The bucket streams of k = 8, 9 live in a per-sample arena of 4 KiB runs.  With the arena cut
    to a handful of runs (VKIMG_SPILL_RUNS_CAP) most blocks find no room and must be counted with
    global atomics instead: same histogram."""
    eng = route_engine(k, route, VKIMG_SPILL_RUNS_CAP=24)
    samples = [synth.sample_fastq(40 + i, 6000, 150, dist=i & 1) for i in range(3)]
    dev, offs, lens = eng.upload(samples)
    for parts in (1, 3):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        got = hist.cpu().numpy().view(np.uint32)
        for i, fq in enumerate(samples):
            assert np.array_equal(got[i], oracle.count_fastq(fq, k)[0]), (parts, i)


@pytest.mark.parametrize("k", (8, 9))
def test_quad_route_with_full_regions_counts_directly(route_engine, k):
    """Quads of which only some windows count (a read's first and last, the neighbours of an N) travel through a region
    per (workgroup, bucket) of pass A; with the regions cut to four entries (VKIMG_SPILL_MISC_CAP = 2: a power of two) most of them find no
    room and are counted window by window with global atomics: same histogram -- reads of every shape, reads riddled
    with N, the edge cases."""
    from fastq_cases import rec
    rng = np.random.default_rng(31 + k)
    holes = b"".join(rec(f"n{i}", "".join(rng.choice(list("ACGTN"), p=[0.24, 0.24, 0.24, 0.24, 0.04], size=int(rng.integers(20, 300)))))
                     for i in range(3000))
    samples = [synth.sample_fastq(90 + i, 6000, 150, dist=i % 3) for i in range(3)] + [holes] + list(edge_cases().values())
    want = [oracle.count_fastq(s, k)[0] for s in samples]
    for env in ({"VKIMG_SPILL_MISC_CAP": 2}, {}):
        eng = route_engine(k, "quads", **env)
        dev, offs, lens = eng.upload(samples)
        for parts in (0, 1, 3):
            hist, status = eng.count(dev, offs, lens, parts=parts)
            assert not status.cpu().numpy().any(), parts
            got = hist.cpu().numpy().view(np.uint32)
            for i in range(len(samples)):
                assert np.array_equal(got[i], want[i]), (env, parts, i)


@pytest.mark.parametrize("k", (8, 9))
def test_spill_replay_pair_counters_wrap_and_the_wide_replay_takes_over(engines, k, monkeypatch):
    """(The pair route, VKIMG_SPILL_PAIRS=1; the quad route's u32 counters cannot wrap: it counts the same sample first.)
    Pass B of k = 8, 9 counts PAIRS in u16 counters; 65536 equal pairs in one bucket stream wrap one, the job's
    counters then do not add up to its entries and the job is replayed into u32 window counters
    (vk_bucket_count_wide_kernel).  A repeat of period 10 (beyond what the low-complexity shortcuts of pass A take
    out: period <= 8) in 60,000 reads does that to the buckets its pairs fall into, the random reads beside it keep
    the other buckets on the u16 path -- one histogram, from both tables.  Also: every job forced through the wide
    replay (VKIMG_SPILL_FORCE_WIDE) gives the same histogram as the shipped path."""
    from fastq_cases import rec
    from varkoder_amd.engine import ImageEngine
    rng = np.random.default_rng(7 + k)
    unit = "ACGTTGCATC"
    reads = []
    for i in range(80000):
        if i % 4 != 3:
            ph = int(rng.integers(0, 10))
            seq = (unit * 17)[ph:ph + 150]
        else:
            seq = "".join(rng.choice(list("ACGT"), size=150))
        reads.append(rec(f"r{i}", seq))
    fq = b"".join(reads)
    want, nwin, st = oracle.count_fastq(fq, k)
    assert st == 0 and int(want.max()) > 3 * 65536
    eng = engines(k)
    dev, offs, lens = eng.upload([fq])
    for parts in (1, 5):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        assert np.array_equal(hist.cpu().numpy().view(np.uint32)[0], want), parts
    monkeypatch.setenv("VKIMG_SPILL_PAIRS", "1")
    e1 = ImageEngine(k=k, mapping="cgr", device=0)
    try:
        d1, o1, l1 = e1.upload([fq])
        for parts in (1, 5):
            hist, status = e1.count(d1, o1, l1, parts=parts)
            assert not status.cpu().numpy().any()
            assert np.array_equal(hist.cpu().numpy().view(np.uint32)[0], want), parts
    finally:
        e1.close()
    monkeypatch.setenv("VKIMG_SPILL_FORCE_WIDE", "1")
    e2 = ImageEngine(k=k, mapping="cgr", device=0)
    try:
        samples = [synth.sample_fastq(70 + i, 5000, 150, dist=i) for i in range(3)]
        d2, o2, l2 = e2.upload(samples + [fq])
        hist, status = e2.count(d2, o2, l2)
        got = hist.cpu().numpy().view(np.uint32)
        for i, smp in enumerate(samples):
            assert np.array_equal(got[i], oracle.count_fastq(smp, k)[0]), i
        assert np.array_equal(got[3], want)
    finally:
        e2.close()


@pytest.mark.parametrize("k", (8, 9))
def test_spill_pair_route_still_counts_like_the_oracle(route_engine, k):
    """VKIMG_SPILL_PAIRS=1: the plain count through the pair kernels (subsampled launches use them whatever the
    environment says): synthetic batches of all three read shapes, the edge cases, several workgroup splits."""
    eng = route_engine(k, "pairs")
    samples = [synth.sample_fastq(60 + i, 8000, 150, dist=i % 3) for i in range(6)] + list(edge_cases().values())
    want = [oracle.count_fastq(s, k)[0] for s in samples]
    dev, offs, lens = eng.upload(samples)
    for parts in (0, 1, 3):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any(), parts
        got = hist.cpu().numpy().view(np.uint32)
        for i in range(len(samples)):
            assert np.array_equal(got[i], want[i]), (parts, i)


@pytest.mark.parametrize("k", (8, 9))
def test_spill_path_through_the_packed_stream(k, monkeypatch):
    """VKIMG_SPILL_PACKED=1: pass A of k = 8, 9 as vk_pack_kernel (the text packed into 2-bit codes + masks, lanes the
    line pass cannot describe set aside and counted by vk_aside_kernel) + the partition of the packed stream --
    slower than the shipped single kernel for a plain count, but the same histogram: synthetic batches of all three
    read shapes, the edge cases, several workgroup splits."""
    from varkoder_amd.engine import ImageEngine
    monkeypatch.setenv("VKIMG_SPILL_PACKED", "1")
    eng = ImageEngine(k=k, mapping="cgr", device=0)
    try:
        samples = [synth.sample_fastq(50 + i, 8000, 150, dist=i % 3) for i in range(6)] + list(edge_cases().values())
        want = [oracle.count_fastq(s, k)[0] for s in samples]
        dev, offs, lens = eng.upload(samples)
        for parts in (0, 1, 3):
            hist, status = eng.count(dev, offs, lens, parts=parts)
            assert not status.cpu().numpy().any(), parts
            got = hist.cpu().numpy().view(np.uint32)
            for i in range(len(samples)):
                assert np.array_equal(got[i], want[i]), (parts, i)
    finally:
        eng.close()


@pytest.mark.parametrize("k", KS)
def test_count_fuzz_batches(engines, k):
    """Hundreds of random adversarial (well-formed) FASTQ samples per launch, random workgroup
    splits: every histogram equals the oracle's."""
    from fastq_cases import random_fastq
    rng = np.random.default_rng(100 + k)
    eng = engines(k)
    for rnd in range(3):
        samples = [random_fastq(rng) for _ in range(150)]
        want = np.stack([oracle.count_fastq(s, k)[0] for s in samples])
        dev, offs, lens = eng.upload(samples)
        for parts in (1, int(rng.integers(2, 6))):
            hist, status = eng.count(dev, offs, lens, parts=parts)
            assert not status.cpu().numpy().any(), (rnd, parts)
            got = hist.cpu().numpy().view(np.uint32)
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert bad.size == 0, (rnd, parts, bad[:5])


def test_giant_sample_ranges_sum_to_the_whole(engines):
    """SURVEY 8e optional row: one sample cut at record boundaries into per-rank byte ranges;
    the per-range GPU histograms sum to the whole-sample histogram (what the RCCL all-reduce of
    engine.count_giant_sample computes across ranks)."""
    from varkoder_amd.engine import count_giant_sample
    from varkoder_amd.shard import split_at_records
    eng = engines(7)
    fq = synth.sample_fastq(500, 40000, 150, dist=1).tobytes()
    want = oracle.count_fastq(fq, 7)[0].astype(np.int64)
    h, st = count_giant_sample(eng, fq)                      # world = 1: no collective
    assert st == 0 and np.array_equal(h.cpu().numpy(), want)
    total = np.zeros_like(want)
    for s, e in split_at_records(fq, 4):
        hist, status = eng.count_host(fq[s:e])
        assert status == 0
        total += hist
    assert np.array_equal(total, want)


@pytest.mark.parametrize("k,mapping", [(7, "cgr"), (7, "varKode"), (8, "cgr"), (8, "varKode"), (9, "cgr"), (9, "varKode")])
def test_large_image_order_statistics_by_counting(engines, k, mapping):
    """Images of k >= 7 take vk_image_count_kernel (counted ranks + a sorted list of outliers) with
    vk_image_kernel's sort as the fallback: every regime against the oracle's sort."""
    import torch
    eng = engines(k, mapping)
    n = 4 ** k
    rng = np.random.default_rng(k * 7 + len(mapping))
    cases = {
        "small": rng.poisson(500, n),                                              # one counting pass
        "both_halves": rng.integers(0, 33000, n),                                  # values on both sides of 32768
        "few_outliers": np.where(rng.random(n) < 0.001, rng.integers(10 ** 5, 10 ** 9, n), rng.poisson(900, n)),
        "many_outliers": np.where(rng.random(n) < 0.7, rng.integers(40000, 2 ** 30, n), rng.poisson(50, n)),  # -> sort
        "empty": np.zeros(n, dtype=np.int64),                                      # every pixel the same value
        "edges": rng.choice(np.array([0, 16382, 16383, 16384, 32766, 32767, 32768, 65534, 65535, 65536]), n),
        "huge": rng.integers(2 ** 30 - 5, 2 ** 30, n),                             # all values listed, none counted
    }
    names = list(cases)
    fwd = np.stack([np.ascontiguousarray(cases[c], dtype=np.uint32) for c in names])
    img = eng.images(torch.from_numpy(fwd.view(np.int32)).cuda()).cpu().numpy()
    lut, s = pixel_lut(k, mapping), side(k, mapping)
    for i, name in enumerate(names):
        want = oracle.image(oracle.strand_merge(fwd[i], k), k, lut, s * s)
        assert np.array_equal(img[i].ravel(), want), name


@pytest.mark.parametrize("mapping", ("varKode", "cgr"))
def test_k7_images_by_counting_equal_the_sorted_ones(monkeypatch, mapping):
    """Round 5 moved k = 7 images from the LDS sort to the counting kernel: both routes on histograms of real
    shape (counts of synthetic samples of very different depth, a poly-A sample, an empty one) against the oracle."""
    import torch
    from varkoder_amd.engine import ImageEngine
    from fastq_cases import random_fastq, rec
    rng = np.random.default_rng(77)
    blobs = [random_fastq(rng, nrec=nrec) for nrec in (3, 400, 20000)]
    blobs.append(b"".join(rec("a%d" % i, "A" * 150) for i in range(5000)))
    blobs.append(b"".join(rec("t%d" % i, "ACGTTGCAAC" * 15) for i in range(5000)))
    blobs.append(rec("n", "N" * 100))
    hist = np.stack([oracle.count_fastq(b, 7)[0] for b in blobs]).astype(np.uint32)
    hist[2] *= 40000                                       # counts far beyond 16 bits: listed, not counted
    lut, s = pixel_lut(7, mapping), side(7, mapping)
    want = [oracle.image(oracle.strand_merge(h, 7), 7, lut, s * s) for h in hist]
    for sort_only in ("0", "1"):
        monkeypatch.setenv("VKIMG_IMAGE_SORT_ONLY", sort_only)
        eng = ImageEngine(k=7, mapping=mapping, device=0)
        try:
            img = eng.images(torch.from_numpy(hist.view(np.int32)).cuda()).cpu().numpy()
        finally:
            eng.close()
        for i in range(len(blobs)):
            assert np.array_equal(img[i].ravel(), want[i]), (sort_only, i)


# ---- the sequence-only heavy stage of the k <= 7 kernel (vk_count_dense_kernel) --------------------

import functools


@functools.lru_cache(maxsize=1)
def _dense_blobs():
    """Inputs large enough to leave the first / last piece of a wave's range (those take the general
    path): reads around the granule (16 B) and block (64 B) sizes, ragged reads, headers that end on
    block borders, very long lines, bytes >= 0x80 in headers, low complexity, CRLF, odd alphabets."""
    from fastq_cases import random_fastq, rec
    rng = np.random.default_rng(31337)
    blobs = [random_fastq(rng, nrec=int(rng.integers(1500, 6000))) for _ in range(10)]
    for n in (14, 15, 16, 17, 31, 33, 45, 47, 48, 49, 63, 64, 65, 129, 150, 151, 250):
        blobs.append(b"".join(rec("x" * int(rng.integers(1, 70)), "".join(rng.choice(list("ACGTN"), size=n)))
                              for _ in range(6000)))
    blobs.append(b"".join(rec("q%d" % i, "ACGT" * 3000) for i in range(120)))
    blobs.append(b"".join(rec("h" * 9000, "ACGTTGCAAC" * 30) for i in range(120)))
    blobs.append(b"".join(rec("u%d \xc3\xa9" % i, "ACGTTGCAAC" * 15) for i in range(6000)))
    blobs.append(b"".join(rec("p%d" % i, "A" * 150 if i % 3 else "ACGTTGCAAC" * 15) for i in range(8000)))
    return blobs


@pytest.mark.parametrize("k", (5, 6, 7))
def test_dense_stage_matches_oracle_and_classic_kernel(engines, k):
    eng = engines(k)
    blobs = _dense_blobs()
    dev, offs, lens = eng.upload(blobs)
    want = np.stack([oracle.count_fastq(b, k)[0] for b in blobs])
    for parts in (0, 1, 2, 5):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any(), parts
        got = hist.cpu().numpy().view(np.uint32)
        bad = [i for i in range(len(blobs)) if not np.array_equal(got[i], want[i])]
        assert not bad, (k, parts, bad)


@pytest.mark.parametrize("k", (5, 6, 7, 8, 9))
def test_tandem_repeats_of_every_short_period_stay_exact(engines, k):
    """Microsatellite-like input: reads that are tandem repeats of period 1..9 (seen at every phase, mixed
    motifs, broken by N, of odd lengths, next to ordinary reads).  The k <= 7 kernels count the lane groups that
    repeat with ONE add per residue class (windows_lds_hot), everything else as usual: same histograms."""
    from fastq_cases import rec
    rng = np.random.default_rng(700 + k)
    motifs = ["A", "AC", "ACG", "ACGT", "AACGT", "ACGTTG", "ACGGTCA", "ACGTTGCA", "ACGTTGCAT", "GT", "TTAGGG", "CAG"]
    recs, i = [], 0
    for block in range(36):
        m = motifs[block % len(motifs)]
        for _ in range(int(rng.integers(150, 500))):
            n = int(rng.integers(20, 260))
            ph = int(rng.integers(0, len(m)))
            seq = ((m * 300)[ph: ph + n])
            r = rng.random()
            if r < 0.1:
                seq = seq[: n // 2] + "N" + seq[n // 2 + 1:]
            elif r < 0.2:
                seq = "".join(rng.choice(list("ACGT"), size=n))
            elif r < 0.25:
                seq = seq[: n // 3] + (motifs[(block + 5) % len(motifs)] * 300)[: n - n // 3]   # the motif changes inside the read
            recs.append(rec(f"t{i}" + "x" * int(rng.integers(0, 40)), seq))
            i += 1
    fq = b"".join(recs)
    uniform = b"".join(rec("u%07d" % j, ("ACGT" * 40)[:150]) for j in range(20000))   # every record 320 bytes, one phase
    eng = engines(k)
    dev, offs, lens = eng.upload([fq, uniform, fq[::1]])
    wants = [oracle.count_fastq(b, k)[0] for b in (fq, uniform)]
    for parts in (1, 2):
        hist, status = eng.count(dev, offs, lens, parts=parts)
        assert not status.cpu().numpy().any()
        got = hist.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[0], wants[0]), (k, parts)
        assert np.array_equal(got[1], wants[1]), (k, parts)
        assert np.array_equal(got[2], wants[0]), (k, parts)


def test_aside_kernel_serves_several_count_workgroups_of_a_sample(engines):
    """Round 6: the count launch is many workgroups per sample and vk_aside_kernel serves several of them (all of one
    sample) per workgroup of its own -- about a thousand in all, so the grouping only happens in launches of a thousand
    workgroups and more.  1100 copies of a few blobs full of short reads (lanes set aside in nearly every piece) at 3 and
    at 7 workgroups per sample: every histogram equals the oracle's."""
    k = 7
    eng = engines(k)
    blobs = [b for b in _dense_blobs() if len(b) > 300_000][:6]
    want = [oracle.count_fastq(b, k)[0] for b in blobs]
    dev, offs, lens = eng.upload(blobs)
    n = 1100
    idx = np.arange(n) % len(blobs)
    o, l = offs[idx].copy(), lens[idx].copy()
    for parts in (3, 7):
        hist, status = eng.count(dev, o, l, parts=parts)
        assert not status.cpu().numpy().any(), parts
        got = hist.cpu().numpy().view(np.uint32)
        bad = [i for i in range(n) if not np.array_equal(got[i], want[idx[i]])]
        assert not bad, (parts, bad[:8])
