"""Read subsampling on the GPU: the ladder arithmetic of split_fastq
(commands/image.py:677-713, pinned by golden `split_cases` captured from the reference), and the
sampled count kernel against the oracle's restatement of the same rule."""
import json
import os

import numpy as np
import pytest

import fastq_cases
from oracle import oracle
from varkoder_amd import subsample, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def manifest():
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)


def test_ladder_and_names_match_split_fastq(manifest):
    for case in manifest["split_cases"]:
        kw = dict(min_bp=case["min_bp"], max_bp=case["max_bp"], is_query=case["is_query"])
        if "raises" in case:
            with pytest.raises(Exception, match="less than minimum data"):
                subsample.sites_ladder(case["nsites"], **kw)
            continue
        sizes = subsample.sites_ladder(case["nsites"], **kw)
        assert sizes == case["sites_per_file"], case
        assert [subsample.split_name(case["prefix"], bp) + ".fq.gz" for bp in sizes] == case["outfiles"]
        assert [int(case["seed"]) + i for i in range(len(sizes))] == case["sampleseeds"]


def test_threshold_is_a_probability():
    assert subsample.threshold(10, 10) == 1 << 32 and subsample.threshold(11, 10) == 1 << 32
    assert subsample.threshold(0, 10) == 0
    assert subsample.threshold(5_000_000, 20_000_000) == 1 << 30
    assert subsample.threshold(1, 0) == 1 << 32


def test_oracle_rule_on_a_hand_case():
    # two reads; thresholds 0 and 2^32 take none / all
    data = fastq_cases.rec("a", "ACGTACGTAC") + fastq_cases.rec("b", "GGGGGGGGCC")
    full = oracle.count_fastq(data, 5)
    none = oracle.count_fastq_sampled(data, 5, 7, 0)
    allr = oracle.count_fastq_sampled(data, 5, 7, 1 << 32)
    assert none[1] == 0 and none[3] == (20, 0)
    assert allr[1] == full[1] and np.array_equal(allr[0], full[0]) and allr[3] == (20, 20)
    # some seed separates the two reads (each read is all-or-nothing)
    seen = set()
    for seed in range(64):
        fwd, nwin, st, sites = oracle.count_fastq_sampled(data, 5, seed, 1 << 31)
        assert st == 0 and nwin in (0, 6, 12) and sites[1] in (0, 10, 20)
        seen.add(sites[1])
    assert seen == {0, 10, 20}


def _break_case():
    """Reads of 1200, 500, 501 and 499 bases (no N): reformat.sh breaklength=500 (commands/image.py:586-588)
    cuts them into 500 + 500 + 200, 500, 500 + 1 and 499."""
    rng = np.random.default_rng(500)
    reads = ["".join(rng.choice(list("ACGT"), size=n)) for n in (1200, 500, 501, 499)]
    return b"".join(fastq_cases.rec("r%d" % i, r) for i, r in enumerate(reads)), reads


def test_no_window_of_a_subsample_spans_a_multiple_of_500_bases():
    """Hand case for breaklength=500: the windows of a read are the windows of its 500-base pieces."""
    data, reads = _break_case()
    for k in (5, 7, 9):
        fwd, nwin, st, sites = oracle.count_fastq_sampled(data, k, 1, 1 << 32)
        pieces = [r[i:i + 500] for r in reads for i in range(0, len(r), 500)]
        want = sum(max(0, len(p) - k + 1) for p in pieces)
        assert st == 0 and nwin == want and sites == (2700, 2700)
        # ... which are exactly the windows of a file that holds the pieces as reads of their own
        broken = b"".join(fastq_cases.rec("p%d" % i, p) for i, p in enumerate(pieces))
        assert np.array_equal(fwd, oracle.count_fastq(broken, k)[0])
        # the unsampled count (split files: already broken by step C) is untouched by the rule
        assert oracle.count_fastq(data, k)[1] == sum(len(r) - k + 1 for r in reads)


def _engine(k):
    from varkoder_amd.engine import ImageEngine
    return ImageEngine(k=k, mapping="cgr")


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 7, 8])
def test_sampled_count_matches_the_oracle(k):
    eng = _engine(k)
    cases = fastq_cases.edge_cases()
    names = ["one_read", "n_in_middle", "lowercase", "crlf", "no_final_newline", "qual_starts_at_plus", "poly_a"]
    blobs = [cases[n] for n in names]
    blobs += [fastq_cases.random_fastq(np.random.default_rng(s), 400) for s in range(3)]
    blobs.append(synth.sample_fastq(3, 4000, 150, dist=1).tobytes())          # 1.28 MB: several waves
    blobs.append(b"".join(fastq_cases.rec(f"r{i}", "ACGTTGCA" * 2) for i in range(3000)))   # short reads
    blobs.append(b"".join(fastq_cases.rec(f"r{i}", "ACGTAC") for i in range(5000)))         # > 3 newlines per block
    fq, offs, lens = eng.upload(blobs)
    for seed, frac in ((1, 0.5), (2, 0.1), (99, 0.9), (5, 1.0), (6, 0.0)):
        thr = min(1 << 32, int(frac * (1 << 32)))
        for parts in (0, 1, 5):
            hist, status, sites = eng.count_sampled(fq, offs, lens, seed, thr, parts=parts)
            h = hist.cpu().numpy().view(np.uint32)
            st = status.cpu().numpy()
            si = sites.cpu().numpy()
            for i, blob in enumerate(blobs):
                want, nwin, wst, wsites = oracle.count_fastq_sampled(blob, k, seed, thr)
                assert st[i] == 0 and wst == 0, (i, seed, parts)
                assert tuple(int(x) for x in si[i]) == wsites, (i, seed, frac, parts)
                assert np.array_equal(h[i], want), (i, seed, frac, parts)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [8, 9])
def test_walker_at_k8_and_k9(k):
    """k = 8, 9: the walker takes a call only when every threshold is a few per cent of the reads (the library's rule,
    kWalkMaxThresholdSpill) -- the calls of the test above hold larger ones and stream.  Small fractions over samples
    large enough to have reads taken, several splits; the launch must be the walker's (its LDS holds no bucket queues)."""
    eng = _engine(k)
    blobs = [synth.sample_fastq(11 + d, 20000, 150, dist=d).tobytes() for d in range(3)]
    blobs.append(b"".join(fastq_cases.rec(f"L{i}", fastq_cases.rand_seq(np.random.default_rng(i), 1700)) for i in range(400)))
    blobs.append(fastq_cases.random_fastq(np.random.default_rng(77), 3000))
    fq, offs, lens = eng.upload(blobs)
    for parts in (0, 1, 3):
        nsites, status = eng.read_index(fq, offs, lens, parts=parts)
        assert not status.any()
        pairs = [(i, seed, den) for i in range(len(blobs)) for seed, den in ((7, 33), (8, 100), (9, 400))]
        idx = [i for i, _, _ in pairs]
        seeds = np.array([s_ for _, s_, _ in pairs], dtype=np.uint64)
        thr = np.array([(1 << 32) // den for _, _, den in pairs], dtype=np.uint64)
        assert int(thr.max()) <= subsample.WALK_MAX_THRESHOLD_SPILL
        hist, st, sites = eng.count_sampled(fq, offs[idx], lens[idx], seeds, thr)
        assert eng.last_count_launch()["lds_bytes"] < 16384            # the walker, not the spill path
        h = hist.cpu().numpy().view(np.uint32)
        si = sites.cpu().numpy()
        assert not st.cpu().numpy().any()
        taken = 0
        for j, (i, seed, den) in enumerate(pairs):
            want, nwin, wst, wsites = oracle.count_fastq_sampled(blobs[i], k, seed, int(thr[j]))
            assert tuple(int(x) for x in si[j]) == wsites, (j, i, seed, den, parts)
            assert int(h[j].sum(dtype=np.uint64)) == nwin and np.array_equal(h[j], want), (j, i, seed, den, parts)
            taken += nwin
        assert taken > 10000
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 7, 9])
def test_read_index_and_walker_match_the_oracle(k):
    """vk_read_index_device + the walker (vk_ladder.h): the same subsampled counts and sites as the streaming kernel and
    the oracle, for the edge cases, all three synthetic read shapes (fixed 150, GC skew + homopolymers, fastp's 0 .. 290),
    long reads across many 500-base breaks, several workgroup splits, the same sample under several (seed, threshold)
    pairs in one call; nsites from the index equals the oracle's; a sample of reads too short for the index (more than one
    per 32 bytes) falls back to the streaming kernel, still exact."""
    eng = _engine(k)
    cases = fastq_cases.edge_cases()
    names = ["one_read", "n_in_middle", "lowercase", "crlf", "no_final_newline", "qual_starts_at_plus", "poly_a", "ragged", "long_read"]
    blobs = [cases[n] for n in names]
    blobs += [fastq_cases.random_fastq(np.random.default_rng(40 + s), 300) for s in range(3)]
    blobs += [synth.sample_fastq(3 + d, 5000, 150, dist=d).tobytes() for d in range(3)]
    blobs.append(b"".join(fastq_cases.rec(f"L{i}", fastq_cases.rand_seq(np.random.default_rng(i), 1700)) for i in range(60)))
    fq, offs, lens = eng.upload(blobs)
    for parts in (0, 1, 3):
        nsites, status = eng.read_index(fq, offs, lens, parts=parts)
        for i, blob in enumerate(blobs):
            _, _, wst, wsites = oracle.count_fastq_sampled(blob, k, 0, 1 << 32)
            assert status[i] == 0 and wst == 0 and int(nsites[i]) == wsites[0], (i, parts)
        # every blob under three (seed, fraction) pairs, one launch
        pairs = [(i, seed, frac) for i in range(len(blobs)) for seed, frac in ((7, 0.3), (8, 1.0), (9, 0.05))]
        idx = [i for i, _, _ in pairs]
        seeds = np.array([s_ for _, s_, _ in pairs], dtype=np.uint64)
        thr = np.array([min(1 << 32, int(f * (1 << 32))) for _, _, f in pairs], dtype=np.uint64)
        hist, st, sites = eng.count_sampled(fq, offs[idx], lens[idx], seeds, thr)
        assert eng.last_count_launch()["grid"] % len(pairs) == 0      # one workgroup per (pair, part): the walker's grid
        h = hist.cpu().numpy().view(np.uint32)
        si = sites.cpu().numpy()
        assert not st.cpu().numpy().any()
        for j, (i, seed, frac) in enumerate(pairs):
            want, nwin, wst, wsites = oracle.count_fastq_sampled(blobs[i], k, seed, int(thr[j]))
            assert tuple(int(x) for x in si[j]) == wsites, (j, i, seed, frac, parts)
            assert int(h[j].sum(dtype=np.uint64)) == nwin and np.array_equal(h[j], want), (j, i, seed, frac, parts)
    # reads of a few bases: more anchors than the index holds -> no index for that sample, the call streams instead
    tiny = b"".join(fastq_cases.rec("t", "ACGTAC") for i in range(4000))
    fq2, o2, l2 = eng.upload([tiny, blobs[-1]])
    nsites, status = eng.read_index(fq2, o2, l2)
    assert int(nsites[0]) == 6 * 4000 and int(nsites[1]) == 1700 * 60
    hist, st, sites = eng.count_sampled(fq2, o2, l2, 5, 1 << 31)
    for i, blob in enumerate((tiny, blobs[-1])):
        want, nwin, wst, wsites = oracle.count_fastq_sampled(blob, k, 5, 1 << 31)
        assert np.array_equal(hist.cpu().numpy().view(np.uint32)[i], want) and tuple(int(x) for x in sites.cpu().numpy()[i]) == wsites
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 6, 7, 8])
def test_count_and_index_in_one_pass(k):
    """vk_count_index_device: the plain count, the number of sites and the read index out of one pass over the text (k <= 7:
    the count kernel lists the anchors on its way -- fast path, general path and the lanes set aside each add theirs);
    subsamples walked from that index equal the oracle's, as from vk_read_index_device's."""
    eng = _engine(k)
    cases = fastq_cases.edge_cases()
    blobs = [cases[n] for n in ("one_read", "crlf", "no_final_newline", "qual_starts_at_plus", "ragged", "long_read", "long_header")]
    blobs += [fastq_cases.random_fastq(np.random.default_rng(70 + s), 300) for s in range(3)]
    blobs += [synth.sample_fastq(13 + d, 6000, 150, dist=d).tobytes() for d in range(3)]
    fq, offs, lens = eng.upload(blobs)
    for parts in (0, 1, 4):
        hist, nsites, status = eng.count_index(fq, offs, lens, parts=parts)
        h = hist.cpu().numpy().view(np.uint32)
        for i, blob in enumerate(blobs):
            want, nwin, wst = oracle.count_fastq(blob, k)
            _, _, _, wsites = oracle.count_fastq_sampled(blob, k, 0, 1 << 32)
            assert status[i] == 0 and wst == 0 and np.array_equal(h[i], want), (i, parts)
            assert int(nsites[i]) == wsites[0], (i, parts, int(nsites[i]), wsites[0])
        idx = list(range(len(blobs))) * 2
        seeds = np.array([3] * len(blobs) + [4] * len(blobs), dtype=np.uint64)
        thr = np.array([1 << 30] * len(blobs) + [(1 << 32) // 50] * len(blobs), dtype=np.uint64)
        sh, st, sites = eng.count_sampled(fq, offs[idx], lens[idx], seeds, thr)
        for j, i in enumerate(idx):
            want, nwin, wst, wsites = oracle.count_fastq_sampled(blobs[i], k, int(seeds[j]), int(thr[j]))
            assert np.array_equal(sh.cpu().numpy().view(np.uint32)[j], want), (j, i, parts)
            assert tuple(int(x) for x in sites.cpu().numpy()[j]) == wsites, (j, i, parts)
    eng.close()


@pytest.mark.gpu
def test_sampling_fraction_and_seed_independence():
    eng = _engine(7)
    blob = synth.sample_fastq(11, 40000, 150, dist=0).tobytes()
    fq, offs, lens = eng.upload([blob])
    fracs = []
    hists = []
    for seed in (10, 11):
        hist, status, sites = eng.count_sampled(fq, offs, lens, seed, 1 << 30)      # p = 1/4
        s = sites.cpu().numpy()[0]
        assert s[0] == 40000 * 150
        fracs.append(s[1] / s[0])
        hists.append(hist.cpu().numpy().copy())
    assert all(abs(f - 0.25) < 0.01 for f in fracs), fracs          # 40000 reads: sd 0.0022
    assert not np.array_equal(hists[0], hists[1])
    eng.close()


@pytest.mark.gpu
def test_ladder_counts_on_device():
    eng = _engine(7)
    blobs = [synth.sample_fastq(s, 20000, 150, dist=1).tobytes() for s in (0, 1)]      # 3 Mbp each
    blobs.append(synth.sample_fastq(2, 200, 150).tobytes())                            # 30 kbp: too little
    fq, offs, lens = eng.upload(blobs)
    recs = subsample.ladder_counts(eng, fq, offs, lens, seed=100, min_bp=500_000, max_bp=2_000_000)
    assert [r["nsites"] for r in recs] == [3_000_000, 3_000_000, 30_000]
    assert recs[2]["error"] and not recs[2]["steps"]
    for i in (0, 1):
        assert [bp for bp, _, _ in recs[i]["steps"]] == [2_000_000, 1_000_000, 500_000]
        for level, (bp, hist, taken) in enumerate(recs[i]["steps"]):
            thr = subsample.threshold(bp, 3_000_000)
            want, nwin, st, sites = oracle.count_fastq_sampled(blobs[i], 7, 100 + level, thr)
            assert np.array_equal(hist.cpu().numpy().view(np.uint32), want)
            assert taken == sites[1] and abs(taken - bp) < 0.05 * bp
    # no max_bp: the first step is the whole file, counted once
    recs = subsample.ladder_counts(eng, fq, offs[:1], lens[:1], seed=3, min_bp=1_000_000, max_bp=None)
    assert [bp for bp, _, _ in recs[0]["steps"]] == [3_000_000, 2_000_000, 1_000_000]
    assert np.array_equal(recs[0]["steps"][0][1].cpu().numpy().view(np.uint32), oracle.count_fastq(blobs[0], 7)[0])
    eng.close()


@pytest.mark.gpu
def test_cli_from_clean_writes_the_ladder(tmp_path):
    """`image --from-clean`: steps C+D+E from cleaned reads; every PNG equals the oracle's image of
    the oracle's sampled counts for the seed / threshold the command derives."""
    import gzip

    import pandas as pd
    from PIL import Image
    from varkoder_amd import cli
    clean = tmp_path / "int" / "clean_reads"
    clean.mkdir(parents=True)
    blobs = {"sampA": synth.sample_fastq(4, 20000, 150, dist=1).tobytes(),
             "sampB": synth.sample_fastq(5, 8000, 150).tobytes()}
    (clean / "sampA.fq").write_bytes(blobs["sampA"])
    with gzip.open(clean / "sampB.fq.gz", "wb") as f:
        f.write(blobs["sampB"])
    out, stats = tmp_path / "images", tmp_path / "stats.csv"
    cli.main(["image", "--from-clean", "-k", "7", "-p", "cgr", "-m", "500K", "-M", "2M", "-R", "5", "-o", str(out),
              "-f", str(stats), str(tmp_path / "int")])
    ladders = {"sampA": [2_000_000, 1_000_000, 500_000], "sampB": [1_200_000, 1_000_000, 500_000]}
    rng = np.random.default_rng(5)
    pix = oracle.cgr_lut(7)
    got = sorted(p.name for p in out.glob("*.png"))
    want_names = sorted(f"{s}@{bp // 1000:08d}K+cgr+k7.png" for s, l in ladders.items() for bp in l)
    assert got == want_names
    for i, s in enumerate(sorted(blobs)):
        seed = int(str(i) + str(rng.integers(low=0, high=2 ** 32))) % (1 << 63)
        nsites = 150 * (20000 if s == "sampA" else 8000)
        for level, bp in enumerate(ladders[s]):
            thr = subsample.threshold(bp, nsites)
            fwd = oracle.count_fastq_sampled(blobs[s], 7, seed + level, thr)[0]
            img = oracle.image(oracle.strand_merge(fwd, 7), 7, pix, 4 ** 7).reshape(128, 128)
            im = Image.open(out / f"{s}@{bp // 1000:08d}K+cgr+k7.png")
            assert np.array_equal(np.array(im), img), (s, bp)
            assert im.info["varkoderMapping"] == "cgr"
    df = pd.read_csv(stats)
    assert list(df["sample"]) == ["sampA", "sampB"]
    assert list(df["splitting_bp_per_file"]) == ["2000000,1000000,500000", "1200000,1000000,500000"]
    assert {"7mer_counting_time", "k7_img_time", "splitting_time"} <= set(df.columns)


@pytest.mark.gpu
@pytest.mark.parametrize("k", (5, 7, 8, 9))
def test_breaklength_on_the_gpu(k):
    """vk_count_sampled_device applies reformat.sh's breaklength=500: hand case and long ragged reads, every
    split of the samples into byte ranges."""
    data, _ = _break_case()
    rng = np.random.default_rng(9)
    long_reads = b"".join(fastq_cases.rec("L%d" % i, "".join(rng.choice(list("ACGTN"), size=int(n),
                                                                            p=[.24, .24, .24, .24, .04])))
                          for i, n in enumerate(rng.integers(1, 6000, size=400)))
    eng = _engine(k)
    try:
        blobs = [data, long_reads, data * 200]
        dev, offs, lens = eng.upload(blobs)
        for seed, thr in ((3, 1 << 32), (4, 1 << 31)):
            for parts in (0, 1, 3):
                hist, status, sites = eng.count_sampled(dev, offs, lens, seed, thr, parts=parts)
                got = hist.cpu().numpy().view(np.uint32)
                for i, b in enumerate(blobs):
                    want, nwin, st, wsites = oracle.count_fastq_sampled(b, k, seed, thr)
                    assert st == 0 and int(status.cpu()[i]) == 0
                    assert tuple(int(x) for x in sites.cpu().numpy()[i]) == wsites, (i, parts)
                    assert np.array_equal(got[i], want), (k, i, seed, parts)
    finally:
        eng.close()
